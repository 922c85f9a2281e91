#!/bin/bash
# kernel time of the default frame under a list of environment settings:
#   tools/sweep_env.sh "bench args" "VAR=a" "VAR=b OTHER=c" ...
ARGS=$1; shift
for rep in 1 2; do
for e in "$@"; do
  echo -n "$e : "; env $e timeout -k 10 300 python -u bench.py $ARGS --timed-only 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel_ms_steps'])"
done
done
