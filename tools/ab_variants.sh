#!/bin/bash
# A/B several builds of the GPU library on one box, interleaved so that clock drift hits all alike:
#   tools/ab_variants.sh "bench args" lib1.so lib2.so ...      (REPS=4 by default)
# prints, per run: library, Mpaths/s, mean radiance (must not change), mean and fastest kernel ms
ARGS=$1; shift
for rep in $(seq 1 ${REPS:-4}); do
  for lib in "$@"; do
    echo -n "$lib "; RTIOW_HIP_LIB=$PWD/$lib timeout -k 10 120 python -u bench.py $ARGS --timed-only 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['mean_radiance'],6), round(d['roofline']['kernel_ms'],2), 'min', round(d['roofline']['kernel_ms_min'],2))"
  done
done
