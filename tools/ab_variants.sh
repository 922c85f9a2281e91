#!/bin/bash
# A/B several builds of the GPU library on one box: tools/ab_variants.sh "bench args" lib1.so lib2.so ...
ARGS=$1; shift
for lib in "$@"; do
  for rep in 1 2; do
    echo -n "$lib "; RTIOW_HIP_LIB=$PWD/$lib timeout -k 10 120 python -u bench.py $ARGS --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['mean_radiance'],6), round(d['roofline']['kernel_ms'],2))"
  done
done
