#!/bin/bash
# kernel time against the number of costing samples: tools/sweep_probe.sh SPP
SPP=$1
for rep in 1 2 3; do
for p in 1 2 4 8; do
  echo -n "spp $SPP probe $p : "; RTIOW_TILE_PROBE=$p timeout -k 10 200 python -u bench.py --spp $SPP --steps 4 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms'],2))"
done
done
