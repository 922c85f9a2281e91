#!/bin/bash
ARGS="--scene stress:1000000 --camera 0,25,60,0,0,0,40 --spp 16 --steps 2 --no-cpu-baseline"
run() { echo -n "$1 : "; env $1 timeout -k 10 200 python -u bench.py $ARGS 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms'],2))"; }
run "X=1"
run "RTIOW_RENDER_LDS_NODES=0"
run "RTIOW_RENDER_LDS_NODES=127"
run "RTIOW_RENDER_LDS_NODES=400"
run "RTIOW_RENDER_STACK_CAP=6"
run "RTIOW_RENDER_STACK_CAP=8 RTIOW_RENDER_LDS_NODES=0"
run "X=2"
