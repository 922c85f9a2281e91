#!/bin/bash
# Kernel times of the three judged scenes under the environment given on the command line:
#   tools/ab_quick.sh TAG [VAR=value ...]      -> gpurun_out/TAG.txt
# (1920x1080: the 485-sphere scene at 1024 spp, book-2 at 1024 spp unless SKIP_C3=1, 10^6 spheres at 64 spp)
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/$TAG.txt
: > "$OUT"
run() {
    timeout -k 10 300 python3 bench.py --timed-only --steps 2 --warmup 1 "$@" 2>> gpurun_out/$TAG.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('%-60s %8.0f Mpaths/s  kernel %8.3f ms  first %8.3f ms' % (d['config']['workload'][:60], d['value'], d['roofline']['kernel_ms'], d['first_frame_kernel_ms']))" >> "$OUT" || exit 1
}
run --spp 1024 && { [ -n "${SKIP_C3:-}" ] || run --config c3; } && run --config c5 && run --config c5 --spp 8 && run --spp 128
cat "$OUT"
