#!/bin/bash
# Kernel ms of the 485-sphere frame (1024 spp), book-2 (256 spp) and 10^6 spheres (64 spp) for each library named: tools/ab_c2.sh LIB...
for lib in "$@"; do
    for cfg in "--spp 1024" "--config c3 --spp 256" "--config c5"; do
        RTIOW_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --timed-only --steps 2 --warmup 1 $cfg 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print('%-28s %-22s kernel %8.3f ms' % ('$(basename $lib)', '$cfg', d['roofline']['kernel_ms']))" || exit 1
    done
done
