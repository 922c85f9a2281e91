#!/bin/bash
# Produces the profile artefacts of a round under gpurun_out/TAG (copy the summaries to profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench run (the judged configuration)
#   2. every PMC pass of tools/pmc_profile.sh over that same configuration (1920x1080, 1024 spp)
#   3. valu.json / traffic.json: what bench.py's `roofline` reads (lane-slots, busy, occupancy; HBM bytes)
# usage: tools/profile_round.sh TAG      (run from the repository root, on the GPU box)
set -u
TAG=$1
R=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python3 "$R/bench.py" --steps 3 --warmup 1 > "$R/$OUT/bench.log" 2>&1 || { echo "trace run failed"; tail -5 "$R/$OUT/bench.log"; exit 1; }
grep '"metric"' "$R/$OUT/bench.log" > "$R/$OUT/bench.json"
cd "$R" && tools/pmc_profile.sh "$OUT/pmc1024" --spp 1024 || exit 1
python3 tools/pmc_summary.py "$OUT/pmc1024" > "$OUT/pmc1024_summary.txt"
python3 tools/roofline_inputs.py "$OUT/pmc1024" "$OUT" "book1 1920x1080 samples [0,1024)" "$TAG" || exit 1
cut -c1-200 "$OUT/trace_kernel_stats.csv"
cut -c1-600 "$OUT/bench.json"
cat "$OUT/pmc1024_summary.txt"
