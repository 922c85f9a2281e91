#!/bin/bash
# Produces the profile artefacts of a round under gpurun_out/TAG (copy the summaries to profiles/):
#   1. every PMC pass of tools/pmc_profile.sh over the judged configuration (1920x1080, 1024 spp)
#      and over the book-2 and 10^6-sphere configurations (BASELINE.json configs[2], configs[4])
#   2. valu.json / traffic.json from those passes, keyed by configuration: what bench.py's
#      `roofline` objects read (lane-slots, busy, occupancy; HBM bytes); also placed in profiles/
#      of this checkout for step 3
#   3. the default bench run, un-profiled, with its roofline filled from step 2 -> bench.json
#   4. rocprofv3 --kernel-trace --stats of the same run without the extra configurations, so that
#      the statistics of k_render<float, ...> hold the timed 1024-spp launches only
# usage: tools/profile_round.sh TAG [main]     (run from the repository root, on the GPU box;
#                                               `main`: the judged configuration only)
set -u
TAG=$1
ONLY=${2:-all}
R=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rm -f "$OUT/valu.json" "$OUT/traffic.json"
pass() {   # pass NAME CONFIG_KEY bench-args...
    name=$1; key=$2; shift 2
    tools/pmc_profile.sh "$OUT/$name" "$@" || exit 1
    python3 tools/pmc_summary.py "$OUT/$name" > "$OUT/${name}_summary.txt"
    python3 tools/roofline_inputs.py "$OUT/$name" "$OUT" "$key" "$TAG" > /dev/null || exit 1
}
pass pmc1024 "book1 1920x1080 samples [0,1024)" --spp 1024
if [ "$ONLY" != main ]; then
    pass c3_pmc1024 "book2 1920x1080 samples [0,1024)" --config c3
    pass c5_pmc64 "stress:1000000 1920x1080 samples [0,64)" --config c5
fi
cp "$OUT/valu.json" "$OUT/traffic.json" profiles/
timeout -k 10 900 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-extra > "$R/$OUT/trace_bench.log" 2>&1 || { echo "trace run failed"; tail -5 "$R/$OUT/trace_bench.log"; exit 1; }
cd "$R"
cut -c1-200 "$OUT/trace_kernel_stats.csv"
cut -c1-900 "$OUT/bench.json"
cat "$OUT/pmc1024_summary.txt"
