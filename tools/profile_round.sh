#!/bin/bash
# Produces the profile artefacts of a round under gpurun_out/ (copy the summaries to profiles/):
#   1. rocprofv3 --kernel-trace --stats of the default bench run (the judged configuration)
#   2. FETCH_SIZE / WRITE_SIZE passes of the same configuration (HBM traffic of one k_render launch)
#   3. SQ / TCC / LDS counter passes at 64 spp (tools/pmc_profile.sh)
# usage: tools/profile_round.sh TAG      (run from the repository root, on the GPU box)
set -u
TAG=$1
R=$(pwd)
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT" -o trace -- python3 "$R/bench.py" --steps 3 --warmup 1 > "$R/$OUT/bench.log" 2>&1 || { echo "trace run failed"; tail -5 "$R/$OUT/bench.log"; exit 1; }
grep '"metric"' "$R/$OUT/bench.log" > "$R/$OUT/bench.json"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d "$R/$OUT" -o full_$c -- python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$R/$OUT/full_$c.log" 2>&1 || { echo "pass $c failed"; exit 1; }
done
cd "$R" && tools/pmc_profile.sh "$OUT/pmc64" --spp 64 && python3 tools/pmc_summary.py "$OUT/pmc64" > "$OUT/pmc64_summary.txt"
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
tot = {}
for f in glob.glob(os.path.join(out, "full_*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "k_render<float" in row["Kernel_Name"]:   # the timed frame, not the small image-check render
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
# FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes,
# so the read side is doubled (MI355X_MICROARCH.md, HBM section).
res = {"config": "bench.py defaults (book1, 1920x1080, 1024 spp), one step: costing pass + main k_render launch",
       "FETCH_SIZE_KiB": tot.get("FETCH_SIZE"), "WRITE_SIZE_KiB": tot.get("WRITE_SIZE"),
       "hbm_bytes_per_launch": (2 * tot.get("FETCH_SIZE", 0) + tot.get("WRITE_SIZE", 0)) * 1024}
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(res)
PY
cat "$OUT/trace_kernel_stats.csv" | cut -c1-200
cat "$OUT/bench.json" | cut -c1-300
