#!/bin/bash
# Runs bench.py with the given arguments and prints its line in short form.
# usage: tools/bench_summary.sh OUTDIR [bench args]       (from the repository root)
OUT=$1; shift
mkdir -p "$OUT"
RTIOW_VERBOSE=1 timeout -k 10 600 python3 bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
python3 - "$OUT/bench.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("value %.0f Mpaths/s, %.2f ms/step, kernel %.2f ms, roofline frac %s lanes %s busy %s occ %s, build %.3f s" % (
    d["value"], d["ms_per_step"], r["kernel_ms"], r.get("frac"), r.get("lanes_active"), r.get("valu_busy"), r.get("occupancy"), d["scene_build_upload_s"]))
if d.get("image_check"): print("image_check", d["image_check"])
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("all_cores", {}).get("value"))
for e in d.get("extra_configs", []):
    print("  %-50s %6.0f Mpaths/s  kernel %8.2f ms  setup %s" % (e["config"][:50], e["value"], e["kernel_ms"], e.get("setup_s")))
PY
grep "rtiow:" "$OUT/bench.err" | tail -8
