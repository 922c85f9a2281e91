#!/usr/bin/env python3
"""Counters of one k_render launch from the CSVs written by tools/pmc_profile.sh.

usage: tools/pmc_summary.py PMC_DIR [all]

Every pass runs a warm-up frame (costing pass + main launch) and a frame like the benchmark's timed
ones (main launch alone, tile order reused): the figures are those of the LAST main launch of each
pass, as in tools/roofline_inputs.py.  With `all`, every k_render dispatch is summed instead."""
import csv, glob, os, re, sys
d = sys.argv[1]
everything = len(sys.argv) > 2 and sys.argv[2] == "all"
main_launch = re.compile(r"k_render<\w+, \w+, \d+, false,")
tot = {}
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    rows = [row for row in csv.DictReader(open(f)) if (("k_render" in row["Kernel_Name"]) if everything else main_launch.search(row["Kernel_Name"]))]
    if not rows:
        continue
    last = max(int(row["Dispatch_Id"]) for row in rows)
    for row in rows:
        if everything or int(row["Dispatch_Id"]) == last:
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
print("# " + ("every k_render dispatch" if everything else "the last main k_render launch of each pass"))
for k in sorted(tot):
    print(f"{k:40s} {tot[k]:.6g}")
g = tot.get
if g("SQ_WAVE_CYCLES"):
    print("--- derived")
    print("VALU active / wave cycles       %.3f" % (g("SQ_ACTIVE_INST_VALU", 0) / g("SQ_WAVE_CYCLES")))
    print("wait_any / wave cycles          %.3f" % (g("SQ_WAIT_ANY", 0) / g("SQ_WAVE_CYCLES")))
    print("wait_inst_any / wave cycles     %.3f" % (g("SQ_WAIT_INST_ANY", 0) / g("SQ_WAVE_CYCLES")))
    if g("SQ_INSTS_VALU"):
        print("lanes active per VALU inst      %.1f of 64" % (g("SQ_THREAD_CYCLES_VALU", 0) / g("SQ_ACTIVE_INST_VALU", 1) / 1.0))
        print("VALU insts per wave             %.0f" % (g("SQ_INSTS_VALU") / g("SQ_WAVES", 1)))
    if g("GRBM_GUI_ACTIVE"):
        clocks = g("GRBM_GUI_ACTIVE") / 8
        print("VALU busy (x4 / SIMD clocks)    %.3f" % (4 * g("SQ_ACTIVE_INST_VALU", 0) / (1024 * clocks)))
        print("waves per SIMD                  %.2f" % (4 * g("SQ_WAVE_CYCLES") / (1024 * clocks)))
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
    print("L2 hit rate                     %.4f" % (g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
