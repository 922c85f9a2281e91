#!/usr/bin/env python3
"""Sums the counters of the k_render dispatch from the CSVs written by tools/pmc_profile.sh."""
import csv, glob, os, sys
d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_render"
tot = {}
for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if kern in row["Kernel_Name"]:
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
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
if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
    print("L2 hit rate                     %.4f" % (g("TCC_HIT_sum") / max(1.0, g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
