#!/usr/bin/env python3
"""Turns the PMC passes of tools/pmc_profile.sh into the two files bench.py's `roofline` reads.

usage: tools/roofline_inputs.py PMC_DIR OUT_DIR CONFIG_KEY TAG

Takes every counter from the last main k_render launch of its pass (the passes run one warm-up frame,
which measures the tile costs, and one frame like the benchmark's timed ones, which reuses the tile
order: one launch, what rt_stats.kernel_ms times) and enters them under CONFIG_KEY into
OUT_DIR/valu.json and OUT_DIR/traffic.json ({"configs": {CONFIG_KEY: {...}}}; other keys already in
those files are kept).

Units (checked against tools/microbench/valu_rates under the same counters): the SQ cycle
counters tick once per 4 clocks; SQ_THREAD_CYCLES_VALU adds, per VALU instruction pass, the
number of active lanes, so a full 64-lane single-pass instruction adds 64 = the lane-slots a SIMD
offers in 4 clocks; GRBM_GUI_ACTIVE is summed over the 8 XCDs.  FETCH_SIZE and WRITE_SIZE are in
KiB, and on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes, so the read side is
doubled.
"""
import csv
import glob
import json
import os
import re
import sys

N_XCD, N_SIMD = 8, 1024


def merge(path, config_key, record):
    doc = {"configs": {}}
    if os.path.exists(path):
        old = json.load(open(path))
        if "configs" in old:
            doc = old
    doc["configs"][config_key] = record
    json.dump(doc, open(path, "w"), indent=1)


def main():
    pmc_dir, out_dir, config_key, tag = sys.argv[1:5]
    # the last main launch of each pass: the frame a timed step renders (tile order on file from the warm-up frame)
    tot, span = {}, {}
    main_launch = re.compile(r"k_render<\w+, \w+, \d+, false,")
    for f in sorted(glob.glob(os.path.join(pmc_dir, "*_counter_collection.csv"))):
        rows = [row for row in csv.DictReader(open(f)) if main_launch.search(row["Kernel_Name"])]
        if not rows:
            continue
        last = max(int(row["Dispatch_Id"]) for row in rows)
        for row in rows:
            if int(row["Dispatch_Id"]) != last:
                continue
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            if row["Counter_Name"] == "SQ_THREAD_CYCLES_VALU":
                span[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    g = tot.get
    need = ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"]
    missing = [k for k in need if g(k) is None]
    if missing:
        sys.exit(f"missing counters {missing} in {pmc_dir}")
    clocks = g("GRBM_GUI_ACTIVE") / N_XCD                    # shader clocks the frame's kernels were resident
    lanes = g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU")
    valu = {
        "config_key": config_key, "source": f"{tag}/{os.path.basename(pmc_dir.rstrip('/'))} (tools/pmc_profile.sh; the second frame's k_render launch)",
        "SQ_THREAD_CYCLES_VALU": g("SQ_THREAD_CYCLES_VALU"), "SQ_ACTIVE_INST_VALU": g("SQ_ACTIVE_INST_VALU"), "SQ_INSTS_VALU": g("SQ_INSTS_VALU"),
        "SQ_WAVE_CYCLES": g("SQ_WAVE_CYCLES"), "SQ_WAVES": g("SQ_WAVES"), "GRBM_GUI_ACTIVE": g("GRBM_GUI_ACTIVE"),
        "lanes_active": lanes,                                                     # of 64, per VALU instruction
        # VALU instruction passes issued per SIMD per 4 clocks; simple 32-bit integer instructions complete two
        # per 4 clocks when several waves offer them, so this can exceed 1 (microbenchmark: 1.55 for v_add_u32)
        "valu_busy": 4.0 * g("SQ_ACTIVE_INST_VALU") / (N_SIMD * clocks),
        "occupancy": 4.0 * g("SQ_WAVE_CYCLES") / (N_SIMD * clocks),                # mean resident waves per SIMD
        "valu_insts_per_launch": g("SQ_INSTS_VALU"),
        "pmc_kernel_ms": sum(span.values()),
        "pmc_shader_clock_ghz": clocks / (sum(span.values()) / 1e3) / 1e9 if span else None,
    }
    f64 = [g("SQ_INSTS_VALU_ADD_F64"), g("SQ_INSTS_VALU_MUL_F64"), g("SQ_INSTS_VALU_FMA_F64"), g("SQ_INSTS_VALU_TRANS_F64")]
    if None not in f64:
        valu["f64_insts_per_launch"] = {"add": f64[0], "mul": f64[1], "fma": f64[2], "trans": f64[3]}
        # wave-instructions x the mean lane count of all VALU instructions (the counters do not split lanes by class)
        valu["f64_flops_per_launch"] = (f64[0] + f64[1] + 2 * f64[2] + f64[3]) * lanes
    merge(os.path.join(out_dir, "valu.json"), config_key, valu)
    print(json.dumps(valu))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        traffic = {"config_key": config_key, "source": f"{tag}/{os.path.basename(pmc_dir.rstrip('/'))}", "FETCH_SIZE_KiB": g("FETCH_SIZE"), "WRITE_SIZE_KiB": g("WRITE_SIZE"),
                   "hbm_bytes_per_launch": (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024}
        merge(os.path.join(out_dir, "traffic.json"), config_key, traffic)
        print(json.dumps(traffic))


if __name__ == "__main__":
    main()
