#!/bin/bash
# clock under the 10^6-sphere frame for a library variant: clk.sh TAG LIB
TAG=$1; LIB=$2
R=$(pwd); OUT=$R/gpurun_out/clk_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RTIOW_HIP_LIB=$LIB timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT -o p -- python3 $R/bench.py --steps 1 --warmup 1 --timed-only --config c5 > $OUT/log 2>&1
cd $R
python3 - <<PY
import csv,glob,re
rows=[r for f in glob.glob("$OUT/*counter_collection.csv") for r in csv.DictReader(open(f)) if re.search(r"k_render<\w+, \w+, \d+, false,", r["Kernel_Name"])]
last=max(int(r["Dispatch_Id"]) for r in rows)
d={r["Counter_Name"]:float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"])==last}
ms=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6 for r in rows if int(r["Dispatch_Id"])==last][0]
print("$TAG", "ms %.2f"%ms, "clock GHz %.3f"%(d["GRBM_GUI_ACTIVE"]/8/ms/1e6), "valu/wave %.3f"%(d["SQ_ACTIVE_INST_VALU"]/d["SQ_WAVE_CYCLES"]), "wait_any %.3f"%(d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"]))
PY
