#!/bin/bash
# Hardware-counter passes over tools/onna_quick.py (the Lisp tracer's features): tools/onna_pmc.sh OUTDIR SCENE
set -u
OUT=$1; SCENE=$2
R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
    name=$1; shift
    timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$R/$OUT" -o "$name" -- python3 "$R/tools/onna_quick.py" "$SCENE" > "$R/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$R/$OUT/$name.log"; exit 1; }
}
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT
run sq_b SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY
run mem TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
cd "$R"
python3 tools/pmc_summary.py "$OUT" all
