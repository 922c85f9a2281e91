#!/bin/bash
# One SQ counter pass over a short bench run.  usage: tools/pmc_quick.sh OUTDIR [bench args]
set -u
OUT=$1; shift
R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH_ARGS="${*:---spp 32}"
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$R/$OUT" -o sq_a -- \
    python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > "$R/$OUT/sq_a.log" 2>&1 || { echo failed; tail -3 "$R/$OUT/sq_a.log"; exit 1; }
timeout -s KILL 180 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d "$R/$OUT" -o sq_b -- \
    python3 "$R/bench.py" --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > "$R/$OUT/sq_b.log" 2>&1 || { echo failed; tail -3 "$R/$OUT/sq_b.log"; exit 1; }
cd "$R" && python3 tools/pmc_summary.py "$OUT"
