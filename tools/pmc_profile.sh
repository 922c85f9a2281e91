#!/bin/bash
# Hardware-counter passes over one short bench run (a warm-up frame and a timed one each), one rocprofv3 process per pass: the counters of a pass must fit the per-block slot
# limits (8 SQ, 4 TCC with FETCH_SIZE = 3 and WRITE_SIZE = 2, 4 TCP, 2 TA, 2 GRBM).
# usage: tools/pmc_profile.sh OUTDIR [bench args...]   (run from the repository root)
set -u
OUT=$1; shift
R=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
    name=$1; shift
    timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$R/$OUT" -o "$name" -- \
        python3 "$R/bench.py" --steps 1 --warmup 1 --timed-only $BENCH_ARGS > "$R/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$R/$OUT/$name.log"; exit 1; }
}
BENCH_ARGS="${*:---spp 32}"
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE GRBM_COUNT
run sq_b SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY
run mem TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT
echo done
