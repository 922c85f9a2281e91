#!/bin/bash
# The parity file and the no-alloc tests with one diagnostic switch exported for the whole run, switch by switch:
#   tools/switch_runs.sh   -> gpurun_out/switch_runs.txt   (run on the GPU box)
OUT=gpurun_out/switch_runs.txt
: > $OUT
run() {
    name=$1; shift
    env "$@" timeout -k 10 600 python3 -u -m pytest tests/test_gpu_parity.py tests/test_zz_no_alloc.py -q -x --timeout 300 --timeout-method thread \
        --deselect tests/test_gpu_parity.py::test_four_wide_walk_changes_nothing --deselect tests/test_gpu_parity.py::test_tree_built_on_the_device_changes_nothing \
        > gpurun_out/switch_$name.log 2>&1
    rc=$?
    echo "$name ($*): rc $rc: $(tail -1 gpurun_out/switch_$name.log)" >> $OUT
    [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc      # (test failures go on to the next switch; a crash or a time limit ends the script)
}
run large_tree_shape RTIOW_RENDER_LDS_NODES=25 RTIOW_RENDER_STACK_CAP=2
run reference_layout RTIOW_TREE=reference
run general_variant RTIOW_STATIC_VARIANT=0
run waves3 RTIOW_RENDER_WAVES=3
run tiles8 RTIOW_TILE_SHIFT=3
run tiles2 RTIOW_TILE_SHIFT=1
run far_slack RTIOW_FAR_SLACK_ALWAYS=1
run wide RTIOW_WIDE=1
run wide_device_memory RTIOW_WIDE=1 RTIOW_RENDER_LDS_NODES=0 RTIOW_WIDE_STACK_CAP=4
run device_build RTIOW_BUILD=device
run device_build_unstaged RTIOW_BUILD=device RTIOW_RENDER_LDS_NODES=0 RTIOW_RENDER_STACK_CAP=2
run strips RTIOW_TILE_REGIONS=8
run strips_large_tree_shape RTIOW_TILE_REGIONS=8 RTIOW_RENDER_LDS_NODES=25 RTIOW_RENDER_STACK_CAP=2
cat $OUT
