#!/bin/bash
# round 5, GPU call 11: full GPU suite on the product / test-build split; why stem2_fused is not engaged (debug print of the test build)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call11
mkdir -p $OUT
cd $ROOT
RIFE_HIP_LIB=$ROOT/rife-ncnn-vulkan_amd/librife_hip_test.so RIFE_HIP_DEBUG_V2=1 timeout 300 python tools/layer_trace.py run --workload v23-1080p --parts 1 --pairs 1 2>&1 | sort | uniq -c | head -20 > $OUT/debug_v2.txt
(time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $OUT/pytest_all.txt 2>&1
cat $OUT/debug_v2.txt; cat $OUT/pytest_all.txt
