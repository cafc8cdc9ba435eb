#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call3
mkdir -p $OUT
cd $ROOT
RIFE_HIP_V2_STEM16=0 timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_base.txt 2>&1
timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_stem16.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_v2.py tests/test_gpu_v3.py -x -q -m gpu 2>&1 | tail -3 > $OUT/pytest_v2.txt
