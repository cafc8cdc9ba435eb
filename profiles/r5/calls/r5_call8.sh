#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call8
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_v2.py tests/test_gpu_v3.py tests/test_gpu_ref_fixtures.py tests/test_gpu_vs_ref_build.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest_v2.txt
RIFE_HIP_CTX0_IMG=0 timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_a.txt 2>&1
timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_b.txt 2>&1
