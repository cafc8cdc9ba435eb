#!/bin/bash
# round 5, GPU call 21: the round-5 v2 kernels at 4K / 1440p / UHD-mode sizes (tests/test_gpu_v2.py incl. the new large-grid cases), v3.1
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call21
mkdir -p $OUT
cd $ROOT
(time timeout 1200 python -m pytest tests/test_gpu_v2.py tests/test_gpu_v3.py -x -q -m gpu 2>&1 | tail -12) > $OUT/pytest.txt 2>&1
cat $OUT/pytest.txt
