#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call4
mkdir -p $OUT
cd $ROOT
RIFE_HIP_NS3_ROWS4=0 timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_a.txt 2>&1
RIFE_HIP_NS3_ROWS4=1 timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_b.txt 2>&1
RIFE_HIP_NS3_ROWS4=1 RIFE_HIP_ROWS4_MAX=100000 timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_c.txt 2>&1
RIFE_HIP_NS3_ROWS4=0 RIFE_HIP_ROWS4_MAX=0 timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_d.txt 2>&1
