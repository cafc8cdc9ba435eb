#!/bin/bash
# round 5, GPU call 18: per-class profile of the 4K and 1080p v4.6 workloads in their bench stream layouts
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call18
mkdir -p $OUT
cd $ROOT
timeout 300 python tools/part_profile.py --workload 4k --parts 2 --per-part 2 --pairs 10 > $OUT/part_4k.txt 2>&1
timeout 300 python tools/part_profile.py --workload 1080p --parts 4 --pairs 16 > $OUT/part_1080p.txt 2>&1
cat $OUT/part_4k.txt
