#!/bin/bash
# round 5, GPU call 16 (bench only): N-tile width of the 192-channel trunk, split-K settings of the small-grid trunks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call16
mkdir -p $OUT
cd $ROOT
TL=$ROOT/rife-ncnn-vulkan_amd/librife_hip_test.so
B="--steps 40 --no-cpu-baseline --no-host-path --no-live-traffic --no-extra --no-configs"
run() { name=$1; shift; env "$@" RIFE_HIP_LIB=$TL timeout 300 python bench.py $B --workload v23-1080p > $OUT/v23_$name.json 2>> $OUT/err.txt; }
for rep in 1 2; do
run base_$rep RIFE_HIP_X=0
run nt96_$rep RIFE_HIP_NT192_96=1
run nt96r4_$rep RIFE_HIP_NT192_96=1 RIFE_HIP_NS3_ROWS4=1
run sk2_$rep RIFE_HIP_SPLITK_N=2
run sk0_$rep RIFE_HIP_SPLITK_NB=0
run sk200_$rep RIFE_HIP_SPLITK_NB=200 RIFE_HIP_SPLITK_N=2
run sk8_$rep RIFE_HIP_SPLITK_N=8
done
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'])
PY
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
