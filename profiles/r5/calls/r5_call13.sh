#!/bin/bash
# round 5, GPU call 13: conv_rowf_kernel (256 / 384 / 128 channels) - unit + end-to-end parity, A/B per channel count on v2.3 1080p
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call13
mkdir -p $OUT
cd $ROOT
TL=$ROOT/rife-ncnn-vulkan_amd/librife_hip_test.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_v2.py tests/test_gpu_v3.py tests/test_gpu_ref_fixtures.py tests/test_gpu_vs_ref_build.py tests/test_gpu_edge_sizes.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest.txt
B="--steps 40 --no-cpu-baseline --no-host-path --no-live-traffic --no-extra --no-configs --workload v23-1080p"
for m in 0 3 1 2 7 0 3 7; do
RIFE_HIP_LIB=$TL RIFE_HIP_V2_ROWF=$m timeout 300 python bench.py $B > $OUT/v23_rowf${m}_$RANDOM.json 2>> $OUT/err.txt
done
RIFE_HIP_LIB=$TL RIFE_HIP_V2_ROWF=0 timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_rowf0.txt 2>&1
RIFE_HIP_LIB=$TL RIFE_HIP_V2_ROWF=7 timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23_rowf7.txt 2>&1
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'])
PY
done > $OUT/summary.txt 2>&1
cat $OUT/pytest.txt $OUT/summary.txt
