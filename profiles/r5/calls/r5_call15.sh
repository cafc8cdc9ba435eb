#!/bin/bash
# round 5, GPU call 15: two-destination skip stores + one-launch ContextNet warps: parity tests, A/B, part profile
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call15
mkdir -p $OUT
cd $ROOT
TL=$ROOT/rife-ncnn-vulkan_amd/librife_hip_test.so
timeout 900 python -m pytest tests/test_gpu_v2.py tests/test_gpu_v3.py tests/test_gpu_ref_fixtures.py tests/test_gpu_vs_ref_build.py tests/test_gpu_edge_sizes.py tests/test_gpu_stream_mode.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest.txt
B="--steps 40 --no-cpu-baseline --no-host-path --no-live-traffic --no-extra --no-configs"
run() { name=$1; shift; env "$@" RIFE_HIP_LIB=$TL timeout 300 python bench.py $B --workload v23-1080p > $OUT/v23_$name.json 2>> $OUT/err.txt; }
for rep in 1 2 3; do
run old_$rep RIFE_HIP_V2_SKIP_COPY=1 RIFE_HIP_V2_CTX_BATCH=0
run copy_$rep RIFE_HIP_V2_SKIP_COPY=1
run new_$rep RIFE_HIP_V2_SKIP_COPY=0
done
timeout 300 python bench.py $B --workload v23-1080p > $OUT/v23_product.json 2>> $OUT/err.txt
timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23.txt 2>&1
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'])
PY
done > $OUT/summary.txt 2>&1
cat $OUT/pytest.txt $OUT/summary.txt
