#!/bin/bash
# round 5, GPU call 12: after the static-initialiser fix (named env helpers): fused v2 stems really on - parity tests + A/B; hipGraph replay really switchable - 1080p A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call12
mkdir -p $OUT
cd $ROOT
TL=$ROOT/rife-ncnn-vulkan_amd/librife_hip_test.so
timeout 900 python -m pytest tests/test_gpu_v2.py tests/test_gpu_v3.py tests/test_gpu_ref_fixtures.py tests/test_gpu_vs_ref_build.py tests/test_gpu_edge_sizes.py tests/test_cli.py tests/test_gpu_stream_mode.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest_v2.txt
B="--steps 40 --no-cpu-baseline --no-host-path --no-live-traffic --no-extra --no-configs"
for rep in 1 2; do
RIFE_HIP_LIB=$TL RIFE_HIP_V2_FUSED_STEM=0 timeout 300 python bench.py --workload v23-1080p $B > $OUT/v23_unfused_$rep.json 2>> $OUT/err.txt
RIFE_HIP_LIB=$TL timeout 300 python bench.py --workload v23-1080p $B > $OUT/v23_fused_$rep.json 2>> $OUT/err.txt
timeout 300 python bench.py --workload 1080p $B > $OUT/1080p_nograph_$rep.json 2>> $OUT/err.txt
RIFE_HIP_GRAPH=1 timeout 300 python bench.py --workload 1080p $B > $OUT/1080p_graph_$rep.json 2>> $OUT/err.txt
done
timeout 300 python bench.py --workload v23-1080p $B > $OUT/v23_product.json 2>> $OUT/err.txt
timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23.txt 2>&1
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'])
PY
done > $OUT/summary.txt 2>&1
cat $OUT/pytest_v2.txt $OUT/summary.txt
