#!/bin/bash
# round 5, GPU call 22: the first flow update merged into block 1's stem + one two-update pass: bit-identity tests, v4 suite, A/B at 4K and 1080p
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call22
mkdir -p $OUT
cd $ROOT
TL=$ROOT/rife-ncnn-vulkan_amd/librife_hip_test.so
timeout 900 python -m pytest tests/test_gpu_v4.py tests/test_gpu_gather.py tests/test_gpu_stream_mode.py -x -q -m gpu -k "not 8k" 2>&1 | tail -8 > $OUT/pytest.txt
B="--steps 30 --no-cpu-baseline --no-host-path --no-live-traffic --no-extra --no-configs"
for rep in 1 2 3; do
for wl in 4k 1080p; do
RIFE_HIP_LIB=$TL RIFE_HIP_MERGE_FLOW0=0 timeout 300 python bench.py $B --workload $wl > $OUT/${wl}_sep_$rep.json 2>> $OUT/err.txt
RIFE_HIP_LIB=$TL timeout 300 python bench.py $B --workload $wl > $OUT/${wl}_merged_$rep.json 2>> $OUT/err.txt
done
done
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'])
PY
done > $OUT/summary.txt 2>&1
cat $OUT/pytest.txt $OUT/summary.txt
