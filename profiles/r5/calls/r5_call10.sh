#!/bin/bash
# round 5, GPU call 10: fused v2 stems (stem_fused_v2.h) - parity tests, A/B against the unfused pair, stream layouts
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call10
mkdir -p $OUT
cd $ROOT
timeout 700 python -m pytest tests/test_gpu_v2.py tests/test_gpu_v3.py tests/test_gpu_ref_fixtures.py tests/test_gpu_vs_ref_build.py tests/test_gpu_edge_sizes.py -x -q -m gpu 2>&1 | tail -15 > $OUT/pytest_v2.txt
B="--workload v23-1080p --steps 40 --no-cpu-baseline --no-host-path --no-live-traffic --no-extra"
for rep in 1 2; do
RIFE_HIP_V2_FUSED_STEM=0 timeout 300 python bench.py $B > $OUT/bench_unfused_$rep.json 2> $OUT/err.txt
timeout 300 python bench.py $B > $OUT/bench_fused_$rep.json 2>> $OUT/err.txt
done
timeout 300 python bench.py $B --streams 8 --cu-parts 4 > $OUT/bench_fused_s8p4.json 2>> $OUT/err.txt
timeout 300 python bench.py $B --streams 6 --cu-parts 2 > $OUT/bench_fused_s6p2.json 2>> $OUT/err.txt
timeout 300 python bench.py $B --streams 8 --cu-parts 8 > $OUT/bench_fused_s8p8.json 2>> $OUT/err.txt
timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23.txt 2>&1
for f in $OUT/bench_*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('pairs_in_flight_per_gpu'), d['config'].get('cu_partition'))
PY
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
