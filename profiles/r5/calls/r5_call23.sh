#!/bin/bash
# round 5, GPU call 23 (bench only): the two rife-v4.6 bench lines on the final commit (after the merged first flow update)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call23
mkdir -p $OUT
cd $ROOT
timeout 200 python bench.py --workload 4k --steps 50 --no-cpu-baseline --no-host-path --no-live-traffic --no-configs > $OUT/bench_4k_final.json 2>> $OUT/err.txt
timeout 200 python bench.py --workload 1080p --steps 50 --no-cpu-baseline --no-host-path --no-live-traffic --no-configs > $OUT/bench_1080p_final.json 2>> $OUT/err.txt
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'], d['roofline']['frac'], d['extra'].get('frames_per_s_with_1_pair_in_flight'))
PY
done
