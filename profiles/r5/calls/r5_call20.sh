#!/bin/bash
# round 5, GPU call 20 (bench only): -x -z at 4K with 1 - 4 pairs in flight
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call20
mkdir -p $OUT
cd $ROOT
B="--workload 4k-tta --steps 6 --warmup 1 --no-cpu-baseline --no-host-path --no-live-traffic --no-extra --no-configs"
for s in 2 1 3 4 2; do
timeout 300 python bench.py $B --streams $s > $OUT/tta_s${s}_$RANDOM.json 2>> $OUT/err.txt
done
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('pairs_in_flight_per_gpu'))
PY
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt; tail -3 $OUT/err.txt
