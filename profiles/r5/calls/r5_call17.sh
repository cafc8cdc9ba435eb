#!/bin/bash
# round 5, GPU call 17 (bench only): stream layouts at 4K and 1080p after this round's changes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call17
mkdir -p $OUT
cd $ROOT
B="--steps 30 --no-cpu-baseline --no-host-path --no-live-traffic --no-extra --no-configs"
run() { name=$1; shift; timeout 300 python bench.py $B "$@" > $OUT/$name.json 2>> $OUT/err.txt; }
for rep in 1 2; do
run 4k_default_$rep --workload 4k
run 4k_s6p2_$rep --workload 4k --streams 6 --cu-parts 2
run 4k_s6p3_$rep --workload 4k --streams 6 --cu-parts 3
run 4k_s8p2_$rep --workload 4k --streams 8 --cu-parts 2
run 4k_s8p4_$rep --workload 4k --streams 8 --cu-parts 4
run 4k_s3p0_$rep --workload 4k --streams 3 --cu-parts 0
run 1080p_default_$rep --workload 1080p
run 1080p_s8p4_$rep --workload 1080p --streams 8 --cu-parts 4
run 1080p_s6p0_$rep --workload 1080p --streams 6 --cu-parts 0
done
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'], d['config'].get('pairs_in_flight_per_gpu'), d['config'].get('cu_partition'))
PY
done > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
