#!/bin/bash
# GPU box: A/B of an engine create-time switch inside one call:  bash tools/r2_ab_env.sh VAR [workloads...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
VAR=$1; shift
WLS=${@:-4k 1080p}
OUT=gpurun_out/ab_$VAR; mkdir -p $OUT
for rep in 1 2; do
for V in ${VALUES:-1 0}; do
  for WL in $WLS; do
    env $VAR=$V timeout 300 python bench.py --workload $WL --steps 40 --no-cpu-baseline --no-host-path > $OUT/bench_${WL}_$V.json 2>/dev/null
    python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${WL}_$V.json"))
    pc=d["extra"]["per_class_ms_per_pair"]
    print("$VAR=$V $WL fps", d["value"], "median", d["extra"]["frames_per_s_repeated_regions"]["median"], "1-in-flight", d["extra"]["frames_per_s_with_1_pair_in_flight"], {k:pc[k] for k in ("trunk_b3","trunk_b2","trunk_b1","trunk_b0","stem1_b1","stem1_b0","head_b1","head_b0") if k in pc})
except Exception as e: print("bench $VAR=$V $WL failed", e)
PY
  done
done
done
