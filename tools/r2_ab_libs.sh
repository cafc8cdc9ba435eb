#!/bin/bash
# GPU box: end-to-end A/B of library variants kept under rife-ncnn-vulkan_amd/alt/lib_<V>.so (boxes differ by up to 10 %: compare inside one call)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/ab_libs; mkdir -p $OUT
cp rife-ncnn-vulkan_amd/librife_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for V in "$@"; do
  T=1; L=$V
  if [ "$V" = "old" ]; then T=0; L=$1; fi
  cp rife-ncnn-vulkan_amd/alt/lib_$L.so rife-ncnn-vulkan_amd/librife_hip.so
  for WL in 4k 1080p; do
    RIFE_HIP_T64=$T timeout 300 python bench.py --workload $WL --steps 40 --no-cpu-baseline > $OUT/bench_${WL}_$V.json 2>/dev/null
    python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_${WL}_$V.json"))
    print("$V $WL fps", d["value"], "1-in-flight", d["extra"]["frames_per_s_with_1_pair_in_flight"], "dom ms", d["roofline"]["avg_launch_ms"], {k:v for k,v in list(d["extra"]["per_class_ms_per_pair"].items())[:4]})
except Exception as e: print("bench $V $WL failed", e)
PY
  done
done
done
cp /tmp/lib_keep.so rife-ncnn-vulkan_amd/librife_hip.so
