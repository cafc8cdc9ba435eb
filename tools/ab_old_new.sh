#!/bin/bash
# A/B of two builds of librife_hip.so inside ONE gpurun call (boxes differ by up to 10 %, so only same-run comparisons count):
#   build the old tree, cp rife-ncnn-vulkan_amd/librife_hip.so rife-ncnn-vulkan_amd/librife_hip_old.so, build the new tree, then
#   gpurun -- 'bash tools/ab_old_new.sh'
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_v4.py tests/test_gpu_v2.py tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -2
for rep in 1 2; do
  for v in new old; do
    if [ $v = old ]; then cp rife-ncnn-vulkan_amd/librife_hip.so /tmp/new.so; cp rife-ncnn-vulkan_amd/librife_hip_old.so rife-ncnn-vulkan_amd/librife_hip.so; fi
    python bench.py --no-cpu-baseline --steps 30 | python -c "
import json,sys; d=json.load(sys.stdin); e=d['extra']['per_class_ms_per_pair']; print('$v', d['value'], d['extra']['frames_per_s_with_1_pair_in_flight'], {k:e[k] for k in ('trunk_b3','trunk_b2','stem1_b3','stem1_b2')})"
    if [ $v = old ]; then cp /tmp/new.so rife-ncnn-vulkan_amd/librife_hip.so; fi
  done
done
