#!/bin/bash
# GPU box: flow updates fused into the stems of blocks 2 / 3 - gather tests, v4 tests, same-call A/B of the 4K and 1080p bench lines
mkdir -p gpurun_out
echo "== pytest gather"; timeout 900 python -m pytest tests/test_gpu_gather.py -q -m gpu -x > gpurun_out/pytest_gather.txt 2>&1; echo "rc=$?"; tail -6 gpurun_out/pytest_gather.txt
echo "== bench A/B fused flow update, 4K"
for ff in 1 0 1 0; do RIFE_HIP_FUSE_FLOW=$ff timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['extra']['per_class_ms_per_pair']; print('FF=$ff', d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'], {k: c[k] for k in c if 'stem' in k or 'flow' in k or 'final' in k or 'head' in k})"; done
echo "== 1080p"
for ff in 1 0; do RIFE_HIP_FUSE_FLOW=$ff timeout 300 python bench.py --workload 1080p --steps 60 --warmup 10 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FF=$ff', d['value'], d['ms_per_step'])"; done
echo "== pytest v4"; timeout 1200 python -m pytest tests/test_gpu_v4.py tests/test_gpu_t64.py -q -m gpu -x > gpurun_out/pytest_v4.txt 2>&1; echo "rc=$?"; tail -6 gpurun_out/pytest_v4.txt
