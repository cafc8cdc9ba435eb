#!/bin/bash
# GPU box: stem_rs_kernel - parity against the two-kernel sequence and the oracle, the other v4 suites, same-call A/B of the 4K and 1080p bench lines
mkdir -p gpurun_out
echo "== pytest"; timeout 1500 python -m pytest tests/test_gpu_stem_rs.py tests/test_gpu_v4.py tests/test_gpu_gather.py tests/test_gpu_t64.py tests/test_gpu_ref_fixtures.py -q -m gpu -x > gpurun_out/pytest_stem_rs.txt 2>&1; echo "rc=$?"; tail -6 gpurun_out/pytest_stem_rs.txt
echo "== bench A/B, 4K"
for v in 1 0 1 0; do RIFE_HIP_STEM_RS=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['extra']['per_class_ms_per_pair']; print('STEM_RS=$v', d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'], {k: c[k] for k in c if 'stem' in k})"; done
echo "== 1080p"
for v in 1 0 1 0; do RIFE_HIP_STEM_RS=$v timeout 300 python bench.py --workload 1080p --steps 60 --warmup 10 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['extra']['per_class_ms_per_pair']; print('STEM_RS=$v', d['value'], d['ms_per_step'], {k: c[k] for k in c if 'stem' in k})"; done
timeout 300 python tools/stem_rs_bench.py > gpurun_out/stem_rs_bench.txt 2>&1; head -14 gpurun_out/stem_rs_bench.txt
