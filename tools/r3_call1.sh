#!/bin/bash
# GPU box, round 3 call 1: the row-streaming trunk kernel (parity with conv_t64, timing, ablations), its pytest, an engine-level A/B, the stem bisect
mkdir -p gpurun_out
echo "== rs_bench"; timeout 300 python tools/rs_bench.py > gpurun_out/rs_bench.txt 2>&1; echo "rc=$?"; tail -70 gpurun_out/rs_bench.txt
echo "== pytest t64/rs"; timeout 600 python -m pytest tests/test_gpu_t64.py -x -q -m gpu > gpurun_out/pytest_t64.txt 2>&1; echo "rc=$?"; tail -15 gpurun_out/pytest_t64.txt
echo "== bench A/B"
for rs in 1 0 1 0; do RIFE_HIP_RS=$rs timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RS=$rs', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_ms'))"; done
echo "== stem bisect"; timeout 420 python tools/stem_bisect.py 330 > gpurun_out/stem_bisect_stdout.txt 2>&1; echo "rc=$?"; tail -40 gpurun_out/stem_bisect_stdout.txt
