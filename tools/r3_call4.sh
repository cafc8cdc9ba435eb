#!/bin/bash
# GPU box, round 3 call 4: conv_rs with three steps of load lookahead / cross-step prefetch, the register-poison probe of the stem instability, the full GPU suite
mkdir -p gpurun_out
echo "== rs_bench"; timeout 400 python tools/rs_bench.py > gpurun_out/rs_bench.txt 2>&1; echo "rc=$?"; sed -n 2,32p gpurun_out/rs_bench.txt; grep -n "clock probe\|199:" gpurun_out/rs_bench.txt
echo "== bench A/B"
for rs in 1 0 1 0; do RIFE_HIP_RS=$rs timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RS=$rs', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_ms'), d['extra']['frames_per_s_repeated_regions']['median'])"; done
echo "== stem poison"; timeout 400 python tools/stem_poison.py 300 > gpurun_out/stem_poison_stdout.txt 2>&1; echo "rc=$?"; tail -30 gpurun_out/stem_poison_stdout.txt
echo "== full GPU suite"; timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -8 gpurun_out/pytest_gpu.txt
