#!/bin/bash
# GPU box: loader waves in conv_t64 for block 2 (A/B), t64 tests, parity report at HEAD, rocprof kernel stats of the v2.3 and -x -z workloads
mkdir -p gpurun_out
echo "== pytest t64"; timeout 600 python -m pytest tests/test_gpu_t64.py -q -m gpu > gpurun_out/pytest_t64.txt 2>&1; echo "rc=$?"; tail -4 gpurun_out/pytest_t64.txt
echo "== bench A/B loader waves"
for lw in 1 0 1 0; do RIFE_HIP_T64_LW=$lw timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-host-path 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LW=$lw', d['value'], d['ms_per_step'], d['extra']['frames_per_s_repeated_regions']['median'], 'trunk_b2', d['extra']['per_class_ms_per_pair'].get('trunk_b2'), 'trunk_b3', d['extra']['per_class_ms_per_pair'].get('trunk_b3'))"; done
echo "== parity report"; timeout 1500 python tools/parity_report.py > gpurun_out/parity_report.txt 2>&1; echo "rc=$?"; tail -30 gpurun_out/parity_report.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for wl in v23-1080p 4k-tta; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$wl -- python $R/tools/prof_run.py --workload $wl --pairs 4 > $R/gpurun_out/kt_$wl.log 2>&1
  cp $(find $R/gpurun_out/kt_$wl -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kernel_stats_$wl.csv 2>/dev/null
  rm -rf $R/gpurun_out/kt_$wl
  head -12 $R/gpurun_out/kernel_stats_$wl.csv | cut -c1-120
done
