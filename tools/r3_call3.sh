#!/bin/bash
# GPU box, round 3 call 3: gather parity taps, rs / t64 tests, the bench line with native-resolution frames, values of the stem instability
mkdir -p gpurun_out
echo "== pytest gather + t64"; timeout 900 python -m pytest tests/test_gpu_gather.py tests/test_gpu_t64.py -q -m gpu > gpurun_out/pytest_gather.txt 2>&1; echo "rc=$?"; tail -25 gpurun_out/pytest_gather.txt
echo "== bench 4k"; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_4k.json 2> gpurun_out/bench_4k.err; echo "rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_4k.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["frames"], (d.get("roofline") or {}).get("avg_launch_ms"), (d.get("roofline") or {}).get("frac"))
print(d["extra"]["frames_per_s_host_buffers"]); print(d["extra"]["per_class_ms_per_pair"]); print(d["cpu_baseline"])
PY
echo "== stem values"; timeout 300 python tools/stem_bisect.py 200 values-only > gpurun_out/stem_values.txt 2>&1; echo "rc=$?"; tail -45 gpurun_out/stem_values.txt
echo "== rs_bench quick"; timeout 300 python tools/rs_bench.py quick > gpurun_out/rs_bench_quick.txt 2>&1; echo "rc=$?"; sed -n 12,40p gpurun_out/rs_bench_quick.txt
