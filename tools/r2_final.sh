#!/bin/bash
# GPU box: the round's closing call - full GPU suite, smoke(), the bench lines of every workload (into gpurun_out/final/)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/final; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 | tee $OUT/smoke.txt
for wl in 4k 1080p v23-1080p 4k-tta; do
    timeout 600 python bench.py --workload $wl --steps 50 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
    tail -c 300 $OUT/bench_$wl.err
done
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
for wl in ("4k","1080p","v23-1080p","4k-tta","default"):
    try:
        d=json.load(open("$OUT/bench_%s.json"%wl)); print(wl, d["value"], d["unit"], d.get("roofline",{}).get("frac"), d["extra"].get("frames_per_s_host_buffers"))
    except Exception as e: print(wl, "failed", e)
PY
