#!/bin/bash
# GPU box: a round's closing call - the full GPU suite and smoke() on the final commit (into gpurun_out/profile_<tag>/)
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
OUT=gpurun_out/profile_$TAG; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.json
