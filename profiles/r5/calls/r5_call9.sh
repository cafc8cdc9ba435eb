#!/bin/bash
# round 5, GPU call 9 (first of the re-entered session): validate HEAD on the v2 family, per-layer profile + per-dispatch trace of rife-v2.3 1080p
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call9
mkdir -p $OUT
cd $ROOT
timeout 700 python -m pytest tests/test_gpu_v2.py tests/test_gpu_v3.py tests/test_gpu_ref_fixtures.py -x -q -m gpu 2>&1 | tail -5 > $OUT/pytest_v2.txt
timeout 300 python tools/part_profile.py --workload v23-1080p --parts 4 --pairs 12 > $OUT/part_v23.txt 2>&1
cd /tmp && export TMPDIR=/tmp
d=$OUT/lt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python $ROOT/tools/layer_trace.py run --workload v23-1080p --parts 1 --pairs 3 > $d.log 2>&1
f=$(find $d -name '*kernel_trace.csv' | head -1)
python $ROOT/tools/layer_trace.py sum $f --pairs 3 > $OUT/layers_v23_p1.txt 2>&1
rm -rf $d
cd $ROOT
timeout 300 python bench.py --workload v23-1080p --steps 40 --no-cpu-baseline --no-host-path --no-live-traffic > $OUT/bench_v23.json 2> $OUT/bench_v23.err
ls -la $OUT
