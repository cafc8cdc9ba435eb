#!/bin/bash
# round 5, GPU call 1: per-layer timelines of rife-v2.3 1080p and rife-v4.6 4K on the whole chip and on one part of it + same-box baselines
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "v23-1080p 1" "v23-1080p 4" "4k 1" "4k 2" "1080p 4"; do
    set -- $cfg
    d=$OUT/lt_$1_p$2
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -- python $ROOT/tools/layer_trace.py run --workload $1 --parts $2 --pairs 3 > $d.log 2>&1
    f=$(find $d -name '*kernel_trace.csv' | head -1)
    python $ROOT/tools/layer_trace.py sum $f --pairs 3 > $OUT/layers_$1_p$2.txt 2>&1
    rm -rf $d
done
cd $ROOT
timeout 600 python bench.py --workload v23-1080p --steps 40 --no-cpu-baseline --no-host-path --no-live-traffic > $OUT/bench_v23.json 2> $OUT/bench_v23.err
timeout 600 python bench.py --workload 4k --steps 30 --no-cpu-baseline --no-host-path --no-live-traffic > $OUT/bench_4k.json 2> $OUT/bench_4k.err
ls -la $OUT
