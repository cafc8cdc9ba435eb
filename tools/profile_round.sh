#!/bin/bash
# One-shot profile refresh on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <round-tag>      -> gpurun_out/profile_<tag>/...
# 1) rocprofv3 --kernel-trace --stats of the bench workload (4K, rife-v4.6, one pair in flight)
# 2) three separate PMC passes (matrix pipe / LDS, FETCH_SIZE, WRITE_SIZE) of the same workload
# 3) bench.py JSON lines for every workload
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $ROOT/tools/prof_run.py --workload 4k --pairs 8 > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_4k.csv 2>/dev/null
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python $ROOT/tools/prof_run.py --workload 4k --pairs 3 > $OUT/pmc_$name.log 2>&1
    f=$(find $OUT/pmc_$name -name '*counter_collection.csv' | head -1)
    python $ROOT/tools/pmc_summary.py $f "conv_h2b_kernel<2, 10, 3, 8>" > $OUT/pmc_${name}_trunk_b3.txt 2>&1
    python $ROOT/tools/pmc_summary.py $f > $OUT/pmc_${name}_all.txt 2>&1
done
# keep only the small summaries (gpurun_out is capped at 64 MiB)
rm -rf $OUT/kt $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
cd $ROOT
for wl in 4k 1080p v23-1080p 4k-tta; do
    python bench.py --workload $wl > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
ls -la $OUT
