#!/bin/bash
# One-shot profile refresh on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <round-tag>      -> gpurun_out/profile_<tag>/...   (copy what is to be judged into profiles/<tag>/)
# 1) rocprofv3 --kernel-trace --stats of the bench workload (4K, rife-v4.6, one pair in flight)
# 2) separate PMC passes of the same workload: matrix pipe / LDS / wave states, FETCH_SIZE, WRITE_SIZE (never combined with other traces)
# 3) ablation timings and clock stamps of the dominant kernel (tools/rs_bench.py) and of the block-3 stem kernel (tools/stem_rs_bench.py)
# 4) bench.py JSON lines for every workload, the host-path sweep
# (tools/collect_profiles.sh <tag> copies the summaries into profiles/<tag>/; tools/round_close.sh runs the GPU suite + smoke())
set -u
TAG=${1:-r4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profile_$TAG
DOM=${DOM:-conv_rs2_kernel}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $ROOT/tools/prof_run.py --workload 4k --pairs 8 > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_4k.csv 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1080 -- python $ROOT/tools/prof_run.py --workload 1080p --pairs 8 > $OUT/kt1080.log 2>&1
cp $(find $OUT/kt1080 -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_1080p.csv 2>/dev/null
for wl in v23-1080p 4k-tta; do      # BASELINE configs 2 and 5
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$wl -- python $ROOT/tools/prof_run.py --workload $wl --pairs 4 > $OUT/kt_$wl.log 2>&1
    cp $(find $OUT/kt_$wl -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_${wl//-/_}.csv 2>/dev/null
    rm -rf $OUT/kt_$wl
done
PASSES=("SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
# the four counter passes for EVERY bench workload (round 4: BASELINE configs 2, 3 and 5 too); counters alone, one pass per rocprofv3 run
for wl in 4k 1080p v23-1080p 4k-tta; do
    sfx=""; [ $wl != 4k ] && sfx="_${wl//-/_}"
    np=3; [ $wl = 4k-tta ] && np=1
    for pass in "${PASSES[@]}"; do
        name=$(echo $pass | cut -d' ' -f1)
        timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name$sfx -- python $ROOT/tools/prof_run.py --workload $wl --pairs $np > $OUT/pmc_$name$sfx.log 2>&1
        f=$(find $OUT/pmc_$name$sfx -name '*counter_collection.csv' | head -1)
        [ $wl = 4k ] && python $ROOT/tools/pmc_summary.py $f "$DOM" > $OUT/pmc_${name}_trunk_b3.txt 2>&1
        python $ROOT/tools/pmc_summary.py $f > $OUT/pmc_${name}${sfx}_all.txt 2>&1
        rm -rf $OUT/pmc_$name$sfx        # keep only the small summaries (gpurun_out is capped at 64 MiB)
    done
done
rm -rf $OUT/kt $OUT/kt1080
cd $ROOT
timeout 400 python tools/rs_bench.py > $OUT/rs_bench.txt 2>&1
timeout 400 python tools/rs2_bench.py > $OUT/rs2_bench.txt 2>&1
timeout 600 python tools/power_workloads.py 8 > $OUT/power_workloads.txt 2>&1
timeout 300 python tools/ks_bench.py > $OUT/ks_bench.txt 2>&1
timeout 300 python tools/stem_rs_bench.py > $OUT/stem_rs_bench.txt 2>&1
timeout 300 python tools/tail_rs_bench.py > $OUT/tail_rs_bench.txt 2>&1

for wl in 4k 1080p v23-1080p 4k-tta; do
    timeout 900 python bench.py --workload $wl --steps 50 > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
done
timeout 300 python tools/host_path_bench2.py > $OUT/host_path.txt 2>&1
python tools/pmc_tables.py $OUT $OUT/tables 3 4k > $OUT/tables.log 2>&1
python tools/pmc_tables.py $OUT $OUT/tables 3 1080p >> $OUT/tables.log 2>&1
python tools/pmc_tables.py $OUT $OUT/tables 3 v23_1080p >> $OUT/tables.log 2>&1
python tools/pmc_tables.py $OUT $OUT/tables 1 4k_tta >> $OUT/tables.log 2>&1
ls -la $OUT
