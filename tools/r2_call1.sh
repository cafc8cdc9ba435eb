#!/bin/bash
# GPU box, round 2 call 1: correctness of the S16 / conv_t64 trunk path + A/B against the per-tile trunk.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2c1
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_t64.py -x -q > $OUT/pytest_t64.log 2>&1; echo "pytest_t64 rc=$?" | tee -a $OUT/summary.txt
tail -5 $OUT/pytest_t64.log
timeout 600 python -m pytest tests/test_gpu_v4.py tests/test_gpu_kernels.py -x -q > $OUT/pytest_v4.log 2>&1; echo "pytest_v4 rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/pytest_v4.log
for t in 1 0 1 0; do
  RIFE_HIP_T64=$t timeout 300 python bench.py --workload 4k --steps 40 --no-cpu-baseline > $OUT/bench_4k_t64_$t.json 2> $OUT/bench_4k_t64_$t.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_4k_t64_$t.json"))
    print("T64=$t 4k fps", d["value"], "1-in-flight", d["extra"]["frames_per_s_with_1_pair_in_flight"], "dom", d["roofline"]["avg_launch_ms"], {k:v for k,v in list(d["extra"]["per_class_ms_per_pair"].items())[:6]})
except Exception as e: print("bench T64=$t failed", e)
PY
done 2>&1 | tee -a $OUT/summary.txt
RIFE_HIP_T64=1 timeout 300 python bench.py --workload 1080p --steps 60 --no-cpu-baseline > $OUT/bench_1080p_t64_1.json 2>/dev/null
RIFE_HIP_T64=0 timeout 300 python bench.py --workload 1080p --steps 60 --no-cpu-baseline > $OUT/bench_1080p_t64_0.json 2>/dev/null
python - <<PY | tee -a $OUT/summary.txt
import json
for t in (1,0):
    try:
        d=json.load(open("$OUT/bench_1080p_t64_%d.json"%t)); print("T64=%d 1080p fps"%t, d["value"], d["extra"]["frames_per_s_with_1_pair_in_flight"], d["roofline"]["avg_launch_ms"])
    except Exception as e: print("1080p fail", t, e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python $ROOT/tools/prof_run.py --workload 4k --pairs 8 > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats_4k.csv 2>/dev/null
head -12 $OUT/kernel_stats_4k.csv | cut -c1-200 | tee -a $OUT/summary.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python $ROOT/tools/prof_run.py --workload 4k --pairs 3 > $OUT/pmc1.log 2>&1
f=$(find $OUT/pmc1 -name '*counter_collection.csv' | head -1)
python $ROOT/tools/pmc_summary.py $f "conv_t64" > $OUT/pmc_t64.txt 2>&1
cat $OUT/pmc_t64.txt | head -30 | tee -a $OUT/summary.txt
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$pass -- python $ROOT/tools/prof_run.py --workload 4k --pairs 3 > $OUT/pmc_$pass.log 2>&1
  f=$(find $OUT/pmc_$pass -name '*counter_collection.csv' | head -1)
  python $ROOT/tools/pmc_summary.py $f > $OUT/pmc_${pass}_all.txt 2>&1
done
rm -rf $OUT/kt $OUT/pmc1 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
grep -i "t64\|h2b" $OUT/pmc_FETCH_SIZE_all.txt $OUT/pmc_WRITE_SIZE_all.txt | head | tee -a $OUT/summary.txt
