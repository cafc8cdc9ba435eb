#!/bin/bash
# round 5, GPU call 19: the parity report of every BASELINE config (tools/parity_report.py --full) and the parity-margin tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_call19
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_margin.py -x -q -m gpu -s 2>&1 | grep -E "margin:|passed|failed|Error" > $OUT/margin.txt
(time timeout 2400 python tools/parity_report.py --full) > $OUT/parity_report.txt 2> $OUT/parity_report.err
cat $OUT/margin.txt; tail -40 $OUT/parity_report.txt
