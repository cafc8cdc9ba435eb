#!/bin/bash
# build container: copy what tools/profile_round.sh <tag> left under gpurun_out/profile_<tag>/ into profiles/<tag>/
TAG=${1:-r4}; SRC=gpurun_out/profile_$TAG; DST=profiles/$TAG
mkdir -p $DST
for f in kernel_stats_4k.csv kernel_stats_1080p.csv kernel_stats_v23_1080p.csv kernel_stats_4k_tta.csv smoke.txt bench_default.json rs_bench.txt rs2_bench.txt power_workloads.txt ks_bench.txt stem_rs_bench.txt tail_rs_bench.txt t64_bench.txt host_path.txt pytest_gpu.txt; do [ -f $SRC/$f ] && cp $SRC/$f $DST/; done
cp $SRC/tables/* $DST/
for j in $DST/pmc_*.json; do sed -i "s#\"source\": \"[^\"]*/\(pmc_trunk_kernels_[a-z0-9_]*\.txt\)#\"source\": \"$DST/\1#" $j; done      # the GPU box's scratch path -> the committed one
for wl in 4k 1080p v23-1080p 4k-tta; do [ -f $SRC/bench_$wl.json ] && cp $SRC/bench_$wl.json $DST/bench_${wl//-/_}.json; done
ls $DST
