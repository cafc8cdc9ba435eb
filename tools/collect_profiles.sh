#!/bin/bash
# build container: copy what tools/profile_round.sh <tag> and tools/r2_final.sh left under gpurun_out/ into profiles/<tag>/
TAG=${1:-r2}; SRC=gpurun_out/profile_$TAG; DST=profiles/$TAG
mkdir -p $DST
cp $SRC/kernel_stats_4k.csv $SRC/kernel_stats_1080p.csv $SRC/t64_bench.txt $SRC/host_path.txt $DST/
cp $SRC/tables/* $DST/
sed -i "s#\"source\": \"[^\"]*tables/#\"source\": \"$DST/#" $DST/pmc_4k.json
for wl in 4k 1080p v23-1080p 4k-tta; do cp gpurun_out/final/bench_$wl.json $DST/bench_${wl//-/_}.json; done
cp gpurun_out/final/pytest_gpu.txt $DST/pytest_gpu.txt
ls $DST
