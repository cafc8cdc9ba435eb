cd $GRAFT_REPO_ROOT
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power|Temperature \(Sensor junction\)|Sensor edge" | tr '\n' ' '; echo; sleep 0.5; done) > gpurun_out/smi_during_bench.txt &
SMI=$!
sleep 1
python bench.py --no-cpu-baseline --steps 400 --no-extra > gpurun_out/bench_long.json 2>/dev/null
kill $SMI 2>/dev/null
head -30 gpurun_out/smi_during_bench.txt
python -c "import json; d=json.load(open('gpurun_out/bench_long.json')); print(d['value'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
