#!/bin/bash
# GPU box, round 3 call 5: run-to-run stability of the stem kernels under both flag sets (old / new staging), tests that touch the stems
mkdir -p gpurun_out
echo "== stem_det_both"; timeout 900 python tools/stem_det_both.py 200 > gpurun_out/stem_det_both_stdout.txt 2>&1; echo "rc=$?"; tail -20 gpurun_out/stem_det_both_stdout.txt
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_gather.py tests/test_gpu_kernels.py tests/test_gpu_v4.py -q -m gpu > gpurun_out/pytest_stems.txt 2>&1; echo "rc=$?"; tail -6 gpurun_out/pytest_stems.txt
