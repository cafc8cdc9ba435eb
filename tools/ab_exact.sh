#!/bin/bash
# GPU box: are the frames of library build rife-ncnn-vulkan_amd/alt/lib_<V>.so and of the build in place byte-identical?  bash tools/ab_exact.sh <V>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
mkdir -p gpurun_out
cp rife-ncnn-vulkan_amd/librife_hip.so /tmp/lib_keep.so
timeout 600 python tools/dump_outputs.py /tmp/out_new.npz | tail -1
cp rife-ncnn-vulkan_amd/alt/lib_$1.so rife-ncnn-vulkan_amd/librife_hip.so
timeout 600 python tools/dump_outputs.py /tmp/out_alt.npz | tail -1
cp /tmp/lib_keep.so rife-ncnn-vulkan_amd/librife_hip.so
python - <<PY
import numpy as np
a, b = np.load("/tmp/out_new.npz"), np.load("/tmp/out_alt.npz")
bad = 0
for k in a.files:
    d = np.abs(a[k].astype(int) - b[k].astype(int))
    print("%-16s differing bytes %d of %d, max %d" % (k, int((d > 0).sum()), d.size, int(d.max())))
    bad += int((d > 0).sum())
print("IDENTICAL" if bad == 0 else "DIFFERENT")
PY
