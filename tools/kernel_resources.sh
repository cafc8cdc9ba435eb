#!/bin/bash
# VGPRs / spills / LDS of the device kernels of engine.hip (device-only compile, no GPU needed):  tools/kernel_resources.sh [name filter]
set -e
cd "$(dirname "$0")/../rife-ncnn-vulkan_amd/csrc"
T=${TMPDIR:-/tmp}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fno-vectorize --cuda-device-only -c engine.hip -o $T/rife_dev.o
(cd $T && /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=rife_dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=rife_dev.co)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/rife_dev.co | grep -E "\.name:|\.vgpr_count|vgpr_spill|\.group_segment_fixed|private_segment_fixed" | paste - - - - - | sed 's/  */ /g' | grep -i "${1:-.}" | c++filt
