// translation unit of tools/probes/stem_bisect.py: the two fused stem kernels that were not run-to-run stable when built with the SLP vectorizer
#include "../../rife-ncnn-vulkan_amd/csrc/stem_fused.h"
template __global__ void rife::stem0_fused_kernel<4, 2, 0>(rife::StemFusedArgs);
template __global__ void rife::stem0_fused_kernel<2, 2, 0>(rife::StemFusedArgs);
template __global__ void rife::stem0_fused_kernel<4, 2, 4096>(rife::StemFusedArgs);      // round-2 staging (lane-divergent second-pixel store)
template __global__ void rife::stem0_fused_kernel<2, 2, 4096>(rife::StemFusedArgs);
