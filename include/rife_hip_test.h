/* rife_hip_test.h - TEST SURFACE of librife_hip.so: stage taps and single-kernel entry points that exist for the parity tests (tests/test_gpu_*.py)
 * and tools/parity_report.py.  The product header include/rife_hip.h does not include this file, and a drop-in for src/rife.cpp needs none of it.
 * The symbols are exported by the same library because they must run the PRODUCT's kernels; tests/test_host_and_sharding.py checks that the library
 * exports exactly what the two headers declare. */
#ifndef RIFE_HIP_TEST_H
#define RIFE_HIP_TEST_H

#include "rife_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- stage taps for parity tests (the reference's Extractor can extract / inject `flow0..flow3`,
 * src/rife.cpp:2653-2669; these do the same for the v4 schedule).  Planar CHW fp32 host arrays. ------------- */
int rife_hip_v4_extract_flow(const rife_hip_t* r, const uint8_t* in0_rgb, const uint8_t* in1_rgb, int w, int h,
                             float timestep, int fi, const float* const* inject, int n_inject, float* out6chw);
/* shape of blob flow{fi} for frames of w x h: rife-v4.6 6 x hp/s x wp/s (PixelShuffle output, models/rife-v4.6/flownet.param:46),
 * rife-v4 5 x hp/2s x wp/2s (Deconvolution output, models/rife-v4/flownet.param:33); s = 8, 4, 2, 1. */
int rife_hip_v4_flow_dims(const rife_hip_t* r, int w, int h, int fi, int* channels, int* fh, int* fw);
/* Taps of the gather code on injected flows (rife-v4.6; parity tests of rife.Warp + Interp + Concat as the hot path runs them,
 * src/warp.cpp:96-168, models/rife-v4.6/flownet.param:52-62, 107-115, 160-165, 202-217).  what = 0: the 12-channel input of IFBlock b
 * (1..3) from the unfused assembly kernel; 1: the same tensor read back through the product's fused stem kernel (one-hot weights; values
 * to 2^-22 relative); both 12 x hp/S x wp/S, n_inject = b.  what = 2: blob out0 before the postproc, 3 x hp x wp, n_inject = 4.
 * what = 4 / 3: the running flow F (4 channels) and mask M that IFBlock b's stem reads, 5 x hp x wp, after the flow-update kernel / as
 * written by the stem kernel that applies the last update itself (blocks 2 and 3; flownet.param:99-105, 152-158).
 * what = 5 (b = 3): the 12-channel input of IFBlock 3 read back through the product's row-streaming stem kernel (both of its convolutions with
 * one-hot weights, eight launches; values to 2^-21 relative), 12 x hp x wp. */
int rife_hip_v4_tap(const rife_hip_t* r, const uint8_t* in0_rgb, const uint8_t* in1_rgb, int w, int h, float timestep, int what, int b,
                    const float* const* inject, int n_inject, float* out_chw);
/* The plain pass with blobs flow0 .. flow{n_inject - 1} injected (n_inject = 0..3): the remaining blocks and the fused tail run as in
 * rife_hip_process.  out_rgb: w x h u8 RGB. */
int rife_hip_v4_process_injected(const rife_hip_t* r, const uint8_t* in0_rgb, const uint8_t* in1_rgb, int w, int h, float timestep,
                                 const float* const* inject, int n_inject, uint8_t* out_rgb);

/* ---- single-kernel entry points for per-kernel parity tests (host arrays, planar CHW fp32 like ncnn::Mat) --- */
/* 3x3 conv, pad 1, stride 1|2, + bias, optional residual add (same shape as output), per-channel negative slope
 * (1.0 = none, 0.2 = LeakyReLU(0.2), PReLU slopes otherwise): ncnn Convolution (+BinaryOp add +ReLU/PReLU). */
int rife_hip_op_conv3x3(int gpuid, const float* x_chw, int c, int h, int w, const float* weight_oihw, const float* bias,
                        int outc, int stride, const float* residual_chw, const float* slope, float* out_chw);
/* 4x4 stride-2 pad-1 transposed conv (ncnn Deconvolution, weights [oc][ic][4][4]) + per-channel slope. */
int rife_hip_op_deconv4x4(int gpuid, const float* x_chw, int c, int h, int w, const float* weight_oihw, const float* bias,
                          int outc, const float* slope, float* out_chw);
/* rife.Warp (src/warp.cpp:96-168): image c x h x w, flow 2 x h x w. */
int rife_hip_op_warp(int gpuid, const float* image_chw, const float* flow_chw, int c, int h, int w, float* out_chw);

/* ---- workspace pool of the host-buffer entry points (csrc/engine_abi.h: lease_ctx / release_ctx): pooled = idle workspaces the engine holds,
 * leased = workspaces in use by callers right now, high_water = the most callers in flight at any of the last 32 leases (the pool is trimmed to it). */
int rife_hip_pool_state(const rife_hip_t* r, int* pooled, int* leased, int* high_water);

#ifdef __cplusplus
}
#endif

#endif  /* RIFE_HIP_TEST_H */
