// conv_rowf_kernel<C, ROWS>: conv_row_kernel (conv_row.h) for the wide, SMALL-grid trunk convolutions of rife-v2.x / v3.x - Convolution 3x3 pad 1
// stride 1 + PReLU, C -> C channels with C = 256, 384 (IFNet blocks 1 and 0: models/rife-v2.3/flownet.param:41-52 and 10-21, 2,040 - 8,160 pixels at 1080p;
// the same shapes in the FusionNet / ContextNet pyramids: fusionnet.param, contextnet.param) - on plain NHWC fp32 tensors (the v2 schedule keeps fp32
// activations between layers; views with a pixel stride / channel offset are served, the FusionNet writes into its concat buffers).
//
// These layers took 20 - 36 us per launch on the per-tile kernels whatever their size (conv_h2b_kernel: a chain of C / 16 K-chunk steps of stage - barrier -
// 36 matrix instructions - barrier per workgroup; block 0 with split-K and a reduction launch on top), 12 launches per pair.  The form of conv_row_kernel
// removes the chain instead of shortening its links:
//   * one workgroup = ROWS x 32 output pixels x ALL C output channels; wave w owns the 32-channel output block w (C / 32 waves = 8 or 12), so a workgroup's
//     whole life is one pass over K with ONE barrier per two K chunks and none inside the matrix loop;
//   * the fp32 halo ((ROWS + 2) x 34 pixels) of two 16-channel chunks per phase is split into f16 hi / lo while it is staged (hi = f16(a), lo = f16(a - hi):
//     the split-f16 scheme of conv_h2_kernel) into the plane / half-swap layout conv_row_kernel reads, two phases alternating between two LDS buffers;
//   * every weight fragment is used by exactly one wave, so the weights never touch LDS: each wave streams its [chunk][tap][k half][32 rows][8 f16] slice
//     from the L2 straight into matrix operand registers through a ring of (chunk, tap) slots one to two K chunks ahead of the matrix pipe.  ROWS is picked so
//     that few workgroups stream the weights (the per-XCD L2 was the limit of conv_row_kernel on 255 workgroups): 4 rows for 256 channels, 2 for 384.
// Products and accumulation order (chunk-major, taps in order, hi then lo) are those of conv_h2b_kernel without split-K.
#pragma once
#include "conv_row.h"

namespace rife {

struct RowfArgs {
    const float* in;             // NHWC fp32, in_ld floats per pixel, first channel at in_coff
    float* out;                  // NHWC fp32, out_ld / out_coff
    const unsigned char* img;    // per 32-channel output block: [chunk C/16][tap 9][k half 2][row 32][8 f16] (rows permuted by s16_row_channel), bias[32], slope[32]
    int H, W;
    int in_ld, in_coff, out_ld, out_coff;
    int tiles_x;
    const float* in1 = nullptr;   // gridDim.y = 2: the second tensor pair of the launch (same geometry, strides and weights)
    float* out1 = nullptr;
};

template <int C> constexpr int rowf_ph() { return 2; }                                  // K chunks per staging phase
template <int C> constexpr int rowf_ring() { return C >= 384 ? 10 : 18; }               // (chunk, tap) weight slots in flight per wave (4 VGPRs each)
template <int C> constexpr int rowf_waves_per_simd() { return C >= 256 ? C / 32 / 4 : 2; }      // one workgroup per CU: 8 waves = 2 per SIMD, 12 = 3; 128 channels: two workgroups of 4 waves
template <int C, int ROWS> constexpr int rowf_lds_bytes() { return 2 * rowf_ph<C>() * (ROWS + 2) * 34 * 64; }
constexpr int rowf_img_bytes(int C) { return (C / 32) * t64_img_nt(1, C / 16); }

template <int C, int ROWS>
__global__ __launch_bounds__(2 * C) __attribute__((amdgpu_waves_per_eu(rowf_waves_per_simd<C>(), rowf_waves_per_simd<C>()))) void conv_rowf_kernel(RowfArgs a) {
    constexpr int NW = C / 32, NTHR = 64 * NW, NCH = C / 16, IH = ROWS + 2, IW = 34, NPX = IH * IW;
    constexpr int PLANE = NPX * 32, CHB = 2 * PLANE;                     // bytes per (chunk, hi | lo) plane / per chunk in LDS
    constexpr int PH = rowf_ph<C>(), NPH = NCH / PH;
    constexpr int PSLOTS = PH * NPX * 2, NLD = (PSLOTS + NTHR - 1) / NTHR;      // (chunk, pixel, k half) staging slots per phase: 8 floats in, 16 B hi + 16 B lo out
    constexpr int WSTRIDE = t64_img_nt(1, NCH);
    static_assert(NCH % PH == 0, "whole staging phases");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* const lds = ldsb;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);              // wave = 32-channel output block
    const float* const tin = blockIdx.y ? a.in1 : a.in;
    float* const tout = blockIdx.y ? a.out1 : a.out;
    const int h = lane >> 5, li = lane & 31;
    int L;
    {
        const int n = gridDim.x, b = blockIdx.x;
        const int q = n >> 3, r = n & 7, xcd = b & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);      // block b runs on XCD b % 8: contiguous bands of tiles per XCD
    }
    const int ty = L / a.tiles_x, tx = L - ty * a.tiles_x;
    const int oy0 = ty * ROWS, ox0 = tx * 32;

    // ---- halo staging
    int goff[NLD];                                                       // float offset of the slot's 8 channels in chunk 0 of a phase; < 0: outside the image (zeros)
    unsigned ldo[NLD];                                                   // byte offset of the slot's hi entry inside an LDS buffer
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int s = min(tid + k * NTHR, PSLOTS - 1);
        const int cc = s / (2 * NPX), rem = s - cc * (2 * NPX);
        const int P = rem >> 1, kh = rem & 1;
        const int py = P / IW, px = P - py * IW;
        const int gy = oy0 - 1 + py, gx = ox0 - 1 + px;
        const bool ok = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[k] = ok ? (gy * a.W + gx) * a.in_ld + a.in_coff + 16 * cc + 8 * kh : -1;
        ldo[k] = (unsigned)(cc * CHB + P * 32 + ((kh ^ ((P >> 3) & 1)) << 4));
    }
    f32x4 stage[NLD][2];
#define ROWF_LOAD(PHASE)                                                                                     \
    _Pragma("unroll") for (int k = 0; k < NLD; k++) {                                                        \
        const float* p_ = tin + (goff[k] < 0 ? a.in_coff : goff[k] + 16 * PH * (PHASE));                    \
        stage[k][0] = *reinterpret_cast<const f32x4*>(p_);                                                   \
        stage[k][1] = *reinterpret_cast<const f32x4*>(p_ + 4);                                               \
    }
#define ROWF_STORE(PHASE)                                                                                    \
    _Pragma("unroll") for (int k = 0; k < NLD; k++) {                                                        \
        f16x8 hv, lv;                                                                                        \
        _Pragma("unroll") for (int e = 0; e < 8; e++) {                                                      \
            const float v_ = goff[k] < 0 ? 0.f : stage[k][e >> 2][e & 3];                                    \
            const _Float16 hh_ = (_Float16)v_;                                                               \
            hv[e] = hh_;                                                                                     \
            lv[e] = (_Float16)(v_ - (float)hh_);                                                             \
        }                                                                                                    \
        if (PSLOTS % NTHR == 0 || tid + k * NTHR < PSLOTS) {                                                 \
            unsigned char* d_ = lds + ((PHASE) & 1) * PH * CHB + ldo[k];                                     \
            *reinterpret_cast<f16x8*>(d_) = hv;                                                              \
            *reinterpret_cast<f16x8*>(d_ + PLANE) = lv;                                                      \
        }                                                                                                    \
    }

    // ---- operands
    unsigned ao[ROWS][9];                                                // fragment offsets inside a chunk (hi plane; lo = + PLANE)
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++)
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int P = (rr + t / 3) * IW + li + t % 3;
            ao[rr][t] = (unsigned)(P * 32 + ((h ^ ((P >> 3) & 1)) << 4));
        }
    const unsigned char* const wsrc = a.img + (size_t)w * WSTRIDE + (h * 32 + li) * 16;      // + (chunk * 9 + tap) * 1024
    f32x16 acc[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[rr][q] = 0.f;

    constexpr int RING = rowf_ring<C>(), NJ = NCH * 9;
    f16x8 wr[RING];
#define ROWF_WSLOT(J) wr[(J) % RING] = *reinterpret_cast<const f16x8*>(wsrc + (J) * 1024);
#define ROWF_TAP(J)                                                                                          \
    {                                                                                                        \
        constexpr int c_ = (J) / 9, t_ = (J) % 9;                                                            \
        const unsigned char* const cb_ = lds + (((c_ / PH) & 1) * PH + c_ % PH) * CHB;                       \
        _Pragma("unroll") for (int rr = 0; rr < ROWS; rr++) {                                                \
            const f16x8 ah = *reinterpret_cast<const f16x8*>(cb_ + ao[rr][t_]);                              \
            const f16x8 al = *reinterpret_cast<const f16x8*>(cb_ + ao[rr][t_] + PLANE);                      \
            acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[(J) % RING], ah, acc[rr], 0, 0, 0);          \
            acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[(J) % RING], al, acc[rr], 0, 0, 0);          \
        }                                                                                                    \
        if ((J) + RING < NJ) ROWF_WSLOT((J) + RING)                                                          \
    }

    ROWF_LOAD(0)
    for_each_slot<0, RING>([&](auto j) { ROWF_WSLOT(decltype(j)::value) });
    ROWF_STORE(0)
    __syncthreads();
    for_each_slot<0, NPH>([&](auto pp) {
        constexpr int p = decltype(pp)::value;
        if (p + 1 < NPH) ROWF_LOAD(p + 1)                                // the next phase's halo chunks travel under this phase's matrix work
        for_each_slot<p * PH * 9, (p + 1) * PH * 9>([&](auto j) { ROWF_TAP(decltype(j)::value) });
        if (p + 1 < NPH) {
            ROWF_STORE(p + 1)                                            // the other buffer: last read in phase p - 1
            __syncthreads();
        }
    });
#undef ROWF_LOAD
#undef ROWF_STORE
#undef ROWF_WSLOT
#undef ROWF_TAP

    // ---- epilogue: y = PReLU(acc + bias); the lane holds the 16 consecutive channels 32 w + 16 h .. + 15 of pixel li
    const float* const bs = reinterpret_cast<const float*>(a.img + (size_t)w * WSTRIDE + (size_t)NCH * t64_wch(1));
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++) {
        const int oy = oy0 + rr, ox = ox0 + li;
        const bool ok = oy < a.H && ox < a.W;
        float* const o = tout + ((size_t)oy * a.W + ox) * a.out_ld + a.out_coff + 32 * w + 16 * h;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bs + 16 * h + 4 * q);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(bs + 32 + 16 * h + 4 * q);
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float y = acc[rr][4 * q + k] + b4[k];
                v[k] = y < 0.f ? y * s4[k] : y;
            }
            if (ok) *reinterpret_cast<f32x4*>(o + 4 * q) = v;
        }
    }
}

}  // namespace rife
