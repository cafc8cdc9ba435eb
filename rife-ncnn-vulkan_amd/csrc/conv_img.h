// conv_img_s2_kernel: the FIRST convolution of the rife-v2.x / v3.x ContextNet (models/rife-v2.3/contextnet.param:5-6: Convolution 3 -> 32, 3x3,
// stride 2, pad 1 + PReLU), twice per pair at full frame resolution (src/rife.cpp:1027-1060), straight from the padded RGBX u8 frame.
//
// Until round 5 this layer ran on the fp32 matrix path over an 8-channel fp32 copy of the frame (k2_image_nhwc8: 67 MB written and read again per
// pass, 2 x (18 + 49) us per 1080p pair).  Here the frame is read as it lies (4 bytes per pixel, 8.4 MB, L2 resident), x * (1 / 255.f) is recomputed per
// tap exactly like every other consumer of the frame does (elementwise.h unpack_rgb) and split into f16 hi + lo for the split-f16 matrix scheme
// (conv_mfma.h conv_h2_kernel); K is ordered (tap, RGBX byte): k = 4 tap + byte, 36 real values padded to 3 K-steps of 16, so a lane's 8 K values of
// a step are the 2 x 4 bytes of two neighbouring taps - two dword loads, no LDS staging, no layout conversion.  One wave = 32 output pixels of a row x
// all 32 output channels = 6 MFMAs; the 32 x 32 fp32 tile is transposed through LDS so that every store instruction writes 1 KB of contiguous NHWC.
// The layer is bound by its 67 MB of output.
#pragma once
#include "conv_mfma.h"

namespace rife {

struct ImgConvArgs {
    const uint32_t* img;      // padded RGBX u8 frame, wp x hp, byte 3 = 0
    float* out;               // NHWC fp32, 32 channels, Ho x Wo
    const uint16_t* wpk;      // f16 [K-step 3][k half 2][out channel 32][8]: k = 16 step + 8 half + e -> tap 4 step + 2 half + (e >> 2), byte e & 3 (byte 3 and taps 9..11: 0)
    const float* bias;        // [32]
    const float* slope;       // [32]
    int wp, hp, Wo, Ho, tiles_x, ntiles;
    const uint32_t* img1 = nullptr;      // gridDim.y = 2: the second frame / output of the launch (the two ContextNet passes)
    float* out1 = nullptr;
};

__global__ __launch_bounds__(256) void conv_img_s2_kernel(ImgConvArgs a) {
    constexpr int ROWF = 36;                                   // floats per pixel row of the transpose tile
    __shared__ __attribute__((aligned(16))) float tlb[4 * 32 * ROWF];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int h = lane >> 5, li = lane & 31;
    float* const tl = tlb + wv * 32 * ROWF;
    const uint32_t* const timg = blockIdx.y ? a.img1 : a.img;
    float* const tout = blockIdx.y ? a.out1 : a.out;
    f16x8 wA[3];
#pragma unroll
    for (int j = 0; j < 3; j++) wA[j] = *reinterpret_cast<const f16x8*>(a.wpk + ((j * 2 + h) * 32 + li) * 8);
    f32x4 b4[4], s4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        b4[q] = *reinterpret_cast<const f32x4*>(a.bias + 8 * q + 4 * h);
        s4[q] = *reinterpret_cast<const f32x4*>(a.slope + 8 * q + 4 * h);
    }
    const float k255 = 1 / 255.f;
    for (int T = blockIdx.x * 4 + wv; T < a.ntiles; T += gridDim.x * 4) {
        const int oy = T / a.tiles_x, ox0 = (T - oy * a.tiles_x) * 32;
        const int ox = ox0 + li;
        uint32_t px[3][2];
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                const int t = 4 * j + 2 * h + tt;              // tap index, 0..11 (9..11 do not exist)
                const int dy = t / 3, dx = t - 3 * dy;
                const int iy = 2 * oy + dy - 1, ix = 2 * ox + dx - 1;
                const bool ok = t < 9 && iy >= 0 && iy < a.hp && ix >= 0 && ix < a.wp;
                px[j][tt] = ok ? timg[(size_t)iy * a.wp + ix] : 0u;
            }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            f16x8 bh, bl;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float v = (float)((px[j][e >> 2] >> (8 * (e & 3))) & 0xffu) * k255;
                const _Float16 hh = (_Float16)v;
                bh[e] = hh;
                bl[e] = (_Float16)(v - (float)hh);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wA[j], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wA[j], bl, acc, 0, 0, 0);
        }
        // epilogue: bias + PReLU, transpose through LDS, 1 KB contiguous per store instruction
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = acc[4 * q + k] + b4[q][k]; v[k] = v[k] < 0.f ? v[k] * s4[q][k] : v[k]; }
            *reinterpret_cast<f32x4*>(tl + li * ROWF + 8 * q + 4 * h) = v;
        }
        __builtin_amdgcn_wave_barrier();
        const int pl = lane >> 3, chunk = lane & 7;           // 8 lanes (16-byte chunks) per pixel, 8 pixels per store instruction
        float* const orow = tout + ((size_t)oy * a.Wo + ox0) * 32 + chunk * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int p = j * 8 + pl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tl + p * ROWF + chunk * 4);
            if (ox0 + p < a.Wo) *reinterpret_cast<f32x4*>(orow + (size_t)p * 32) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace rife
