// stem2_fused_kernel<S, IMG, FSCALE>: the rife-v2.x / v3.x twin of stem0_fused_kernel (stem_fused.h) - the block-input assembly of IFNet blocks 1..
// (2 x Interp x2 of the running flow, Mul 2, 2 x rife.Warp, Concat, Interp 1/S: models/rife-v2.3/flownet.param:27-37, 58-68, 90-99) and of the
// FusionNet (fusionnet.param:14-23) fused into the stride-2 3 x 3 convolution + PReLU that consumes it (flownet.param:38-39, 69-70, 100-101;
// fusionnet.param:24-25).  The 10-channel block input is computed per halo pixel by assemble2_pixel<S>() - the very code of k2_assemble - split into
// f16 hi / lo and written to LDS only; the convolution runs on the f16 matrix pipe with fp32 accumulation in the tap / hi / lo order of
// conv_h2s2_kernel on the one zero-padded 16-channel chunk, so the result is the unfused path's.  Saves writing and re-reading the NHWC16 fp32 block
// input (2 x 134 MB per pair for IFNet block 3 and again for the FusionNet at 1920 x 1088) and one launch each.
//   512 threads = 8 waves: all of them gather the 9 x 65 halo tile (the kernel is gather-latency bound), then wave w computes output row w & 3 of
//   the 4 x 32 tile for the 32-channel output subtiles (w >> 2) and (w >> 2) + 2.  Weights of all subtiles wait in LDS from the prologue on.
#pragma once
#include "conv_mfma.h"
#include "elementwise_v2.h"

namespace rife {

template <typename IMG>
struct Stem2Args {
    IMG img0, img1;
    const float4* acc;         // running flow, float4 per pixel at (hp / 2) x (wp / 2)
    const unsigned char* wpk;  // f16 [n-tile][tap 9][half 2][n NT][8] (pack_weights_h2, one chunk)
    const float *bias, *slope; // padded to nsub * 32 or more
    float* out;                // NHWC, out_ld floats per pixel
    int wp, hp;                // padded frame
    int Ho, Wo, out_ld, Cout, tiles_x;
    int NS;                    // 32-channel subtiles per n-tile of the weight packing
    int nsub;                  // 32-channel output subtiles (<= 4)
};

// R64 (scale 1 only: the gather needs 56 VGPRs): 64-byte halo records with the 16-byte quarters XOR-swizzled by pixel index (quarter q of pixel P at position
// q ^ ((P >> 2) & 3): 2-way on the stride-2 operand reads like the padded 80-byte records, conflict-free staging writes - the block-3 layout of stem_fused.h) and the
// weights read from the L2 straight into the operand registers after the gather instead of waiting in LDS: 38.5 KB of LDS per workgroup, THREE workgroups = 24 waves
// per CU (85 VGPRs) instead of two - the kernel is gather-latency bound.  At most two output subtiles (one per wave group): the scale-1 layers have 48 and 32 channels.
constexpr int stem2_lds_bytes(int nsub, bool r64 = false) { return r64 ? (9 * 65 + 1) * 64 + 2 * 128 * 4 : (9 * 65 + 1) * 80 + nsub * 32 * 9 * 2 * 16 + 2 * 128 * 4; }

template <int S, typename IMG, bool FSCALE, bool R64 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(R64 ? 6 : 4, R64 ? 6 : 4))) void stem2_fused_kernel(Stem2Args<IMG> a) {
    constexpr int IH = 9, IW = 65, PIXB = R64 ? 64 : 80, NPIX = IH * IW, ROWF = 36;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int cp = a.nsub * 32;
    unsigned char* const lw = ldsb + (NPIX + 1) * PIXB;                  // record NPIX: dummy target of the lanes without a second pixel
    float* const lbs = reinterpret_cast<float*>(lw + (R64 ? 0 : cp * 9 * 2 * 16));  // bias[128], slope[128]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv8 = tid >> 6;
    const int wv = wv8 & 3, grp = wv8 >> 2;
    const int half = lane >> 5, li = lane & 31;
    int L;
    {
        const int n = gridDim.x, b = blockIdx.x;
        const int q = n >> 3, r = n & 7, xcd = b & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int ty = L / a.tiles_x, tx = L - ty * a.tiles_x;
    const int oy0 = ty * 4, ox0 = tx * 32;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const int Hb = a.hp / S, Wb = a.wp / S;

    // weights -> LDS as [tap * 2 + half][cp rows][8 f16]; bias and slopes next to them
    {
        const int NT = a.NS * 32;
        if (!R64) for (int idx = tid; idx < 18 * cp; idx += 512) {
            const int t2 = idx / cp, n = idx - t2 * cp;
            const int nt = n / NT, nin = n - nt * NT;
            reinterpret_cast<f32x4*>(lw)[idx] = reinterpret_cast<const f32x4*>(a.wpk)[(nt * 18 + t2) * NT + nin];
        }
        if (tid < cp) { lbs[tid] = a.bias[tid]; lbs[128 + tid] = a.slope[tid]; }
    }

    // block-input halo tile -> LDS as f16 hi | lo (80-byte records).  585 pixels over 512 threads: every thread takes pixel `tid`, waves 0-1 also pixel
    // 512 + tid; straight-line code on clamped coordinates (conv zero padding = a select afterwards), no lane-divergent control flow around the staging
    // (stem_fused.h, round 3)
#define STEM2_GATHER(P, O)                                                                                        \
    {                                                                                                             \
        const int p_ = (P) < NPIX ? (P) : NPIX - 1;                                                               \
        const int py_ = p_ / IW, px_ = p_ - py_ * IW;                                                             \
        const int by_ = iy0 + py_, bx_ = ix0 + px_;                                                               \
        const bool in_ = by_ >= 0 && by_ < Hb && bx_ >= 0 && bx_ < Wb;                                            \
        assemble2_pixel<S, IMG, FSCALE>(a.img0, a.img1, a.acc, a.wp, a.hp, min(max(bx_, 0), Wb - 1), min(max(by_, 0), Hb - 1), O); \
        _Pragma("unroll") for (int c = 0; c < 10; c++) O[c] = in_ ? O[c] : 0.f;                                   \
    }
#define STEM2_STAGE(P, O)                                                                                         \
    {                                                                                                             \
        f16x8 h0, h1, l0, l1;                                                                                     \
        _Pragma("unroll") for (int c = 0; c < 8; c++) {                                                           \
            const _Float16 ha = (_Float16)O[c];                                                                   \
            h0[c] = ha; l0[c] = (_Float16)(O[c] - (float)ha);                                                     \
            const float vb = c < 2 ? O[8 + c] : 0.f;                                                              \
            const _Float16 hb = (_Float16)vb;                                                                     \
            h1[c] = hb; l1[c] = (_Float16)(vb - (float)hb);                                                       \
        }                                                                                                         \
        unsigned char* dst = ldsb + (P) * PIXB;                                                                   \
        const int sw_ = R64 ? (((P) >> 2) & 3) : 0;                                                               \
        *reinterpret_cast<f16x8*>(dst + ((0 ^ sw_) << 4)) = h0; *reinterpret_cast<f16x8*>(dst + ((1 ^ sw_) << 4)) = h1; \
        *reinterpret_cast<f16x8*>(dst + ((2 ^ sw_) << 4)) = l0; *reinterpret_cast<f16x8*>(dst + ((3 ^ sw_) << 4)) = l1; \
    }
    const bool two_px = __builtin_amdgcn_readfirstlane(wv8) < 2;
    if (two_px) {
        float o0[10], o1[10];
        STEM2_GATHER(tid, o0)
        STEM2_GATHER(tid + 512, o1)
        STEM2_STAGE(tid, o0)
        const int p1 = tid + 512 < NPIX ? tid + 512 : NPIX;              // record NPIX is the dummy
        STEM2_STAGE(p1, o1)
    } else {
        float o0[10];
        STEM2_GATHER(tid, o0)
        STEM2_STAGE(tid, o0)
    }
#undef STEM2_GATHER
#undef STEM2_STAGE
    const bool on0 = grp < a.nsub, on1 = grp + 2 < a.nsub;              // wave-uniform
    f16x8 wq0[R64 ? 9 : 1];                                            // R64: this wave's weight fragments, issued before the barrier
    if (R64) {
        const int NT = a.NS * 32;
        const int g0 = min(grp, a.nsub - 1);
        const int nt0 = g0 / a.NS;
        const f16x8* const w0 = reinterpret_cast<const f16x8*>(a.wpk) + (size_t)(nt0 * 18 + half) * NT + (g0 - nt0 * a.NS) * 32 + li;
#pragma unroll
        for (int t = 0; t < 9; t++) wq0[t] = w0[(size_t)(2 * t) * NT];
    }
    __syncthreads();

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
    const unsigned char* ab = ldsb + ((2 * wv) * IW + 2 * li) * PIXB + half * 16;
    const unsigned char* bb = lw + (half * cp + grp * 32 + li) * 16;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int dy = t / 3, dx = t % 3;
        f16x8 ah, al;
        if (R64) {
            const int P = (2 * wv + dy) * IW + 2 * li + dx;
            const int ph = half ^ ((P >> 2) & 3);
            ah = *reinterpret_cast<const f16x8*>(ldsb + P * 64 + (ph << 4));
            al = *reinterpret_cast<const f16x8*>(ldsb + P * 64 + ((ph ^ 2) << 4));
        } else {
            ah = *reinterpret_cast<const f16x8*>(ab + (dy * IW + dx) * PIXB);
            al = *reinterpret_cast<const f16x8*>(ab + (dy * IW + dx) * PIXB + 32);
        }
        if (on0) {
            const f16x8 bw = R64 ? wq0[t] : *reinterpret_cast<const f16x8*>(bb + (t * 2 * cp) * 16);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw, ah, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw, al, acc0, 0, 0, 0);
        }
        if (!R64 && on1) {
            const f16x8 bw = *reinterpret_cast<const f16x8*>(bb + (t * 2 * cp + 64) * 16);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw, ah, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw, al, acc1, 0, 0, 0);
        }
    }

    // epilogue: bias + PReLU, then every wave transposes its 32 x 32 tile through LDS (the halo tile is dead by now) so that 8 consecutive lanes store
    // one pixel's 128 contiguous bytes
    __syncthreads();
    float* const tl = reinterpret_cast<float*>(ldsb) + wv8 * 32 * ROWF;
    const int oy = oy0 + wv;
    const int pl = lane >> 3, chunk = lane & 7;
#pragma unroll
    for (int j = 0; j < 2; j++) {
        if (!(j ? (!R64 && on1) : on0)) continue;
        const int g = grp + 2 * j;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c0 = g * 32 + 8 * q + 4 * half;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbs + c0);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(lbs + 128 + c0);
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = (j ? acc1[4 * q + k] : acc0[4 * q + k]) + b4[k]; v[k] = v[k] < 0.f ? v[k] * s4[k] : v[k]; }
            *reinterpret_cast<f32x4*>(tl + li * ROWF + 8 * q + 4 * half) = v;
        }
        const int c0 = g * 32 + chunk * 4;
        float* const orow = a.out + ((size_t)oy * a.Wo + ox0) * a.out_ld + c0;
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int px = jj * 8 + pl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tl + px * ROWF + chunk * 4);
            if (oy < a.Ho && ox0 + px < a.Wo && c0 < a.Cout) *reinterpret_cast<f32x4*>(orow + (size_t)px * a.out_ld) = v;
        }
    }
}

}  // namespace rife
