// stem0_fused_kernel<S, NS>: blocks 1..3 of the v4 IFNet — the block-input assembly (2x rife.Warp, Concat, Interp 1/S,
// div; flownet.param:52-62, 107-115, 160-165) fused into the first stride-2 3x3 convolution (convrelu_2/4/6,
// flownet.param:63, 116, 166).  The 12-channel block input is computed per halo pixel by assemble_pixel<S>() (the very
// code of k_assemble), split into f16 hi/lo and written to LDS only; the convolution runs on the f16 matrix pipe with
// fp32 accumulation (see conv_h2_kernel for the split-f16 scheme).  Saves writing and re-reading the 16-channel block
// input (535 MB per pair for block 3 at 4K) and one launch.
//   512 threads = 8 waves: all of them gather the 9 x 65 halo tile (the kernel is gather-latency bound), then wave w
//   computes output row w & 3 of the 4 x 32 tile for the N-subtile w >> 2 (waves 4-7 are idle in the tiny MFMA phase if NS = 1).
#pragma once
#include "conv_mfma.h"
#include "elementwise.h"

namespace rife {

struct StemFusedArgs {
    const uint32_t *img0, *img1;
    const float4* F; const float* M;
    const void* wpk;       // f16 [tap 9][half 2][n NT][8]
    const float *bias, *slope;
    float* out;            // NHWC, out_ld floats per pixel
    float timestep;
    int wp, hp;            // padded full resolution
    int Ho, Wo, out_ld, Cout, tiles_x;
};

constexpr int STEMF_PIXB = 64;     // no pad: the tiny MFMA phase tolerates 8-way conflicts, the gather phase wants 3 workgroups per CU
template <int NS>
constexpr int stemf_lds_bytes() { return 9 * 65 * STEMF_PIXB + 9 * 2 * NS * 32 * 16; }

// ABL (bench only): 1 = skip MFMA + epilogue, 2 = skip the image gathers, 4 = skip F/M loads too, 8 = skip LDS staging, 16 = skip stores, 32 = skip MFMAs
template <int S, int NS, int ABL = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void stem0_fused_kernel(StemFusedArgs a) {
    constexpr int IH = 9, IW = 65, PIXB = STEMF_PIXB, NT = NS * 32;
    constexpr int NPIX = IH * IW;
    constexpr int W_16 = 9 * 2 * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* const lw = ldsb + NPIX * PIXB;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv8 = tid >> 6;
    const int wv = wv8 & 3, nsel = wv8 >> 2;          // output row, N-subtile of this wave
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

    // weights -> LDS (one 16-channel chunk covers the 12 input channels)
    for (int idx = tid; idx < W_16; idx += 512) reinterpret_cast<f32x4*>(lw)[idx] = reinterpret_cast<const f32x4*>(a.wpk)[idx];

    // block-input halo tile -> LDS as f16 hi | lo
    for (int p = tid; p < NPIX; p += 512) {
        const int py = p / IW, px = p - py * IW;
        const int by = iy0 + py, bx = ix0 + px;
        float o[12];
        if (ABL & 2) {
            if (by >= 0 && by < Hb && bx >= 0 && bx < Wb) {
                const size_t i = (size_t)by * a.wp + bx;
                float4 f = make_float4(1.f, 2.f, 3.f, 4.f); float m = 0.5f;
                if (!(ABL & 4)) { f = a.F[i]; m = a.M[i]; }
#pragma unroll
                for (int c = 0; c < 12; c++) o[c] = f.x * (float)c + f.y + f.z + f.w + m;
            } else {
#pragma unroll
                for (int c = 0; c < 12; c++) o[c] = 0.f;
            }
        } else
        if (by >= 0 && by < Hb && bx >= 0 && bx < Wb) assemble_pixel<S>(a.img0, a.img1, a.timestep, a.F, a.M, a.wp, a.hp, bx, by, o);
        else {
#pragma unroll
            for (int c = 0; c < 12; c++) o[c] = 0.f;        // conv zero padding
        }
        f16x8 h0, h1, l0, l1;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const _Float16 ha = (_Float16)o[c];
            h0[c] = ha; l0[c] = (_Float16)(o[c] - (float)ha);
            const float vb = c < 4 ? o[8 + c] : 0.f;
            const _Float16 hb = (_Float16)vb;
            h1[c] = hb; l1[c] = (_Float16)(vb - (float)hb);
        }
        unsigned char* dst = ldsb + p * PIXB;
        if (ABL & 8) { if (h0[0] == (_Float16)123.f) *reinterpret_cast<f16x8*>(dst) = h0; continue; }
        *reinterpret_cast<f16x8*>(dst) = h0; *reinterpret_cast<f16x8*>(dst + 16) = h1;
        *reinterpret_cast<f16x8*>(dst + 32) = l0; *reinterpret_cast<f16x8*>(dst + 48) = l1;
    }
    __syncthreads();

    if (ABL & 1) return;
    if (nsel >= NS) return;                                // NS = 1: waves 4-7 only helped with the gather
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const unsigned char* ab = ldsb + ((2 * wv) * IW + 2 * li) * PIXB + half * 16;
    const unsigned char* bb = lw + (half * NT + nsel * 32 + li) * 16;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int dy = t / 3, dx = t % 3;
        const f16x8 ah = *reinterpret_cast<const f16x8*>(ab + (dy * IW + dx) * PIXB);
        const f16x8 al = *reinterpret_cast<const f16x8*>(ab + (dy * IW + dx) * PIXB + 32);
        const f16x8 bw = *reinterpret_cast<const f16x8*>(bb + (t * 2 * NT) * 16);
        if (ABL & 32) { acc[0] += (float)ah[0] + (float)al[1] + (float)bw[2]; continue; }     // ablation: LDS reads without the matrix pipe
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw, ah, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw, al, acc, 0, 0, 0);
    }

    // epilogue: bias + leaky, then transpose the wave's 32 x 32 tile through LDS (the halo tile is dead by now) so that
    // 8 consecutive lanes store one pixel's 128 contiguous bytes (a full line per pixel, 1 KB per instruction if out_ld = 32)
    __syncthreads();                                       // waves 4-7 (NS = 1) have exited; exited waves do not count
    constexpr int ROWF = 36;
    float* const tl = reinterpret_cast<float*>(ldsb) + wv8 * 32 * ROWF;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int c0 = nsel * 32 + 8 * q + 4 * half;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c0);
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(a.slope + c0);
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; k++) { v[k] = acc[4 * q + k] + b4[k]; v[k] = v[k] < 0.f ? v[k] * s4[k] : v[k]; }
        *reinterpret_cast<f32x4*>(tl + li * ROWF + 8 * q + 4 * half) = v;
    }
    const int oy = oy0 + wv;
    const int pl = lane >> 3, chunk = lane & 7;
    const int c0 = nsel * 32 + chunk * 4;
    float* const orow = a.out + ((size_t)oy * a.Wo + ox0) * a.out_ld + c0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int px = j * 8 + pl;
        const f32x4 v = *reinterpret_cast<const f32x4*>(tl + px * ROWF + chunk * 4);
        if (ABL & 16) { if (v[0] == 123.456f) a.out[0] = v[1]; }                               // ablation: no stores
        else if (oy < a.Ho && ox0 + px < a.Wo && c0 < a.Cout) *reinterpret_cast<f32x4*>(orow + (size_t)px * a.out_ld) = v;
    }
}

}  // namespace rife
