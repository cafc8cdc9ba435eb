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
    const float* tsp;      // != null: timestep read from device memory (hipGraph replays)
    int wp, hp;            // padded full resolution
    int Ho, Wo, out_ld, Cout, tiles_x;
    float* dbg = nullptr;  // bench builds (ABL & 1024): the gathered block-input pixel of every thread, [workgroup][512][12]
    FlowPending pend;      // UPD kernels: the flow update of the previous block, applied while gathering (elementwise.h); F, M are then the OLD tensors
};

// LDS pixel record: 32 B hi + 32 B lo (+ 16 B pad: 80-byte records are conflict-free for the 16-byte staging writes and 2-way for
// the stride-2 operand reads; plain 64-byte records are 4-way / 8-way and measured 8 % slower).
template <int ABL> constexpr int stemf_pixb() { return (ABL & 256) ? 64 : 80; }
template <int NS, int ABL = 0>
constexpr int stemf_lds_bytes() { return (9 * 65 + 1) * stemf_pixb<ABL>() + 9 * 2 * NS * 32 * 16; }      // + one dummy pixel record

// ABL, bench builds only (RIFE_ABL): 1 = skip MFMA + epilogue, 16 = skip stores, 32 = skip MFMAs, 1024 = dump the gathered pixels, 4096 = round-2
// staging of the second halo pixel under a lane-divergent branch.  (Settled A/Bs removed: second halo pixel in a second dependent round,
// direct-store epilogue, unswizzled 64-byte records - all slower.)
// ABL 256 (a layout, not an ablation: block 3 of the product): 64-byte records with the 16-byte quarters XOR-swizzled by pixel index (quarter q of pixel P at position q ^ ((P >> 2) & 3): 2-way on the
// stride-2 operand reads like the padded records, conflict-free staging writes) -> 46.6 KB for NS = 1: three workgroups (24 waves) per CU
// UPD: k_flow_update<2 S> of the block before (flownet.param:99-105, 152-158) happens here: the kernel visits every full-resolution pixel (S <= 2)
// anyway, so F, M make one round trip less through HBM per block and the launch disappears; halo pixels shared by tiles are written twice with
// the same value, into the other F, M buffer.
template <int S, int NS, int ABL = 0, int UPD = 0>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((ABL & 256) ? 6 : 4, (ABL & 256) ? 6 : 4))) void stem0_fused_kernel(StemFusedArgs a) {
    static_assert(UPD != 1 || S <= 2, "the scale-4 stem samples a quarter of the full-resolution pixels: it cannot write the updated tensors (UPD = 1), only sample the first update (UPD = 2)");
    constexpr int IH = 9, IW = 65, PIXB = stemf_pixb<ABL>(), NT = NS * 32;
    constexpr int NPIX = IH * IW;
    constexpr int W_16 = 9 * 2 * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* const lw = ldsb + (NPIX + 1) * PIXB;                  // record NPIX: dummy target of the lanes without a second pixel

    const float timestep = a.tsp ? *a.tsp : a.timestep;
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

    // block-input halo tile -> LDS as f16 hi | lo.  585 pixels over 512 threads: every thread takes pixel `tid`, waves 0-1 also
    // pixel 512 + tid.  The gather is straight-line code on clamped coordinates (conv zero padding = a select afterwards) so that
    // the two pixels of waves 0-1 are in flight together instead of costing a second round of dependent F -> image-tap loads.
#define STEM_GATHER(P, O)                                                                                         \
    {                                                                                                             \
        const int p_ = (P) < NPIX ? (P) : NPIX - 1;                                                               \
        const int py_ = p_ / IW, px_ = p_ - py_ * IW;                                                             \
        const int by_ = iy0 + py_, bx_ = ix0 + px_;                                                               \
        const bool in_ = by_ >= 0 && by_ < Hb && bx_ >= 0 && bx_ < Wb;                                            \
        assemble_pixel<S, UPD>(a.img0, a.img1, timestep, a.F, a.M, a.wp, a.hp, min(max(bx_, 0), Wb - 1), min(max(by_, 0), Hb - 1), O, a.pend); \
        _Pragma("unroll") for (int c = 0; c < 12; c++) O[c] = in_ ? O[c] : 0.f;                                   \
    }
#define STEM_STAGE(P, O)                                                                                          \
    {                                                                                                             \
        f16x8 h0, h1, l0, l1;                                                                                     \
        _Pragma("unroll") for (int c = 0; c < 8; c++) {                                                           \
            const _Float16 ha = (_Float16)O[c];                                                                   \
            h0[c] = ha; l0[c] = (_Float16)(O[c] - (float)ha);                                                     \
            const float vb = c < 4 ? O[8 + c] : 0.f;                                                              \
            const _Float16 hb = (_Float16)vb;                                                                     \
            h1[c] = hb; l1[c] = (_Float16)(vb - (float)hb);                                                       \
        }                                                                                                         \
        unsigned char* dst = ldsb + (P) * PIXB;                                                                   \
        const int sw_ = (ABL & 256) ? (((P) >> 2) & 3) : 0;                                                       \
        *reinterpret_cast<f16x8*>(dst + ((0 ^ sw_) << 4)) = h0; *reinterpret_cast<f16x8*>(dst + ((1 ^ sw_) << 4)) = h1; \
        *reinterpret_cast<f16x8*>(dst + ((2 ^ sw_) << 4)) = l0; *reinterpret_cast<f16x8*>(dst + ((3 ^ sw_) << 4)) = l1; \
    }
    // Round 3: no lane-divergent control flow around the staging.  The 73 second pixels (waves 0-1) used to be staged under
    // `if (tid + 512 < NPIX)`; built with the SLP vectorizer the compiler tail-merged that store block with the other paths' and the kernel
    // was not run-to-run stable (whole groups of 16 halo pixels came out with stale LDS contents: tools/probes/stem_bisect.py, stem_poison.py,
    // DESIGN.md (d)-8).  Now the wave role is a scalar (readfirstlane) and the lanes without a second pixel write a dummy record instead
    // of branching.
    const bool two_px = __builtin_amdgcn_readfirstlane(wv8) < 2;
    if (RIFE_ABL(ABL & 4096) && wv8 < 2) {                                       // bench builds: the round-2 form (tools/probes/stem_det_both.py reproduces the instability with it)
        float o0[12], o1[12];
        STEM_GATHER(tid, o0)
        STEM_GATHER(tid + 512, o1)
        STEM_STAGE(tid, o0)
        if (tid + 512 < NPIX) STEM_STAGE(tid + 512, o1)
    } else if (two_px) {
        float o0[12], o1[12];
        STEM_GATHER(tid, o0)
        STEM_GATHER(tid + 512, o1)
        if (RIFE_ABL(ABL & 1024)) { _Pragma("unroll") for (int c = 0; c < 12; c++) a.dbg[((size_t)blockIdx.x * 512 + tid) * 12 + c] = o0[c]; }
        STEM_STAGE(tid, o0)
        const int p1 = tid + 512 < NPIX ? tid + 512 : NPIX;              // record NPIX is the dummy
        STEM_STAGE(p1, o1)
    } else {
        float o0[12];
        STEM_GATHER(tid, o0)
        if (RIFE_ABL(ABL & 1024)) { _Pragma("unroll") for (int c = 0; c < 12; c++) a.dbg[((size_t)blockIdx.x * 512 + tid) * 12 + c] = o0[c]; }
        STEM_STAGE(tid, o0)
    }
#undef STEM_GATHER
#undef STEM_STAGE
    __syncthreads();

    if (RIFE_ABL(ABL & 1)) return;
    if (nsel >= NS) return;                                // NS = 1: waves 4-7 only helped with the gather
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const unsigned char* ab = ldsb + ((2 * wv) * IW + 2 * li) * PIXB + half * 16;
    const unsigned char* bb = lw + (half * NT + nsel * 32 + li) * 16;
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int dy = t / 3, dx = t % 3;
        f16x8 ah, al;
        if (ABL & 256) {
            const int P = (2 * wv + dy) * IW + 2 * li + dx;
            const int ph = half ^ ((P >> 2) & 3);
            ah = *reinterpret_cast<const f16x8*>(ldsb + P * 64 + (ph << 4));
            al = *reinterpret_cast<const f16x8*>(ldsb + P * 64 + ((ph ^ 2) << 4));
        } else {
            ah = *reinterpret_cast<const f16x8*>(ab + (dy * IW + dx) * PIXB);
            al = *reinterpret_cast<const f16x8*>(ab + (dy * IW + dx) * PIXB + 32);
        }
        const f16x8 bw = *reinterpret_cast<const f16x8*>(bb + (t * 2 * NT) * 16);
        if (RIFE_ABL(ABL & 32)) { acc[0] += (float)ah[0] + (float)al[1] + (float)bw[2]; continue; }     // ablation: LDS reads without the matrix pipe
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
        if (RIFE_ABL(ABL & 16)) { if (v[0] == 123.456f) a.out[0] = v[1]; }                               // ablation: no stores
        else if (oy < a.Ho && ox0 + px < a.Wo && c0 < a.Cout) *reinterpret_cast<f32x4*>(orow + (size_t)px * a.out_ld) = v;
    }
}

}  // namespace rife
