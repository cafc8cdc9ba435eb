// conv_row_kernel<C, ROWS>: the residual trunk convolutions of the two COARSE IFBlocks of rife-v4.6 (block 0: 192 -> 192 channels at 1/32
// resolution, block 1: 128 -> 128 at 1/16; and block 2, 96 -> 96 at 1/8, while its grid has fewer 8 x 32 tiles than the chip has CUs: 1080p; reference models/rife-v4.6/flownet.param:14-42, 66-94: Split, Convolution 3x3 pad 1, BinaryOp add,
// ReLU slope 0.2), 16 launches per pair, on S16 tensors (conv_t64.h) with the split-f16 matrix scheme of conv_h2b_kernel.
//
// These layers are small (4K: 8,160 and 32,640 pixels; 5.4 and 9.6 GFLOP) and were latency bound: the per-tile kernels need 27 - 37 us per
// launch whatever the size, a chain of 8 - 12 dependent K-chunk steps (stage, barrier, MFMA, barrier) in each of 100 - 270 workgroups, and
// the persistent LDS-DMA kernel of the fine blocks is no better here (its per-step cost exceeds a step's matrix work).  Skipping both coarse
// trunks altogether raised the 4K rate from 431 to 514 frames/s with two pairs in flight: they cost their full time.  This kernel removes the
// chain instead of shortening its links:
//   * one workgroup = ROWS x 32 output pixels x ALL C output channels; wave w owns the 32-channel output block w (C / 32 waves), so the
//     grid is as wide as the layer allows (4K: 255 workgroups for block 0, 510 for block 1) and a workgroup's whole life is one pass over K;
//   * the S16 halo of the tile ((ROWS + 2) x 34 pixels x C channels: 78 / 70 KB) goes through LDS in phases of 4 / 2 K chunks of plain 16-byte
//     copies (S16 entries need no conversion) alternating between two buffers (52 / 35 KB: three workgroups per CU for block 1, so that the
//     544 workgroups of a 4K layer are resident at once instead of 512 + a second round of 32), one barrier per phase and none inside the K loop;
//   * every weight fragment is used by exactly one wave (its output block), so the weights never touch LDS: each wave streams its
//     [chunk][tap][k half][32 rows][8 f16] slice from the L2 straight into MFMA operand registers, through a ring of 18 (block 0) / 10
//     (block 1: 162 VGPRs = three waves per SIMD) (chunk, tap) slots, one to two K chunks ahead of the matrix pipe.
// Measured (MI355X, ms per pair for the 8 launches of a block, same-call A/B against the per-tile kernels): 4K block 1 0.255 (0.242 with the
// two-buffer phases and three workgroups per CU) vs 0.285,
// block 0 0.239 vs 0.208 (272 workgroups streaming the same 663 KB of weights at once: the per-XCD L2 becomes the limit - the engine keeps
// the per-tile kernel there); 1080p block 1 0.119 vs 0.156, block 0 0.133 vs 0.152: 1418 vs 1307 frames/s.
// Output channels are permuted inside the 32-row block like in conv_t64_kernel (a lane ends up with 16 consecutive channels = one S16
// entry); products, accumulation order (chunk-major, taps in order, hi then lo, identity tap of the skip connection last) and epilogue
// are those of the per-tile kernels: results are bit-identical to them wherever they do not split K.
#pragma once
#include <type_traits>
#include "conv_t64.h"

namespace rife {

struct RowArgs {
    const unsigned char* in;     // S16 tensor, allocation start
    unsigned char* out;          // S16 tensor of the same geometry
    const unsigned char* img;    // weight image: per 32-channel output block [chunk C/16][tap 9][k half 2][row 32][8 f16], then bias[32], slope[32] (t64_img_nt(1, C / 16) bytes each)
    int H, W;                    // valid pixels
    int pitch;                   // pixels per plane row
    unsigned plane;              // bytes per plane
    int tiles_x, ntiles;
    // batched launch (rife_hip_process_batch): gridDim.y = nb > 0 pairs in flight, one S16 tensor pair each; the weights are shared,
    // so the workgroups of all pairs stream them from the L2 together and the small grids of the coarse blocks fill the chip in one round
    int nb = 0;
    const unsigned char* inb[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned char* outb[4] = {nullptr, nullptr, nullptr, nullptr};
};

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void for_each_slot(F&& f) {
    if constexpr (B < E) { f(std::integral_constant<int, B>{}); for_each_slot<B + 1, E>(f); }
}

// K chunks per halo load phase; the phases alternate between two LDS buffers (phase p + 1 is written at the end of phase p into the buffer phase
// p - 1 was read from: every wave has passed the barrier that ended phase p - 1 by then)
// (TAG bit 16, A/B: block 1 with phases of 4 chunks = 70 KB, two workgroups per CU)
template <int C, int TAG> constexpr int convrow_ph() { return (C == 128 && !(TAG & 16)) || C < 128 ? 2 : 4; }
template <int C, int ROWS, int TAG = 0> constexpr int convrow_lds_bytes() { return 2 * convrow_ph<C, TAG>() * (ROWS + 2) * 34 * 64; }

// waves per SIMD the register budget is set for: block 0 two workgroups x 6 waves, block 1 three workgroups x 4 waves per CU
template <int C, int TAG> constexpr int convrow_waves_per_simd() { return C == 128 && (TAG & 16) ? 2 : 3; }
template <int C, int ROWS, int TAG>
__global__ __launch_bounds__(2 * C) __attribute__((amdgpu_waves_per_eu(convrow_waves_per_simd<C, TAG>(), convrow_waves_per_simd<C, TAG>()))) void conv_row_kernel(RowArgs a) {
    constexpr int NW = C / 32, NTHR = 64 * NW, NCH = C / 16, IH = ROWS + 2, IW = 34, NPX = IH * IW;
    constexpr int PLANE = NPX * 32, CHB = 2 * PLANE;                     // bytes per (chunk, hi | lo) plane / per chunk in LDS
    constexpr int PH = convrow_ph<C, TAG>(), NPH = NCH / PH;            // K chunks per load phase, phases
    constexpr int PSLOTS = PH * 4 * NPX, NLD = (PSLOTS + NTHR - 1) / NTHR;      // 16-byte slots per phase, loads per thread and phase
    constexpr int WSTRIDE = t64_img_nt(1, NCH);                          // bytes per output block of the weight image
    static_assert(NCH % PH == 0, "whole load phases");
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* const lds = ldsb;
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned char* const tin = a.nb ? a.inb[blockIdx.y] : a.in;   // batched: the tensors of pair blockIdx.y
    unsigned char* const tout = a.nb ? a.outb[blockIdx.y] : a.out;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);              // wave = 32-channel output block
    const int h = lane >> 5, li = lane & 31;
    int L;
    {
        const int n = gridDim.x, b = blockIdx.x;
        const int q = n >> 3, r = n & 7, xcd = b & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);      // block b runs on XCD b % 8: contiguous bands of tiles per XCD
    }
    const int ty = L / a.tiles_x, tx = L - ty * a.tiles_x;
    const int oy0 = ty * ROWS, ox0 = tx * 32;
    const unsigned tb = (unsigned)(oy0 * a.pitch + ox0) * 32u;           // halo origin inside a plane: padded pixel (oy0, ox0) = pixel (oy0 - 1, ox0 - 1)

    // ---- halo staging: slot s of a phase = (chunk, plane, halo pixel P, 16-byte half); LDS is linear in s, the half-swap swizzle of
    // conv_t64.h (pos = half ^ bit 3 of P) is applied to the source address
    unsigned soff[NLD];
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int s = min(tid + k * NTHR, PSLOTS - 1);
        const int cc = s / (4 * NPX), rem = s - cc * (4 * NPX);
        const int pl = rem / (2 * NPX), rem2 = rem - pl * (2 * NPX);
        const int P = rem2 >> 1, pos = rem2 & 1;
        const int kh = pos ^ ((P >> 3) & 1);
        const int py = P / IW, px = P - py * IW;
        soff[k] = (unsigned)(2 * cc + pl) * a.plane + (unsigned)(py * a.pitch + px) * 32u + (unsigned)(kh * 16);
    }
    f32x4 stage[NLD];
#define ROW_LOAD(PHASE)                                                                                      \
    _Pragma("unroll") for (int k = 0; k < NLD; k++)                                                          \
        stage[k] = *reinterpret_cast<const f32x4*>(tin + (tb + (unsigned)(2 * PH * (PHASE)) * a.plane + soff[k]));
#define ROW_STORE(PHASE)                                                                                     \
    _Pragma("unroll") for (int k = 0; k < NLD; k++)                                                          \
        if (PSLOTS % NTHR == 0 || tid + k * NTHR < PSLOTS) *reinterpret_cast<f32x4*>(lds + ((PHASE) & 1) * PH * CHB + (tid + k * NTHR) * 16) = stage[k];

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
    f16x8 idf[2];                                                        // identity A fragments of the skip connection (conv_t64.h)
    {
        const int ch = s16_row_channel(li);
#pragma unroll
        for (int hc = 0; hc < 2; hc++)
#pragma unroll
            for (int e = 0; e < 8; e++) idf[hc][e] = ch == 16 * hc + 8 * h + e ? (_Float16)1.f : (_Float16)0.f;
    }
    f32x16 acc[ROWS];
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++)
#pragma unroll
        for (int q = 0; q < 16; q++) acc[rr][q] = 0.f;

    // weight fragments: a ring of RING (chunk, tap) slots, every slot refilled right after its use, i.e. always RING taps (two K chunks)
    // ahead of the matrix pipe: an L2 round trip is longer than the 18 MFMAs of one chunk
    constexpr int RING = (C == 128 && !(TAG & 16)) || C < 128 ? 10 : 18, NJ = NCH * 9;       // block 1: 10 slots keep the kernel inside 168 VGPRs
    f16x8 wr[RING];
#define ROW_WSLOT(J) wr[(J) % RING] = *reinterpret_cast<const f16x8*>(wsrc + (J) * 1024);
#define ROW_TAP(J)                                                                                           \
    {                                                                                                        \
        constexpr int c_ = (J) / 9, t_ = (J) % 9;                                                            \
        const unsigned char* const cb_ = lds + (((c_ / PH) & 1) * PH + c_ % PH) * CHB;                       \
        _Pragma("unroll") for (int rr = 0; rr < ROWS; rr++) {                                                \
            const f16x8 ah = *reinterpret_cast<const f16x8*>(cb_ + ao[rr][t_]);                              \
            const f16x8 al = *reinterpret_cast<const f16x8*>(cb_ + ao[rr][t_] + PLANE);                      \
            acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[(J) % RING], ah, acc[rr], 0, 0, 0);          \
            acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[(J) % RING], al, acc[rr], 0, 0, 0);          \
        }                                                                                                    \
        if ((J) + RING < NJ) ROW_WSLOT((J) + RING)                                                           \
        if (t_ == 8 && (c_ >> 1) == w) {      /* K chunk c_ carries the input channels of output block c_ >> 1: the skip connection */ \
            _Pragma("unroll") for (int rr = 0; rr < ROWS; rr++) {                                            \
                const f16x8 ah = *reinterpret_cast<const f16x8*>(cb_ + ao[rr][4]);                           \
                const f16x8 al = *reinterpret_cast<const f16x8*>(cb_ + ao[rr][4] + PLANE);                   \
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf[c_ & 1], ah, acc[rr], 0, 0, 0);         \
                acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(idf[c_ & 1], al, acc[rr], 0, 0, 0);         \
            }                                                                                                \
        }                                                                                                    \
    }

    ROW_LOAD(0)
    for_each_slot<0, RING>([&](auto j) { ROW_WSLOT(decltype(j)::value) });
    ROW_STORE(0)
    __syncthreads();
    for_each_slot<0, NPH>([&](auto pp) {
        constexpr int p = decltype(pp)::value;
        if (p + 1 < NPH) ROW_LOAD(p + 1)                                 // the next phase's halo chunks travel under this phase's matrix work
        for_each_slot<p * PH * 9, (p + 1) * PH * 9>([&](auto j) { ROW_TAP(decltype(j)::value) });
        if (p + 1 < NPH) {
            ROW_STORE(p + 1)                                             // the other buffer: last read in phase p - 1
            __syncthreads();
        }
    });
#undef ROW_LOAD
#undef ROW_STORE
#undef ROW_WSLOT
#undef ROW_TAP

    // ---- epilogue: y = slope(acc + bias) -> the hi / lo entries of chunk 2 w + h
    const float* const bs = reinterpret_cast<const float*>(a.img + (size_t)w * WSTRIDE + (size_t)NCH * t64_wch(1));
#pragma unroll
    for (int rr = 0; rr < ROWS; rr++) {
        const int oy = oy0 + rr, ox = ox0 + li;
        const bool ok = oy < a.H && ox < a.W;
        float v[16];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bs + 16 * h + 4 * q);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(bs + 32 + 16 * h + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float y = acc[rr][4 * q + k] + b4[k];
                v[4 * q + k] = y < 0.f ? y * s4[k] : y;
            }
        }
        s16_store_chunk(v, tout + ((size_t)(2 * (2 * w + h)) * a.plane + ((size_t)(oy + 1) * a.pitch + ox + 1) * 32), a.plane, ok);
    }
}

}  // namespace rife
