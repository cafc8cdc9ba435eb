// conv_rs_kernel: the 64 -> 64 channel residual trunk convolution of the finest IFBlock of rife-v4.6 (reference
// models/rife-v4.6/flownet.param:169-197: Split, Convolution 3x3 pad 1, BinaryOp add, ReLU slope 0.2; 8 launches per pair, 44 % of the pair's
// MACs) as a ROW-STREAMING persistent kernel with specialised waves.  Round 3; replaces conv_t64_kernel<3, 2> (conv_t64.h), whose load, store
// and matrix phases added up (22 + 27 + 43 us = the 82 - 88 us of a 4K launch) instead of overlapping, because every wave both fed the matrix
// pipe and issued the tile's LDS-DMA pieces and stores, and a wave that waits for a slot in the CU's memory queue issues no MFMAs.
//
// One workgroup per CU, 8 waves, three jobs:
//   * waves 0-3, "consumers" (one per SIMD): wave k owns output block n = k & 1 (32 channels) of output row (k >> 1) of the current row
//     pair.  Its 36 weight fragments ([K chunk 4][tap 9] x 16 bytes per lane = 144 VGPRs) are loaded ONCE per launch and stay in registers:
//     no weight ever passes through LDS (conv_t64 re-streamed the 72 KB of weights per 8 x 32 tile: 1.2 GB of L2 -> LDS traffic per 4K
//     launch, 4 x the activations).  A consumer issues ds_read_b128 (pixel fragments, three MFMA pairs ahead of their use - across the step
//     boundary too) and MFMAs in two accumulation chains (hi products, lo products), and writes the raw fp32 sums of its row to an LDS
//     staging buffer.  It never touches vector memory after the prologue and does no epilogue arithmetic.
//   * waves 4-5, "loaders": LDS-DMA (global_load_lds_dwordx4, 1 KiB per instruction, no registers) of the halo ROWS three steps ahead into a
//     ring of 12 row slots.  A workgroup walks DOWN (or up) a 32-column strip, so every input row is loaded once per strip instead of once
//     per 8-row tile: 34 / 32 of the tensor instead of 10 x 34 / (8 x 32) = 1.33 x.
//   * waves 6-7, "storers": bias, LeakyReLU and the {hi, lo} split of the row pair staged two steps ago, then 1 KiB contiguous per store.
//   Loads and stores are issued by different waves because vmcnt only retires in order within one kind of access: the loaders' counted
//   waits (vmcnt(9) / vmcnt(18): the rows of the steps after next may still be in flight) would be meaningless with stores in the same queue.
// A step = one row pair of one strip = 76 MFMAs per consumer (2,432 cycles of its SIMD's matrix pipe) and ends with ONE s_barrier:
//   iteration it:  consumers  MFMAs of step it -> staging[it % 3]; first fragments of step it + 1
//                  loaders    rows of step it + 3 -> ring; wait for the rows of step it + 2
//                  storers    staging[(it - 2) % 3] -> epilogue -> global
// Work split: the tiles_x x ceil(H / 2) (strip, row pair) units in strip-major order are cut into gridDim.x equal contiguous ranges (4K: 8,160
// units over 256 workgroups = 31 or 32 steps each); a range that crosses into the next strip restarts the ring there (4 fresh rows).
// Tensors are the S16 tensors of conv_t64.h ({hi, lo} f16 planes per 16-channel chunk, one pixel of zero border), the weight image is
// conv_t64's (pack_t64_image), the products are conv_t64_kernel's.  The sums differ from conv_t64's in the last bits (two chains added at the
// end instead of one): tests/test_gpu_t64.py and tools/rs_bench.py hold the two kernels together (<= 1 LSB on the u8 frame, < 1e-3 of the
// bytes differ, block-3 flows to 1e-4) and check that conv_rs itself is run-to-run identical.
// LDS: ring 12 x 8,704 B (row slot = [chunk 4][hi | lo][34 px][32 B], halves swapped where bit 3 of the column is set: conflict-free
// ds_read_b128, applied to the DMA source addresses) + staging 3 x 2 rows x 8 KiB of fp32 sums + bias / slopes = 154,112 B.
// Measured (MI355X, 4K, same-call A/B, profiles/r3/rs_bench.txt): 73.5 - 76 us per launch in the pass against 81.6 - 82.9 for conv_t64; matrix work
// alone 45.8 us (2.22 GHz), memory alone 49.9 us, both 60.1 us at the 1.88 GHz the chip settles at under this load: the phases overlap.
#pragma once
#include <type_traits>
#include "conv_t64.h"
#include "conv_row.h"

namespace rife {

constexpr int RS_NR = 12;                                  // ring row slots
constexpr int RS_SEG = 34 * 32;                            // one (chunk, hi | lo) segment of a halo row: 1,088 B
constexpr int RS_ROWB = 8 * RS_SEG;                        // 8,704 B per halo row (64 channels x {hi, lo})
constexpr int RS_STG_ROW = 8 * 1024;                       // staged output row: [chunk 4][hi | lo][32 px][32 B]
constexpr int RS_LDS_RING = 0;
constexpr int RS_LDS_STG = RS_NR * RS_ROWB;                // 104,448
constexpr int RS_NSTG = 3;                                 // staging buffers (a row pair each): written in iteration it, read in it + 2
constexpr int RS_LDS_BS = RS_LDS_STG + RS_NSTG * 2 * RS_STG_ROW; // 153,600: bias[64] | slope[64]
constexpr int RS_LDS = RS_LDS_BS + 512;                    // 154,112 B: one workgroup per CU
constexpr int RS_AHEAD = 3;                                // the loaders run three steps ahead of the consumers
constexpr int RS_MIN_PAIRS = 4;                            // ring capacity: at most one strip change among RS_AHEAD + 1 consecutive steps (4 + 4 + 2 + 2 rows)
constexpr int RS_NTHR = 512;

struct RsArgs {
    const unsigned char* in;     // S16 tensor, allocation start (= pixel (-1, -1) of plane 0)
    unsigned char* out;          // S16 tensor of the same geometry
    const unsigned char* img;    // conv_t64's weight image of a 64 -> 64 layer (T64_IMG bytes)
    int H, W;                    // valid pixels
    int pitch;                   // pixels per plane row
    unsigned plane;              // bytes per plane
    int npairs;                  // row pairs per strip = ceil(H / 2)
    int nunits;                  // tiles_x * npairs
    int descend;                 // 1: every workgroup walks its range last unit first, rows bottom-up (consecutive layers alternate)
    long long* stamps = nullptr; // bench builds RIFE_ABL(TAG & RS_CLK): [workgroup][4] = shader cycles of the workgroup's life, start, end (100 MHz counter)
};
// bench-only ablation bits of TAG (timing experiments; results are garbage).  The product instantiates TAG = 0.
enum { RS_NOSTORE = 0x100, RS_NODMA = 0x200, RS_NOMATH = 0x400, RS_CLK = 0x40000, RS_PRIO = 0x2000, RS_NTLOAD = 0x4000, RS_NTSTORE = 0x8000 };

// (strip, pair) cursor over a workgroup's unit range, walking up or down
struct RsCursor {
    int strip, p;
    __device__ __forceinline__ void init(int u, int npairs) { strip = u / npairs; p = u - strip * npairs; }
    // advance by one step; returns true if the next step starts a new strip (its ring rows are all fresh)
    __device__ __forceinline__ bool advance(int npairs, int descend) {
        if (descend) { if (--p < 0) { p = npairs - 1; --strip; return true; } }
        else if (++p >= npairs) { p = 0; ++strip; return true; }
        return false;
    }
};

// 64 lanes x 16 bytes global -> LDS, not tracked by the compiler: LDS destination = dst (wave-uniform) + lane * 16, source = base + voff
template <int TAG>
__device__ __forceinline__ void rs_dma16(const unsigned char* base, unsigned voff, unsigned dst) {
    unsigned keep;
    if RIFE_ABL(TAG & RS_NTLOAD)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(base) : "memory");
}

#define RS_SYNC_LGKM() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define RS_SYNC_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")
#define RS_SYNC_BARE() asm volatile("s_barrier" ::: "memory")

// The matrix work of one (row, output block N) is a fixed sequence of 38 MFMA pairs (A fragment, pixel fragment {hi, lo}): per K chunk c the
// nine taps in order, then - if chunk c carries the input channels of output block N - the identity tap of the skip connection
// (conv_t64_kernel's order: the accumulation is bit-identical).
struct RsPairDesc { int c, t; bool idn; };
__host__ __device__ constexpr RsPairDesc rs_pair(int N, int m) {
    for (int c = 0; c < 4; c++) {
        const int cnt = 9 + ((c >> 1) == N ? 1 : 0);
        if (m < cnt) return m < 9 ? RsPairDesc{c, m, false} : RsPairDesc{c, 4, true};
        m -= cnt;
    }
    return RsPairDesc{0, 0, false};
}
constexpr int RS_NPAIR = 38, RS_PF = 3;      // pairs per row; fragment prefetch distance in pairs (RS_PF + 1 fragment register sets)

// consumer wave: output block N of row `ro` of every row pair of the workgroup's range (see the header of this file)
template <int N, int TAG>
__device__ __forceinline__ void rs_consumer(const RsArgs& a, unsigned char* const ldsb, const int ro, const int lane, const int S, const int ufirst) {
    const int h = lane >> 5, li = lane & 31;
    // weights of output block N: [chunk][tap] A fragments, lane (li, h) = row li, k half h (conv_t64's image: [chunk][tap][k half][64 rows][8 f16])
    f16x8 W[4][9];
    {
        const unsigned char* wsrc = a.img + h * 1024 + (N * 32 + li) * 16;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int t = 0; t < 9; t++) W[c][t] = *reinterpret_cast<const f16x8*>(wsrc + c * t64_wch(2) + t * 2048);
    }
    f16x8 idf[2];                                                        // identity A fragments of the skip connection (conv_t64.h)
    {
        const int ch = s16_row_channel(li);
#pragma unroll
        for (int hc = 0; hc < 2; hc++)
#pragma unroll
            for (int e = 0; e < 8; e++) idf[hc][e] = ch == 16 * hc + 8 * h + e ? (_Float16)1.f : (_Float16)0.f;
    }
    unsigned colo[3];                                                    // column part of the fragment addresses
#pragma unroll
    for (int dx = 0; dx < 3; dx++) { const int px = li + dx; colo[dx] = (unsigned)(px * 32 + ((h ^ ((px >> 3) & 1)) << 4)); }
    // staged record of this lane: 16 fp32 = channels 32 N + 16 h .. + 15 of pixel li, the four 16-byte quads XOR-swizzled by (li >> 1) & 3
    // (the eight lanes of a ds_write_b128 group then cover all 32 banks)
    unsigned char* const stg = ldsb + RS_LDS_STG + ro * RS_STG_ROW + ((2 * N + h) * 32 + li) * 64;
    const int qs = (li >> 1) & 3;

    RsCursor cur; cur.init(ufirst, a.npairs);
    int sq = 0;                                                          // rows loaded through the step `cur` points at, mod RS_NR
    unsigned ad[9];                                                      // fragment addresses of the nine taps, chunk 0, hi plane
    // addresses of the step at `cur` (fresh: it starts a strip, all four halo rows are new)
    auto step_addresses = [&](const bool fresh) {
        sq += fresh ? 4 : 2; if (sq >= RS_NR) sq -= RS_NR;
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int j = ro + dy;                                       // row of the step's four halo rows, from the top
            int sl = a.descend ? sq + RS_NR - 1 - j : sq + RS_NR - 4 + j;
            if (sl >= RS_NR) sl -= RS_NR;
            const unsigned rb = (unsigned)(RS_LDS_RING + sl * RS_ROWB);
#pragma unroll
            for (int dx = 0; dx < 3; dx++) ad[dy * 3 + dx] = rb + colo[dx];
        }
    };
    constexpr int NF = RS_PF + 1;                                        // fragment register sets; pair m of a step of parity PAR uses set (m + 2 PAR) % NF (38 % 4 = 2)
    static_assert(NF == 4 && RS_NPAIR % NF == 2, "fragment set rotation across steps");
    f16x8 fh[NF], fl[NF];
    int it = 0;
    // One step.  Two accumulation chains, the hi products and the lo products: an MFMA never waits for the result of the one issued just
    // before it (one wave per SIMD: nobody else would fill the gap).  The pixel fragments are read RS_PF pairs ahead of their MFMAs -
    // across the step boundary too: the last RS_PF pairs of a step read the first fragments of the NEXT step, whose rows the loaders
    // guarantee one barrier early - and sched_barrier keeps the compiler from sinking the reads back to their uses.
    auto step = [&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        auto frag_read = [&](auto mc) {                                  // pair m of the step whose addresses are in ad[]
            constexpr int m = decltype(mc)::value;
            constexpr RsPairDesc d = rs_pair(N, m % RS_NPAIR);
            constexpr int st = (m + 2 * PAR) % NF;
            fh[st] = *reinterpret_cast<const f16x8*>(ldsb + ad[d.t] + d.c * (2 * RS_SEG));
            fl[st] = *reinterpret_cast<const f16x8*>(ldsb + ad[d.t] + d.c * (2 * RS_SEG) + RS_SEG);
        };
        f32x16 accH, accL;
        if (!RIFE_ABL(TAG & RS_NOMATH)) {
            for_each_slot<0, RS_NPAIR>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr RsPairDesc d = rs_pair(N, m);
                constexpr int st = (m + 2 * PAR) % NF;
                if constexpr (m == RS_NPAIR - RS_PF) step_addresses(cur.advance(a.npairs, a.descend));     // ad[] is dead: every read of this step is issued
                frag_read(std::integral_constant<int, m + RS_PF>{});     // m + RS_PF >= 38: pair m + RS_PF - 38 of the next step (sets continue to rotate)
                const f16x8 A = d.idn ? idf[d.c & 1] : W[d.c][d.t];
                if constexpr (m == 0) {
                    f32x16 z;
#pragma unroll
                    for (int q = 0; q < 16; q++) z[q] = 0.f;
                    accH = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, fh[st], z, 0, 0, 0);
                    accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, fl[st], z, 0, 0, 0);
                } else {
                    accH = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, fh[st], accH, 0, 0, 0);
                    accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, fl[st], accL, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) { accH[q] = (float)lane; accL[q] = 0.f; }
        }
        // raw sums -> staging[it % 3]; bias, LeakyReLU and the {hi, lo} split are the storers' work two iterations later, by when these
        // writes have long completed (no wait here: the LDS executes a wave's operations in order, and the next step's reads follow them)
        int sb = it % RS_NSTG;
        f32x4* const d4 = reinterpret_cast<f32x4*>(stg + sb * (2 * RS_STG_ROW));
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = accH[4 * q + k] + accL[4 * q + k];
            d4[q ^ qs] = v;
        }
        RS_SYNC_BARE();
        it++;
    };
    RS_SYNC_LGKM();                                                      // ring rows of steps 0 and 1 landed (loaders), bias in LDS
    step_addresses(true);
    if (!RIFE_ABL(TAG & RS_NOMATH)) for_each_slot<0, RS_PF>([&](auto mc) {       // first fragments of step 0 (parity 0)
        constexpr int m = decltype(mc)::value;
        constexpr RsPairDesc d = rs_pair(N, m);
        fh[m % NF] = *reinterpret_cast<const f16x8*>(ldsb + ad[d.t] + d.c * (2 * RS_SEG));
        fl[m % NF] = *reinterpret_cast<const f16x8*>(ldsb + ad[d.t] + d.c * (2 * RS_SEG) + RS_SEG);
    });
    while (it + 1 < S) { step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); }
    if (it < S) step(std::integral_constant<int, 0>{});
    RS_SYNC_LGKM();                                                      // the last step's staging writes have landed
}

// SPLIT (round 4, A/B RIFE_HIP_RS_SPLIT): 1 = the epilogue of a row pair is shared by all four io waves - loader j finishes the chunks 0, 1 of row j
// (before it issues the step's LDS-DMA: its stores are then OLDER than every load its counted vmcnt wait leaves in flight, so the wait stays
// sufficient, merely stricter), storer j the chunks 2, 3: bias / LeakyReLU / split VALU and the global stores on all four SIMDs instead of two.
template <int TAG, int SPLIT = 0>
__global__ __launch_bounds__(RS_NTHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_rs_kernel(RsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);             // wave role (wave-uniform by construction)
    long long clk0 = 0, rt0 = 0;
    if RIFE_ABL(TAG & RS_CLK) { clk0 = (long long)__builtin_readcyclecounter(); rt0 = (long long)__builtin_amdgcn_s_memrealtime(); }

    const int nwg = gridDim.x, b = blockIdx.x;
    const int u0 = (int)((long long)a.nunits * b / nwg), u1 = (int)((long long)a.nunits * (b + 1) / nwg);
    const int S = u1 - u0;                                               // steps of this workgroup
    if (S <= 0) return;
    const int ufirst = a.descend ? u1 - 1 : u0;
    // every iteration ends with RS_SYNC_*: all eight waves execute the same S + 2 of them

    // epilogue of chunks [cc0, cc1) of output row j of the step staged in buffer `sb`, at (strip, pair) = (cstrip, cp): y = slope(sum + bias), split into
    // {hi, lo}, zeros outside the valid pixels; lane = (pixel l >> 1, half l & 1) of a 16-channel chunk: 16 bytes of the hi plane and 16 of the lo plane
    auto epilogue = [&](const int j, const int sb, const int cstrip, const int cp, const int cc0, const int cc1, const float slope) {
        const int px = lane >> 1, jh = lane & 1, qs = (px >> 1) & 3;
        const int y = 2 * cp + j, x0 = 32 * cstrip;
        if (!(y < a.H) || RIFE_ABL(TAG & RS_NOSTORE)) return;
        const unsigned okmask = x0 + px < a.W ? 0xffffffffu : 0u;
        const unsigned char* src = ldsb + RS_LDS_STG + sb * (2 * RS_STG_ROW) + j * RS_STG_ROW + px * 64;
        unsigned char* dst = a.out + ((unsigned)((y + 1) * a.pitch + x0 + 1) * 32u + (unsigned)(lane * 16));
        for (int cc = cc0; cc < cc1; cc++) {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(src + cc * 2048 + (((2 * jh) ^ qs) << 4));
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(src + cc * 2048 + (((2 * jh + 1) ^ qs) << 4));
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(ldsb + RS_LDS_BS + (16 * cc + 8 * jh) * 4);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(ldsb + RS_LDS_BS + (16 * cc + 8 * jh + 4) * 4);
            f16x8 hv, lv;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float yv = (e < 4 ? r0[e & 3] : r1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
                float v = yv < 0.f ? yv * slope : yv;
                v = __uint_as_float(__float_as_uint(v) & okmask);
                const _Float16 hh = (_Float16)v;
                hv[e] = hh;
                lv[e] = (_Float16)(v - (float)hh);
            }
            *reinterpret_cast<f16x8*>(dst + (size_t)(2 * cc) * a.plane) = hv;
            *reinterpret_cast<f16x8*>(dst + (size_t)(2 * cc + 1) * a.plane) = lv;
        }
    };
    if (wv < 4) {
        // ------------------------------------------------------------------------------------------------ consumers
        if RIFE_ABL(TAG & RS_PRIO) __builtin_amdgcn_s_setprio(2);             // A/B only: raised priority of the matrix waves measured 3 - 5 % SLOWER (their loader / storer starve, the barrier waits)
        if (wv == 0 && lane < 32) reinterpret_cast<f32x4*>(ldsb + RS_LDS_BS)[lane] = reinterpret_cast<const f32x4*>(a.img + 4 * t64_wch(2))[lane];
        if (wv & 1) rs_consumer<1, TAG>(a, ldsb, wv >> 1, lane, S, ufirst);
        else rs_consumer<0, TAG>(a, ldsb, wv >> 1, lane, S, ufirst);
    } else if (wv < 6) {
        // ------------------------------------------------------------------------------------------------ loaders
        const int j = wv - 4;
        // piece i of a row covers LDS units 64 i .. 64 i + 63 (16 bytes each) of the 544 of a row slot: unit = (segment, pixel, half)
        unsigned soff[9];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int u = min(i * 64 + lane, 543);
            const int seg = u / 68, within = u - seg * 68;
            const int px = within >> 1, pos = within & 1;
            const int kh = pos ^ ((px >> 3) & 1);
            soff[i] = (unsigned)seg * a.plane + (unsigned)(px * 32 + kh * 16);
        }
        RsCursor cur; cur.init(ufirst, a.npairs);
        int sq = 0;                                                      // rows loaded so far, mod RS_NR
        // the new rows of the step at `cur` (4 if fresh, else 2), in ring order; this wave takes rows j and j + 2.  Returns its piece count.
        auto load_step = [&](bool fresh) -> int {
            const int nnew = fresh ? 4 : 2;
            const int y = 2 * cur.p, x0 = 32 * cur.strip;
            int mine = 0;
            for (int i = j; i < nnew; i += 2) {
                const int prow = a.descend ? y + nnew - 1 - i : y + 4 - nnew + i;       // padded row index (pixel row prow - 1)
                int sl = sq + i; if (sl >= RS_NR) sl -= RS_NR;
                if (!RIFE_ABL(TAG & RS_NODMA)) {
                    const unsigned rowoff = (unsigned)(prow * a.pitch + x0) * 32u;
                    const unsigned dst = (unsigned)(RS_LDS_RING + sl * RS_ROWB);
#pragma unroll
                    for (int k = 0; k < 8; k++) rs_dma16<TAG>(a.in, rowoff + soff[k], dst + k * 1024);
                    if (lane < 32) rs_dma16<TAG>(a.in, rowoff + soff[8], dst + 8 * 1024);
                }
                mine += 9;
            }
            sq += nnew; if (sq >= RS_NR) sq -= RS_NR;
            return mine;
        };
        if RIFE_ABL(TAG & RS_NODMA) { for (int i = lane + 64 * j; i < RS_LDS_STG / 16; i += 128) reinterpret_cast<f32x4*>(ldsb)[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // prologue: rows of steps 0 .. RS_AHEAD - 1; steps 0 and 1 must have landed before the first barrier
        load_step(true);
        int ahead = 0;                                                   // my pieces of the newest step issued
        int loaded = 1;                                                  // steps issued
        for (; loaded < RS_AHEAD && loaded < S; loaded++) { const bool f = cur.advance(a.npairs, a.descend); ahead = load_step(f); }
        if (loaded < RS_AHEAD) ahead = 0;                                // fewer than RS_AHEAD steps in the range: wait for everything
#define RS_WAIT_AHEAD()                                                                                      \
        if RIFE_ABL(TAG & RS_NODMA) RS_SYNC_LGKM();                                                                  \
        else if (ahead == 0) RS_SYNC_VM(0);                                                                  \
        else if (ahead == 9) RS_SYNC_VM(9);                                                                  \
        else RS_SYNC_VM(18);
        RS_WAIT_AHEAD()                                                  // rows of steps 0 and 1 landed
        if (SPLIT) {
            // my half of the epilogue (chunks 0, 1 of row j of the step staged two iterations ago), then the loads: S + 2 iterations like the storers
            const float slope = reinterpret_cast<const float*>(a.img + 4 * t64_wch(2))[64];
            RsCursor ecur; ecur.init(ufirst, a.npairs);
            for (int it = 0; it <= S + 1; it++) {
                if (it >= 2) { epilogue(j, (it - 2) % RS_NSTG, ecur.strip, ecur.p, 0, 2, slope); ecur.advance(a.npairs, a.descend); }
                if (it < S) {
                    if (it + RS_AHEAD < S) { const bool f = cur.advance(a.npairs, a.descend); ahead = load_step(f); }
                    else ahead = 0;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of the staging buffer are done before the barrier lets the consumers reuse it
                    RS_WAIT_AHEAD()
                } else if (it == S) RS_SYNC_LGKM();                      // the consumers' final barrier
            }
        } else {
        for (int it = 0; it < S; it++) {
            if (it + RS_AHEAD < S) { const bool f = cur.advance(a.npairs, a.descend); ahead = load_step(f); }
            else ahead = 0;
            RS_WAIT_AHEAD()                                              // rows of step it + 2 landed: the consumers prefetch step it + 1's first fragments before the NEXT barrier
        }
        RS_SYNC_BARE();                                                  // the consumers' final barrier
        }
#undef RS_WAIT_AHEAD
    } else {
        // ------------------------------------------------------------------------------------------------ storers
        // wave j finishes output row j of the pair staged by the consumers in the previous iteration: y = slope(sum + bias), split into
        // {hi, lo}, zeros outside the valid pixels; lane = (pixel l >> 1, half l & 1) of a 16-channel chunk: 8 values in, 16 bytes of the
        // hi plane and 16 of the lo plane out, 1 KiB contiguous per store instruction
        const int j = wv - 6;
        const int px = lane >> 1, jh = lane & 1, qs = (px >> 1) & 3;
        const float slope = reinterpret_cast<const float*>(a.img + 4 * t64_wch(2))[64];      // one LeakyReLU slope for the whole layer (pack_t64_image)
        RsCursor cur; cur.init(ufirst, a.npairs);
        RS_SYNC_LGKM();
        for (int it = 0; it <= S + 1; it++) {
            if (it >= 2) {                                               // step it - 2, staged at the end of iteration it - 2
                const int y = 2 * cur.p + j, x0 = 32 * cur.strip;
                if (SPLIT) epilogue(j, (it - 2) % RS_NSTG, cur.strip, cur.p, 2, 4, slope);
                else if (y < a.H && !RIFE_ABL(TAG & RS_NOSTORE)) {
                    const unsigned okmask = x0 + px < a.W ? 0xffffffffu : 0u;
                    const unsigned char* src = ldsb + RS_LDS_STG + ((it - 2) % RS_NSTG) * (2 * RS_STG_ROW) + j * RS_STG_ROW + px * 64;
                    unsigned char* dst = a.out + ((unsigned)((y + 1) * a.pitch + x0 + 1) * 32u + (unsigned)(lane * 16));
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) {
                        const f32x4 r0 = *reinterpret_cast<const f32x4*>(src + cc * 2048 + (((2 * jh) ^ qs) << 4));
                        const f32x4 r1 = *reinterpret_cast<const f32x4*>(src + cc * 2048 + (((2 * jh + 1) ^ qs) << 4));
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(ldsb + RS_LDS_BS + (16 * cc + 8 * jh) * 4);
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(ldsb + RS_LDS_BS + (16 * cc + 8 * jh + 4) * 4);
                        f16x8 hv, lv;
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const float yv = (e < 4 ? r0[e & 3] : r1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
                            float v = yv < 0.f ? yv * slope : yv;
                            v = __uint_as_float(__float_as_uint(v) & okmask);
                            const _Float16 hh = (_Float16)v;
                            hv[e] = hh;
                            lv[e] = (_Float16)(v - (float)hh);
                        }
                        if RIFE_ABL(TAG & RS_NTSTORE) {
                            __builtin_nontemporal_store(hv, reinterpret_cast<f16x8*>(dst + (size_t)(2 * cc) * a.plane));
                            __builtin_nontemporal_store(lv, reinterpret_cast<f16x8*>(dst + (size_t)(2 * cc + 1) * a.plane));
                        } else {
                            *reinterpret_cast<f16x8*>(dst + (size_t)(2 * cc) * a.plane) = hv;
                            *reinterpret_cast<f16x8*>(dst + (size_t)(2 * cc + 1) * a.plane) = lv;
                        }
                    }
                }
                cur.advance(a.npairs, a.descend);
            }
            if (it < S) RS_SYNC_LGKM();                                  // barrier of iteration it
            else if (it == S) RS_SYNC_LGKM();                            // the consumers' final barrier: the last step's staging writes have landed
        }
    }
    if (RIFE_ABL(TAG & RS_CLK) && tid == 0) {
        a.stamps[4 * blockIdx.x] = (long long)__builtin_readcyclecounter() - clk0;
        a.stamps[4 * blockIdx.x + 1] = rt0;
        a.stamps[4 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace rife
