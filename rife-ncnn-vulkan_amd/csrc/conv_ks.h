// conv_ks_kernel<C, NB, CPW>: the C -> C residual trunk convolutions of the COARSE IFBlocks of rife-v4.6 (block 1: 128 channels at 1/16 resolution,
// block 0: 192 at 1/32, block 2: 96 at 1/8; reference models/rife-v4.6/flownet.param:14-42, 66-94, 119-147: Split, Convolution 3x3 pad 1,
// BinaryOp add, ReLU slope 0.2) as a ROW-STREAMING, WEIGHT-STATIONARY kernel with the K dimension split over the consumer waves.  Round 4.
//
// Why: conv_row_kernel (round 2) gives every 32-pixel tile its own workgroup and streams ALL C x C x 9 weights through it from the L2:
// 544 workgroups x 295 KB = 160 MB of L2 -> register traffic per 4K launch of block 1, 28 - 30 us per launch against a matrix floor of
// 9 us; conv_rs_kernel (round 3) keeps the weights in registers but only has room for 64 channels (144 VGPRs per consumer).  Here the
// weights stay in registers for wider layers by cutting BOTH dimensions of the weight matrix:
//   * N: a workgroup computes NB 32-channel output blocks (C = 128: 64 of the 128 output channels); the C / (32 NB) "N groups" are separate
//     workgroups that read the same input rows (coarse tensors are small and L2 resident);
//   * K: consumer wave (n, kq) holds the weights of output block n for CPW of the C / 16 input-channel chunks only: 9 CPW fragments = 18 - 27
//     x 4 VGPRs, loaded ONCE per launch.  Its partial sums (fp32) go to an LDS staging buffer; the storer waves add the KS = C / (16 CPW) partials in a fixed
//     order, then bias, LeakyReLU / per-channel slopes, the {hi, lo} split and the stores (conv_rs.h's epilogue).
// The rest is conv_rs_kernel's machinery: S16 tensors (conv_t64.h), a ring of halo ROWS in LDS filled by loader waves with LDS-DMA (1 KiB per
// instruction, counted vmcnt waits), two s_barriers per step (X "staging full" / Y "staging free": ONE staging buffer, the LDS goes to a deeper ring; the consumers keep the sums of the previous step in a second accumulator set, so the handshake hides under the next step's MFMAs), storers one step behind.  A step = ONE output row of one 32-column strip (the coarse
// layers have 34 - 136 rows per strip: row pairs would leave a third of a workgroup's range as remainder).
// Work split: (strip, row) units in strip-major order are cut into equal contiguous ranges, one per workgroup of an N group; a range is walked
// as SEGMENTS (consecutive rows of one strip), each with its own prologue and drain (the host picks a range count that is a multiple of the strip
// count, so a range normally is one segment).
// Waves: NCON = NB * KS consumers (waves 0 ..), then KS_NLD = 2 loaders, then NST = 2 (or 1) storers: at most twelve, three per SIMD.
//   C = 128: NB 2, CPW 2 -> KS 4, 8 consumers, 12 waves (three per SIMD: 168 VGPRs), 2 N groups; LDS: ring 7 x 17,408 + staging 32 KiB
//   C =  96: NB 3, CPW 2 -> KS 3, 9 consumers, 12 waves (one storer), 1 N group;                                LDS: ring 9 x 13,056 + staging 36 KiB
//   C = 192: NB 2, CPW 3 -> KS 4, 8 consumers, 12 waves, 3 N groups;                               LDS: ring 4 x 26,112 + staging 1 x 32 KiB ... (see KsCfg)
// Arithmetic: the products are those of conv_row_kernel / conv_rs_kernel (fp16 weights x {hi, lo} activations, fp32 accumulation, identity tap
// for the skip connection); the summation order is this kernel's own (per consumer: chunk-major, taps in order, hi then lo; then the partials kq = 0 .. KS - 1;
// then the bias), deterministic by construction.  OPT-IN (RIFE_HIP_KS) and compiled into the test / bench builds only: measured slower with pairs in
// flight (profiles/r4/ks_ab.txt), so the product does not carry it.  tests/test_gpu_ks.py holds it against the kernels it replaces (<= 1 LSB on the
// frame, flows to 1e-4) and, directly, within 1 LSB of the CPU restatement of the reference that the tests check against.
#pragma once
#include <type_traits>
#include "conv_rs.h"

namespace rife {

constexpr int KS_NLD = 2;

template <int C, int NB, int CPW>
struct KsCfg {
    static constexpr int NCH = C / 16;                       // K chunks of 16 input channels
    static constexpr int KS = NCH / CPW;                     // K slices = partial sums per output
    static constexpr int NCON = NB * KS;                     // consumer waves
    static constexpr int NG = C / (32 * NB);                 // N groups (separate workgroups)
    static constexpr int NST = NCON + KS_NLD + 2 <= 12 ? 2 : 1;   // storer waves: twelve waves = three per SIMD = 168 VGPRs for the consumers
    static constexpr int NWAVES = NCON + KS_NLD + NST;
    static constexpr int NTHR = 64 * NWAVES;
    static constexpr int ROWB = NCH * 2 * RS_SEG;            // one halo row in the ring: [chunk][hi | lo][34 px][32 B]
    static constexpr int UNITS = ROWB / 16;                  // 16-byte units per row
    static constexpr int PIECES = (UNITS + 63) / 64;         // LDS-DMA instructions per row
    static constexpr int STGB = NCON * 4096;                 // one staging buffer: [kq][n][half h][32 px][64 B] of fp32 partial sums
    static constexpr int NSTG = 1;                           // one staging buffer: two barriers per step ("staging free", "staging full"), a deeper ring instead
    static constexpr int NR = (160 * 1024 - NSTG * STGB - NB * 256 - 1024) / ROWB;      // ring row slots
    static constexpr int LA = NR - 3;                        // rows the loaders run ahead of the step that needs them
    static constexpr int LDS_RING = 0;
    static constexpr int LDS_STG = NR * ROWB;
    static constexpr int LDS_BS = LDS_STG + NSTG * STGB;     // per output block: bias[32] | slope[32]
    static constexpr int LDS = LDS_BS + NB * 256;
    static constexpr int WAVES_PER_SIMD = (NWAVES + 3) / 4;
    static_assert(NCH % CPW == 0 && C % (32 * NB) == 0, "whole K slices and N groups");
    static_assert(NR >= 4 && LA >= 1 && LA <= 7, "ring: three rows in use and at least one ahead");
    static_assert(NWAVES <= 16, "workgroup size");
};

struct KsArgs {
    const unsigned char* in;     // S16 tensor, allocation start (= pixel (-1, -1) of plane 0)
    unsigned char* out;          // S16 tensor of the same geometry
    const unsigned char* img;    // conv_row's weight image: per 32-channel output block [chunk][tap][k half][row 32][8 f16], bias[32], slope[32]
    int H, W;                    // valid pixels
    int pitch;                   // pixels per plane row
    unsigned plane;              // bytes per plane
    int nunits;                  // tiles_x * H
    int skip;                    // 1: residual layer (identity tap); 0: plain convolution (rife-v2.3 trunks)
    // batched launch (rife_hip_process_batch): gridDim.y = nb > 0 pairs in flight, one S16 tensor pair each, shared weights
    int nb = 0;
    const unsigned char* inb[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned char* outb[4] = {nullptr, nullptr, nullptr, nullptr};
    long long* stamps = nullptr; // bench builds RIFE_ABL(TAG & KS_CLK): [workgroup][16] times on the constant 100 MHz counter (tools/ks_bench.py)
};
// bench-only ablation bits of TAG (timing experiments; results are garbage).  The product instantiates TAG = 0.
enum { KS_NOSTORE = 0x100, KS_NODMA = 0x200, KS_NOMATH = 0x400, KS_NOWEIGHTS = 0x800, KS_CLK = 0x40000 };
#define KS_STAMP(I) if (RIFE_ABL(TAG & KS_CLK) && lane == 0) a.stamps[16 * (blockIdx.x + gridDim.x * blockIdx.y) + (I)] = (long long)__builtin_amdgcn_s_memrealtime();

// s_waitcnt vmcnt(N * P) for a wave-uniform N in 0 .. 6 and a compile-time P, followed by a barrier
template <int P>
__device__ __forceinline__ void ks_wait_rows_and_sync(int rows_in_flight) {
    constexpr int M = 63;                                                // vmcnt is a 6-bit counter: a larger allowance is clamped (waits a little early)
#define KS_WAIT_CASE(N) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((N) * P < M ? (N) * P : M) : "memory")
    switch (rows_in_flight < 0 ? 0 : rows_in_flight > 6 ? 6 : rows_in_flight) {
        case 0: KS_WAIT_CASE(0); break;
        case 1: KS_WAIT_CASE(1); break;
        case 2: KS_WAIT_CASE(2); break;
        case 3: KS_WAIT_CASE(3); break;
        case 4: KS_WAIT_CASE(4); break;
        case 5: KS_WAIT_CASE(5); break;
        default: KS_WAIT_CASE(6); break;
    }
#undef KS_WAIT_CASE
}

template <int C, int NB, int CPW, int TAG>
__global__ __launch_bounds__((KsCfg<C, NB, CPW>::NTHR)) __attribute__((amdgpu_waves_per_eu(KsCfg<C, NB, CPW>::WAVES_PER_SIMD, KsCfg<C, NB, CPW>::WAVES_PER_SIMD)))
void conv_ks_kernel(KsArgs a) {
    using K = KsCfg<C, NB, CPW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);             // wave role (wave-uniform by construction)
    const unsigned char* const tin = a.nb ? a.inb[blockIdx.y] : a.in;
    unsigned char* const tout = a.nb ? a.outb[blockIdx.y] : a.out;

    // workgroup b = N group g, range r of G
    // Workgroup b runs on XCD b % 8 (round-robin dispatch), and every XCD has its own L2: the NG workgroups that read the same input rows (one
    // per N group) are placed on the SAME XCD - b = xcd + 8 k, k = (range / 8) * NG + g - so that only the first of them fetches the rows from
    // memory.  (Ranges that do not fill whole rounds of eight fall back to the plain order.)
    const int G = gridDim.x / K::NG;
    int g, r;
    if ((G & 7) == 0) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        g = k % K::NG; r = (k / K::NG) * 8 + xcd;
    } else { g = blockIdx.x % K::NG; r = blockIdx.x / K::NG; }
    if (r >= G) return;
    const int u0 = (int)((long long)a.nunits * r / G), u1 = (int)((long long)a.nunits * (r + 1) / G);
    if (u1 <= u0) return;
    constexpr int WSTRIDE = t64_img_nt(1, K::NCH);                       // bytes per output block of the weight image

    if (wv == 0) { KS_STAMP(0) }
    if (wv < K::NCON) {
        // ------------------------------------------------------------------------------------------------ consumers
        const int n = wv % NB, kq = wv / NB;                             // output block within the group, K slice
        const int nbg = g * NB + n;                                      // output block of the layer
        const int h = lane >> 5, li = lane & 31;
        f16x8 Wf[CPW][9];
        {
            const unsigned char* wsrc = a.img + (size_t)nbg * WSTRIDE + (size_t)(kq * CPW) * 9 * 1024 + (h * 32 + li) * 16;
#pragma unroll
            for (int cc = 0; cc < CPW; cc++)
#pragma unroll
                for (int t = 0; t < 9; t++) {
                    if RIFE_ABL(TAG & KS_NOWEIGHTS) { for (int e = 0; e < 8; e++) Wf[cc][t][e] = (_Float16)(0.001f * (lane + t)); }
                    else Wf[cc][t] = *reinterpret_cast<const f16x8*>(wsrc + (cc * 9 + t) * 1024);
                }
            if RIFE_ABL(TAG & KS_CLK) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (wv == 0) { KS_STAMP(1) } }
        }
        // identity A fragments of the skip connection: chunk c carries the input channels of output block c >> 1 (conv_t64.h)
        f16x8 idf[2];
        {
            const int ch = s16_row_channel(li);
#pragma unroll
            for (int hc = 0; hc < 2; hc++)
#pragma unroll
                for (int e = 0; e < 8; e++) idf[hc][e] = ch == 16 * hc + 8 * h + e ? (_Float16)1.f : (_Float16)0.f;
        }
        if (wv == 0) {                                                   // bias | slopes of the group's output blocks -> LDS (read by the storers)
            for (int i = lane; i < NB * 16; i += 64) {
                const int nn = i >> 4, q = i & 15;
                reinterpret_cast<f32x4*>(ldsb + K::LDS_BS + nn * 256)[q] =
                    reinterpret_cast<const f32x4*>(a.img + (size_t)(g * NB + nn) * WSTRIDE + (size_t)K::NCH * t64_wch(1))[q];
            }
        }
        unsigned colo[3];                                                // column + chunk part of the fragment addresses (hi plane of my first chunk)
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int px = li + dx;
            colo[dx] = (unsigned)(K::LDS_RING + kq * CPW * (2 * RS_SEG) + px * 32 + ((h ^ ((px >> 3) & 1)) << 4));
        }
        unsigned char* const stg = ldsb + K::LDS_STG + (kq * NB + n) * 4096 + (h * 32 + li) * 64;
        const int qs = (li >> 1) & 3;
        bool own[CPW];                                                   // my chunk cc carries the input channels of my output block: identity tap
#pragma unroll
        for (int cc = 0; cc < CPW; cc++) own[cc] = a.skip && ((kq * CPW + cc) >> 1) == nbg;

        // Software pipeline in registers: the sums of step it - 1 stay in one accumulator set while the MFMAs of step it run into the other; an
        // iteration = [stage the sums of step it - 1] X [MFMAs of step it] Y, with X = "staging full" and Y = "staging free" (the storers finish
        // step it - 1 between X and Y, under the MFMAs).  One accumulation chain per set (hi and lo products into the same registers): the second
        // consumer wave of the SIMD fills the issue slots a dependent MFMA leaves.
        f32x16 accA, accB;
        auto iteration = [&](auto parc, const int it, const int nrow) {
            constexpr int PAR = decltype(parc)::value;
            f32x16& acc = PAR ? accB : accA;
            f32x16& pend = PAR ? accA : accB;
            if (it > 0) {
                f32x4* const d4 = reinterpret_cast<f32x4*>(stg);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] = pend[4 * q + k];
                    d4[q ^ qs] = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // X: the partial sums of step it - 1 are in the staging buffer
            if (wv == 0 && it < 4) { KS_STAMP(3 + 2 * it) }
            if (it < nrow) {
                unsigned rb[3];
#pragma unroll
                for (int dy = 0; dy < 3; dy++) rb[dy] = (unsigned)(((it + dy) % K::NR) * K::ROWB);
#pragma unroll
                for (int q = 0; q < 16; q++) acc[q] = 0.f;
                // fragment reads run PF (chunk, tap) pairs ahead of their MFMAs
                constexpr int NP = CPW * 9, PF = 2, NF = PF + 1;
                f16x8 fh[NF], fl[NF];
                auto frag_read = [&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    constexpr int cc = m / 9, t = m % 9;
                    const unsigned ad = rb[t / 3] + colo[t % 3] + cc * (2 * RS_SEG);
                    fh[m % NF] = *reinterpret_cast<const f16x8*>(ldsb + ad);
                    fl[m % NF] = *reinterpret_cast<const f16x8*>(ldsb + ad + RS_SEG);
                };
                if (!RIFE_ABL(TAG & KS_NOMATH)) for_each_slot<0, PF>([&](auto mc) { frag_read(mc); });
                if (!RIFE_ABL(TAG & KS_NOMATH)) for_each_slot<0, NP>([&](auto mc) {
                    constexpr int m = decltype(mc)::value;
                    constexpr int cc = m / 9, t = m % 9;
                    if constexpr (m + PF < NP) frag_read(std::integral_constant<int, m + PF>{});
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[cc][t], fh[m % NF], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[cc][t], fl[m % NF], acc, 0, 0, 0);
                    if constexpr (t == 4) {
                        if (own[cc]) {                                   // + x: the centre tap's pixel fragment through the identity matrix
                            const f16x8 idA = ((kq * CPW + cc) & 1) ? idf[1] : idf[0];
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(idA, fh[m % NF], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(idA, fl[m % NF], acc, 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            if (wv == 0 && it < 4) { KS_STAMP(4 + 2 * it) }
            asm volatile("s_barrier" ::: "memory");                      // Y: the storers have read the partial sums of step it - 1; rows of step it + 1 landed
        };
        for (int u = u0; u < u1;) {
            const int strip = u / a.H, y0 = u - strip * a.H;
            const int nrow = min(a.H - y0, u1 - u);
            u += nrow;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // prologue barrier: rows 0 .. 2 of the segment landed, bias in LDS
            if (wv == 0) { KS_STAMP(2) }
            for (int it = 0; it <= nrow; it += 2) {
                iteration(std::integral_constant<int, 0>{}, it, nrow);
                if (it + 1 <= nrow) iteration(std::integral_constant<int, 1>{}, it + 1, nrow);
            }
            if (wv == 0) { KS_STAMP(11) }
        }
    } else if (wv < K::NCON + KS_NLD) {
        // ------------------------------------------------------------------------------------------------ loaders
        // loader J issues the pieces J, J + 2, .. of every row: PJ instructions per row, a compile-time count (the counted vmcnt waits below)
        auto loader = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            constexpr int PJ = (K::PIECES - J + KS_NLD - 1) / KS_NLD;
            unsigned soff[PJ];
            bool act[PJ];
#pragma unroll
            for (int i = 0; i < PJ; i++) {
                const int uu = (J + KS_NLD * i) * 64 + lane;              // 16-byte unit of the row slot: (segment = chunk, hi | lo; pixel; half)
                act[i] = uu < K::UNITS;                                  // the last piece of a row may be partial: at least one lane is active
                const int uc = min(uu, K::UNITS - 1);
                const int seg = uc / 68, within = uc - seg * 68;
                const int px = within >> 1, pos = within & 1;
                const int kh = pos ^ ((px >> 3) & 1);
                soff[i] = (unsigned)seg * a.plane + (unsigned)(px * 32 + kh * 16);
            }
            for (int u = u0; u < u1;) {
                const int strip = u / a.H, y0 = u - strip * a.H;
                const int nrow = min(a.H - y0, u1 - u);
                u += nrow;
                const int nin = nrow + 2;                                // input rows of the segment: padded rows y0 .. y0 + nrow + 1
                int issued = 0;
                auto issue_row = [&]() {                                 // row `issued` of the segment -> ring slot issued % NR
                    const unsigned rowoff = (unsigned)((y0 + issued) * a.pitch + 32 * strip) * 32u;
                    const unsigned dst = (unsigned)(K::LDS_RING + (issued % K::NR) * K::ROWB);
#pragma unroll
                    for (int i = 0; i < PJ; i++)
                        if (act[i] && !RIFE_ABL(TAG & KS_NODMA)) rs_dma16<0>(tin, rowoff + soff[i], dst + (J + KS_NLD * i) * 1024);
                    issued++;
                };
                while (issued < nin && issued < 4) issue_row();          // the three rows of step 0 and one more
                if (J == 0) { KS_STAMP(12) }
                ks_wait_rows_and_sync<PJ>(issued - 3);                   // prologue barrier: rows 0 .. 2 landed
                if (J == 0) { KS_STAMP(13) }
                for (int it = 0; it <= nrow; it++) {
                    // rows up to it + NR - 1 may be in the ring during iteration it (the slot of row it + NR - 1 held row it - 1, last read in step it - 1);
                    // at most two new rows per iteration, so that a loader never keeps the workgroup waiting at X
                    for (int k = 0; k < 2 && issued < nin && issued < it + K::NR; k++) issue_row();
                    asm volatile("s_barrier" ::: "memory");              // X
                    ks_wait_rows_and_sync<PJ>(issued - min(it + 4, nin));        // Y; rows 0 .. it + 3 landed: step it + 1 may start after this barrier
                }
            }
        };
        if (wv == K::NCON) loader(std::integral_constant<int, 0>{});
        else loader(std::integral_constant<int, 1>{});
    } else {
        // ------------------------------------------------------------------------------------------------ storers
        // storer j finishes the 16-channel chunks cc = j, j + KS_NST, .. of the row staged one step ago: sum of the KS partials, bias, slope,
        // {hi, lo} split, zeros outside the valid pixels; lane = (pixel l >> 1, half l & 1): 16 bytes of the hi plane and 16 of the lo plane,
        // 1 KiB contiguous per store instruction
        const int j = wv - K::NCON - KS_NLD;
        constexpr int KS_NST = K::NST;
        const int px = lane >> 1, jh = lane & 1, qs = (px >> 1) & 3;
        for (int u = u0; u < u1;) {
            const int strip = u / a.H, y0 = u - strip * a.H;
            const int nrow = min(a.H - y0, u1 - u);
            u += nrow;
            const int x0 = 32 * strip;
            const unsigned okmask = x0 + px < a.W ? 0xffffffffu : 0u;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // prologue barrier
            for (int it = 0; it <= nrow; it++) {
                asm volatile("s_barrier" ::: "memory");                  // X: the partial sums of step it - 1 are staged
                if (it >= 1) {                                           // step it - 1
                    const int y = y0 + it - 1;
                    const unsigned char* const sbuf = ldsb + K::LDS_STG;
                    unsigned char* const dst = tout + ((unsigned)((y + 1) * a.pitch + x0 + 1) * 32u + (unsigned)(lane * 16));
                    for (int cc = j; cc < 2 * NB; cc += KS_NST) {        // chunk cc of the group = output block cc >> 1, half cc & 1
                        const int nn = cc >> 1, hh = cc & 1;
                        f32x4 s0, s1;
#pragma unroll
                        for (int kq = 0; kq < K::KS; kq++) {
                            const unsigned char* const rec = sbuf + (kq * NB + nn) * 4096 + (hh * 32 + px) * 64;
                            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rec + (((2 * jh) ^ qs) << 4));
                            const f32x4 r1 = *reinterpret_cast<const f32x4*>(rec + (((2 * jh + 1) ^ qs) << 4));
                            if (kq == 0) { s0 = r0; s1 = r1; }
                            else {
#pragma unroll
                                for (int k = 0; k < 4; k++) { s0[k] += r0[k]; s1[k] += r1[k]; }
                            }
                        }
                        const float* const bsp = reinterpret_cast<const float*>(ldsb + K::LDS_BS + nn * 256);
                        const f32x4 b0 = *reinterpret_cast<const f32x4*>(bsp + 16 * hh + 8 * jh);
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(bsp + 16 * hh + 8 * jh + 4);
                        const f32x4 l0 = *reinterpret_cast<const f32x4*>(bsp + 32 + 16 * hh + 8 * jh);
                        const f32x4 l1 = *reinterpret_cast<const f32x4*>(bsp + 32 + 16 * hh + 8 * jh + 4);
                        f16x8 hv, lv;
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const float yv = (e < 4 ? s0[e & 3] : s1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
                            const float sl = e < 4 ? l0[e & 3] : l1[e & 3];
                            float v = yv < 0.f ? yv * sl : yv;
                            v = __uint_as_float(__float_as_uint(v) & okmask);
                            const _Float16 hq = (_Float16)v;
                            hv[e] = hq;
                            lv[e] = (_Float16)(v - (float)hq);
                        }
                        const int oc = g * (2 * NB) + cc;                // 16-channel chunk of the layer's output
                        if (!RIFE_ABL(TAG & KS_NOSTORE)) {
                            *reinterpret_cast<f16x8*>(dst + (size_t)(2 * oc) * a.plane) = hv;
                            *reinterpret_cast<f16x8*>(dst + (size_t)(2 * oc + 1) * a.plane) = lv;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                 // Y: my reads of the staging buffer are done
            }
            if (j == 0) { KS_STAMP(14) }
        }
        if RIFE_ABL(TAG & KS_CLK) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (j == 0) { KS_STAMP(15) } }
    }
}

}  // namespace rife
