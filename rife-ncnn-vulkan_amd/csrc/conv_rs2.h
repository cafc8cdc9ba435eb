// conv_rs2_kernel: TWO consecutive 64 -> 64 channel residual trunk convolutions of the finest IFBlock of rife-v4.6 in one launch (reference
// models/rife-v4.6/flownet.param:169-197: 8 x {Split, Convolution 3x3 pad 1, BinaryOp add, ReLU slope 0.2} = 4 launches of this kernel), the
// rows of the first layer ("A") never leaving the CU: they go, already bias + LeakyReLU + {hi, lo} split, into a second LDS ring that the
// second layer ("B") consumes.  Round 6; depth-fused form of conv_rs_kernel (conv_rs.h), which round-trips the whole 267 MB trunk tensor of a
// 4K pair through HBM once per layer (measured 290 MB of traffic per launch, 2.3 GB per pair for the eight layers): this kernel reads A's input
// and writes B's output only - half the traffic per layer - for ~ 7 % more matrix work (column halo) and a pipeline fill per workgroup.
//
// One workgroup per CU, 8 waves, the waves of conv_rs_kernel with other jobs:
//   * wave 0, 1 "CA": consumer of layer A, output block N = wave (32 channels), ONE row per step.  36 weight fragments in registers, pixel
//     fragments from ring A (the input rows, LDS-DMA), raw fp32 sums -> staging A.  The code of rs_consumer (conv_rs.h): the same products in
//     the same order, two accumulation chains.
//   * wave 2, 3 "CB": the same for layer B on ring B (A's rows), raw sums -> staging B.
//   * wave 4 "L": LDS-DMA of one input row per step into ring A (8 slots), RS2_AH rows in flight beyond the one the consumers need next.
//   * wave 5, 6 "EA": epilogue of layer A, chunks {0, 1} / {2, 3}: staging A -> bias, LeakyReLU, {hi, lo} split, ZERO outside the image (B's zero
//     padding) -> ring B (5 slots).  The arithmetic of conv_rs_kernel's storers: ring B holds exactly the bytes conv_rs would have stored to HBM.
//   * wave 7 "EB": epilogue of layer B: staging B -> global, 960 contiguous bytes per plane row (conv_rs_kernel's storer).
// Geometry: a workgroup walks down (or up: `descend`) a strip of RS2_SW = 30 output columns.  The matrix instruction covers 32 pixel columns:
// layer A is computed for columns x0 - 1 .. x0 + 30 (one column of halo on each side of B's 30), from input columns x0 - 2 .. x0 + 31 (34 pixels
// per ring row, like conv_rs); layer B computes x0 .. x0 + 31 and keeps the first 30 (its last two read ring-B pixels 32, 33 that nobody writes:
// every pixel is its own column of the matrix product, garbage stays in columns that are never stored).  Rows: a segment = rows [r0, r1) of one
// strip; A is computed for rows r0 - 1 .. r1 (one row of halo recomputed per segment end), rows / columns of A outside the image are stored as
// zeros by EA (they are B's padding), so the DMA of rows outside the tensor is simply clamped to a valid row.
// Schedule, one s_barrier per iteration `it` (all eight waves), sequence numbers in walking order:
//   L    by barrier it: input rows .. it + 4 landed (the consumers prefetch the first fragments of step it + 1 before barrier it), rows .. it + 7 issued
//   CA   it = 0 .. R + 1        A row j = it from input rows it .. it + 2                       -> staging A[it % 3]
//   EA   it = 2 .. R + 3        A row j = it - 2: staging A[j % 3] -> ring B[j % 5]            (two iterations later: the consumers do not wait for
//                                                                                               their staging writes before the barrier, like conv_rs)
//   CB   it = LAG .. LAG + R-1  B row it - LAG from A rows it - LAG .. it - LAG + 2             -> staging B[it % 3]
//   EB   it = LAG+2 .. LAG+R+1  B row it - LAG - 2 -> global
//   LAG = 5 walking down, 6 walking up (the first fragments a consumer prefetches are tap row dy = 0: the NEWEST row when walking up).
// A segment costs R + LAG + 2 iterations for R rows; the host cuts every strip into equal parts so that all CUs have one segment (4K, 256 CUs: 32
// strips x 8 parts of 68 rows; on a CU-masked stream of 128 CUs: 4 parts of 136 rows).
// Bit-exactness: per layer the products, their order and the epilogue arithmetic are conv_rs_kernel's, and ring B holds the {hi, lo} f16 pairs
// conv_rs_kernel would have written: the output equals two conv_rs launches byte for byte (tools/rs2_bench.py, tests/test_gpu_t64.py).
// LDS: ring A 8 x 8,704 + ring B 5 x 8,704 + staging 2 x 3 x 8 KiB + bias / slopes 1 KiB = 163,328 B of the CU's 163,840.
#pragma once
#include "conv_rs.h"

namespace rife {

constexpr int RS2_SW = 30;                                 // output columns per strip
constexpr int RS2_NRA = 8;                                 // ring A: input rows
constexpr int RS2_NRB = 5;                                 // ring B: rows of layer A
constexpr int RS2_AH = 3;                                  // input rows in flight beyond the newest one the next barrier must see
constexpr int RS2_NSTG = 3;                                // staging buffers per layer (one row each): written in iteration it, read in it + 2
constexpr int RS2_LDS_RA = 0;
constexpr int RS2_LDS_RB = RS2_LDS_RA + RS2_NRA * RS_ROWB; // 69,632
constexpr int RS2_LDS_SA = RS2_LDS_RB + RS2_NRB * RS_ROWB; // 113,152
constexpr int RS2_LDS_SB = RS2_LDS_SA + RS2_NSTG * RS_STG_ROW;      // 137,728
constexpr int RS2_LDS_BS = RS2_LDS_SB + RS2_NSTG * RS_STG_ROW;      // 162,304: bias A[64] | slope A[64] | bias B[64] | slope B[64]
constexpr int RS2_LDS = RS2_LDS_BS + 1024;                 // 163,328 B
static_assert(RS2_LDS <= 160 * 1024, "LDS budget of one CU");
static_assert(RS2_NRA == 4 + RS2_AH + 1, "the row issued in iteration it (sequence it + 4 + AH) takes the slot of row it - 1");
constexpr int RS2_NTHR = 512;
constexpr int RS2_MIN_ROWS = 8;                            // host: rows per segment below which two conv_rs launches are used instead

struct Rs2Args {
    const unsigned char* in;     // S16 tensor, allocation start (= pixel (-1, -1) of plane 0)
    unsigned char* out;          // S16 tensor of the same geometry
    const unsigned char* imgA;   // conv_t64's weight image of the first layer (T64_IMG bytes)
    const unsigned char* imgB;   // ... of the second layer
    int H, W;                    // valid pixels
    int pitch;                   // pixels per plane row
    unsigned plane;              // bytes per plane
    int rowmax;                  // columns with storage in a plane row (pitch - 2)
    int kparts;                  // segments per strip
    int nseg;                    // strips * kparts
    int descend;                 // 1: every segment is walked bottom-up (consecutive launches alternate)
    int limit;                   // tensor bytes - 16: the DMA source offsets are clamped to [0, limit] (the strips' column halo reaches 32 B before / after the tensor)
    long long* stamps = nullptr; // bench builds RIFE_ABL(TAG & RS_CLK): [workgroup][4]; RIFE_ABL(TAG & RS2_STAMPS): see RS2_BAR
    int stamp_wg = 0;
};

struct Rs2Seg { int x0, ystart, dir, R; };
__device__ __forceinline__ Rs2Seg rs2_segment(const Rs2Args& a, const int seg) {
    const int strip = seg / a.kparts, part = seg - strip * a.kparts;
    const int r0 = (int)((long long)a.H * part / a.kparts), r1 = (int)((long long)a.H * (part + 1) / a.kparts);
    Rs2Seg s;
    s.x0 = RS2_SW * strip; s.R = r1 - r0; s.dir = a.descend ? -1 : 1; s.ystart = a.descend ? r1 - 1 : r0;
    return s;
}

#define RS2_NEXT(V, MOD) { V = V + 1 == (MOD) ? 0 : V + 1; }
// bench builds, RIFE_ABL(TAG & RS2_STAMPS): lane 0 of every wave of workgroup a.stamp_wg records the shader clock when it arrives at a barrier and when it
// leaves it: stamps[(wave * RS2_NSTAMP + barrier) * 2 + {0, 1}] (tools/rs2_bench.py prints who the others waited for, step by step)
enum { RS2_STAMPS = 0x1000, RS2_NOFRAG = 0x10, RS2_NOLO = 0x20 };      // NOFRAG: no LDS fragment reads (MFMAs on whatever the registers hold); NOLO: hi products only
constexpr int RS2_NSTAMP = 128;
#define RS2_BAR(KIND)                                                                                                  \
    {                                                                                                                  \
        if (RIFE_ABL(TAG & RS2_STAMPS) && (int)blockIdx.x == a.stamp_wg && bar < RS2_NSTAMP && (threadIdx.x & 63) == 0)         \
            a.stamps[((threadIdx.x >> 6) * RS2_NSTAMP + bar) * 2] = (long long)__builtin_readcyclecounter();           \
        RS_SYNC_##KIND;                                                                                                \
        if (RIFE_ABL(TAG & RS2_STAMPS) && (int)blockIdx.x == a.stamp_wg && bar < RS2_NSTAMP && (threadIdx.x & 63) == 0)         \
            a.stamps[((threadIdx.x >> 6) * RS2_NSTAMP + bar) * 2 + 1] = (long long)__builtin_readcyclecounter();       \
        bar++;                                                                                                         \
    }

// consumer wave of one layer: output block N, one row per step, `lead` idle iterations before the first step of a segment, R + extra steps
template <int N, int NR, int TAG>
__device__ __forceinline__ void rs2_consumer(const Rs2Args& a, const unsigned char* const img, unsigned char* const ldsb, const unsigned ring, const unsigned stgbase,
                                             const int lead, const int extra, const int lag, const int lane) {
    const int h = lane >> 5, li = lane & 31;
    f16x8 W[4][9];                                                       // rs_consumer's register-resident weights
    {
        const unsigned char* wsrc = img + h * 1024 + (N * 32 + li) * 16;
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int t = 0; t < 9; t++) W[c][t] = *reinterpret_cast<const f16x8*>(wsrc + c * t64_wch(2) + t * 2048);
    }
    f16x8 idf[2];
    {
        const int ch = s16_row_channel(li);
#pragma unroll
        for (int hc = 0; hc < 2; hc++)
#pragma unroll
            for (int e = 0; e < 8; e++) idf[hc][e] = ch == 16 * hc + 8 * h + e ? (_Float16)1.f : (_Float16)0.f;
    }
    unsigned colo[3];
#pragma unroll
    for (int dx = 0; dx < 3; dx++) { const int px = li + dx; colo[dx] = (unsigned)(px * 32 + ((h ^ ((px >> 3) & 1)) << 4)); }
    unsigned char* const stg = ldsb + stgbase + ((2 * N + h) * 32 + li) * 64;
    const int qs = (li >> 1) & 3;

    int bar = 0;
    int sq = 0;                                                          // ring slot of the step's first row in walking order
    int sbuf = 0;                                                        // staging buffer of the step (iteration % 3)
    unsigned ad[9];
    auto step_addresses = [&]() {
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            int sl = sq + (a.descend ? 2 - dy : dy);                     // walking up: the first row of the sequence is the row BELOW (tap row dy = 2)
            if (sl >= NR) sl -= NR;
            const unsigned rb = ring + (unsigned)(sl * RS_ROWB);
#pragma unroll
            for (int dx = 0; dx < 3; dx++) ad[dy * 3 + dx] = rb + colo[dx];
        }
    };
    constexpr int NF = RS_PF + 1;
    static_assert(NF == 4 && RS_NPAIR % NF == 2, "fragment set rotation across steps");
    f16x8 fh[NF], fl[NF];
    auto step = [&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        auto frag_read = [&](auto mc) {
            constexpr int m = decltype(mc)::value;
            constexpr RsPairDesc d = rs_pair(N, m % RS_NPAIR);
            constexpr int st = (m + 2 * PAR) % NF;
            if RIFE_ABL(TAG & RS2_NOFRAG) { asm volatile("" : "+v"(fh[st]), "+v"(fl[st])); return; }
            fh[st] = *reinterpret_cast<const f16x8*>(ldsb + ad[d.t] + d.c * (2 * RS_SEG));
            if (!RIFE_ABL(TAG & RS2_NOLO)) fl[st] = *reinterpret_cast<const f16x8*>(ldsb + ad[d.t] + d.c * (2 * RS_SEG) + RS_SEG);
        };
        f32x16 accH, accL;
        if (!RIFE_ABL(TAG & RS_NOMATH)) {
            for_each_slot<0, RS_NPAIR>([&](auto mc) {
                constexpr int m = decltype(mc)::value;
                constexpr RsPairDesc d = rs_pair(N, m);
                constexpr int st = (m + 2 * PAR) % NF;
                if constexpr (m == RS_NPAIR - RS_PF) { RS2_NEXT(sq, NR) step_addresses(); }      // ad[] is dead: every read of this step is issued
                frag_read(std::integral_constant<int, m + RS_PF>{});
                const f16x8 A = d.idn ? idf[d.c & 1] : W[d.c][d.t];
                if constexpr (m == 0) {
                    f32x16 z;
#pragma unroll
                    for (int q = 0; q < 16; q++) z[q] = 0.f;
                    accH = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, fh[st], z, 0, 0, 0);
                    if (!RIFE_ABL(TAG & RS2_NOLO)) accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, fl[st], z, 0, 0, 0); else accL = z;
                } else {
                    accH = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, fh[st], accH, 0, 0, 0);
                    if (!RIFE_ABL(TAG & RS2_NOLO)) accL = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, fl[st], accL, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) { accH[q] = (float)lane; accL[q] = 0.f; }
        }
        f32x4* const d4 = reinterpret_cast<f32x4*>(stg + sbuf * RS_STG_ROW);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = accH[4 * q + k] + accL[4 * q + k];
            d4[q ^ qs] = v;
        }
        RS2_NEXT(sbuf, RS2_NSTG)
        RS2_BAR(BARE())
    };
    for (int seg = blockIdx.x; seg < a.nseg; seg += gridDim.x) {
        const int R = rs2_segment(a, seg).R;
        const int S = R + extra, NIT = R + lag + 2;
        RS2_BAR(LGKM())                                                  // the segment's first rows have landed (L), bias in LDS
        for (int i = 0; i < lead; i++) RS2_BAR(BARE())
        sq = 0; sbuf = lead % RS2_NSTG;
        step_addresses();
        if (!RIFE_ABL(TAG & RS_NOMATH)) for_each_slot<0, RS_PF>([&](auto mc) {       // first fragments of the first step (parity 0)
            constexpr int m = decltype(mc)::value;
            constexpr RsPairDesc d = rs_pair(N, m);
            fh[m % NF] = *reinterpret_cast<const f16x8*>(ldsb + ad[d.t] + d.c * (2 * RS_SEG));
            fl[m % NF] = *reinterpret_cast<const f16x8*>(ldsb + ad[d.t] + d.c * (2 * RS_SEG) + RS_SEG);
        });
        int k = 0;
        for (; k + 1 < S; k += 2) { step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); }
        if (k < S) step(std::integral_constant<int, 0>{});
        for (int i = lead + S; i < NIT; i++) RS2_BAR(BARE())
    }
}

// y = slope(sum + bias) of one 16-channel chunk of one staged pixel, split into {hi, lo}; zero where okmask is
__device__ __forceinline__ void rs2_finish(const unsigned char* const src, const unsigned char* const bs, const int cc, const int jh, const int qs, const float slope, const unsigned okmask,
                                           f16x8& hv, f16x8& lv) {
    const f32x4 r0 = *reinterpret_cast<const f32x4*>(src + cc * 2048 + (((2 * jh) ^ qs) << 4));
    const f32x4 r1 = *reinterpret_cast<const f32x4*>(src + cc * 2048 + (((2 * jh + 1) ^ qs) << 4));
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bs + (16 * cc + 8 * jh) * 4);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(bs + (16 * cc + 8 * jh + 4) * 4);
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float yv = (e < 4 ? r0[e & 3] : r1[e & 3]) + (e < 4 ? b0[e & 3] : b1[e & 3]);
        float v = yv < 0.f ? yv * slope : yv;
        v = __uint_as_float(__float_as_uint(v) & okmask);
        const _Float16 hh = (_Float16)v;
        hv[e] = hh;
        lv[e] = (_Float16)(v - (float)hh);
    }
}

template <int TAG>
__global__ __launch_bounds__(RS2_NTHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_rs2_kernel(Rs2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    long long clk0 = 0, rt0 = 0;
    if RIFE_ABL(TAG & RS_CLK) { clk0 = (long long)__builtin_readcyclecounter(); rt0 = (long long)__builtin_amdgcn_s_memrealtime(); }
    if ((int)blockIdx.x >= a.nseg) return;
    const int lag = a.descend ? 6 : 5;
    // every wave executes, per segment, one prologue barrier + R + lag + 2 iteration barriers
    if (wv < 4) {
        // ------------------------------------------------------------------------------------------------ consumers
        if ((wv == 0 || wv == 2) && lane < 32)
            reinterpret_cast<f32x4*>(ldsb + RS2_LDS_BS + (wv >> 1) * 512)[lane] = reinterpret_cast<const f32x4*>((wv ? a.imgB : a.imgA) + 4 * t64_wch(2))[lane];
        if (wv == 0) rs2_consumer<0, RS2_NRA, TAG>(a, a.imgA, ldsb, RS2_LDS_RA, RS2_LDS_SA, 0, 2, lag, lane);
        else if (wv == 1) rs2_consumer<1, RS2_NRA, TAG>(a, a.imgA, ldsb, RS2_LDS_RA, RS2_LDS_SA, 0, 2, lag, lane);
        else if (wv == 2) rs2_consumer<0, RS2_NRB, TAG>(a, a.imgB, ldsb, RS2_LDS_RB, RS2_LDS_SB, lag, 0, lag, lane);
        else rs2_consumer<1, RS2_NRB, TAG>(a, a.imgB, ldsb, RS2_LDS_RB, RS2_LDS_SB, lag, 0, lag, lane);
    } else if (wv == 4) {
        // ------------------------------------------------------------------------------------------------ loader
        // piece i of a row covers LDS units 64 i .. 64 i + 63 (16 bytes each) of the 544 of a row slot: unit = (segment, pixel, half)   [conv_rs.h]
        int bar = 0;
        int soff[9];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int u = min(i * 64 + lane, 543);
            const int sg = u / 68, within = u - sg * 68;
            const int px = within >> 1, pos = within & 1;
            const int kh = pos ^ ((px >> 3) & 1);
            soff[i] = (int)((unsigned)sg * a.plane) + px * 32 + kh * 16;
        }
        if RIFE_ABL(TAG & RS_NODMA) { for (int i = lane; i < RS2_LDS_SA / 16; i += 64) reinterpret_cast<f32x4*>(ldsb)[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        for (int seg = blockIdx.x; seg < a.nseg; seg += gridDim.x) {
            const Rs2Seg s = rs2_segment(a, seg);
            const int needed = s.R + 4, NIT = s.R + lag + 2;            // input rows r0 - 2 .. r1 + 1 in walking order
            int issued = 0;
            auto issue = [&]() {
                const int y = s.ystart + s.dir * (issued - 2);          // pixel row; padded row y + 1, clamped into the tensor (rows outside feed masked rows of A only)
                const int prow = min(max(y + 1, 0), a.H + 1);
                if (!RIFE_ABL(TAG & RS_NODMA)) {
                    const int rowoff = (prow * a.pitch + s.x0 - 1) * 32;        // ring pixel 0 = column x0 - 2 = padded column x0 - 1
                    const unsigned dst = (unsigned)(RS2_LDS_RA + (issued & (RS2_NRA - 1)) * RS_ROWB);
#pragma unroll
                    for (int k = 0; k < 8; k++) rs_dma16<TAG>(a.in, (unsigned)min(max(rowoff + soff[k], 0), a.limit), dst + k * 1024);
                    if (lane < 32) rs_dma16<TAG>(a.in, (unsigned)min(max(rowoff + soff[8], 0), a.limit), dst + 8 * 1024);
                }
                issued++;
            };
            auto wait_barrier = [&](const int pending) {                 // all but the newest `pending` rows have landed (9 DMA instructions per row)
                if RIFE_ABL(TAG & RS_NODMA) RS2_BAR(LGKM())
                else if (pending <= 0) RS2_BAR(VM(0))
                else if (pending == 1) RS2_BAR(VM(9))
                else if (pending == 2) RS2_BAR(VM(18))
                else RS2_BAR(VM(27))
            };
            static_assert(RS2_AH == 3, "counted waits above");
            while (issued < min(needed, 4 + RS2_AH)) issue();
            wait_barrier(issued - min(needed, 4));                       // rows of steps 0 and 1
            for (int it = 0; it < NIT; it++) {
                if (issued < needed) issue();
                wait_barrier(issued - min(needed, it + 5));              // rows of step it + 2
            }
        }
    } else if (wv < 7) {
        // ------------------------------------------------------------------------------------------------ epilogue of layer A -> ring B
        const int e = wv - 5;
        int bar = 0;
        const int px = lane >> 1, jh = lane & 1, qs = (px >> 1) & 3;
        const float slope = reinterpret_cast<const float*>(a.imgA + 4 * t64_wch(2))[64];
        const unsigned dcol = (unsigned)(px * 32 + ((jh ^ ((px >> 3) & 1)) << 4));
        for (int seg = blockIdx.x; seg < a.nseg; seg += gridDim.x) {
            const Rs2Seg s = rs2_segment(a, seg);
            const int NIT = s.R + lag + 2;
            RS2_BAR(LGKM())
            RS2_BAR(BARE()) RS2_BAR(BARE())                              // iterations 0, 1
            int slot = 0, sb = 0;
            const int xA = s.x0 - 1 + px;
            const bool colok = xA >= 0 && xA < a.W;
            for (int j = 0; j < s.R + 2; j++) {                          // iteration j + 2: row j of A in walking order = pixel row ystart + dir (j - 1)
                const int yA = s.ystart + s.dir * (j - 1);
                const unsigned okmask = (colok && yA >= 0 && yA < a.H) ? 0xffffffffu : 0u;
                const unsigned char* src = ldsb + RS2_LDS_SA + sb * RS_STG_ROW + px * 64;
                unsigned char* dst = ldsb + RS2_LDS_RB + slot * RS_ROWB + dcol;
#pragma unroll
                for (int cc = 2 * e; cc < 2 * e + 2; cc++) {
                    f16x8 hv, lv;
                    rs2_finish(src, ldsb + RS2_LDS_BS, cc, jh, qs, slope, okmask, hv, lv);
                    *reinterpret_cast<f16x8*>(dst + (2 * cc) * RS_SEG) = hv;
                    *reinterpret_cast<f16x8*>(dst + (2 * cc + 1) * RS_SEG) = lv;
                }
                RS2_NEXT(slot, RS2_NRB) RS2_NEXT(sb, RS2_NSTG)
                RS2_BAR(LGKM())                                          // the row is in ring B before the barrier
            }
            for (int i = s.R + 4; i < NIT; i++) RS2_BAR(BARE())
        }
    } else {
        // ------------------------------------------------------------------------------------------------ epilogue of layer B -> global
        const int px = lane >> 1, jh = lane & 1, qs = (px >> 1) & 3;
        const float slope = reinterpret_cast<const float*>(a.imgB + 4 * t64_wch(2))[64];
        int bar = 0;
        for (int seg = blockIdx.x; seg < a.nseg; seg += gridDim.x) {
            const Rs2Seg s = rs2_segment(a, seg);
            RS2_BAR(LGKM())
            for (int i = 0; i < lag + 2; i++) RS2_BAR(BARE())
            int sb = lag % RS2_NSTG;
            const int x = s.x0 + px;
            const bool store = px < RS2_SW && x < a.rowmax && !RIFE_ABL(TAG & RS_NOSTORE);
            const unsigned okmask = x < a.W ? 0xffffffffu : 0u;
            for (int rho = 0; rho < s.R; rho++) {                        // iteration lag + 2 + rho
                const int y = s.ystart + s.dir * rho;
                if (store) {
                    const unsigned char* src = ldsb + RS2_LDS_SB + sb * RS_STG_ROW + px * 64;
                    unsigned char* dst = a.out + ((unsigned)((y + 1) * a.pitch + s.x0 + 1) * 32u + (unsigned)(lane * 16));
#pragma unroll
                    for (int cc = 0; cc < 4; cc++) {
                        f16x8 hv, lv;
                        rs2_finish(src, ldsb + RS2_LDS_BS + 512, cc, jh, qs, slope, okmask, hv, lv);
                        *reinterpret_cast<f16x8*>(dst + (size_t)(2 * cc) * a.plane) = hv;
                        *reinterpret_cast<f16x8*>(dst + (size_t)(2 * cc + 1) * a.plane) = lv;
                    }
                }
                RS2_NEXT(sb, RS2_NSTG)
                RS2_BAR(LGKM())                                          // my reads of the staging buffer are done before the consumers may reuse it
            }
        }
    }
    if (RIFE_ABL(TAG & RS_CLK) && tid == 0) {
        a.stamps[4 * blockIdx.x] = (long long)__builtin_readcyclecounter() - clk0;
        a.stamps[4 * blockIdx.x + 1] = rt0;
        a.stamps[4 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace rife
