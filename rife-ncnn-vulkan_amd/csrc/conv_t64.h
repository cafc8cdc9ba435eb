// conv_t64_kernel: the 64 -> 64 channel residual trunk convolution of the finest IFBlock
// (reference models/rife-v4.6/flownet.param:169-197: Split, Convolution 3x3 pad 1, BinaryOp add, ReLU slope 0.2; 8 per pair,
// 44 % of the pair's MACs) as persistent workgroups, two per CU, on the split-f16 matrix path of conv_h2b_kernel.
//
// What changed against conv_h2b_kernel (whose phase trace showed ~10 of the 21 us of every workgroup in latency-bound prologue /
// epilogue, matrix pipe 42 % busy):
//   * "S16" activations: the trunk tensor is stored pre-split, hi = f16(x) and lo = f16(x - hi) (the same 4 bytes per element as
//     fp32, and exactly the values the old kernel computed at LDS-staging time), as eight planes [16-channel chunk][hi | lo] of
//     32 bytes per pixel, each plane a zero-bordered image: rows x pitch pixels, pixel (y, x) at (y + 1, x + 1).  The halo of a
//     tile is then plain memory (no bounds tests, no conversion), one row of one chunk is 34 x 32 contiguous bytes, and
//     global_load_lds_dwordx4 moves it 1 KiB at a time touching 8-9 cache lines (per-pixel 64-byte records of all chunks
//     interleaved touched 32: the loads were address-processing bound).
//   * persistent: 2 x #CU workgroups of 8 waves walk their own sequences of 8-row x 32-column tiles.  Every step (one 16-channel
//     K chunk of one tile: 38 MFMAs per wave) starts by sending the NEXT step's halo chunk and weight chunk on their way with
//     LDS-DMA (22 + 18 pieces of 1 KiB, no registers, no VALU) and ends with vmcnt(0) + one s_barrier: the loads have the whole
//     step to land, a tile switch costs nothing, and the epilogue (registers -> global, no LDS, no barrier) of tile i runs at the
//     start of tile i + 1 while the co-resident workgroup has the matrix pipe.
//   * output channels are permuted inside each 32-row MFMA block (a property of the weight packing only) so that a lane ends
//     up with 16 CONSECUTIVE channels of one pixel = the pixel's 32 hi and 32 lo bytes of one chunk: consecutive lanes store
//     consecutive 32-byte entries (s16_store_chunk, conv_mfma.h), no LDS transpose, no barrier.
//   * the identity tap of the skip connection needs no LDS: its A fragment (a permutation matrix) is built in registers.
// Measured (MI355X, 544 x 960 tensor = 3840 x 2160 frame, per launch, tools/t64_bench.py; profiles/r2/): conv_h2b 91-95 us ->
// this kernel 81-83 us.  Parts, same launch geometry: MFMAs on registers only 40 us (= the matrix floor at the ~1.95 GHz the
// chip sustains), + LDS fragment reads 43 us, loads alone 22-26 us, stores alone 27 us, loads + stores 49-52 us (= 5.2-5.5 TB/s:
// the HBM floor of 134 MB in + 134 MB out).  The parts still add up more than they overlap: every VMEM instruction (40 DMA
// pieces per step, 64 stores per tile) blocks its wave while the memory queue is full, and the waves that wait are the waves
// that feed the matrix pipe.  Variants measured and dropped (git history): ONE 16-wave workgroup per CU with resident weights
// and two phase-shifted 8-wave groups behind one s_barrier (100 us with 64-byte per-pixel records stored at a 256-byte lane
// stride - the stores alone took 45 us, store-issue bound; 82-88 us after a DPP quad transpose made them whole lines: each
// group waited at the barrier for the other group's epilogue); one 16-wave workgroup, 16 x 32 tiles, ring of three halo
// buffers with counted vmcnt waits (loads fully hidden, but all 16 epilogues collide: 86-91 us; best frame rate with two pairs
// in flight, 409 vs 397 frames/s at 4K); 1-byte LDS-DMA "touches" that pull the chunk after next into the L2 (+5 us).
// Round-2 probes (tools/t64_bench.py, profiles/r2/t64_bench.txt; T64_CLK records every workgroup's life on the 100 MHz counter and its shader
// cycles): (1) the chip's clock is not constant under this kernel - a burst of launches starts at 1.77 GHz, sags to 1.2 GHz after ~3 ms (110 us
// per launch), and settles at 2.0-2.05 GHz after ~12 ms: 68 us first-start-to-last-end + 3 us to the next launch; the same burst of the
// matrix-only ablation holds 2.1-2.25 GHz, loads + stores only 2.25-2.37 GHz.  (2) Workgroup lives are bimodal: the first-placed workgroup of a
// CU ends at ~51 us, the second-placed one at ~67 us (the older waves win the issue arbitration) - but that is no imbalance to fix: with the
// second-placed workgroups at s_setprio 2 on every other tile both end at ~60-64 us and the launch still takes 68 us (the CU's total rate is the
// limit), and per-XCD atomic tile counters (dynamic schedule; the reset by the last workgroup adds 4 us between launches, 4 tiles per workgroup
// are too coarse to balance) measured 105 us.  (3) Issuing the DMA pieces one per tap between the MFMAs instead of all at the step start lets
// the MFMAs start at once but stretches them from ~2,500 to ~4,500 cycles per step: same 84 us; epilogue stores before / after the DMA issue:
// same.  (4) MFMAs on one fragment set per step (no LDS fragment reads) + loads = 65 us = with the reads (64 us): the LDS is not what keeps
// loads and matrix work from overlapping.
// LDS image of a halo chunk: hi plane [340 px][32 B] then lo plane [340 px][32 B]; the two 16-byte halves of a 32-byte entry are
// swapped when bit 3 of the pixel index is set, which makes the ds_read_b128 fragment reads of 16 consecutive pixels hit 64
// distinct banks (SQ_LDS_BANK_CONFLICT = 0).  The DMA writes LDS lane-linear, so the swap is applied to each lane's SOURCE
// address (cdna guide, rule 21).
// Arithmetic (products, accumulation order, epilogue) is that of conv_h2b_kernel<2, 10>: results are bit-identical.
#pragma once
#include "conv_mfma.h"

namespace rife {

constexpr int T64_TH = 8, T64_NTHR = 64 * T64_TH;                        // tile rows = waves per workgroup
constexpr int T64_IH = T64_TH + 2, T64_IW = 34, T64_NPX = T64_IH * T64_IW;      // halo tile of an 8 x 32 output tile
constexpr int T64_PLANE = T64_NPX * 32;                                  // 10,880 B: hi (or lo) halves of one 16-channel chunk
constexpr int T64_INB = 2 * T64_PLANE;                                   // 21,760 B per halo chunk buffer
// C = 32 NS channels (NS = 2: the 64-channel trunk of the finest block, NS = 3: the 96-channel trunk of block 2)
constexpr int t64_wch(int NS) { return 9 * 2 * 32 * NS * 16; }          // weights of one K chunk [tap][k half][C rows][8 f16]: 18,432 / 27,648 B
constexpr int t64_wb(int NS) { return 2 * NS * t64_wch(NS); }           // C / 16 chunks
constexpr int t64_bsb(int NS) { return 2 * 32 * NS * 4; }               // bias[C], slope[C]
constexpr int t64_img(int NS) { return t64_wb(NS) + t64_bsb(NS); }      // weight image of a C = 32 NS layer in global memory, built on the host
// C = 128 / 192 (blocks 1 / 0): NS = 2 with C / 64 N-tiles of 64 output channels; a work item = (pixel tile, N-tile), K = C / 16 chunks;
// the image holds per N-tile: nchunks x t64_wch(2) of weights, then that N-tile's bias and slopes
constexpr int t64_img_nt(int NS, int nchunks) { return nchunks * t64_wch(NS) + t64_bsb(NS); }
constexpr int T64_LDS_IN = 0, T64_LDS_W = 2 * T64_INB;
constexpr int t64_lds_bs(int NS) { return T64_LDS_W + 2 * t64_wch(NS); }
constexpr int t64_lds(int NS) { return t64_lds_bs(NS) + 2 * t64_bsb(NS); }  // two bias / slope buffers; 81,408 B (two workgroups per CU) / 100,352 B (one)
constexpr int t64_wg_per_cu(int NS) { return 160 * 1024 / t64_lds(NS); }
static_assert(t64_wg_per_cu(2) == 2 && t64_wg_per_cu(3) == 1, "LDS budget");
constexpr int T64_IMG = t64_img(2), T64_WB = t64_wb(2), T64_LDS = t64_lds(2);

// row i of a 32-row MFMA block <-> output channel (within the block): a lane's 16 accumulator registers are 16 consecutive channels
__host__ __device__ constexpr int s16_row_channel(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

struct T64Args {
    const unsigned char* in;     // S16 tensor, allocation start (= pixel (-1, -1) of plane 0)
    unsigned char* out;          // S16 tensor of the same geometry
    const unsigned char* img;    // T64_IMG bytes
    int H, W;                    // valid pixels
    int pitch;                   // pixels per plane row (tiles_x * 32 + 2)
    unsigned plane;              // bytes per plane (rows * pitch * 32)
    int tiles_x, ntiles;         // pixel tiles
    int nchunks, nnt;            // K chunks (input channels / 16, even), N-tiles of 32 NS output channels; work items = ntiles * nnt
    int reverse;                 // 1: walk the tiles from the last to the first (see launch_t64: consecutive layers alternate)
    long long* stamps = nullptr; // bench builds only: TAG & T64_STAMPS: [workgroup][wave 8][step 32][4] shader-clock stamps; TAG & T64_CLK: [workgroup][4] = shader
                                 // cycles of the workgroup's life, its start and its end on the constant 100 MHz counter (tools/t64_bench.py)
};
// bench-only ablation bits of TAG (timing experiments; the results of all but T64_STAMPS are garbage).  The product instantiates TAG = 3.
enum { T64_NOSTORE = 0x100, T64_NODMA = 0x200, T64_NOMATH = 0x400, T64_NOVMWAIT = 0x800, T64_STAMPS = 0x1000, T64_CLK = 0x40000 };

typedef __attribute__((address_space(3))) unsigned char t64_lds_u8;
typedef __attribute__((address_space(1))) const unsigned char t64_glb_u8;
// 64 lanes x 16 bytes, global -> LDS without passing through registers; LDS destination = wave-uniform base + lane * 16
__device__ __forceinline__ void t64_glds16(const unsigned char* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((t64_glb_u8*)g, (t64_lds_u8*)l, 16, 0, 0);
}

// LW (round 3): number of extra LOADER waves (0 or 2).  With LW = 2 the workgroup has ten waves: waves 8 and 9 issue every LDS-DMA piece of
// the halo / weight ring and wait for them, the eight matrix waves issue none and never wait on the vector-memory counter - the stall
// round 2 measured (a wave that waits for a slot in the CU's memory queue issues no MFMAs) is taken off the matrix waves, as in
// conv_rs_kernel, without touching the arithmetic: results are bit-identical to LW = 0.
template <int TAG, int NS = 2, int LW = 0>
__global__ __launch_bounds__(T64_NTHR + 64 * LW) __attribute__((amdgpu_waves_per_eu(2 * t64_wg_per_cu(NS), 2 * t64_wg_per_cu(NS) + (LW ? 1 : 0)))) void conv_t64_kernel(T64Args a) {
    constexpr int T64_WCH = t64_wch(NS), T64_BSB = t64_bsb(NS), T64_LDS_BS = t64_lds_bs(NS), T64_LDS = t64_lds(NS), CH = 32 * NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* const lds = ldsb;
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = __builtin_amdgcn_readfirstlane(tid >> 6);              // wave = output row of the tile (wave-uniform by construction)
    const int h = lane >> 5, li = lane & 31;
    long long clk0 = 0, rt0 = 0;
    if RIFE_ABL(TAG & T64_CLK) { clk0 = (long long)__builtin_readcyclecounter(); rt0 = (long long)__builtin_amdgcn_s_memrealtime(); }
    if (LW > 0 && r >= T64_TH) {
        // ---- loader waves: the work-item stream and the ring schedule of the matrix waves below, DMA and waits only
        const int l = r - T64_TH;
        const int nwg = gridDim.x, b = blockIdx.x;
        const int slot = (b & 7) * (nwg >> 3) + (b >> 3);
        const int nitems = a.ntiles * a.nnt;
        const int mine = nitems > slot ? (nitems - slot + nwg - 1) / nwg : 0;
        const int imgstride = a.nchunks * T64_WCH + T64_BSB;
        auto item = [&](int w_, unsigned& tb_, int& nt_) {
            const int wv = a.reverse ? nitems - 1 - w_ : w_;
            const int t_ = wv / a.nnt; nt_ = wv - t_ * a.nnt;
            const int ty_ = t_ / a.tiles_x;
            tb_ = (unsigned)(ty_ * T64_TH * a.pitch + (t_ - ty_ * a.tiles_x) * 32) * 32u;
        };
        auto dma_in = [&](unsigned tb_, int c, int par) {                // the 22 pieces of a halo chunk, every LW-th one
            const unsigned char* src_ = a.in + (tb_ + (unsigned)(2 * c) * a.plane);
            for (int i = l; i < 22; i += LW) {
                const int sidx = i * 64 + lane;
                const int pl = sidx >= 2 * T64_NPX ? 1 : 0, s1 = sidx - pl * 2 * T64_NPX;
                const int P = min(s1 >> 1, T64_NPX - 1), pos = s1 & 1;
                const int kh = pos ^ ((P >> 3) & 1);
                const int py = P / T64_IW, px = P - py * T64_IW;
                const unsigned so = (unsigned)pl * a.plane + (unsigned)(py * a.pitch + px) * 32u + (unsigned)(kh * 16);
                if (sidx < 4 * T64_NPX) t64_glds16(src_ + so, lds + T64_LDS_IN + par * T64_INB + i * 1024);
            }
        };
        auto dma_w = [&](int nt_, int c, int par) {
            const unsigned char* src_ = a.img + (nt_ * imgstride + c * T64_WCH) + lane * 16;
            for (int i = l; i < T64_WCH / 1024; i += LW) t64_glds16(src_ + i * 1024, lds + T64_LDS_W + par * T64_WCH + i * 1024);
        };
        auto dma_bs = [&](int nt_, int buf) {
            if (l == 0 && lane < T64_BSB / 16) t64_glds16(a.img + (nt_ * imgstride + a.nchunks * T64_WCH) + lane * 16, lds + T64_LDS_BS + buf * T64_BSB);
        };
        auto sync = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); };
        unsigned tb = 0; int nt = 0;
        if (mine > 0) { item(slot, tb, nt); dma_bs(nt, 0); dma_w(nt, 0, 0); dma_in(tb, 0, 0); }
        sync();
        for (int k = 0; k < mine; k++) {
            const bool more = k + 1 < mine;
            unsigned tbn = 0; int ntn = 0;
            if (more) item(slot + (k + 1) * nwg, tbn, ntn);
            for (int c = 0; c < a.nchunks; c++) {
                if (c + 1 < a.nchunks) { dma_in(tb, c + 1, (c & 1) ^ 1); dma_w(nt, c + 1, (c & 1) ^ 1); }
                else if (more) { dma_in(tbn, 0, 0); dma_w(ntn, 0, 0); }
                if (c == 1 && more) dma_bs(ntn, (k + 1) & 1);
                sync();
            }
            tb = tbn; nt = ntn;
        }
        return;
    }

    // ---- per-lane constants
    // halo DMA: piece i = r + 8 j of the chunk buffer covers LDS slots 64 i .. 64 i + 63 (16 bytes each)
    unsigned soff[3];
    bool s2ok;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int s = (r + 8 * j) * 64 + lane;
        const int pl = s >= 2 * T64_NPX ? 1 : 0, s1 = s - pl * 2 * T64_NPX;
        const int P = min(s1 >> 1, T64_NPX - 1), pos = s1 & 1;
        const int kh = pos ^ ((P >> 3) & 1);
        const int py = P / T64_IW, px = P - py * T64_IW;
        soff[j] = (unsigned)pl * a.plane + (unsigned)(py * a.pitch + px) * 32u + (unsigned)(kh * 16);
        if (j == 2) s2ok = s < 4 * T64_NPX;
    }
    // fragment addresses of the nine taps (hi plane; lo = + T64_PLANE), pixel P = (r + dy) * 34 + li + dx
    const unsigned char* ap[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int P = (r + t / 3) * T64_IW + li + t % 3;
        ap[t] = lds + T64_LDS_IN + P * 32 + ((h ^ ((P >> 3) & 1)) << 4);
    }
    const unsigned char* const wb = lds + T64_LDS_W + h * (CH * 16) + li * 16;
    const float* const bs0 = reinterpret_cast<const float*>(lds + T64_LDS_BS);      // bias | slopes of the work item, two buffers (item parity)
    // identity A fragments of the skip connection: K chunk c = 2 n + hc carries input channels 32 n + 16 hc .. + 15, i.e. the rows
    // i of block n with s16_row_channel(i) = 16 hc + k; lane (row li, k half h) holds A[li][8 h .. 8 h + 7]
    f16x8 idf[2];
    {
        const int ch = s16_row_channel(li);
#pragma unroll
        for (int hc = 0; hc < 2; hc++)
#pragma unroll
            for (int e = 0; e < 8; e++) idf[hc][e] = ch == 16 * hc + 8 * h + e ? (_Float16)1.f : (_Float16)0.f;
    }

    // ---- work-item stream: workgroup b runs on XCD b % 8; per round every XCD walks a contiguous band of (pixel tile, N-tile) items
    const int nwg = gridDim.x, b = blockIdx.x;
    const int slot = (b & 7) * (nwg >> 3) + (b >> 3);
    const int nitems = a.ntiles * a.nnt;
    const int mine = nitems > slot ? (nitems - slot + nwg - 1) / nwg : 0;
    const int imgstride = a.nchunks * T64_WCH + T64_BSB;                 // bytes per N-tile of the weight image
    f32x16 acc[NS];
    int oy0 = 0, ox0 = 0, nt = 0;
    unsigned tb = 0;                                                     // byte offset of the tile's halo origin (tensors stay below 4 GB)
#define T64_ITEM(W, OY, OX, TB, NT)                                                                          \
    {                                                                                                        \
        const int w_ = a.reverse ? nitems - 1 - (W) : (W);                                                   \
        const int t_ = w_ / a.nnt; NT = w_ - t_ * a.nnt;                                                     \
        const int ty_ = t_ / a.tiles_x; OY = ty_ * T64_TH; OX = (t_ - ty_ * a.tiles_x) * 32;                 \
        TB = (unsigned)(OY * a.pitch + OX) * 32u;                                                            \
    }
    if (mine > 0) T64_ITEM(slot, oy0, ox0, tb, nt)
    int poy0 = 0, pox0 = 0, pnt = 0;

    int stepno = 0;
#define T64_STAMP(K)                                                                                         \
    if (RIFE_ABL(TAG & T64_STAMPS) && lane == 0 && stepno < 32)                                                      \
        a.stamps[(((size_t)blockIdx.x * 8 + r) * 32 + stepno) * 4 + (K)] = (long long)__builtin_readcyclecounter();
    // halo chunk C of the tile at TB -> in[PAR]; weight chunk C -> w[PAR]
#define T64_DMA_IN(TB, C, PAR)                                                                               \
    if (!RIFE_ABL(TAG & T64_NODMA) && LW == 0) {                                                                                \
        const unsigned char* src_ = a.in + ((TB) + (unsigned)(2 * (C)) * a.plane);       /* wave-uniform base + 32-bit lane offset */ \
        unsigned char* dst_ = lds + T64_LDS_IN + (PAR) * T64_INB + r * 1024;                                 \
        t64_glds16(src_ + soff[0], dst_);                                                                    \
        t64_glds16(src_ + soff[1], dst_ + 8 * 1024);                                                         \
        if (r < 6 && s2ok) t64_glds16(src_ + soff[2], dst_ + 16 * 1024);                                     \
    }
#define T64_DMA_W(NT, C, PAR)                                                                                \
    if (!RIFE_ABL(TAG & T64_NODMA) && LW == 0) {                                                                                \
        const unsigned char* src_ = a.img + ((NT) * imgstride + (C) * T64_WCH + r * 1024) + lane * 16;       \
        unsigned char* dst_ = lds + T64_LDS_W + (PAR) * T64_WCH + r * 1024;                                  \
        t64_glds16(src_, dst_);                                                                              \
        t64_glds16(src_ + 8 * 1024, dst_ + 8 * 1024);                                                        \
        if (r + 16 < T64_WCH / 1024) t64_glds16(src_ + 16 * 1024, dst_ + 16 * 1024);                         \
        if (r + 24 < T64_WCH / 1024) t64_glds16(src_ + 24 * 1024, dst_ + 24 * 1024);                         \
    }
#define T64_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
    // IDN: the 32-channel output block (of this work item) that K chunk C is the skip connection of, or -1
#define T64_TAPS(C, PAR, IDN)                                                                                \
    T64_STAMP(3)                                                                                             \
    if (!RIFE_ABL(TAG & T64_NOMATH)) {                                                                               \
        _Pragma("unroll") for (int t = 0; t < 9; t++) {                                                      \
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ap[t] + (PAR) * T64_INB);                       \
            const f16x8 al = *reinterpret_cast<const f16x8*>(ap[t] + (PAR) * T64_INB + T64_PLANE);           \
            f16x8 bw_[NS];                                                                                   \
            _Pragma("unroll") for (int n = 0; n < NS; n++) bw_[n] = *reinterpret_cast<const f16x8*>(wb + (PAR) * T64_WCH + t * (2 * CH * 16) + n * 512); \
            _Pragma("unroll") for (int n = 0; n < NS; n++) acc[n] = T64_MFMA(bw_[n], ah, acc[n]);            \
            _Pragma("unroll") for (int n = 0; n < NS; n++) acc[n] = T64_MFMA(bw_[n], al, acc[n]);            \
        }                                                                                                    \
        if ((IDN) >= 0) {   /* skip connection: identity on the centre pixel; K chunk C only feeds output block C >> 1 */ \
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ap[4] + (PAR) * T64_INB);                       \
            const f16x8 al = *reinterpret_cast<const f16x8*>(ap[4] + (PAR) * T64_INB + T64_PLANE);           \
            _Pragma("unroll") for (int n = 0; n < NS; n++)                                                   \
                if ((IDN) == n) { acc[n] = T64_MFMA(idf[(PAR)], ah, acc[n]); acc[n] = T64_MFMA(idf[(PAR)], al, acc[n]); } \
        }                                                                                                    \
    }
#define T64_SYNC()                                                                                           \
    {                                                                                                        \
        T64_STAMP(0)                                                                                         \
        if (!RIFE_ABL(TAG & T64_NOVMWAIT) && LW == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* LW > 0: only epilogue stores are in flight here */ \
        T64_STAMP(1)                                                                                         \
        __builtin_amdgcn_s_barrier();                                                                        \
        T64_STAMP(2)                                                                                         \
        stepno++;                                                                                            \
    }
    // y = slope(acc + bias) -> hi / lo entries of chunk 2 n + h, 32 bytes per pixel and plane
#define T64_EPILOGUE(OY0, OX0, NT, BUF)                                                                      \
    {                                                                                                        \
        const float* const bs = bs0 + (BUF) * (T64_BSB / 4);                                                 \
        const int oy_ = (OY0) + r, ox_ = (OX0) + li;                                                         \
        const bool ok_ = oy_ < a.H && ox_ < a.W && (!RIFE_ABL(TAG & T64_NOSTORE) || acc[0][0] == 123.456f);          /* ablation: (almost) never true, keeps the matrix work alive */ \
        unsigned char* const o_ = a.out + ((unsigned)(2 * h + 4 * NS * (NT)) * a.plane + (unsigned)((oy_ + 1) * a.pitch + ox_ + 1) * 32u); \
        _Pragma("unroll") for (int n = 0; n < NS; n++) {                                                     \
            float v_[16];                                                                                    \
            _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                  \
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bs + n * 32 + 16 * h + 4 * q);              \
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(bs + CH + n * 32 + 16 * h + 4 * q);         \
                _Pragma("unroll") for (int k = 0; k < 4; k++) {                                              \
                    const float y = acc[n][4 * q + k] + b4[k];                                               \
                    v_[4 * q + k] = y < 0.f ? y * s4[k] : y;                                                 \
                }                                                                                            \
            }                                                                                                \
            s16_store_chunk(v_, o_ + (unsigned)(4 * n) * a.plane, a.plane, ok_);                             \
        }                                                                                                    \
    }

    // ---- prologue: bias / slopes, weight chunk 0, halo chunk 0 of the first work item
#define T64_DMA_BS(NT, BUF)                                                                                  \
    if (r == 7 && lane < T64_BSB / 16 && !RIFE_ABL(TAG & T64_NODMA) && LW == 0)                                      \
        t64_glds16(a.img + ((NT) * imgstride + a.nchunks * T64_WCH) + lane * 16, lds + T64_LDS_BS + (BUF) * T64_BSB);
    if RIFE_ABL(TAG & T64_NODMA) { for (int i = tid; i < T64_LDS / 16; i += T64_NTHR) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    if (mine > 0) { T64_DMA_BS(nt, 0) T64_DMA_W(nt, 0, 0) T64_DMA_IN(tb, 0, 0) }
    T64_SYNC()

    // one step = one K chunk of one work item: send chunk c + 1 (or the next item's chunk 0) and its weights to the other buffers, then
    // the matrix work of chunk c.  The chunk count is even, so chunk c always sits in buffer c & 1.
#define T64_STEP(C, PAR, LAST)                                                                               \
    {                                                                                                        \
        if (!(LAST)) { T64_DMA_IN(tb, (C) + 1, (PAR) ^ 1) T64_DMA_W(nt, (C) + 1, (PAR) ^ 1) }                \
        else if (more) { T64_DMA_IN(tbn, 0, 0) T64_DMA_W(ntn, 0, 0) }                                        \
        if ((C) == 1 && more) T64_DMA_BS(ntn, (k + 1) & 1)      /* that buffer served the epilogue of item k - 1 during step 0 */ \
        if ((C) == 0) {                                                                                      \
            if (k > 0) T64_EPILOGUE(poy0, pox0, pnt, (k - 1) & 1)                                            \
            _Pragma("unroll") for (int n = 0; n < NS; n++)                                                   \
                _Pragma("unroll") for (int q = 0; q < 16; q++) acc[n][q] = 0.f;                              \
        }                                                                                                    \
        const int idn_ = ((C) >> 1) - nt * NS;                                                               \
        T64_TAPS(C, PAR, (idn_ >= 0 && idn_ < NS) ? idn_ : -1)                                               \
        T64_SYNC()                                                                                           \
    }
    for (int k = 0; k < mine; k++) {
        const bool more = k + 1 < mine;
        int oy0n = 0, ox0n = 0, ntn = 0;
        unsigned tbn = 0;
        if (more) T64_ITEM(slot + (k + 1) * nwg, oy0n, ox0n, tbn, ntn)
        for (int c = 0; c < a.nchunks; c += 2) {
            T64_STEP(c, 0, false)
            T64_STEP(c + 1, 1, c + 2 >= a.nchunks)
        }
        poy0 = oy0; pox0 = ox0; pnt = nt;
        oy0 = oy0n; ox0 = ox0n; tb = tbn; nt = ntn;
    }
    if (mine > 0) T64_EPILOGUE(poy0, pox0, pnt, (mine - 1) & 1)
    if (RIFE_ABL(TAG & T64_CLK) && tid == 0) {
        a.stamps[4 * blockIdx.x] = (long long)__builtin_readcyclecounter() - clk0;
        a.stamps[4 * blockIdx.x + 1] = rt0;
        a.stamps[4 * blockIdx.x + 2] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#undef T64_ITEM
#undef T64_DMA_BS
#undef T64_STEP
#undef T64_STAMP
#undef T64_DMA_IN
#undef T64_DMA_W
#undef T64_MFMA
#undef T64_TAPS
#undef T64_SYNC
#undef T64_EPILOGUE
}

}  // namespace rife
