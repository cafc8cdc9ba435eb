// conv_t64_kernel: the 64 -> 64 channel residual trunk convolution of the finest IFBlock
// (reference models/rife-v4.6/flownet.param:169-197: Split, Convolution 3x3 pad 1, BinaryOp add, ReLU slope 0.2; 8 per pair,
// 44 % of the pair's MACs) as ONE persistent workgroup per CU on the split-f16 matrix path of conv_h2b_kernel.
//
// What changed against conv_h2b_kernel (whose phase trace showed ~10 of the 21 us of every workgroup in latency-bound prologue /
// epilogue, two workgroups per CU, matrix pipe 42 % busy):
//   * "S16" activations: the trunk tensor is stored pre-split, per pixel and 16-channel chunk one 64-byte record
//     {hi = f16(x) x 16, lo = f16(x - hi) x 16} (the same 4 bytes per element as fp32, and exactly the values the old kernel
//     computed at LDS-staging time), in a zero-bordered allocation: (rows + 2) x pitch pixels, pixel (y, x) at (y + 1, x + 1).
//     The halo of a tile is then plain memory: no bounds tests, no conversion, and global_load_lds_dwordx4 can move it.
//   * the weights of all four K chunks (9 taps x 64 x 64 f16 = 72 KB), the two identity slabs of the skip connection, bias and
//     slopes are loaded into LDS once per workgroup and stay there; nothing but the halo tiles moves in the steady state.
//   * 16 waves = two groups of 8, each walking its own sequence of 8-row x 32-column tiles; group 1 runs two K chunks behind
//     group 0, so one group's epilogue (VALU + stores) and tile switch sit under the other group's matrix work.
//   * every step (one K chunk of one tile: 38 MFMAs per wave) starts by sending the NEXT step's halo chunk on its way with
//     LDS-DMA (22 x 1 KiB per group) and ends with vmcnt(0) + one s_barrier; the load has the whole step to land.
//   * output channels are permuted inside each 32-row MFMA block (a property of the weight packing only) so that a lane ends
//     up with 16 CONSECUTIVE channels of one pixel = one whole S16 record: the epilogue needs no LDS transpose and no barrier.
// LDS image of a halo chunk: hi plane [340 px][32 B] then lo plane [340 px][32 B]; the two 16-byte halves of a 32-byte entry are
// swapped when bit 3 of the pixel index is set, which makes the ds_read_b128 fragment reads of 16 consecutive pixels hit 64
// distinct banks.  The DMA writes LDS lane-linear, so the swap is applied to each lane's SOURCE address (cdna guide, rule 21).
// Arithmetic (products, accumulation order, epilogue) is that of conv_h2b_kernel<2, 10>: results are bit-identical.
#pragma once
#include "conv_mfma.h"

namespace rife {

constexpr int T64_IH = 10, T64_IW = 34, T64_NPX = T64_IH * T64_IW;      // halo tile of an 8 x 32 output tile
constexpr int T64_PLANE = T64_NPX * 32;                                  // 10,880 B: hi (or lo) halves of one 16-channel chunk
constexpr int T64_INB = 2 * T64_PLANE;                                   // 21,760 B per chunk buffer
constexpr int T64_WB = 4 * 9 * 2048;                                     // 73,728 B: [chunk][tap][k half][64 rows][8 f16]
constexpr int T64_IDB = 2 * 1024;                                        // identity slabs [chunk parity][k half][32 rows][8 f16]
constexpr int T64_BSB = 2 * 64 * 4;                                      // bias[64], slope[64]
constexpr int T64_IMG = T64_WB + T64_IDB + T64_BSB;                      // 76,288 B: static LDS image, built on the host
constexpr int T64_LDS = T64_IMG + 4 * T64_INB;                           // 163,328 B (limit 163,840)
static_assert(T64_LDS <= 160 * 1024, "LDS budget");

// row i of a 32-row MFMA block <-> output channel (within the block): a lane's 16 accumulator registers are 16 consecutive channels
__host__ __device__ constexpr int s16_row_channel(int i) { return 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3); }

struct T64Args {
    const unsigned char* in;     // S16 tensor, allocation start (= pixel (-1, -1))
    unsigned char* out;          // S16 tensor of the same geometry
    const unsigned char* img;    // T64_IMG bytes
    int H, W;                    // valid pixels
    int pitch;                   // pixels per allocation row (tiles_x * 32 + 2)
    int tiles_x, ntiles;
    int rounds;                  // tiles per group stream (ceil(ntiles / (2 * gridDim.x)))
    long long* stamps = nullptr; // bench builds only (TAG & T64_STAMPS): [workgroup][wave 16][step 32][4] shader-clock stamps
};
// bench-only ablation bits of TAG (timing experiments; the results of all but T64_STAMPS are garbage).  The product instantiates TAG = 3.
enum { T64_NOSTORE = 0x100, T64_NODMA = 0x200, T64_NOMATH = 0x400, T64_NOVMWAIT = 0x800, T64_STAMPS = 0x1000, T64_INPHASE = 0x2000 };

typedef __attribute__((address_space(3))) unsigned char t64_lds_u8;
typedef __attribute__((address_space(1))) const unsigned char t64_glb_u8;
// 64 lanes x 16 bytes, global -> LDS without passing through registers; LDS destination = wave-uniform base + lane * 16
__device__ __forceinline__ void t64_glds16(const unsigned char* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((t64_glb_u8*)g, (t64_lds_u8*)l, 16, 0, 0);
}

template <int TAG>
__global__ __launch_bounds__(1024) void conv_t64_kernel(T64Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* const lds = ldsb;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);            // provably wave-uniform: the barrier counts below depend on it
    const int g = wv >> 3, r = wv & 7;                                    // group, output row of the group's tile
    const int h = lane >> 5, li = lane & 31;

    // ---- static image: 74.5 KiB, lane-linear
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int i = wv + 16 * j;
        if (i * 1024 + lane * 16 < T64_IMG) t64_glds16(a.img + i * 1024 + lane * 16, lds + i * 1024);
    }

    // ---- per-lane constants
    // DMA: piece i = r + 8 j of the group's chunk buffer covers LDS slots 64 i .. 64 i + 63 (16 bytes each)
    unsigned soff[3];
    bool s2ok;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const int s = (r + 8 * j) * 64 + lane;
        const int pl = s >= 2 * T64_NPX ? 1 : 0, s1 = s - pl * 2 * T64_NPX;
        const int P = min(s1 >> 1, T64_NPX - 1), pos = s1 & 1;
        const int kh = pos ^ ((P >> 3) & 1);
        const int py = P / T64_IW, px = P - py * T64_IW;
        soff[j] = (unsigned)(py * a.pitch + px) * 256u + (unsigned)(pl * 32 + kh * 16);
        if (j == 2) s2ok = s < 4 * T64_NPX;
    }
    unsigned char* const inb = lds + T64_IMG + g * 2 * T64_INB;
    // fragment addresses of the nine taps (hi plane; lo = + T64_PLANE), pixel P = (r + dy) * 34 + li + dx
    const unsigned char* ap[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int P = (r + t / 3) * T64_IW + li + t % 3;
        ap[t] = inb + P * 32 + ((h ^ ((P >> 3) & 1)) << 4);
    }
    const unsigned char* const wb = lds + h * 1024 + li * 16;
    const unsigned char* const idb = lds + T64_WB + h * 512 + li * 16;
    const float* const bs = reinterpret_cast<const float*>(lds + T64_WB + T64_IDB);

    // ---- tile streams: workgroup b runs on XCD b % 8; per round every XCD walks a contiguous band of tiles
    const int nwg = gridDim.x, b = blockIdx.x;
    const int per_xcd = (nwg >> 3) * 2;
    const int slot = (b & 7) * per_xcd + (b >> 3) * 2 + g;
    const int per_round = nwg * 2;

    // this group's stream: tiles slot, slot + per_round, ... (mine of them exist); the other rounds only keep the barrier count
    const int mine = a.ntiles > slot ? (a.ntiles - slot + per_round - 1) / per_round : 0;
    f32x16 acc[2];
    int oy0 = 0, ox0 = 0;
    unsigned tb = 0;                                                     // byte offset of the tile's halo origin (tensors stay below 4 GB)
    if (mine > 0) { const int ty = slot / a.tiles_x; oy0 = ty * 8; ox0 = (slot - ty * a.tiles_x) * 32; tb = (unsigned)(oy0 * a.pitch + ox0) * 256u; }
    int poy0 = 0, pox0 = 0;

    int stepno = 0;
#define T64_STAMP(K)                                                                                         \
    if ((TAG & T64_STAMPS) && lane == 0 && stepno < 32)                                                      \
        a.stamps[(((size_t)blockIdx.x * 16 + wv) * 32 + stepno) * 4 + (K)] = (long long)__builtin_readcyclecounter();
#define T64_DMA(TB, C, PAR)                                                                                  \
    if (!(TAG & T64_NODMA)) {                                                                                \
        const unsigned char* src_ = a.in + (TB) + (C) * 64;       /* wave-uniform base + 32-bit lane offset */ \
        unsigned char* dst_ = inb + (PAR) * T64_INB + r * 1024;                                              \
        t64_glds16(src_ + soff[0], dst_);                                                                    \
        t64_glds16(src_ + soff[1], dst_ + 8 * 1024);                                                         \
        if (r < 6 && s2ok) t64_glds16(src_ + soff[2], dst_ + 16 * 1024);                                     \
    }
#define T64_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define T64_TAPS(C, PAR)                                                                                     \
    T64_STAMP(3)                                                                                             \
    if (!(TAG & T64_NOMATH)) {                                                                               \
        _Pragma("unroll") for (int t = 0; t < 9; t++) {                                                      \
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ap[t] + (PAR) * T64_INB);                       \
            const f16x8 al = *reinterpret_cast<const f16x8*>(ap[t] + (PAR) * T64_INB + T64_PLANE);           \
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(wb + ((C) * 9 + t) * 2048);                     \
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(wb + ((C) * 9 + t) * 2048 + 512);               \
            acc[0] = T64_MFMA(b0, ah, acc[0]);                                                               \
            acc[1] = T64_MFMA(b1, ah, acc[1]);                                                               \
            acc[0] = T64_MFMA(b0, al, acc[0]);                                                               \
            acc[1] = T64_MFMA(b1, al, acc[1]);                                                               \
        }                                                                                                    \
        {   /* skip connection: identity on the centre pixel; chunk C only feeds output block C >> 1 */      \
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ap[4] + (PAR) * T64_INB);                       \
            const f16x8 al = *reinterpret_cast<const f16x8*>(ap[4] + (PAR) * T64_INB + T64_PLANE);           \
            const f16x8 bi = *reinterpret_cast<const f16x8*>(idb + ((C) & 1) * 1024);                        \
            acc[(C) >> 1] = T64_MFMA(bi, ah, acc[(C) >> 1]);                                                 \
            acc[(C) >> 1] = T64_MFMA(bi, al, acc[(C) >> 1]);                                                 \
        }                                                                                                    \
    }
#define T64_SYNC()                                                                                           \
    {                                                                                                        \
        T64_STAMP(0)                                                                                         \
        if (!(TAG & T64_NOVMWAIT)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* this wave's LDS-DMA pieces have landed */ \
        T64_STAMP(1)                                                                                         \
        if (TAG & T64_NOVMWAIT) __builtin_amdgcn_s_barrier(); else __syncthreads();                          \
        T64_STAMP(2)                                                                                         \
        stepno++;                                                                                            \
    }
    // y = slope(acc + bias) -> {hi, lo} records, quad-transposed so that every store instruction writes whole 128-byte lines
#define T64_EPILOGUE(OY0, OX0)                                                                               \
    {                                                                                                        \
        const int oy_ = (OY0) + r, oxq_ = (OX0) + (li & ~3);                                                 \
        const bool ok_ = oy_ < a.H && oxq_ < a.W && (!(TAG & T64_NOSTORE) || acc[0][0] == 123.456f);         /* ablation: (almost) never true, keeps the matrix work alive */ \
        unsigned char* const o_ = a.out + ((unsigned)((oy_ + 1) * a.pitch + oxq_ + 1) * 256u + (unsigned)(h * 64)); \
        _Pragma("unroll") for (int n = 0; n < 2; n++) {                                                      \
            float v_[16];                                                                                    \
            _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                  \
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bs + n * 32 + 16 * h + 4 * q);              \
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(bs + 64 + n * 32 + 16 * h + 4 * q);         \
                _Pragma("unroll") for (int k = 0; k < 4; k++) {                                              \
                    const float y = acc[n][4 * q + k] + b4[k];                                               \
                    v_[4 * q + k] = y < 0.f ? y * s4[k] : y;                                                 \
                }                                                                                            \
            }                                                                                                \
            s16_store_record(v_, o_ + n * 128, lane, ok_);                                                   \
        }                                                                                                    \
    }

    if (mine > 0) T64_DMA(tb, 0, 0)
    T64_SYNC()
    if (g == 1 && !(TAG & T64_INPHASE)) { __syncthreads(); __syncthreads(); }      // group 1 runs two steps behind group 0

    for (int k = 0; k < mine; k++) {
        const int Tn = slot + (k + 1) * per_round;
        const bool more = k + 1 < mine;
        int oy0n = 0, ox0n = 0;
        unsigned tbn = 0;
        if (more) { const int ty = Tn / a.tiles_x; oy0n = ty * 8; ox0n = (Tn - ty * a.tiles_x) * 32; tbn = (unsigned)(oy0n * a.pitch + ox0n) * 256u; }

        T64_DMA(tb, 1, 1)
        if (k > 0) T64_EPILOGUE(poy0, pox0)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[n][q] = 0.f;
        T64_TAPS(0, 0)
        T64_SYNC()

        T64_DMA(tb, 2, 0)
        T64_TAPS(1, 1)
        T64_SYNC()

        T64_DMA(tb, 3, 1)
        T64_TAPS(2, 0)
        T64_SYNC()

        if (more) T64_DMA(tbn, 0, 0)
        T64_TAPS(3, 1)
        T64_SYNC()

        poy0 = oy0; pox0 = ox0;
        oy0 = oy0n; ox0 = ox0n; tb = tbn;
    }
    if (mine > 0) T64_EPILOGUE(poy0, pox0)
    for (int k = mine; k < a.rounds; k++) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
    if (g == 0 && !(TAG & T64_INPHASE)) { __syncthreads(); __syncthreads(); }
#undef T64_DMA
#undef T64_MFMA
#undef T64_TAPS
#undef T64_SYNC
#undef T64_EPILOGUE
}

}  // namespace rife
