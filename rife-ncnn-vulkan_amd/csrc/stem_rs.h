// stem_rs_kernel: the whole input side of the finest IFBlock of rife-v4.6 in ONE row-streaming kernel - the block-input assembly (2x rife.Warp,
// Concat; reference models/rife-v4.6/flownet.param:160-165, src/warp.cpp:96-168) and BOTH stride-2 stem convolutions (convrelu_6: 12 -> 32
// channels, flownet.param:166, and the 32 -> 64 one, :167-168), from the frames and F, M straight to the S16 trunk tensor at quarter resolution.
// Round 3.  Before: stem0_fused_kernel<1, 1, 256> wrote the 32-channel half-resolution tensor as fp32 (267 MB per 4K pair) and
// conv_h2s2_kernel<2, true> read it back: 188 + 108 us per pair and 534 MB of the pair's 7.6 GB of traffic for a tensor nobody else wants.
//
// A workgroup (512 threads, two per CU) walks DOWN a strip of 31 quarter-resolution columns; one step = one quarter-resolution output row:
//   gather   every thread assembles ONE block-input pixel of the step's four new full-resolution rows (4 x 127 pixels = 508 of 512 threads;
//            assemble_pixel's arithmetic in two halves, warp_issue / warp_finish): the F, M loads run two steps ahead of their use and the
//            image taps one step ahead, across the matrix phases below - the gather's two dependent memory round trips (what the tile kernel
//            waited for 70 % of its time) are in flight while the workgroup does something else.  Pixels go to LDS ring A as split f16
//            {hi, lo} (5 row slots; columns split by parity so that the stride-2 operand reads are contiguous).
//   stem 0   waves 4-7: the two new half-resolution rows x two 32-pixel tiles, 18 MFMAs each (weights in registers) -> bias, LeakyReLU, {hi, lo}
//            -> LDS ring B (3 row slots).
//   stem 1   waves 0-3: output block n = wave & 1 (32 of the 64 channels) x K chunk wave >> 1 (16 of the 32 input channels), 18 MFMAs each on
//            9 register-resident weight fragments; the two K partial sums meet through 8 KB of LDS, waves 0-1 finish (bias, LeakyReLU) and store
//            S16 entries (conv_t64.h) - 992 contiguous bytes per plane and instruction.
// Three barriers per step; every wave runs the same loop (wave-uniform role tests), so the barrier counts cannot diverge.  The first step of a
// strip (and of a workgroup's range) needs three more input rows and one more half-resolution row: a blocking "pre-step".
// Arithmetic: the block input is assemble_pixel<1>'s bit for bit; stem 0 accumulates in stem0_fused_kernel's order (bit-identical values before
// the split); stem 1 adds two K partial sums instead of one chain over both chunks: last-bit differences against conv_h2s2_kernel, held together
// by tests/test_gpu_stem_rs.py.
// LDS: ring A 5 x 8,448 B + ring B 3 x 8,448 B + partial sums 8 KB + biases = 76,544 B: two workgroups per CU.
#pragma once
#include "conv_mfma.h"
#include "elementwise.h"

namespace rife {

constexpr int SRS_SW = 31;                           // quarter-resolution columns per strip
constexpr int SRS_AW = 4 * SRS_SW + 3;               // 127 block-input columns of a strip row
constexpr int SRS_BW = 2 * SRS_SW + 1;               // 63 stem-0 columns
constexpr int SRS_A_PAR = 66 * 32;                   // ring A, one column parity of one (hi | lo) plane: [m 66][32 B]
constexpr int SRS_A_HL = 2 * SRS_A_PAR;              // 4,224
constexpr int SRS_A_ROW = 2 * SRS_A_HL;              // 8,448: [hi | lo][parity][m][16 channels f16]
constexpr int SRS_A_SLOTS = 5;
constexpr int SRS_B_PAR = 33 * 32;                   // ring B, one column parity of one (chunk, hi | lo) plane: [m 33][32 B]
constexpr int SRS_B_HL = 2 * SRS_B_PAR;              // 2,112
constexpr int SRS_B_ROW = 4 * SRS_B_HL;              // 8,448: [chunk 2][hi | lo][parity][m][16 channels f16]
constexpr int SRS_B_SLOTS = 3;
constexpr int SRS_LDS_B = SRS_A_SLOTS * SRS_A_ROW;                   // 42,240
constexpr int SRS_LDS_STG = SRS_LDS_B + SRS_B_SLOTS * SRS_B_ROW;     // 67,584: K partial sums of stem 1, [n 2][quad 4][lane 64][16 B]
constexpr int SRS_LDS_BS = SRS_LDS_STG + 2 * 4096;                   // 75,776: bias0[32] slope0[32] bias1[64] slope1[64]
constexpr int SRS_LDS = SRS_LDS_BS + 768;                            // 76,544
constexpr int SRS_NTHR = 512;

struct StemRsArgs {
    const uint32_t *img0, *img1;
    const float4* F; const float* M;
    const void* w0;              // stem-0 weights: f16 [tap 9][k half 2][32][8] (pack_weights_h2)
    const float *bias0, *slope0;
    const void* w1;              // stem-1 weights: f16 [chunk 2][tap 9][k half 2][64][8], rows permuted by s16_row_channel (pack_weights_h2_perm)
    const float *bias1, *slope1;
    unsigned char* out;          // S16 tensor, 64 channels, quarter resolution
    float timestep;
    const float* tsp;            // != null: timestep read from device memory (hipGraph replays)
    int wp, hp;                  // padded full resolution = resolution of the block input (scale 1)
    int Hq, Wq;                  // valid pixels of out
    int pitch; unsigned plane;   // S16 geometry of out
    int nunits;                  // strips x Hq
    long long* stamps = nullptr; // bench builds (TAG & SRS_CLK): [workgroup][wave 8][phase 8] shader cycles summed over the workgroup's steps
};

// bench-only ablation bits of TAG (timing experiments through rife_hip_bench_stem_rs; results are garbage).  The product instantiates TAG = 0.
enum { SRS_NOTAPS = 1, SRS_NOFM = 2, SRS_NOMATH = 4, SRS_NOSTORE = 8, SRS_NOFINISH = 16, SRS_CLK = 32 };

struct SrsCursor {
    int strip, q;
    __device__ __forceinline__ void init(int u, int Hq) { strip = u / Hq; q = u - strip * Hq; }
    __device__ __forceinline__ bool advance(int Hq) { if (++q >= Hq) { q = 0; ++strip; return true; } return false; }     // true: the next step starts a strip
};
struct SrsFM { float4 f; float m; };
struct SrsTaps { WarpLoads a, b; };

// workgroup barrier for LDS traffic only: __syncthreads() would also wait for every global load in flight (vmcnt(0)) - the prefetched
// F, M and image taps this kernel keeps in flight ACROSS its barriers
#define SRS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// 8-byte global store the compiler does not see (tail_rs.h, trs_store_dword: a store the compiler knows to be in flight turns its next wait for
// a loaded value into vmcnt(0), which drains the prefetched F, M and taps of waves 0-1 every step)
__device__ __forceinline__ void srs_store_b64(unsigned char* base, unsigned off, f16x4 v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 d;
    __builtin_memcpy(&d, &v, 8);
    asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(off), "v"(d), "s"(base) : "memory");
}

template <int TAG>
__global__ __launch_bounds__(SRS_NTHR) __attribute__((amdgpu_waves_per_eu(4, 4))) void stem_rs_kernel(StemRsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const float timestep = a.tsp ? *a.tsp : a.timestep;
    const int Hb = a.hp, Wb = a.wp, Hh = Hb / 2, Wh = Wb / 2;

    const int nwg = gridDim.x, wg = blockIdx.x;
    const int u0 = (int)((long long)a.nunits * wg / nwg), u1 = (int)((long long)a.nunits * (wg + 1) / nwg);
    const int S = u1 - u0;
    if (S <= 0) return;

    float* const lbs = reinterpret_cast<float*>(ldsb + SRS_LDS_BS);
    if (tid < 32) lbs[tid] = a.bias0[tid];
    else if (tid < 64) lbs[tid] = a.slope0[tid - 32];
    else if (tid < 128) lbs[tid] = a.bias1[tid - 64];
    else if (tid < 192) lbs[tid] = a.slope1[tid - 128];

    // nine weight fragments per wave, in registers for the whole launch: stem-1 waves (n, K chunk), stem-0 waves the one 32-channel block
    f16x8 Wf[9];
    {
        const unsigned char* src = wv < 4 ? reinterpret_cast<const unsigned char*>(a.w1) + ((size_t)((wv >> 1) * 9 * 2 + half) * 64 + (wv & 1) * 32 + li) * 16
                                          : reinterpret_cast<const unsigned char*>(a.w0) + ((size_t)half * 32 + li) * 16;
        const int tstride = wv < 4 ? 2 * 64 * 16 : 2 * 32 * 16;
#pragma unroll
        for (int t = 0; t < 9; t++) Wf[t] = *reinterpret_cast<const f16x8*>(src + t * tstride);
    }

    // gather role of this thread: pixel (row grow, column gcol) of the rows a step adds
    const int grow = tid / SRS_AW, gcol = tid - grow * SRS_AW;
    const unsigned gdst = (unsigned)((gcol & 1) * SRS_A_PAR + (gcol >> 1) * 32);
    const int gsw = ((gcol >> 1) >> 3) & 1;

    auto slot5 = [](int r) { int s = r % 5; return s < 0 ? s + 5 : s; };
    auto slot3 = [](int r) { int s = r % 3; return s < 0 ? s + 3 : s; };

    // F, M of this thread's pixel of the rows rb .. of strip `strip` (clamped coordinates: the zero padding is a select at the end)
    auto load_fm = [&](int strip, int rb) -> SrsFM {
        const int by = rb + grow, bx = 4 * SRS_SW * strip - 3 + gcol;
        const int cx = min(max(bx, 0), Wb - 1), cy = min(max(by, 0), Hb - 1);
        const unsigned i = (unsigned)(cy * a.wp + cx);                   // 32-bit offsets: scalar base + one VGPR (F: 16 B x 4K pixels = 134 MB)
        SrsFM r;
        if (RIFE_ABL(TAG & SRS_NOFM)) { r.f = make_float4(0.25f * (float)(bx & 7), -0.5f, 1.5f, 0.75f); r.m = 0.1f; return r; }
        r.f = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(a.F) + i * 16u);
        r.m = *reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.M) + i * 4u);
        return r;
    };
    auto issue_taps = [&](int strip, int rb, const SrsFM& fm) -> SrsTaps {
        const int by = rb + grow, bx = 4 * SRS_SW * strip - 3 + gcol;
        const int cx = min(max(bx, 0), Wb - 1), cy = min(max(by, 0), Hb - 1);
        SrsTaps t;
        if (RIFE_ABL(TAG & SRS_NOTAPS)) {
            t.a.r0 = t.a.r1 = t.b.r0 = t.b.r1 = make_uint2((unsigned)cx * 0x10101u, (unsigned)cy * 0x10101u);
            t.a.alpha = t.b.alpha = fm.f.x - floorf(fm.f.x); t.a.beta = t.b.beta = fm.f.y - floorf(fm.f.y); t.a.l0 = t.b.l0 = true; t.a.l1 = t.b.l1 = false;
            return t;
        }
        t.a = warp_issue(a.img0, cx, cy, fm.f.x, fm.f.y, a.wp, a.hp);
        t.b = warp_issue(a.img1, cx, cy, fm.f.z, fm.f.w, a.wp, a.hp);
        return t;
    };
    // the 12 channels {warp(in0, F.xy) rgb, warp(in1, F.zw) rgb, t, M, F} (assemble_pixel<1>), zero outside the block input, split, -> ring A
    auto finish = [&](int strip, int rb, const SrsFM& fm, const SrsTaps& t, int nactive) {
        const int by = rb + grow, bx = 4 * SRS_SW * strip - 3 + gcol;
        const bool in = by >= 0 && by < Hb && bx >= 0 && bx < Wb;
        if (RIFE_ABL(TAG & SRS_NOFINISH)) { if (fm.m == 123.456f && t.a.alpha == 7.f && t.b.r0.x == 99u) ldsb[tid] = 1; return; }
        unsigned char* const d = ldsb + slot5(by) * SRS_A_ROW + gdst;
        const bool wr = tid < nactive;
        {   // channels 8 .. 11 (F) + four zeros first: the F, M registers die here
            f16x8 h1, l1;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const float vb = !in ? 0.f : c == 0 ? fm.f.x : c == 1 ? fm.f.y : c == 2 ? fm.f.z : c == 3 ? fm.f.w : 0.f;
                const _Float16 hb = (_Float16)vb;
                h1[c] = hb; l1[c] = (_Float16)(vb - (float)hb);
            }
            if (wr) { *reinterpret_cast<f16x8*>(d + ((1 ^ gsw) << 4)) = h1; *reinterpret_cast<f16x8*>(d + SRS_A_HL + ((1 ^ gsw) << 4)) = l1; }
        }
        __builtin_amdgcn_sched_barrier(0);
        const float3 w0 = warp_finish(t.a);
        __builtin_amdgcn_sched_barrier(0);
        const float3 w1 = warp_finish(t.b);
        __builtin_amdgcn_sched_barrier(0);
        const float O[8] = {w0.x, w0.y, w0.z, w1.x, w1.y, w1.z, timestep, fm.m};
        f16x8 h0, l0;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float va = in ? O[c] : 0.f;
            const _Float16 ha = (_Float16)va;
            h0[c] = ha; l0[c] = (_Float16)(va - (float)ha);
        }
        if (wr) { *reinterpret_cast<f16x8*>(d + ((0 ^ gsw) << 4)) = h0; *reinterpret_cast<f16x8*>(d + SRS_A_HL + ((0 ^ gsw) << 4)) = l0; }
    };

    // stem 0: half-resolution row h, columns tile * 32 + li of the strip -> ring B (zeros outside the half-resolution tensor: stem 1's padding)
    // (half, li as parameters: the main loop passes copies the compiler cannot see through, so that it does not hoist the dozens of loop-invariant
    // LDS addresses out of the loop - and then spill them: the budget is 128 registers)
    auto stem0_job = [&](int strip, int h, int tile, const int half, const int li) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        const int c = tile * 32 + li;
        unsigned rowb[3];
#pragma unroll
        for (int dy = 0; dy < 3; dy++) rowb[dy] = (unsigned)(slot5(2 * h - 1 + dy) * SRS_A_ROW);
        unsigned colb[3];
#pragma unroll
        for (int dx = 0; dx < 3; dx++) { const int m = c + (dx >> 1); colb[dx] = (unsigned)((dx & 1) * SRS_A_PAR + m * 32 + ((half ^ ((m >> 3) & 1)) << 4)); }
        f16x8 ah[3], al[3];                                              // fragments two taps ahead of their MFMAs (LDS latency > one tap's 64 cycles), no more: registers
#pragma unroll
        for (int t = 0; t < 2; t++) {
            ah[t] = *reinterpret_cast<const f16x8*>(ldsb + rowb[t / 3] + colb[t % 3]);
            al[t] = *reinterpret_cast<const f16x8*>(ldsb + rowb[t / 3] + colb[t % 3] + SRS_A_HL);
        }
#pragma unroll
        for (int t = 0; t < 9; t++) {
            if (t + 2 < 9) {
                ah[(t + 2) % 3] = *reinterpret_cast<const f16x8*>(ldsb + rowb[(t + 2) / 3] + colb[(t + 2) % 3]);
                al[(t + 2) % 3] = *reinterpret_cast<const f16x8*>(ldsb + rowb[(t + 2) / 3] + colb[(t + 2) % 3] + SRS_A_HL);
            }
            if (!RIFE_ABL(TAG & SRS_NOMATH)) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[t], ah[t % 3], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[t], al[t % 3], acc, 0, 0, 0);
            } else acc[t] += (float)ah[t % 3][0] + (float)al[t % 3][1] + (float)Wf[t][2];
            __builtin_amdgcn_sched_barrier(0);
        }
        const int hc = 2 * SRS_SW * strip - 1 + c;
        const bool ok = h >= 0 && h < Hh && hc >= 0 && hc < Wh && c < SRS_BW;
        const int m = c >> 1;
        unsigned char* const d = ldsb + SRS_LDS_B + slot3(h) * SRS_B_ROW + (c & 1) * SRS_B_PAR + m * 32 + half * 8;
        const int sw = (m >> 3) & 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbs + 8 * q + 4 * half);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(lbs + 32 + 8 * q + 4 * half);
            f16x4 hi4, lo4;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float y = acc[4 * q + k] + b4[k];
                float v = y < 0.f ? y * s4[k] : y;
                v = ok ? v : 0.f;
                const _Float16 hh = (_Float16)v;
                hi4[k] = hh; lo4[k] = (_Float16)(v - (float)hh);
            }
            unsigned char* const dq = d + (q >> 1) * (2 * SRS_B_HL) + (((q & 1) ^ sw) << 4);
            *reinterpret_cast<f16x4*>(dq) = hi4;
            *reinterpret_cast<f16x4*>(dq + SRS_B_HL) = lo4;
        }
    };

    // the first step of a strip: rows 4q - 3 .. 4q - 1 of the block input (blocking gather) and half-resolution row 2q - 1
    auto prestep = [&](const SrsCursor& u) {
        const int rb = 4 * u.q - 3;
        const SrsFM fm = load_fm(u.strip, rb);
        const SrsTaps t = issue_taps(u.strip, rb, fm);
        finish(u.strip, rb, fm, t, 3 * SRS_AW);
        SRS_SYNC();
        if (wv == 4 || wv == 5) stem0_job(u.strip, 2 * u.q - 1, wv - 4, half, li);
        SRS_SYNC();
    };

    SrsCursor cur; cur.init(u0, a.Hq);
    SrsFM fm_next;
    {
        const SrsFM fm0 = load_fm(cur.strip, 4 * cur.q);
        SRS_SYNC();                                                 // biases in LDS
        prestep(cur);
        const SrsTaps t0 = issue_taps(cur.strip, 4 * cur.q, fm0);
        finish(cur.strip, 4 * cur.q, fm0, t0, 4 * SRS_AW);
    }
    SrsCursor nxt = cur;
    bool has_next = S > 1, fresh_next = false;
    if (has_next) { fresh_next = nxt.advance(a.Hq); fm_next = load_fm(nxt.strip, 4 * nxt.q); }
    SRS_SYNC();

    // waves 0-1: output row q of the strip: own K partial sum + the partner's (LDS), bias, LeakyReLU, {hi, lo}, S16 entries.  A lane holds
    // channels 32 n + 16 half .. + 15 of pixel ox = the pixel's entry of chunk 2 n + half: four 8-byte pieces per plane.
    // (Measured and dropped: deferring this to the top of the next iteration, where these waves idle behind the stem-0 phase - 234 vs 227-231 us.)
    auto finish_row = [&](const f32x16& acc, int strip, int q, const int half, const int li, const int lane) {
        const int n = wv;
        const f32x4* const sd = reinterpret_cast<const f32x4*>(ldsb + SRS_LDS_STG + n * 4096 + lane * 16);
        const int ox = SRS_SW * strip + li;
        bool pok = li < SRS_SW && ox < a.Wq;
        const unsigned o = (unsigned)(2 * half + 4 * n) * a.plane + (unsigned)((q + 1) * a.pitch + ox + 1) * 32u;      // S16 tensors stay below 4 GB (block_on_s16)
#pragma unroll
        for (int qd = 0; qd < 4; qd++) {
            const f32x4 p = sd[qd * 64];
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbs + 64 + n * 32 + 16 * half + 4 * qd);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(lbs + 128 + n * 32 + 16 * half + 4 * qd);
            f16x4 hi4, lo4;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const float y = (acc[4 * qd + kk] + p[kk]) + b4[kk];
                const float v = y < 0.f ? y * s4[kk] : y;
                const _Float16 hh = (_Float16)v;
                hi4[kk] = hh; lo4[kk] = (_Float16)(v - (float)hh);
            }
            if (RIFE_ABL(TAG & SRS_NOSTORE)) pok = pok && p[0] == 123.456f;
            if (pok) { srs_store_b64(a.out, o + 8u * qd, hi4); srs_store_b64(a.out, o + a.plane + 8u * qd, lo4); }
        }
    };

    long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#define SRS_STAMP(I) if (RIFE_ABL(TAG & SRS_CLK)) { const long long now_ = (long long)__builtin_readcyclecounter(); tph[I] += now_ - tprev; tprev = now_; }
    if (RIFE_ABL(TAG & SRS_CLK)) tprev = (long long)__builtin_readcyclecounter();
    for (int k = 0; k < S; k++) {
        int hf = half, l32 = li, ln = lane;
        asm volatile("" : "+v"(hf), "+v"(l32), "+v"(ln));               // opaque copies of the lane coordinates (see stem0_job)
        // ring A holds the rows of step k (cur).  Loads of the steps after it first: they fly during the matrix phases.
        SrsTaps tp_next;
        SrsFM fm_nn;
        SrsCursor nn = nxt;
        const bool has_nn = k + 2 < S;
        bool fresh_nn = false;
        if (has_next) tp_next = issue_taps(nxt.strip, 4 * nxt.q, fm_next);
        if (has_nn) { fresh_nn = nn.advance(a.Hq); fm_nn = load_fm(nn.strip, 4 * nn.q); }

        SRS_STAMP(0)                                                     // load issue
        if (wv >= 4) stem0_job(cur.strip, 2 * cur.q + ((wv - 4) >> 1), (wv - 4) & 1, hf, l32);
        SRS_STAMP(1)                                                     // stem 0
        SRS_SYNC();
        SRS_STAMP(2)                                                     // barrier

        f32x16 acc;
        if (wv < 4) {
            const int ck = wv >> 1, n = wv & 1;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            unsigned rowb[3];
#pragma unroll
            for (int dy = 0; dy < 3; dy++) rowb[dy] = (unsigned)(SRS_LDS_B + slot3(2 * cur.q - 1 + dy) * SRS_B_ROW + ck * (2 * SRS_B_HL));
            unsigned colb[3];
#pragma unroll
            for (int dx = 0; dx < 3; dx++) { const int m = l32 + (dx >> 1); colb[dx] = (unsigned)((dx & 1) * SRS_B_PAR + m * 32 + ((hf ^ ((m >> 3) & 1)) << 4)); }
            f16x8 ah[3], al[3];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                ah[t] = *reinterpret_cast<const f16x8*>(ldsb + rowb[t / 3] + colb[t % 3]);
                al[t] = *reinterpret_cast<const f16x8*>(ldsb + rowb[t / 3] + colb[t % 3] + SRS_B_HL);
            }
#pragma unroll
            for (int t = 0; t < 9; t++) {
                if (t + 2 < 9) {
                    ah[(t + 2) % 3] = *reinterpret_cast<const f16x8*>(ldsb + rowb[(t + 2) / 3] + colb[(t + 2) % 3]);
                    al[(t + 2) % 3] = *reinterpret_cast<const f16x8*>(ldsb + rowb[(t + 2) / 3] + colb[(t + 2) % 3] + SRS_B_HL);
                }
                if (!RIFE_ABL(TAG & SRS_NOMATH)) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[t], ah[t % 3], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[t], al[t % 3], acc, 0, 0, 0);
                } else acc[t] += (float)ah[t % 3][0] + (float)al[t % 3][1] + (float)Wf[t][2];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ck == 1) {
                f32x4* const sd = reinterpret_cast<f32x4*>(ldsb + SRS_LDS_STG + n * 4096 + ln * 16);
#pragma unroll
                for (int q = 0; q < 4; q++) sd[q * 64] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            }
        }
        // waves 4-7 are done with their matrix work: they finish their pixel of the next step while waves 0-3 run stem 1 (ring A's rows of step
        // k are dead since the barrier above; a strip change comes with a pre-step that everybody executes together below)
        SRS_STAMP(3)                                                     // stem 1
        const bool early = has_next && !fresh_next;
        if (wv >= 4 && early) finish(nxt.strip, 4 * nxt.q, fm_next, tp_next, 4 * SRS_AW);
        SRS_STAMP(4)                                                     // early gather finish (waves 4-7)
        SRS_SYNC();
        SRS_STAMP(5)                                                     // barrier
        if (wv < 2) finish_row(acc, cur.strip, cur.q, hf, l32, ln);
        if (has_next) {
            if (fresh_next) prestep(nxt);
            if (wv < 4 || fresh_next) finish(nxt.strip, 4 * nxt.q, fm_next, tp_next, 4 * SRS_AW);
        }
        SRS_STAMP(6)                                                     // combine, gather finish (waves 0-3), stores
        SRS_SYNC();
        SRS_STAMP(7)                                                     // barrier
        cur = nxt; nxt = nn; fm_next = fm_nn; has_next = has_nn; fresh_next = fresh_nn;
    }
    if (RIFE_ABL(TAG & SRS_CLK) && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) a.stamps[((size_t)blockIdx.x * 8 + wv) * 8 + i] = tph[i];
    }
#undef SRS_STAMP
}

}  // namespace rife
