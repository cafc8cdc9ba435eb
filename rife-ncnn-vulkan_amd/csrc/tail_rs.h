// tail_rs_kernel: the output side of rife-v4.6 in one row-streaming kernel - the head of the finest IFBlock (Deconvolution 4x4 stride 2 pad 1,
// 64 -> 24 channels, + PixelShuffle(2): reference models/rife-v4.6/flownet.param:200-201) and the tail of the graph (:202-217: F += flow3[0:4],
// M += flow3[4], Sigmoid, 2x rife.Warp (src/warp.cpp:96-168), blend) + the postproc (src/rife_postproc.comp:39-62), from the last S16 trunk tensor,
// F, M and the frames straight to the u8 frame.  Round 3; replaces head_h2_kernel<EPI_FINAL, true> (head_h2.h), a tile kernel that re-staged its
// weights per 8 x 32 tile and finished its eight pixels per lane one after the other - sixteen dependent memory round trips per lane.
//
// A workgroup (512 threads, two per CU) walks DOWN a strip of 32 quarter-resolution columns; one step = one quarter-resolution row of the trunk
// = a 4 x 128 block of output pixels, ONE PIXEL PER THREAD:
//   deconv   wave w: output parity w & 3 (a 2 x 2-tap convolution over the 3 x 3 neighbourhood, head_uses) x K half w >> 2 (two of the four
//            16-channel chunks): 16 MFMAs on 8 register-resident weight fragments; the pixel operands come from an LDS ring of four trunk
//            rows (conv_rs.h's row layout; one new row per step, loaded a step ahead).  Waves 4-7 hand their partial sums to waves 0-3
//            through LDS, which add the bias and scatter the PixelShuffle result as five planes (dx, dy, dz, dw, dm) x 4 rows x 128 columns.
//   tail     every thread: F, M of its pixel (loaded a step ahead) + the deltas -> the two warps' tap loads are ISSUED (warp_issue) and stay
//            in flight until the next step, where the pixel is finished (warp_finish, sigmoid, blend, quantise) and a wave's 64 pixels leave
//            as 48 dwords.
// Three LDS-only barriers per step.  Arithmetic per pixel is k_final's in its order; the deconvolution sums its four K chunks as two partial
// sums instead of one chain: last-bit differences of the flow deltas against head_h2_kernel (tests/test_gpu_tail_rs.py holds the two together).
// LDS: ring 4 x 8,704 B + partial sums 16 KB + delta planes 10 KB + bias = 61,568 B: two workgroups per CU.
#pragma once
#include "head_h2.h"

namespace rife {

constexpr int TRS_ROWB = 8 * 34 * 32;                 // one trunk row of the strip: [chunk 4][hi | lo][34 px][32 B] = 8,704
constexpr int TRS_SEG = 34 * 32;
constexpr int TRS_LDS_PART = 4 * TRS_ROWB;            // 34,816: partial sums of waves 4-7, [parity 4][quad 4][lane 64][16 B]
constexpr int TRS_LDS_DELTA = TRS_LDS_PART + 4 * 4096; // 51,200: [plane 5][row 4][column 128] fp32
constexpr int TRS_LDS_BIAS = TRS_LDS_DELTA + 5 * 4 * 128 * 4;    // 61,440: bias[32]
constexpr int TRS_LDS = TRS_LDS_BIAS + 128;               // 61,568
constexpr int TRS_NTHR = 512;

struct TailRsArgs {
    const unsigned char* in;     // S16 trunk tensor (64 channels), allocation start
    const void* w;               // head weights: f16 [chunk 4][pair 16][k half 2][32][8] (pack_weights_head_h2)
    const float* bias;           // [32] (24 real)
    const uint32_t *img0, *img1;
    const float4* F; const float* M;
    uint8_t* out;                // u8 HWC RGB, w x h
    int w_, h_, wp, hp;          // frame, padded frame
    int Hq, Wq;                  // trunk resolution (hp / 4, wp / 4)
    int pitch; unsigned plane;   // S16 geometry
    int nunits;                  // strips x Hq
};

// bench-only ablation bits of TAG (rife_hip_bench_tail_rs; results are garbage).  The product instantiates TAG = 0.
enum { TRS_NOTAPS = 1, TRS_NOFM = 2, TRS_NOMATH = 4, TRS_NOSTORE = 8, TRS_NOPIX = 16, TRS_NOROW = 32 };

// Global stores the compiler does not see.  vmcnt counts loads and stores together and they retire out of order with respect to each other, so
// with a store it knows to be in flight the compiler waits vmcnt(0) before the next use of ANY loaded value - in a loop that keeps loads in flight
// across iterations that drains the prefetch every step (measured: 165 -> ... us).  An untracked store only makes the compiler's counted waits
// conservative (the counter also holds the store: reaching "at most N outstanding" then needs one more completion, and loads still complete in
// order), and nobody in this kernel reads what it stored.
__device__ __forceinline__ void trs_store_dword(uint8_t* base, unsigned off, uint32_t v) {
    asm volatile("global_store_dword %0, %1, %2" :: "v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void trs_store_byte(uint8_t* base, unsigned off, uint32_t v) {
    asm volatile("global_store_byte %0, %1, %2" :: "v"(off), "v"(v), "s"(base) : "memory");
}

#define TRS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int TAG>
__global__ __launch_bounds__(TRS_NTHR) __attribute__((amdgpu_waves_per_eu(4, 4))) void tail_rs_kernel(TailRsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;

    const int nwg = gridDim.x, wg = blockIdx.x;
    const int u0 = (int)((long long)a.nunits * wg / nwg), u1 = (int)((long long)a.nunits * (wg + 1) / nwg);
    const int S = u1 - u0;
    if (S <= 0) return;

    if (tid < 32) reinterpret_cast<float*>(ldsb + TRS_LDS_BIAS)[tid] = a.bias[tid];      // (a global load in the combine phase would wait for every prefetch in flight)

    // this wave's matrix role: parity par, K half kh (chunks 2 kh, 2 kh + 1); its four taps and eight weight fragments
    const int par = wv & 3, kh = wv >> 2;
    f16x8 Wf[2][4];
    int tapdy[4], tapdx[4];
    {
        int n = 0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
            // head_uses(t, par) with a runtime parity: offsets 0 and (p ? +1 : -1) in each axis
            const int dy = t / 3 - 1, dx = t % 3 - 1, py = par >> 1, px = par & 1;
            const bool use = (dy == 0 || dy == (py ? 1 : -1)) && (dx == 0 || dx == (px ? 1 : -1));
            if (use) {
                // position of (t, par) in the packed pair order: pairs of earlier taps + earlier parities of this tap
                int idx = 0;
                for (int tt = 0; tt < 9; tt++)
                    for (int pp = 0; pp < 4; pp++) {
                        if (tt > t || (tt == t && pp >= par)) continue;
                        const int ddy = tt / 3 - 1, ddx = tt % 3 - 1, ppy = pp >> 1, ppx = pp & 1;
                        if ((ddy == 0 || ddy == (ppy ? 1 : -1)) && (ddx == 0 || ddx == (ppx ? 1 : -1))) idx++;
                    }
                // n is wave-uniform but not a compile-time constant per t: select by comparison below
#pragma unroll
                for (int s = 0; s < 4; s++)
                    if (s == n) {
                        tapdy[s] = dy + 1; tapdx[s] = dx + 1;
#pragma unroll
                        for (int c = 0; c < 2; c++)
                            Wf[c][s] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const unsigned char*>(a.w) + ((size_t)(((2 * kh + c) * 16 + idx) * 2 + half) * 32 + li) * 16);
                    }
                n++;
            }
        }
    }

    // row loader role: 16-byte unit tid (and 512 + tid for tid < 32) of the 544 of a ring row: (segment, pixel, half)
    unsigned lsrc[2], ldst[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int u = min(tid + 512 * k, 543);
        const int seg = u / 68, within = u - seg * 68;
        const int px = within >> 1, pos = within & 1;
        lsrc[k] = (unsigned)seg * a.plane + (unsigned)(px * 32 + ((pos ^ ((px >> 3) & 1)) << 4));
        ldst[k] = (unsigned)(seg * TRS_SEG + px * 32 + pos * 16);
    }
    auto row_load = [&](int strip, int y, f32x4 (&r)[2]) {              // trunk row y (-1 .. Hq: the zero border rows exist) of the strip
        const unsigned base = (unsigned)((y + 1) * a.pitch + 32 * strip) * 32u;
        if (RIFE_ABL(TAG & TRS_NOROW)) { r[0] = r[1] = f32x4{0.f, 0.f, 0.f, 0.f}; return; }
        r[0] = *reinterpret_cast<const f32x4*>(a.in + base + lsrc[0]);
        if (tid < 32) r[1] = *reinterpret_cast<const f32x4*>(a.in + base + lsrc[1]);
    };
    auto row_store = [&](int y, const f32x4 (&r)[2]) {
        unsigned char* const d = ldsb + ((y + 1) & 3) * TRS_ROWB;
        *reinterpret_cast<f32x4*>(d + ldst[0]) = r[0];
        if (tid < 32) *reinterpret_cast<f32x4*>(d + ldst[1]) = r[1];
    };

    // pixel role: row prow (0..3) and column pcol (0..127) of the step's output block; a wave = 64 consecutive pixels of one row
    const int prow = tid >> 7, pcol = tid & 127;
    struct PixIn { float4 f; float m; };
    auto fm_load = [&](int strip, int q) -> PixIn {
        const int fy = min(4 * q + prow, a.hp - 1), fx = min(128 * strip + pcol, a.wp - 1);
        const size_t i = (size_t)fy * a.wp + fx;
        PixIn r;
        if (RIFE_ABL(TAG & TRS_NOFM)) { r.f = make_float4(0.25f * (float)(fx & 7), -0.5f, 1.5f, 0.75f); r.m = 0.1f; return r; }
        r.f = a.F[i]; r.m = a.M[i];
        return r;
    };

    int strip = u0 / a.Hq, q = u0 - strip * a.Hq;
    {   // first step of the range: rows q - 1, q, q + 1
        f32x4 r[2];
#pragma unroll
        for (int dy = -1; dy <= 1; dy++) { row_load(strip, q + dy, r); row_store(q + dy, r); }
    }
    PixIn fm_cur = fm_load(strip, q);
    WarpLoads wa, wb; float mm_prev = 0.f; int pstrip = 0, pq = 0; bool have_prev = false;
    wa.r0 = wa.r1 = wb.r0 = wb.r1 = make_uint2(0u, 0u); wa.alpha = wa.beta = wb.alpha = wb.beta = 0.f; wa.l0 = wa.l1 = wb.l0 = wb.l1 = false;
    TRS_SYNC();

    for (int k = 0; k < S; k++) {
        int hf = half, l32 = li, ln = lane;
        asm volatile("" : "+v"(hf), "+v"(l32), "+v"(ln));               // opaque copies: nothing loop-invariant to hoist and spill (stem_rs.h)
        // the step after this one: its new trunk row and its F, M - loads that fly during the matrix phase
        int nstrip = strip, nq = q + 1;
        bool fresh = false;
        if (nq >= a.Hq) { nq = 0; nstrip++; fresh = true; }
        const bool has_next = k + 1 < S;
        f32x4 rnext[2];
        PixIn fm_next;
        if (has_next && !fresh) row_load(nstrip, nq + 1, rnext);
        if (has_next) fm_next = fm_load(nstrip, nq);

        // ---- deconvolution of trunk row q: this wave's parity and K half
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; c++) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int pxl = l32 + tapdx[s];
                const unsigned ad = (unsigned)(((q + tapdy[s]) & 3) * TRS_ROWB + (2 * (2 * kh + c)) * TRS_SEG + pxl * 32 + ((hf ^ ((pxl >> 3) & 1)) << 4));
                const f16x8 ah = *reinterpret_cast<const f16x8*>(ldsb + ad);
                const f16x8 al = *reinterpret_cast<const f16x8*>(ldsb + ad + TRS_SEG);
                if (RIFE_ABL(TAG & TRS_NOMATH)) { acc[s] += (float)ah[0] + (float)al[1] + (float)Wf[c][s][2]; continue; }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[c][s], ah, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[c][s], al, acc, 0, 0, 0);
            }
        }
        if (kh == 1) {
            f32x4* const sd = reinterpret_cast<f32x4*>(ldsb + TRS_LDS_PART + par * 4096 + ln * 16);
#pragma unroll
            for (int qd = 0; qd < 4; qd++) sd[qd * 64] = f32x4{acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
        }
        TRS_SYNC();
        if (kh == 0) {
            // + the other K half + bias, PixelShuffle: deconv channel 8 qd + 4 half + kk = flow channel 2 qd + half at sub-position kk of the
            // trunk pixel's 2 x 2 block of this parity
            const f32x4* const sd = reinterpret_cast<const f32x4*>(ldsb + TRS_LDS_PART + par * 4096 + ln * 16);
            float* const dl = reinterpret_cast<float*>(ldsb + TRS_LDS_DELTA);
            const int py = par >> 1, px = par & 1;
#pragma unroll
            for (int qd = 0; qd < 3; qd++) {
                const f32x4 p = sd[qd * 64];
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(ldsb + TRS_LDS_BIAS + (8 * qd + 4 * hf) * 4);
                const int ch = 2 * qd + hf;
                if (ch < 5) {
#pragma unroll
                    for (int kk = 0; kk < 4; kk++)
                        dl[(ch * 4 + 2 * py + (kk >> 1)) * 128 + 4 * l32 + 2 * px + (kk & 1)] = (acc[4 * qd + kk] + p[kk]) + b4[kk];
                }
            }
        }
        TRS_SYNC();

        // ---- this thread's pixel of step k: F, M + deltas, tap loads issued
        const int fy = 4 * q + prow, fx = 128 * strip + pcol;
        WarpLoads na, nb; float mm_now;
        {
            const float* const dl = reinterpret_cast<const float*>(ldsb + TRS_LDS_DELTA) + prow * 128 + pcol;
            float4 f = fm_cur.f;
            f.x = f.x + dl[0 * 512]; f.y = f.y + dl[1 * 512]; f.z = f.z + dl[2 * 512]; f.w = f.w + dl[3 * 512];
            mm_now = fm_cur.m + dl[4 * 512];
            const int cy = min(fy, a.hp - 1), cx = min(fx, a.wp - 1);    // out-of-frame lanes compute on clamped coordinates and store nothing
            if (RIFE_ABL(TAG & TRS_NOTAPS)) {
                na.r0 = na.r1 = nb.r0 = nb.r1 = make_uint2((unsigned)cx * 0x10101u, (unsigned)cy * 0x10101u);
                na.alpha = nb.alpha = f.x - floorf(f.x); na.beta = nb.beta = f.y - floorf(f.y); na.l0 = nb.l0 = true; na.l1 = nb.l1 = false;
            } else {
            nb = warp_issue(a.img1, cx, cy, f.z, f.w, a.wp, a.hp);
            na = warp_issue(a.img0, cx, cy, f.x, f.y, a.wp, a.hp);
            }
        }
        // ---- the pixel of step k - 1: its taps have been in flight for a whole step
        auto finish_pixel = [&](const WarpLoads& ta, const WarpLoads& tb, float mm, int ostrip, int oq) {
            const int oy = 4 * oq + prow, oxb = 128 * ostrip + 64 * ((tid >> 6) & 1), ox = oxb + ln;
            if (oy >= a.h_ || oxb >= a.w_) return;                       // wave-uniform
            if (RIFE_ABL(TAG & TRS_NOPIX)) { if (mm == 123.456f && ta.alpha == 7.f && tb.r0.x == 99u) a.out[0] = 1; return; }
            const bool valid = ox < a.w_ && (!RIFE_ABL(TAG & TRS_NOSTORE) || mm == 123.456f);
            const float m = 1.f / (1.f + expf(-mm));
            const float rm = 1.0f - m;
            const float3 w1 = warp_finish(tb);
            const float3 w0 = warp_finish(ta);
            const float r = w0.x * m + w1.x * rm, g = w0.y * m + w1.y * rm, b = w0.z * m + w1.z * rm;
            const uint32_t pk = (uint32_t)min(max((int)(r * 255.f + 0.5f), 0), 255) | ((uint32_t)min(max((int)(g * 255.f + 0.5f), 0), 255) << 8) |
                                ((uint32_t)min(max((int)(b * 255.f + 0.5f), 0), 255) << 16);
            const unsigned orow = (unsigned)(oy * a.w_ + oxb) * 3u;     // byte offset of the segment (frames stay below 4 GB)
            if ((a.w_ & 3) == 0 && oxb + 64 <= a.w_) {
                // 64 pixels = 192 bytes = 48 dwords: dword d takes bytes from pixels 4d/3 and 4d/3 + 1
                const int d = ln < 48 ? ln : 0;
                const int pa = (4 * d) / 3, sh = 8 * (4 * d - 3 * pa);
                const uint32_t va = (uint32_t)__shfl((int)pk, pa), vb = (uint32_t)__shfl((int)pk, pa + 1);
                const uint32_t word = sh == 0 ? (va | (vb << 24)) : ((va >> sh) | (vb << (24 - sh)));
                if (ln < 48 && (!RIFE_ABL(TAG & TRS_NOSTORE) || mm == 123.456f)) trs_store_dword(a.out, orow + 4u * (unsigned)ln, word);
            } else if (valid) {
                const unsigned o = orow + 3u * (unsigned)ln;
                trs_store_byte(a.out, o, pk & 255u); trs_store_byte(a.out, o + 1, (pk >> 8) & 255u); trs_store_byte(a.out, o + 2, pk >> 16);
            }
        };
        if (have_prev) finish_pixel(wa, wb, mm_prev, pstrip, pq);
        wa = na; wb = nb; mm_prev = mm_now; pstrip = strip; pq = q; have_prev = true;

        // ---- ring for the next step
        if (has_next) {
            if (fresh) {
                TRS_SYNC();                                              // everybody is done with this strip's rows
                f32x4 r[2];
#pragma unroll
                for (int dy = -1; dy <= 1; dy++) { row_load(nstrip, nq + dy, r); row_store(nq + dy, r); }
            } else row_store(nq + 1, rnext);
            fm_cur = fm_next;
        }
        TRS_SYNC();
        strip = nstrip; q = nq;
    }
    // the last pixel
    {
        const int ln = lane;
        const int oy = 4 * pq + prow, oxb = 128 * pstrip + 64 * ((tid >> 6) & 1), ox = oxb + ln;
        if (!(oy >= a.h_ || oxb >= a.w_)) {
            const bool valid = ox < a.w_;
            const float m = 1.f / (1.f + expf(-mm_prev));
            const float rm = 1.0f - m;
            const float3 w1 = warp_finish(wb);
            const float3 w0 = warp_finish(wa);
            const float r = w0.x * m + w1.x * rm, g = w0.y * m + w1.y * rm, b = w0.z * m + w1.z * rm;
            const uint32_t pk = (uint32_t)min(max((int)(r * 255.f + 0.5f), 0), 255) | ((uint32_t)min(max((int)(g * 255.f + 0.5f), 0), 255) << 8) |
                                ((uint32_t)min(max((int)(b * 255.f + 0.5f), 0), 255) << 16);
            const unsigned orow = (unsigned)(oy * a.w_ + oxb) * 3u;
            if ((a.w_ & 3) == 0 && oxb + 64 <= a.w_) {
                const int d = ln < 48 ? ln : 0;
                const int pa = (4 * d) / 3, sh = 8 * (4 * d - 3 * pa);
                const uint32_t va = (uint32_t)__shfl((int)pk, pa), vb = (uint32_t)__shfl((int)pk, pa + 1);
                const uint32_t word = sh == 0 ? (va | (vb << 24)) : ((va >> sh) | (vb << (24 - sh)));
                if (ln < 48) trs_store_dword(a.out, orow + 4u * (unsigned)ln, word);
            } else if (valid) {
                const unsigned o = orow + 3u * (unsigned)ln;
                trs_store_byte(a.out, o, pk & 255u); trs_store_byte(a.out, o + 1, (pk >> 8) & 255u); trs_store_byte(a.out, o + 2, pk >> 16);
            }
        }
    }
}

}  // namespace rife
