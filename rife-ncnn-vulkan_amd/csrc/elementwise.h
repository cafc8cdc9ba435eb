// Bandwidth-bound kernels of the RIFE v4 schedule: pre/post-processing, the backward bilinear warp,
// bilinear resampling, flow/mask update and the final blend.  Each restates the arithmetic of one
// reference shader / CPU loop literally (same operation order, two roundings per a*b+c: this TU is
// compiled with -ffp-contract=off), so that given identical inputs the outputs are bit-identical to the
// CPU path they are tested against:
//   rife_preproc.comp:33-66 / rife.cpp:4152-4211      -> k_preproc (stores the u8 image, /255 applied at use)
//   warp.comp:24-69 / warp.cpp:96-168                  -> warp_sample(), k_warp_chw
//   ncnn Interp bilinear (flownet.param:10,47,52,...)  -> fused into k_assemble* / k_flow_update
//   flownet.param:49-62,99-115,152-165,202-217         -> k_assemble<S>, k_flow_update<S>, k_final
//   rife_postproc.comp:33-63 / rife.cpp:4373-4397      -> k_final
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rife {

// ---- images are kept as padded RGBX u8 (4 B / pixel): x * (1/255.f) is recomputed at every use, which is
// bit-identical to storing the fp32 plane (rife.cpp:4167: `*outptr++ = *ptr++ * (1 / 255.f)`) and makes the
// 4-tap warp gather a single dword per tap for all three channels. ----
__global__ void k_preproc(const uint8_t* __restrict__ rgb, int w, int h, uint32_t* __restrict__ out, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wp) return;
    uint32_t v = 0;
    if (x < w && y < h) {
        const uint8_t* p = rgb + ((size_t)y * w + x) * 3;
        v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
    out[(size_t)y * wp + x] = v;
}

// the same for four pixels per lane (w % 4 == 0, 4-byte aligned frame): 12 contiguous bytes in, one 16-byte store out
__global__ void k_preproc4(const uint8_t* __restrict__ rgb, int w, int h, uint32_t* __restrict__ out, int wp, int hp) {
    const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x4 >= wp) return;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (x4 < w && y < h) {
        typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
        u32x3 q;
        __builtin_memcpy(&q, __builtin_assume_aligned(rgb + ((size_t)y * w + x4) * 3, 4), 12);
        v.x = q.x & 0xffffffu; v.y = (q.x >> 24) | ((q.y & 0xffffu) << 8); v.z = (q.y >> 16) | ((q.z & 0xffu) << 16); v.w = q.z >> 8;
    }
    *reinterpret_cast<uint4*>(out + (size_t)y * wp + x4) = v;
}

__device__ __forceinline__ float3 unpack_rgb(uint32_t v) {
    const float k = 1 / 255.f;
    return make_float3((float)(v & 0xff) * k, (float)((v >> 8) & 0xff) * k, (float)((v >> 16) & 0xff) * k);
}

// rife.Warp for a 3-channel RGBX image: out = bilerp(img, x + fx, y + fy) with the reference's
// clamp-then-alpha rule (warp.cpp:126-146).
struct WarpTaps { int i00, i01, i10, i11; float alpha, beta; };

__device__ __forceinline__ WarpTaps warp_taps(int x, int y, float flow_x, float flow_y, int w, int h) {
    const float sample_x = (float)x + flow_x;
    const float sample_y = (float)y + flow_y;
    int x0 = (int)floorf(sample_x);
    int y0 = (int)floorf(sample_y);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), w - 1);
    y0 = min(max(y0, 0), h - 1);
    x1 = min(max(x1, 0), w - 1);
    y1 = min(max(y1, 0), h - 1);
    WarpTaps t;
    t.alpha = sample_x - (float)x0;
    t.beta = sample_y - (float)y0;
    t.i00 = y0 * w + x0; t.i01 = y0 * w + x1; t.i10 = y1 * w + x0; t.i11 = y1 * w + x1;
    return t;
}

__device__ __forceinline__ float warp_lerp(float v0, float v1, float v2, float v3, float alpha, float beta) {
    const float v4 = v0 * (1 - alpha) + v1 * alpha;
    const float v5 = v2 * (1 - alpha) + v3 * alpha;
    return v4 * (1 - beta) + v5 * beta;
}

// The two taps of a row are neighbours - or the same pixel where the sample leaves the frame - so each row is ONE 8-byte load at column
// xb = min(x0, w - 2) and a select (w >= 2: padded widths are multiples of 32): half the gather instructions of four dword loads, same values.
// In two halves, so that a kernel can put other work between the loads and their use (stem_rs.h): warp_issue() computes the taps and starts
// the two loads, warp_finish() does the arithmetic.  warp_rgbx() is the two back to back.
struct WarpLoads { uint2 r0, r1; float alpha, beta; bool l0, l1; };
__device__ __forceinline__ WarpLoads warp_issue(const uint32_t* __restrict__ img, int x, int y, float fx, float fy, int w, int h) {
    const float sample_x = (float)x + fx;
    const float sample_y = (float)y + fy;
    int x0 = (int)floorf(sample_x);
    int y0 = (int)floorf(sample_y);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), w - 1);
    y0 = min(max(y0, 0), h - 1);
    x1 = min(max(x1, 0), w - 1);
    y1 = min(max(y1, 0), h - 1);
    WarpLoads t;
    t.alpha = sample_x - (float)x0;
    t.beta = sample_y - (float)y0;
    const int xb = min(x0, w - 2);
    // (scalar base + 32-bit byte offset per lane instead of these 64-bit lane addresses: 35 fewer vector instructions per pixel pair, measured
    // 0.5 - 1 % SLOWER in the tile stems and the fused tail - same call, two library builds; not used)
    __builtin_memcpy(&t.r0, __builtin_assume_aligned(img + (y0 * w + xb), 4), 8);
    __builtin_memcpy(&t.r1, __builtin_assume_aligned(img + (y1 * w + xb), 4), 8);
    t.l0 = x0 == xb; t.l1 = x1 == xb;
    return t;
}
__device__ __forceinline__ float3 warp_finish(const WarpLoads& t) {
    const float3 a = unpack_rgb(t.l0 ? t.r0.x : t.r0.y), b = unpack_rgb(t.l1 ? t.r0.x : t.r0.y), c = unpack_rgb(t.l0 ? t.r1.x : t.r1.y), d = unpack_rgb(t.l1 ? t.r1.x : t.r1.y);
    return make_float3(warp_lerp(a.x, b.x, c.x, d.x, t.alpha, t.beta), warp_lerp(a.y, b.y, c.y, d.y, t.alpha, t.beta), warp_lerp(a.z, b.z, c.z, d.z, t.alpha, t.beta));
}
__device__ __forceinline__ float3 warp_rgbx(const uint32_t* __restrict__ img, int x, int y, float fx, float fy, int w, int h) {
    return warp_finish(warp_issue(img, x, y, fx, fy, w, h));
}

// generic rife.Warp on planar CHW fp32 (per-kernel parity test entry point; v2.3 context features)
__global__ void k_warp_chw(const float* __restrict__ image, const float* __restrict__ flow, float* __restrict__ out, int c, int h, int w) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const size_t plane = (size_t)w * h;
    const WarpTaps t = warp_taps(x, y, flow[(size_t)y * w + x], flow[plane + (size_t)y * w + x], w, h);
    for (int q = 0; q < c; q++) {
        const float* im = image + q * plane;
        out[q * plane + (size_t)y * w + x] = warp_lerp(im[t.i00], im[t.i01], im[t.i10], im[t.i11], t.alpha, t.beta);
    }
}

// ncnn bilinear downscale by 2^k (align_corner=0): source centre falls exactly between two pixels, so both
// weights are 0.5: rows = S[sx]*a0 + S[sx+1]*a1 (horizontal), D = rows0*b0 + rows1*b1 (vertical).
__device__ __forceinline__ float down4(float v00, float v01, float v10, float v11) {
    const float r0 = v00 * 0.5f + v01 * 0.5f;
    const float r1 = v10 * 0.5f + v11 * 0.5f;
    return r0 * 0.5f + r1 * 0.5f;
}

// Block-0 input: x = Interp(1/8)(Concat(in0, in1, in2)) -> NHWC8 {in0.rgb, in1.rgb, t, 0}   (flownet.param:9-10)
// `tsp` != null: the timestep is read from device memory (hipGraph replays need launch parameters that never change)
__global__ void k_assemble0(const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, float timestep_arg, const float* __restrict__ tsp,
                            float* __restrict__ X, int wp, int hp) {
    const float timestep = tsp ? *tsp : timestep_arg;
    const int Wb = wp / 8, Hb = hp / 8;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wb || y >= Hb) return;
    const int sx = 8 * x + 3, sy = 8 * y + 3;
    const size_t i00 = (size_t)sy * wp + sx, i10 = i00 + wp;
    const float3 a0 = unpack_rgb(img0[i00]), a1 = unpack_rgb(img0[i00 + 1]), a2 = unpack_rgb(img0[i10]), a3 = unpack_rgb(img0[i10 + 1]);
    const float3 b0 = unpack_rgb(img1[i00]), b1 = unpack_rgb(img1[i00 + 1]), b2 = unpack_rgb(img1[i10]), b3 = unpack_rgb(img1[i10 + 1]);
    float4 o0, o1;
    o0.x = down4(a0.x, a1.x, a2.x, a3.x); o0.y = down4(a0.y, a1.y, a2.y, a3.y); o0.z = down4(a0.z, a1.z, a2.z, a3.z);
    o0.w = down4(b0.x, b1.x, b2.x, b3.x); o1.x = down4(b0.y, b1.y, b2.y, b3.y); o1.y = down4(b0.z, b1.z, b2.z, b3.z);
    o1.z = down4(timestep, timestep, timestep, timestep);
    o1.w = 0.f;
    float4* dst = reinterpret_cast<float4*>(X + ((size_t)y * Wb + x) * 8);
    dst[0] = o0; dst[1] = o1;
}

// ncnn linear_coeffs for an upscale by S (power of two): fx = (dx + 0.5) / S - 0.5 is exact in fp32 here
// (the reference computes it in double and rounds to float; all intermediates are dyadic and short).
__device__ __forceinline__ void up_coeff(int d, int S, int in, int& s0, float& a0, float& a1) {
    float f = ((float)d + 0.5f) * (1.0f / (float)S) - 0.5f;
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= in - 1) { s = in - 2; f = 1.f; }
    s0 = s; a0 = 1.f - f; a1 = f;
}

// One full-resolution pixel of the flow update after a block (flownet.param:47-58, 99-105, 152-158): u = Interp(S)(flow_b) at (x, y);
// F = F * 1 + u[0:4] * S (Eltwise with coefficients), M = M + u[4].  flow_b is kept as [hp / S][wp / S][8] fp32 {x, y, z, w, mask, unused, 0, 0}.
// Shared by k_flow_update<S> and by the fused stems that apply the update of the block before them while they gather (stem_fused.h).
template <int S>
__device__ __forceinline__ void flow_upsampled(const float* __restrict__ flow, int wp, int hp, int x, int y, float4& u, float& um) {
    const int Wb = wp / S, Hb = hp / S;
    int sx, sy; float a0, a1, b0, b1;
    up_coeff(x, S, Wb, sx, a0, a1);
    up_coeff(y, S, Hb, sy, b0, b1);
    const float* p00 = flow + ((size_t)sy * Wb + sx) * 8;
    const float* p10 = p00 + (size_t)Wb * 8;
    const float4 q00 = *reinterpret_cast<const float4*>(p00), q01 = *reinterpret_cast<const float4*>(p00 + 8);
    const float4 q10 = *reinterpret_cast<const float4*>(p10), q11 = *reinterpret_cast<const float4*>(p10 + 8);
    const float m00 = p00[4], m01 = p00[12], m10 = p10[4], m11 = p10[12];
#define RIFE_UP(c00, c01, c10, c11) (((c00) * a0 + (c01) * a1) * b0 + ((c10) * a0 + (c11) * a1) * b1)
    u.x = RIFE_UP(q00.x, q01.x, q10.x, q11.x);
    u.y = RIFE_UP(q00.y, q01.y, q10.y, q11.y);
    u.z = RIFE_UP(q00.z, q01.z, q10.z, q11.z);
    u.w = RIFE_UP(q00.w, q01.w, q10.w, q11.w);
    um = RIFE_UP(m00, m01, m10, m11);
#undef RIFE_UP
}
// the FIRST update (after block 0): F = u[0:4] * S, M = u[4] (k_flow_update<S, true>)
template <int S>
__device__ __forceinline__ void flow_first(const float* __restrict__ flow, int wp, int hp, int x, int y, float4& f, float& m) {
    float4 u; float um;
    flow_upsampled<S>(flow, wp, hp, x, y, u, um);
    const float s = (float)S;
    f = make_float4(u.x * s, u.y * s, u.z * s, u.w * s);
    m = um;
}
template <int S>
__device__ __forceinline__ void flow_accumulate(const float4 u, const float um, float4& f, float& m) {
    const float s = (float)S;
    f = make_float4(f.x * 1.0f + u.x * s, f.y * 1.0f + u.y * s, f.z * 1.0f + u.z * s, f.w * 1.0f + u.w * s);
    m = m + um;
}

// A flow update that the consumer of F, M applies itself: F, M are read from the old tensors, updated with Interp(US)(flow) and written to
// Fw, Mw (different buffers: neighbouring tiles re-read the old values of their shared halo pixels).  flow == nullptr: F, M are current.
struct FlowPending {
    const float* flow = nullptr;
    float4* Fw = nullptr;
    float* Mw = nullptr;
};

// Blocks 1..3 input (flownet.param:52-62, 107-115, 160-165):
//   x = Concat(Interp(1/S)(Concat(warp(in0,F.xy), warp(in1,F.zw), in2, M)), Interp(1/S)(F)/S)  -> NHWC16 (12 + 4 zero)
// one pixel (bx, by) of the block input at 1/S resolution: 12 channels {warp(in0,F.xy) rgb, warp(in1,F.zw) rgb, t, M, F/S xyzw}
// UPD = 1: the flow update of the previous block (scale 2 S) is applied on the way: every full-resolution pixel read here is updated and written
// to pend.Fw / pend.Mw (same arithmetic as k_flow_update<2 S>: the 12 channels are those of the unfused sequence bit for bit).
// UPD = 2 (round 5; block 1): F, M do not exist yet - they are what k_flow_update<2 S, FIRST> would have written from the first flow, computed here for the pixels
// that are sampled (F = Interp(2 S)(flow0) * 2 S, M = its mask channel: flow0 is [hp / 8][wp / 8][8] fp32, L2-resident) and written nowhere; k_flow_update2
// produces the full-resolution tensors after this block in one pass together with this block's own update.
template <int S, int UPD = 0>
__device__ __forceinline__ void assemble_pixel(const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, float timestep,
                                               const float4* __restrict__ F, const float* __restrict__ M, int wp, int hp, int x, int y, float o[12],
                                               const FlowPending& pend = FlowPending{}) {
    if (S == 1) {
        const size_t i = (size_t)y * wp + x;
        float4 f; float mk;
        if (UPD == 2) flow_first<2 * S>(pend.flow, wp, hp, x, y, f, mk);
        else { f = F[i]; mk = M[i]; }
        if (UPD == 1) {
            float4 u; float um;
            flow_upsampled<2 * S>(pend.flow, wp, hp, x, y, u, um);
            flow_accumulate<2 * S>(u, um, f, mk);
            pend.Fw[i] = f; pend.Mw[i] = mk;
        }
        const float3 w0 = warp_rgbx(img0, x, y, f.x, f.y, wp, hp);
        const float3 w1 = warp_rgbx(img1, x, y, f.z, f.w, wp, hp);
        o[0] = w0.x; o[1] = w0.y; o[2] = w0.z; o[3] = w1.x; o[4] = w1.y; o[5] = w1.z; o[6] = timestep; o[7] = mk;
        o[8] = f.x; o[9] = f.y; o[10] = f.z; o[11] = f.w;
    } else {
        const int sx = S * x + S / 2 - 1, sy = S * y + S / 2 - 1;
        float v[4][12];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int px = sx + (k & 1), py = sy + (k >> 1);
            const size_t i = (size_t)py * wp + px;
            float4 f; float mk;
            if (UPD == 2) flow_first<2 * S>(pend.flow, wp, hp, px, py, f, mk);
            else { f = F[i]; mk = M[i]; }
            if (UPD == 1) {
                float4 u; float um;
                flow_upsampled<2 * S>(pend.flow, wp, hp, px, py, u, um);
                flow_accumulate<2 * S>(u, um, f, mk);
                pend.Fw[i] = f; pend.Mw[i] = mk;
            }
            const float3 w0 = warp_rgbx(img0, px, py, f.x, f.y, wp, hp);
            const float3 w1 = warp_rgbx(img1, px, py, f.z, f.w, wp, hp);
            v[k][0] = w0.x; v[k][1] = w0.y; v[k][2] = w0.z; v[k][3] = w1.x; v[k][4] = w1.y; v[k][5] = w1.z;
            v[k][6] = timestep; v[k][7] = mk;
            v[k][8] = f.x; v[k][9] = f.y; v[k][10] = f.z; v[k][11] = f.w;
        }
#pragma unroll
        for (int c = 0; c < 12; c++) o[c] = down4(v[0][c], v[1][c], v[2][c], v[3][c]);
#pragma unroll
        for (int c = 8; c < 12; c++) o[c] = o[c] / (float)S;        // BinaryOp div by scalar (flownet.param:53,108)
    }
}

template <int S>
__global__ void k_assemble(const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, float timestep_arg, const float* __restrict__ tsp,
                           const float4* __restrict__ F, const float* __restrict__ M, float* __restrict__ X, int wp, int hp) {
    const float timestep = tsp ? *tsp : timestep_arg;
    const int Wb = wp / S, Hb = hp / S;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wb || y >= Hb) return;
    float o[12];
    assemble_pixel<S>(img0, img1, timestep, F, M, wp, hp, x, y, o);
    float4* dst = reinterpret_cast<float4*>(X + ((size_t)y * Wb + x) * 16);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    dst[2] = make_float4(o[8], o[9], o[10], o[11]);
    dst[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// flow_b is kept as [Hb][Wb][8] fp32 {x, y, z, w, mask, unused, 0, 0}.
// After block b < 3 (flownet.param:47-58, 99-105, 152-158):
//   u = Interp(S)(flow_b);  b == 0: F = u[0:4] * S, M = u[4];   b > 0: F = F*1 + u[0:4]*S (Eltwise), M = M + u[4]
template <int S, bool FIRST>
__global__ void k_flow_update(const float* __restrict__ flow, float4* __restrict__ F, float* __restrict__ M, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wp) return;
    float4 u; float um;
    flow_upsampled<S>(flow, wp, hp, x, y, u, um);
    const size_t i = (size_t)y * wp + x;
    const float s = (float)S;
    if (FIRST) {
        F[i] = make_float4(u.x * s, u.y * s, u.z * s, u.w * s);
        M[i] = um;
    } else {
        float4 f = F[i];
        float m = M[i];
        flow_accumulate<S>(u, um, f, m);
        F[i] = f;
        M[i] = m;
    }
}

// The updates after blocks 0 and 1 in ONE pass over the full-resolution tensors (round 5): F, M = k_flow_update<S1, false> applied to what
// k_flow_update<S0, true> would have written - the same two expressions in the same order, so the tensors are the two-kernel sequence's bit for bit, without the
// 20 B / pixel written after block 0 and read again after block 1 (4K: 200 + 178 MB per pair and one launch).  Block 1's stem samples the first update itself
// (assemble_pixel UPD = 2).
template <int S0, int S1>
__global__ void k_flow_update2(const float* __restrict__ flow0, const float* __restrict__ flow1, float4* __restrict__ F, float* __restrict__ M, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wp) return;
    float4 f; float m;
    flow_first<S0>(flow0, wp, hp, x, y, f, m);
    float4 u; float um;
    flow_upsampled<S1>(flow1, wp, hp, x, y, u, um);
    flow_accumulate<S1>(u, um, f, m);
    const size_t i = (size_t)y * wp + x;
    F[i] = f;
    M[i] = m;
}

// Tail of the graph + postproc (flownet.param:202-217, rife.cpp:4373-4397 / rife_postproc.comp:39-62):
//   F += flow3[0:4]; M += flow3[4]; m = sigmoid(M); out = warp(in0,F.xy)*m + warp(in1,F.zw)*(1-m);
//   u8 = clamp((int)(out*255 + 0.5), 0, 255), cropped to w x h, HWC RGB.
__global__ void k_final(const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, const float4* __restrict__ F,
                        const float* __restrict__ M, const float* __restrict__ flow3, uint8_t* __restrict__ out, int w, int h, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const size_t i = (size_t)y * wp + x;
    const float* fl = flow3 + i * 8;
    const float4 d = *reinterpret_cast<const float4*>(fl);
    float4 f = F[i];
    f.x = f.x + d.x; f.y = f.y + d.y; f.z = f.z + d.z; f.w = f.w + d.w;
    const float mm = M[i] + fl[4];
    const float m = 1.f / (1.f + expf(-mm));
    const float rm = 1.0f - m;
    const float3 w1 = warp_rgbx(img1, x, y, f.z, f.w, wp, hp);
    const float3 w0 = warp_rgbx(img0, x, y, f.x, f.y, wp, hp);
    const float r = w0.x * m + w1.x * rm, g = w0.y * m + w1.y * rm, b = w0.z * m + w1.z * rm;
    uint8_t* o = out + ((size_t)y * w + x) * 3;
    o[0] = (uint8_t)min(max((int)(r * 255.f + 0.5f), 0), 255);
    o[1] = (uint8_t)min(max((int)(g * 255.f + 0.5f), 0), 255);
    o[2] = (uint8_t)min(max((int)(b * 255.f + 0.5f), 0), 255);
}

// rife-v4 (4.0) tail (models/rife-v4/flownet.param:154-168): F and M are final after the block-3 flow update;
//   m = sigmoid(M); out = warp(in0,F.xy)*m + warp(in1,F.zw)*(1-m); then the same postproc as k_final.
__global__ void k_blend_final(const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, const float4* __restrict__ F,
                              const float* __restrict__ M, uint8_t* __restrict__ out, int w, int h, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const size_t i = (size_t)y * wp + x;
    const float4 f = F[i];
    const float m = 1.f / (1.f + expf(-M[i]));
    const float rm = 1.0f - m;
    const float3 w1 = warp_rgbx(img1, x, y, f.z, f.w, wp, hp);
    const float3 w0 = warp_rgbx(img0, x, y, f.x, f.y, wp, hp);
    const float r = w0.x * m + w1.x * rm, g = w0.y * m + w1.y * rm, b = w0.z * m + w1.z * rm;
    uint8_t* o = out + ((size_t)y * w + x) * 3;
    o[0] = (uint8_t)min(max((int)(r * 255.f + 0.5f), 0), 255);
    o[1] = (uint8_t)min(max((int)(g * 255.f + 0.5f), 0), 255);
    o[2] = (uint8_t)min(max((int)(b * 255.f + 0.5f), 0), 255);
}

// the same without postproc: out0 as float4 per padded pixel (TTA averaging)
__global__ void k_blend_final_float(const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, const float4* __restrict__ F,
                                    const float* __restrict__ M, float4* __restrict__ out, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wp) return;
    const size_t i = (size_t)y * wp + x;
    const float4 f = F[i];
    const float m = 1.f / (1.f + expf(-M[i]));
    const float rm = 1.0f - m;
    const float3 w1 = warp_rgbx(img1, x, y, f.z, f.w, wp, hp);
    const float3 w0 = warp_rgbx(img0, x, y, f.x, f.y, wp, hp);
    out[i] = make_float4(w0.x * m + w1.x * rm, w0.y * m + w1.y * rm, w0.z * m + w1.z * rm, 0.f);
}

// a += b (BinaryOp add closing a rife-v4 (4.0) IFBlock trunk)
__global__ void k_add_inplace(float4* __restrict__ a, const float4* __restrict__ b, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 x = a[i];
    const float4 y = b[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    a[i] = x;
}

// ------------------------------------------------------------------------------------------------------------
// TTA (-x spatial, -z temporal), rife-v4:  rife_preproc_tta.comp:40-93, rife_v4_flow_tta_avg.comp:25-129,
// rife_v4_flow_tta_temporal_avg.comp:19-59, rife_postproc_tta.comp:40-81, rife_out_tta_temporal_avg.comp:19-36
// and their CPU twins rife.cpp:3319-3413, 3477-3512, 3515-3821, 4056-4144.
// ------------------------------------------------------------------------------------------------------------

// index of base pixel (i = row, j = col) of a W x H plane inside orientation ti's buffer (ti >= 4: H wide, W tall)
__device__ __forceinline__ size_t tta_index(int ti, int i, int j, int W, int H) {
    switch (ti) {
        case 0: return (size_t)i * W + j;
        case 1: return (size_t)i * W + (W - 1 - j);
        case 2: return (size_t)(H - 1 - i) * W + (W - 1 - j);
        case 3: return (size_t)(H - 1 - i) * W + j;
        case 4: return (size_t)j * H + i;
        case 5: return (size_t)j * H + (H - 1 - i);
        case 6: return (size_t)(W - 1 - j) * H + (H - 1 - i);
        default: return (size_t)(W - 1 - j) * H + i;
    }
}

struct Ptr8 { void* p[8]; };
struct Ptr16 { const void* p[16]; };

// u8 HWC RGB (w x h) -> the 8 orientations of the zero-padded RGBX image
// (thread blocks of the orientation kernels are 2-D TILES, tta_block(): the transposed orientations 4..7 walk their buffers along the other axis,
// and a 256 x 1 block reads / writes them one element per cache line)
__global__ void k_preproc_tta(const uint8_t* __restrict__ rgb, int w, int h, Ptr8 outs, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= wp || y >= hp) return;
    uint32_t v = 0;
    if (x < w && y < h) {
        const uint8_t* p = rgb + ((size_t)y * w + x) * 3;
        v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    }
#pragma unroll
    for (int ti = 0; ti < 8; ti++) reinterpret_cast<uint32_t*>(outs.p[ti])[tta_index(ti, y, x, wp, hp)] = v;
}

// forward / reversed flow consensus, in place on both [H][W][8] tensors
__global__ void k_v4_temporal_merge(float* __restrict__ f, float* __restrict__ r, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float* a = f + i * 8; float* b = r + i * 8;
    const float x = (a[0] + b[2]) * 0.5f, y = (a[1] + b[3]) * 0.5f, z = (a[2] + b[0]) * 0.5f, w = (a[3] + b[1]) * 0.5f;
    const float m = (a[4] - b[4]) * 0.5f;
    a[0] = x; a[1] = y; a[2] = z; a[3] = w; a[4] = m;
    b[0] = z; b[1] = w; b[2] = x; b[3] = y; b[4] = -m;
}

// 8-orientation flow / mask consensus, in place on the eight [.][.][8] tensors; W x H = size of orientation 0
__global__ void k_v4_spatial_avg(Ptr8 fl, int W, int H) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (j >= W || i >= H) return;
    float* q[8];
#pragma unroll
    for (int ti = 0; ti < 8; ti++) q[ti] = reinterpret_cast<float*>(fl.p[ti]) + tta_index(ti, i, j, W, H) * 8;
    const float x = (q[0][0] + -q[1][0] + -q[2][0] + q[3][0] + q[4][1] + q[5][1] + -q[6][1] + -q[7][1]) * 0.125f;
    const float y = (q[0][1] + q[1][1] + -q[2][1] + -q[3][1] + q[4][0] + -q[5][0] + -q[6][0] + q[7][0]) * 0.125f;
    const float z = (q[0][2] + -q[1][2] + -q[2][2] + q[3][2] + q[4][3] + q[5][3] + -q[6][3] + -q[7][3]) * 0.125f;
    const float w = (q[0][3] + q[1][3] + -q[2][3] + -q[3][3] + q[4][2] + -q[5][2] + -q[6][2] + q[7][2]) * 0.125f;
    const float m = (q[0][4] + q[1][4] + q[2][4] + q[3][4] + q[4][4] + q[5][4] + q[6][4] + q[7][4]) * 0.125f;
    q[0][0] = x;  q[0][1] = y;  q[0][2] = z;  q[0][3] = w;
    q[1][0] = -x; q[1][1] = y;  q[1][2] = -z; q[1][3] = w;
    q[2][0] = -x; q[2][1] = -y; q[2][2] = -z; q[2][3] = -w;
    q[3][0] = x;  q[3][1] = -y; q[3][2] = z;  q[3][3] = -w;
    q[4][0] = y;  q[4][1] = x;  q[4][2] = w;  q[4][3] = z;
    q[5][0] = -y; q[5][1] = x;  q[5][2] = -w; q[5][3] = z;
    q[6][0] = -y; q[6][1] = -x; q[6][2] = -w; q[6][3] = -z;
    q[7][0] = y;  q[7][1] = -x; q[7][2] = w;  q[7][3] = -z;
#pragma unroll
    for (int ti = 0; ti < 8; ti++) q[ti][4] = m;
}

// Both consensus steps of a `-x -z` block in one pass over the sixteen flow tensors (reference src/rife.cpp:3477-3512 temporal, 3515-3821 spatial;
// shaders rife_v4_flow_tta_temporal_avg.comp, rife_v4_flow_tta_avg.comp): k_v4_temporal_merge on (forward, reversed) of every orientation, then
// k_v4_spatial_avg on the eight forward tensors; the reversed direction's spatial consensus is the forward one with (x, y) <-> (z, w) and -m - the
// same sums term by term (the temporal step leaves b = (a.z, a.w, a.x, a.y, -a.m)), so it is written, not recomputed.  Same operations in the
// same order as the two kernels: bit-identical tensors; every entry is read once and written once instead of twice.
struct Ptr8x2 { void* f[8]; void* r[8]; };
__global__ void k_v4_consensus(Ptr8x2 fl, int W, int H) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;
    if (j >= W || i >= H) return;
    float *qf[8], *qr[8];
    float ax[8], ay[8], az[8], aw[8], am[8];
#pragma unroll
    for (int ti = 0; ti < 8; ti++) {
        const size_t o = tta_index(ti, i, j, W, H) * 8;
        qf[ti] = reinterpret_cast<float*>(fl.f[ti]) + o; qr[ti] = reinterpret_cast<float*>(fl.r[ti]) + o;
        const float* a = qf[ti]; const float* b = qr[ti];
        ax[ti] = (a[0] + b[2]) * 0.5f; ay[ti] = (a[1] + b[3]) * 0.5f; az[ti] = (a[2] + b[0]) * 0.5f; aw[ti] = (a[3] + b[1]) * 0.5f;
        am[ti] = (a[4] - b[4]) * 0.5f;
    }
    const float x = (ax[0] + -ax[1] + -ax[2] + ax[3] + ay[4] + ay[5] + -ay[6] + -ay[7]) * 0.125f;
    const float y = (ay[0] + ay[1] + -ay[2] + -ay[3] + ax[4] + -ax[5] + -ax[6] + ax[7]) * 0.125f;
    const float z = (az[0] + -az[1] + -az[2] + az[3] + aw[4] + aw[5] + -aw[6] + -aw[7]) * 0.125f;
    const float w = (aw[0] + aw[1] + -aw[2] + -aw[3] + az[4] + -az[5] + -az[6] + az[7]) * 0.125f;
    const float m = (am[0] + am[1] + am[2] + am[3] + am[4] + am[5] + am[6] + am[7]) * 0.125f;
    // the reversed tensors' own consensus: sums over b = (az, aw, ax, ay, -am) in the same order
    const float xr = (az[0] + -az[1] + -az[2] + az[3] + aw[4] + aw[5] + -aw[6] + -aw[7]) * 0.125f;
    const float yr = (aw[0] + aw[1] + -aw[2] + -aw[3] + az[4] + -az[5] + -az[6] + az[7]) * 0.125f;
    const float zr = (ax[0] + -ax[1] + -ax[2] + ax[3] + ay[4] + ay[5] + -ay[6] + -ay[7]) * 0.125f;
    const float wr = (ay[0] + ay[1] + -ay[2] + -ay[3] + ax[4] + -ax[5] + -ax[6] + ax[7]) * 0.125f;
    const float mr = (-am[0] + -am[1] + -am[2] + -am[3] + -am[4] + -am[5] + -am[6] + -am[7]) * 0.125f;
#define RIFE_ORI(q, X, Y, Z, Wv)                                                                         \
    q[0][0] = X;  q[0][1] = Y;  q[0][2] = Z;  q[0][3] = Wv;                                              \
    q[1][0] = -X; q[1][1] = Y;  q[1][2] = -Z; q[1][3] = Wv;                                              \
    q[2][0] = -X; q[2][1] = -Y; q[2][2] = -Z; q[2][3] = -Wv;                                             \
    q[3][0] = X;  q[3][1] = -Y; q[3][2] = Z;  q[3][3] = -Wv;                                             \
    q[4][0] = Y;  q[4][1] = X;  q[4][2] = Wv; q[4][3] = Z;                                               \
    q[5][0] = -Y; q[5][1] = X;  q[5][2] = -Wv; q[5][3] = Z;                                              \
    q[6][0] = -Y; q[6][1] = -X; q[6][2] = -Wv; q[6][3] = -Z;                                             \
    q[7][0] = Y;  q[7][1] = -X; q[7][2] = Wv; q[7][3] = -Z;
    RIFE_ORI(qf, x, y, z, w)
    RIFE_ORI(qr, xr, yr, zr, wr)
#undef RIFE_ORI
#pragma unroll
    for (int ti = 0; ti < 8; ti++) { qf[ti][4] = m; qr[ti][4] = mr; }
}

// tail of the graph without postproc: out0 (3 x hp x wp) kept as float4 per padded pixel, for the TTA averaging
__global__ void k_final_float(const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, const float4* __restrict__ F,
                              const float* __restrict__ M, const float* __restrict__ flow3, float4* __restrict__ out, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wp) return;
    const size_t i = (size_t)y * wp + x;
    const float* fl = flow3 + i * 8;
    const float4 d = *reinterpret_cast<const float4*>(fl);
    float4 f = F[i];
    f.x = f.x + d.x; f.y = f.y + d.y; f.z = f.z + d.z; f.w = f.w + d.w;
    const float mm = M[i] + fl[4];
    const float m = 1.f / (1.f + expf(-mm));
    const float rm = 1.0f - m;
    const float3 w1 = warp_rgbx(img1, x, y, f.z, f.w, wp, hp);
    const float3 w0 = warp_rgbx(img0, x, y, f.x, f.y, wp, hp);
    out[i] = make_float4(w0.x * m + w1.x * rm, w0.y * m + w1.y * rm, w0.z * m + w1.z * rm, 0.f);
}

// gather nori (1 | 8) orientations x ntemp (1 | 2) directions of out0 back to the base frame, average, postproc.
// outs.p[ti] = forward outputs, outs.p[8 + ti] = time-reversed outputs.
__global__ void k_postproc_tta(Ptr16 outs, int nori, int ntemp, uint8_t* __restrict__ out, int w, int h, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= w || y >= h) return;
    float v[3], vr[3];
#pragma unroll
    for (int dir = 0; dir < 2; dir++) {
        if (dir >= ntemp) break;
        float s[3];
        if (nori == 8) {
            float4 t[8];
#pragma unroll
            for (int ti = 0; ti < 8; ti++) t[ti] = reinterpret_cast<const float4*>(outs.p[dir * 8 + ti])[tta_index(ti, y, x, wp, hp)];
            s[0] = (t[0].x + t[1].x + t[2].x + t[3].x + t[4].x + t[5].x + t[6].x + t[7].x) / 8;
            s[1] = (t[0].y + t[1].y + t[2].y + t[3].y + t[4].y + t[5].y + t[6].y + t[7].y) / 8;
            s[2] = (t[0].z + t[1].z + t[2].z + t[3].z + t[4].z + t[5].z + t[6].z + t[7].z) / 8;
        } else {
            const float4 t = reinterpret_cast<const float4*>(outs.p[dir * 8])[(size_t)y * wp + x];
            s[0] = t.x; s[1] = t.y; s[2] = t.z;
        }
#pragma unroll
        for (int c = 0; c < 3; c++) { if (dir == 0) v[c] = s[c]; else vr[c] = s[c]; }
    }
    uint8_t* o = out + ((size_t)y * w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float r = ntemp == 2 ? (v[c] + vr[c]) * 0.5f * 255.f + 0.5f : v[c] * 255.f + 0.5f;
        o[c] = (uint8_t)min(max((int)r, 0), 255);
    }
}

// layout converters for the parity taps (planar CHW fp32 <-> NHWC with a channel stride)
__global__ void k_chw_to_nhwc(const float* __restrict__ src, float* __restrict__ dst, int c, int h, int w, int ld) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    for (int q = 0; q < ld; q++) dst[((size_t)y * w + x) * ld + q] = q < c ? src[((size_t)q * h + y) * w + x] : 0.f;
}

__global__ void k_nhwc_to_chw(const float* __restrict__ src, float* __restrict__ dst, int c, int h, int w, int ld) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    for (int q = 0; q < c; q++) dst[((size_t)q * h + y) * w + x] = src[((size_t)y * w + x) * ld + q];
}

}  // namespace rife
