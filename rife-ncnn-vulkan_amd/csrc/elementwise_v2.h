// Bandwidth-bound kernels of the rife-v2.x schedule (IFNet + ContextNet + FusionNet; reference
// models/rife-v2.3/{flownet,contextnet,fusionnet}.param, orchestration src/rife.cpp:878-1183 / 2139-2457).
// Same rules as elementwise.h: literal restatement of the reference arithmetic, -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "elementwise.h"

namespace rife {

// Image accessors: the padded frame is RGBX u8 (normal mode) or, for the UHD half-resolution flow estimate, a float4
// image produced by Interp(0.5) of the fp32 frame (rife_uhd_downscale_image, rife.cpp:294-306, 928-931).
struct ImgU8 { const uint32_t* p; __device__ __forceinline__ float3 at(size_t i) const { return unpack_rgb(p[i]); } };
struct ImgF4 { const float4* p; __device__ __forceinline__ float3 at(size_t i) const { const float4 v = p[i]; return make_float3(v.x, v.y, v.z); } };

template <typename IMG>
__device__ __forceinline__ float3 warp_img(const IMG& img, int x, int y, float fx, float fy, int w, int h) {
    const WarpTaps t = warp_taps(x, y, fx, fy, w, h);
    const float3 a = img.at(t.i00), b = img.at(t.i01), c = img.at(t.i10), d = img.at(t.i11);
    return make_float3(warp_lerp(a.x, b.x, c.x, d.x, t.alpha, t.beta), warp_lerp(a.y, b.y, c.y, d.y, t.alpha, t.beta),
                       warp_lerp(a.z, b.z, c.z, d.z, t.alpha, t.beta));
}

// UHD: in_downscaled = Interp(0.5)(in_padded) as float4 per pixel (rife.cpp:928-931)
__global__ void k2_image_half(const uint32_t* __restrict__ img, float4* __restrict__ out, int wp, int hp) {
    const int wo = wp / 2, ho = hp / 2;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wo || y >= ho) return;
    const size_t i00 = (size_t)(2 * y) * wp + 2 * x, i10 = i00 + wp;
    const float3 a = unpack_rgb(img[i00]), b = unpack_rgb(img[i00 + 1]), c = unpack_rgb(img[i10]), d = unpack_rgb(img[i10 + 1]);
    out[(size_t)y * wo + x] = make_float4(down4(a.x, b.x, c.x, d.x), down4(a.y, b.y, c.y, d.y), down4(a.z, b.z, c.z, d.z), 0.f);
}

// UHD: flow = Interp(x2)(flow_downscaled) * 2   (rife_uhd_upscale_flow + rife_uhd_double_flow, rife.cpp:308-332, 940-944)
__global__ void k2_flow_up2_double(const float4* __restrict__ in, float4* __restrict__ out, int wo, int ho);

// 2 * Interp(x2)(flow) at full-resolution pixel (x, y); flow is float4 per pixel at (hp/2 x wp/2)
// (flownet.param:27-28 "Resize_22, Mul_24", fusionnet.param:14-15)
__device__ __forceinline__ float4 flow_up2x2(const float4* __restrict__ flow, int x, int y, int wh, int hh) {
    int sx, sy; float a0, a1, b0, b1;
    up_coeff(x, 2, wh, sx, a0, a1);
    up_coeff(y, 2, hh, sy, b0, b1);
    const float4 q00 = flow[(size_t)sy * wh + sx], q01 = flow[(size_t)sy * wh + sx + 1];
    const float4 q10 = flow[(size_t)(sy + 1) * wh + sx], q11 = flow[(size_t)(sy + 1) * wh + sx + 1];
    float4 u;
    u.x = ((q00.x * a0 + q01.x * a1) * b0 + (q10.x * a0 + q11.x * a1) * b1) * 2.0f;
    u.y = ((q00.y * a0 + q01.y * a1) * b0 + (q10.y * a0 + q11.y * a1) * b1) * 2.0f;
    u.z = ((q00.z * a0 + q01.z * a1) * b0 + (q10.z * a0 + q11.z * a1) * b1) * 2.0f;
    u.w = ((q00.w * a0 + q01.w * a1) * b0 + (q10.w * a0 + q11.w * a1) * b1) * 2.0f;
    return u;
}

// IFNet block 0 input: Interp(1/S)(Concat(input0, input1)) -> NHWC8 {rgb0, rgb1, 0, 0}   (S = 8: rife-v2.3 flownet.param:5-7;
// S = 4: rife-v3.x flownet.param:6-8)
template <int S, typename IMG>
__global__ void k2_assemble0(IMG img0, IMG img1, float* __restrict__ X, int wp, int hp) {
    const int Wb = wp / S, Hb = hp / S;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wb || y >= Hb) return;
    const size_t i00 = (size_t)(S * y + S / 2 - 1) * wp + S * x + S / 2 - 1, i10 = i00 + wp;
    const float3 a0 = img0.at(i00), a1 = img0.at(i00 + 1), a2 = img0.at(i10), a3 = img0.at(i10 + 1);
    const float3 b0 = img1.at(i00), b1 = img1.at(i00 + 1), b2 = img1.at(i10), b3 = img1.at(i10 + 1);
    float4* dst = reinterpret_cast<float4*>(X + ((size_t)y * Wb + x) * 8);
    dst[0] = make_float4(down4(a0.x, a1.x, a2.x, a3.x), down4(a0.y, a1.y, a2.y, a3.y), down4(a0.z, a1.z, a2.z, a3.z), down4(b0.x, b1.x, b2.x, b3.x));
    dst[1] = make_float4(down4(b0.y, b1.y, b2.y, b3.y), down4(b0.z, b1.z, b2.z, b3.z), 0.f, 0.f);
}

// IFNet blocks 1..3 and FusionNet input (flownet.param:27-37, 58-68, 90-99; fusionnet.param:14-23):
//   Ff = 2*Interp(x2)(acc);  x = Interp(1/S)(Concat(warp(img0, Ff.xy), warp(img1, Ff.zw), Ff))  -> NHWC16 (10 + 6 zero)
// FSCALE (rife-v3.x, flownet.param:44-46, 83-85): the flow channels are additionally multiplied by 1/S after the resize
// one pixel (x, y) of that block input at 1/S resolution: o[0..5] = the two warped frames, o[6..9] = the flow (shared by k2_assemble and the
// fused stem kernel of stem_fused_v2.h, so that both run the same arithmetic)
template <int S, typename IMG, bool FSCALE>
__device__ __forceinline__ void assemble2_pixel(const IMG& img0, const IMG& img1, const float4* __restrict__ acc, int wp, int hp, int x, int y, float (&o)[10]) {
    if (S == 1) {
        const float4 f = flow_up2x2(acc, x, y, wp / 2, hp / 2);
        const float3 w0 = warp_img(img0, x, y, f.x, f.y, wp, hp);
        const float3 w1 = warp_img(img1, x, y, f.z, f.w, wp, hp);
        o[0] = w0.x; o[1] = w0.y; o[2] = w0.z; o[3] = w1.x; o[4] = w1.y; o[5] = w1.z; o[6] = f.x; o[7] = f.y; o[8] = f.z; o[9] = f.w;
    } else {
        const int sx = S * x + S / 2 - 1, sy = S * y + S / 2 - 1;
        float v[4][10];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int px = sx + (k & 1), py = sy + (k >> 1);
            const float4 f = flow_up2x2(acc, px, py, wp / 2, hp / 2);
            const float3 w0 = warp_img(img0, px, py, f.x, f.y, wp, hp);
            const float3 w1 = warp_img(img1, px, py, f.z, f.w, wp, hp);
            v[k][0] = w0.x; v[k][1] = w0.y; v[k][2] = w0.z; v[k][3] = w1.x; v[k][4] = w1.y; v[k][5] = w1.z;
            v[k][6] = f.x; v[k][7] = f.y; v[k][8] = f.z; v[k][9] = f.w;
        }
#pragma unroll
        for (int c = 0; c < 10; c++) o[c] = down4(v[0][c], v[1][c], v[2][c], v[3][c]);
        if (FSCALE) {
#pragma unroll
            for (int c = 6; c < 10; c++) o[c] = o[c] * (1.0f / (float)S);
        }
    }
}

template <int S, typename IMG, bool FSCALE = false>
__global__ void k2_assemble(IMG img0, IMG img1, const float4* __restrict__ acc, float* __restrict__ X, int wp, int hp) {
    const int Wb = wp / S, Hb = hp / S;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= Wb || y >= Hb) return;
    float o[10];
    assemble2_pixel<S, IMG, FSCALE>(img0, img1, acc, wp, hp, x, y, o);
    float4* dst = reinterpret_cast<float4*>(X + ((size_t)y * Wb + x) * 16);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]);
    dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    dst[2] = make_float4(o[8], o[9], 0.f, 0.f);
    dst[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// running flow sum at half resolution: acc = (FIRST ? 0 : acc) + Interp(xS)(D)   (flownet.param:25, 56, 87, 117-119)
// D = deconv output, float4 per pixel at (hh/S x wh/S); acc float4 at (hh x wh)
// MUL (rife-v3.x, flownet.param:31-32, 71-72): the upsampled head output is multiplied by S before the sum
template <int S, bool FIRST, bool MUL = false>
__global__ void k2_flow_accum(const float4* __restrict__ D, float4* __restrict__ acc, int wh, int hh) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wh) return;
    float4 u;
    if (S == 1) u = D[(size_t)y * wh + x];
    else {
        const int Wd = wh / S, Hd = hh / S;
        int sx, sy; float a0, a1, b0, b1;
        up_coeff(x, S, Wd, sx, a0, a1);
        up_coeff(y, S, Hd, sy, b0, b1);
        const float4 q00 = D[(size_t)sy * Wd + sx], q01 = D[(size_t)sy * Wd + sx + 1];
        const float4 q10 = D[(size_t)(sy + 1) * Wd + sx], q11 = D[(size_t)(sy + 1) * Wd + sx + 1];
        u.x = (q00.x * a0 + q01.x * a1) * b0 + (q10.x * a0 + q11.x * a1) * b1;
        u.y = (q00.y * a0 + q01.y * a1) * b0 + (q10.y * a0 + q11.y * a1) * b1;
        u.z = (q00.z * a0 + q01.z * a1) * b0 + (q10.z * a0 + q11.z * a1) * b1;
        u.w = (q00.w * a0 + q01.w * a1) * b0 + (q10.w * a0 + q11.w * a1) * b1;
        if (MUL) { const float sc = (float)S; u.x = u.x * sc; u.y = u.y * sc; u.z = u.z * sc; u.w = u.w * sc; }
    }
    const size_t i = (size_t)y * wh + x;
    if (FIRST) acc[i] = u;
    else { const float4 a = acc[i]; acc[i] = make_float4(a.x + u.x, a.y + u.y, a.z + u.z, a.w + u.w); }
}

// ContextNet input: the padded frame as NHWC8 fp32 {r, g, b, 0...}   (contextnet.param:3)
__global__ void k2_image_nhwc8(const uint32_t* __restrict__ img, float* __restrict__ X, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float3 c = unpack_rgb(img[i]);
    float4* dst = reinterpret_cast<float4*>(X + i * 8);
    dst[0] = make_float4(c.x, c.y, c.z, 0.f);
    dst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// flow pyramid of the ContextNet: out = Interp(1/2)(in) * 0.5, 2 channels   (contextnet.param:14-15, 23-24, 32-33, 40-41)
// FROM4: level 0 reads channels (c0, c0+1) of the float4 flow (the Slice into flow0 / flow1, rife.cpp:1008-1016)
template <bool FROM4>
__global__ void k2_flow_half(const float* __restrict__ in, int c0, float2* __restrict__ out, int win, int hin) {
    const int wo = win / 2, ho = hin / 2;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wo || y >= ho) return;
    const int st = FROM4 ? 4 : 2;
    const float* p00 = in + ((size_t)(2 * y) * win + 2 * x) * st + (FROM4 ? c0 : 0);
    const float* p10 = p00 + (size_t)win * st;
    out[(size_t)y * wo + x] = make_float2(down4(p00[0], p00[st], p10[0], p10[st]) * 0.5f, down4(p00[1], p00[st + 1], p10[1], p10[st + 1]) * 0.5f);
}

// rife.Warp on NHWC features (C % 4 == 0) with a 2-channel flow, written into a channel slice of a concat buffer
// (contextnet.param:17, 26, 35, 42 -> fusionnet inputs "3".."10")
__global__ void k2_warp_nhwc(const float* __restrict__ feat, int C, const float2* __restrict__ flow, float* __restrict__ out, int out_ld, int out_coff,
                             int w, int h) {
    const int q = threadIdx.x;                       // float4 lane within the pixel
    const int nq = C / 4;
    const int pix_per_block = blockDim.x / nq;
    const int x = blockIdx.x * pix_per_block + threadIdx.x / nq, y = blockIdx.y;
    if (x >= w || (int)threadIdx.x >= pix_per_block * nq) return;
    const int c4 = q % nq;
    const float2 f = flow[(size_t)y * w + x];
    const WarpTaps t = warp_taps(x, y, f.x, f.y, w, h);
    const float4 a = *reinterpret_cast<const float4*>(feat + (size_t)t.i00 * C + c4 * 4);
    const float4 b = *reinterpret_cast<const float4*>(feat + (size_t)t.i01 * C + c4 * 4);
    const float4 c = *reinterpret_cast<const float4*>(feat + (size_t)t.i10 * C + c4 * 4);
    const float4 d = *reinterpret_cast<const float4*>(feat + (size_t)t.i11 * C + c4 * 4);
    float4 o;
    o.x = warp_lerp(a.x, b.x, c.x, d.x, t.alpha, t.beta); o.y = warp_lerp(a.y, b.y, c.y, d.y, t.alpha, t.beta);
    o.z = warp_lerp(a.z, b.z, c.z, d.z, t.alpha, t.beta); o.w = warp_lerp(a.w, b.w, c.w, d.w, t.alpha, t.beta);
    *reinterpret_cast<float4*>(out + ((size_t)y * w + x) * out_ld + out_coff + c4 * 4) = o;
}

// the eight ContextNet warps of a pair (4 pyramid levels x 2 frames: contextnet.param:17, 26, 35, 42 run twice, src/rife.cpp:1027-1060) in ONE launch:
// blockIdx.z = 4 * frame + level selects the tensors, the grid is sized for level 0 and the blocks beyond a level's extent leave at once.  Same arithmetic
// as k2_warp_nhwc; 32 / 64 / 128 / 256 channels = 8 / 16 / 32 / 64 float4 lanes per pixel of a 256-thread block.
struct WarpBatch {
    const float* feat[8]; const float2* flow[8]; float* out[8];
    int C[8], out_ld[8], out_coff[8], w[8], h[8];
};
__global__ void k2_warp_nhwc_batch(WarpBatch b) {
    const int z = blockIdx.z;
    const int C = b.C[z], w = b.w[z], h = b.h[z];
    const int nq = C / 4, pix_per_block = 256 / nq;
    const int x = blockIdx.x * pix_per_block + threadIdx.x / nq, y = blockIdx.y;
    if (y >= h || x >= w) return;
    const int c4 = threadIdx.x % nq;
    const float* __restrict__ feat = b.feat[z];
    const float2 f = b.flow[z][(size_t)y * w + x];
    const WarpTaps t = warp_taps(x, y, f.x, f.y, w, h);
    const float4 a = *reinterpret_cast<const float4*>(feat + (size_t)t.i00 * C + c4 * 4);
    const float4 bb = *reinterpret_cast<const float4*>(feat + (size_t)t.i01 * C + c4 * 4);
    const float4 c = *reinterpret_cast<const float4*>(feat + (size_t)t.i10 * C + c4 * 4);
    const float4 d = *reinterpret_cast<const float4*>(feat + (size_t)t.i11 * C + c4 * 4);
    float4 o;
    o.x = warp_lerp(a.x, bb.x, c.x, d.x, t.alpha, t.beta); o.y = warp_lerp(a.y, bb.y, c.y, d.y, t.alpha, t.beta);
    o.z = warp_lerp(a.z, bb.z, c.z, d.z, t.alpha, t.beta); o.w = warp_lerp(a.w, bb.w, c.w, d.w, t.alpha, t.beta);
    *reinterpret_cast<float4*>(b.out[z] + ((size_t)y * w + x) * b.out_ld[z] + b.out_coff[z] + c4 * 4) = o;
}

// channel-slice copy between NHWC views (U-Net skip connections, fusionnet.param:53, 56, 59)
__global__ void k2_copy_view(const float* __restrict__ src, int src_ld, int src_coff, float* __restrict__ dst, int dst_ld, int dst_coff, int C, size_t npix) {
    const int nq = C / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * nq) return;
    const size_t p = i / nq; const int q = (int)(i - p * nq);
    *reinterpret_cast<float4*>(dst + p * dst_ld + dst_coff + q * 4) = *reinterpret_cast<const float4*>(src + p * src_ld + src_coff + q * 4);
}

// FusionNet tail + postproc (fusionnet.param:64-74; rife.cpp:1167-1182 / 2434-2456):
//   o = sigmoid head (4 ch, applied in the deconv epilogue); res = o.rgb*2 - 1; m = o.w;
//   out = clip(warp(img0,Ff.xy)*m + warp(img1,Ff.zw)*(1-m) + res, 0, 1);  u8 = clamp((int)(out*255 + 0.5))
__global__ void k2_final(const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, const float4* __restrict__ flow,
                         const float4* __restrict__ head, uint8_t* __restrict__ out, int w, int h, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const float4 f = flow_up2x2(flow, x, y, wp / 2, hp / 2);
    const float3 w0 = warp_rgbx(img0, x, y, f.x, f.y, wp, hp);
    const float3 w1 = warp_rgbx(img1, x, y, f.z, f.w, wp, hp);
    const float4 o = head[(size_t)y * wp + x];
    const float m = o.w, rm = 1.0f - m;
    float v[3];
    v[0] = (w0.x * m + w1.x * rm) + (o.x * 2.0f - 1.0f);
    v[1] = (w0.y * m + w1.y * rm) + (o.y * 2.0f - 1.0f);
    v[2] = (w0.z * m + w1.z * rm) + (o.z * 2.0f - 1.0f);
    uint8_t* dst = out + ((size_t)y * w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float t = v[c];
        if (t < 0.f) t = 0.f;
        if (t > 1.f) t = 1.f;
        dst[c] = (uint8_t)min(max((int)(t * 255.f + 0.5f), 0), 255);
    }
}

// tail of the v2 graph without postproc: the clipped output (fusionnet.param:63-74) kept as float4 per padded pixel, for the
// TTA averaging (rife.cpp:804-876 GPU, 2050-2137 CPU)
__global__ void k2_final_float(const uint32_t* __restrict__ img0, const uint32_t* __restrict__ img1, const float4* __restrict__ flow,
                               const float4* __restrict__ head, float4* __restrict__ out, int wp, int hp) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wp) return;
    const float4 f = flow_up2x2(flow, x, y, wp / 2, hp / 2);
    const float3 w0 = warp_rgbx(img0, x, y, f.x, f.y, wp, hp);
    const float3 w1 = warp_rgbx(img1, x, y, f.z, f.w, wp, hp);
    const float4 o = head[(size_t)y * wp + x];
    const float m = o.w, rm = 1.0f - m;
    float v[3];
    v[0] = (w0.x * m + w1.x * rm) + (o.x * 2.0f - 1.0f);
    v[1] = (w0.y * m + w1.y * rm) + (o.y * 2.0f - 1.0f);
    v[2] = (w0.z * m + w1.z * rm) + (o.z * 2.0f - 1.0f);
#pragma unroll
    for (int c = 0; c < 3; c++) v[c] = fminf(fmaxf(v[c], 0.f), 1.f);
    out[(size_t)y * wp + x] = make_float4(v[0], v[1], v[2], 0.f);
}

// v2 forward / reversed flow consensus, in place on both float4 fields (rife.cpp:1484-1540 and 1898-1948;
// GPU twin rife_v2_flow_tta_temporal_avg.comp)
__global__ void k2_temporal_merge(float4* __restrict__ f, float4* __restrict__ r, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float4 a = f[i], b = r[i];
    const float x = (a.x + b.z) * 0.5f, y = (a.y + b.w) * 0.5f, z = (a.z + b.x) * 0.5f, w = (a.w + b.y) * 0.5f;
    f[i] = make_float4(x, y, z, w);
    r[i] = make_float4(z, w, x, y);
}

// v2 8-orientation flow consensus, in place on eight float4 fields; W x H = size of orientation 0 (rife.cpp:1543-1667;
// GPU twin rife_v2_flow_tta_avg.comp)
__global__ void k2_spatial_avg(Ptr8 fl, int W, int H) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;      // 2-D thread tiles (tta_block)
    if (j >= W || i >= H) return;
    float4* q[8];
    float4 v[8];
#pragma unroll
    for (int ti = 0; ti < 8; ti++) { q[ti] = reinterpret_cast<float4*>(fl.p[ti]) + tta_index(ti, i, j, W, H); v[ti] = *q[ti]; }
    const float x = (v[0].x + -v[1].x + -v[2].x + v[3].x + v[4].y + v[5].y + -v[6].y + -v[7].y) * 0.125f;
    const float y = (v[0].y + v[1].y + -v[2].y + -v[3].y + v[4].x + -v[5].x + -v[6].x + v[7].x) * 0.125f;
    const float z = (v[0].z + -v[1].z + -v[2].z + v[3].z + v[4].w + v[5].w + -v[6].w + -v[7].w) * 0.125f;
    const float w = (v[0].w + v[1].w + -v[2].w + -v[3].w + v[4].z + -v[5].z + -v[6].z + v[7].z) * 0.125f;
    *q[0] = make_float4(x, y, z, w);
    *q[1] = make_float4(-x, y, -z, w);
    *q[2] = make_float4(-x, -y, -z, -w);
    *q[3] = make_float4(x, -y, z, -w);
    *q[4] = make_float4(y, x, w, z);
    *q[5] = make_float4(-y, x, -w, z);
    *q[6] = make_float4(-y, -x, -w, -z);
    *q[7] = make_float4(y, -x, w, -z);
}

__global__ void k2_flow_up2_double(const float4* __restrict__ in, float4* __restrict__ out, int wo, int ho) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= wo) return;
    out[(size_t)y * wo + x] = flow_up2x2(in, x, y, wo / 2, ho / 2);      // (Interp x2) * 2.0: the same arithmetic as the in-graph "Resize, Mul 2"
}

}  // namespace rife
