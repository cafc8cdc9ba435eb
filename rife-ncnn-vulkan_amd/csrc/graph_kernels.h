// Layer-wise kernels of the generic graph executor (graph_exec.h): one kernel per ncnn layer type that the hand-scheduled
// families (rife-v4.x, rife-v2.x / v3.x) fuse away.  Used for the v1 family (models/rife, rife-HD, rife-UHD, rife-anime), whose
// graphs need squeeze-and-excitation blocks (Pooling + 2 x InnerProduct + channel-wise BinaryOp), 5 x 5 convolutions, UnaryOp
// and bias-free convolutions.  Every blob is NHWC fp32 with a pixel stride `ld` = channels rounded up to 16 (pad channels are
// written once, as zeros, when the blob is allocated; no kernel here touches them), or a plain vector for the 1 x 1 x C blobs
// behind Pooling / InnerProduct.  Arithmetic follows the ncnn layer semantics of SURVEY.md App. C operation for operation.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "elementwise.h"

namespace rife {

struct GView { float* p; int c, h, w, ld; };   // NHWC view (h = w = 1, ld = c for vectors)

// padded RGBX u8 frame -> 3-channel NHWC fp32 blob, x * (1 / 255.f) like the reference's preproc (rife.cpp:2144-2203)
__global__ void kg_from_rgbx(const uint32_t* __restrict__ img, float* __restrict__ out, int ld, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const float3 c = unpack_rgb(img[i]);
    float* o = out + i * ld;
    o[0] = c.x; o[1] = c.y; o[2] = c.z;
}

// channel-slice copy: Concat (one call per bottom), Crop
__global__ void kg_copy_channels(const float* __restrict__ src, int src_ld, int src_c0, float* __restrict__ dst, int dst_ld, int dst_c0, int c, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * c) return;
    const size_t p = i / c; const int q = (int)(i - p * c);
    dst[p * dst_ld + dst_c0 + q] = src[p * src_ld + src_c0 + q];
}

// ncnn Interp resize_type 2 (bilinear), align_corner 0: linear_coeffs in double like the reference, horizontal pass then vertical
__device__ __forceinline__ void gi_coeff(int d, double scale, int in, int& s0, float& a0, float& a1) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= in - 1) { s = in - 2; f = 1.f; }
    s0 = s; a0 = 1.f - f; a1 = f;
}

__global__ void kg_interp(GView in, GView out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= out.w) return;
    int sx, sy; float a0, a1, b0, b1;
    gi_coeff(x, (double)in.w / out.w, in.w, sx, a0, a1);
    gi_coeff(y, (double)in.h / out.h, in.h, sy, b0, b1);
    const float* p00 = in.p + ((size_t)sy * in.w + sx) * in.ld;
    const float* p01 = in.w > 1 ? p00 + in.ld : p00;
    const float* p10 = in.h > 1 ? p00 + (size_t)in.w * in.ld : p00;
    const float* p11 = in.w > 1 ? p10 + in.ld : p10;
    float* o = out.p + ((size_t)y * out.w + x) * out.ld;
    for (int q = 0; q < out.c; q++) o[q] = (p00[q] * a0 + p01[q] * a1) * b0 + (p10[q] * a0 + p11[q] * a1) * b1;
}

// ncnn BinaryOp codes: 0 add, 1 sub, 2 mul, 3 div, 7 rsub (b - a)
__device__ __forceinline__ float g_binop(int op, float a, float b) {
    switch (op) {
        case 0: return a + b;
        case 1: return a - b;
        case 2: return a * b;
        case 3: return a / b;
        case 7: return b - a;
    }
    return a;
}

__global__ void kg_binary_scalar(GView a, GView out, int op, float b) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)a.h * a.w * a.c;
    if (i >= n) return;
    const size_t p = i / a.c; const int q = (int)(i - p * a.c);
    out.p[p * out.ld + q] = g_binop(op, a.p[p * a.ld + q], b);
}

// bmode 0: same shape; 1: b is a per-channel vector (SE scale); 2: b has one channel (broadcast over channels)
__global__ void kg_binary(GView a, GView b, GView out, int op, int bmode) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)a.h * a.w * a.c;
    if (i >= n) return;
    const size_t p = i / a.c; const int q = (int)(i - p * a.c);
    const float bv = bmode == 0 ? b.p[p * b.ld + q] : (bmode == 1 ? b.p[q] : b.p[p * b.ld]);
    out.p[p * out.ld + q] = g_binop(op, a.p[p * a.ld + q], bv);
}

__global__ void kg_eltwise2(GView a, GView b, GView out, float ca, float cb, int has_coeff) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)a.h * a.w * a.c;
    if (i >= n) return;
    const size_t p = i / a.c; const int q = (int)(i - p * a.c);
    const float x = a.p[p * a.ld + q], y = b.p[p * b.ld + q];
    out.p[p * out.ld + q] = has_coeff ? x * ca + y * cb : x + y;
}

// mode 0: neg (UnaryOp 1); 1: sigmoid; 2: clip(lo, hi); 3: leaky / ReLU with one slope; 4: PReLU with per-channel slopes
__global__ void kg_pointwise(GView a, GView out, int mode, float p0, float p1, const float* __restrict__ slopes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)a.h * a.w * a.c;
    if (i >= n) return;
    const size_t p = i / a.c; const int q = (int)(i - p * a.c);
    float v = a.p[p * a.ld + q];
    if (mode == 0) v = -v;
    else if (mode == 1) v = 1.f / (1.f + expf(-v));
    else if (mode == 2) { if (v < p0) v = p0; if (v > p1) v = p1; }
    else if (mode == 3) { if (v < 0.f) v = v * p0; }
    else { if (v < 0.f) v = v * slopes[q]; }
    out.p[p * out.ld + q] = v;
}

// tail of a squeeze-and-excitation residual block in one pass: BinaryOp mul (per-channel scale) -> BinaryOp add (skip) -> PReLU
__global__ void kg_se_tail(GView y, const float* __restrict__ scale, GView skip, const float* __restrict__ slopes, GView out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)y.h * y.w * y.c;
    if (i >= n) return;
    const size_t p = i / y.c; const int q = (int)(i - p * y.c);
    float v = y.p[p * y.ld + q] * scale[q];
    v = v + skip.p[p * skip.ld + q];
    if (v < 0.f) v = v * slopes[q];
    out.p[p * out.ld + q] = v;
}

// ncnn PixelShuffle (mode 0): out[c][y*r+i][x*r+j] = in[c*r*r + i*r + j][y][x]
__global__ void kg_pixelshuffle(GView in, GView out, int r) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)out.h * out.w * out.c;
    if (i >= n) return;
    const size_t p = i / out.c; const int q = (int)(i - p * out.c);
    const int oy = (int)(p / out.w), ox = (int)(p - (size_t)oy * out.w);
    const int y = oy / r, x = ox / r, ii = oy - y * r, jj = ox - x * r;
    out.p[p * out.ld + q] = in.p[((size_t)y * in.w + x) * in.ld + q * r * r + ii * r + jj];
}

// rife.Warp (src/warp.cpp:96-168) on NHWC blobs: image c channels, flow channels 0 / 1 of its blob
__global__ void kg_warp(GView img, GView flow, GView out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= img.w) return;
    const float* f = flow.p + ((size_t)y * img.w + x) * flow.ld;
    const WarpTaps t = warp_taps(x, y, f[0], f[1], img.w, img.h);
    const float* a = img.p + (size_t)t.i00 * img.ld; const float* b = img.p + (size_t)t.i01 * img.ld;
    const float* c = img.p + (size_t)t.i10 * img.ld; const float* d = img.p + (size_t)t.i11 * img.ld;
    float* o = out.p + ((size_t)y * img.w + x) * out.ld;
    for (int q = 0; q < img.c; q++) o[q] = warp_lerp(a[q], b[q], c[q], d[q], t.alpha, t.beta);
}

// Global average pooling, stage 1: per-channel partial sums of one contiguous pixel chunk per block (double accumulation: the
// result is the correctly rounded mean to well below the fp32 sequential sum of the reference, whose own error is ~1e-6 relative).
// 256 threads = (256 / (c/4)) pixel lanes x (c/4) channel quads, float4 loads; needs c % 4 == 0 and c <= 1024.
__global__ void kg_pool_partial(GView in, double* __restrict__ partial, int nchunks) {
    __shared__ double sm[256 * 4];
    const int cq = in.c / 4, ppb = 256 / cq;
    const int pl = threadIdx.x / cq, q = threadIdx.x - pl * cq;
    const size_t npix = (size_t)in.h * in.w;
    const size_t per = (npix + nchunks - 1) / nchunks;
    const size_t p0 = (size_t)blockIdx.x * per, p1 = min(npix, p0 + per);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (pl < ppb)
        for (size_t p = p0 + pl; p < p1; p += ppb) {
            const float4 v = *reinterpret_cast<const float4*>(in.p + p * in.ld + q * 4);
            a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
        }
    sm[threadIdx.x * 4 + 0] = a0; sm[threadIdx.x * 4 + 1] = a1; sm[threadIdx.x * 4 + 2] = a2; sm[threadIdx.x * 4 + 3] = a3;
    __syncthreads();
    for (int c = threadIdx.x; c < in.c; c += 256) {
        const int qq = c >> 2, e = c & 3;
        double s = 0.0;
        for (int l = 0; l < ppb; l++) s += sm[(l * cq + qq) * 4 + e];
        partial[(size_t)blockIdx.x * in.c + c] = s;
    }
}

__global__ void kg_pool_finish(const double* __restrict__ partial, int nchunks, int c, double inv_npix, float* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= c) return;
    double s = 0.0;
    for (int k = 0; k < nchunks; k++) s += partial[(size_t)k * c + q];
    out[q] = (float)(s * inv_npix);
}

// InnerProduct on a vector: out[o] = act(bias[o] + sum_i w[o][i] x[i]); act 0 none, 1 relu, 2 leaky(p0), 4 sigmoid
__global__ void kg_inner(const float* __restrict__ x, int n, const float* __restrict__ w, const float* __restrict__ bias, int outc, int act, float p0,
                         float* __restrict__ out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= outc) return;
    float s = bias ? bias[o] : 0.f;
    const float* wr = w + (size_t)o * n;
    for (int i = 0; i < n; i++) s += x[i] * wr[i];
    if (act == 1) s = s > 0.f ? s : 0.f;
    else if (act == 2) s = s > 0.f ? s : s * p0;
    else if (act == 4) s = 1.f / (1.f + expf(-s));
    out[o] = s;
}

// Direct convolution for the kernel sizes the MFMA kernels do not cover (5 x 5, stride 1 | 2, pad 2: the rife-HD IFNet).
// weights repacked [ky][kx][ic][oc]; thread = (pixel, 4 output channels); fp32 FMA-free (mul + add) accumulation.
__global__ void kg_conv_direct(GView in, GView out, const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ slope,
                               int k, int stride, int pad) {
    const int ocg = out.c / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)out.h * out.w * ocg;
    if (i >= n) return;
    const size_t p = i / ocg; const int o4 = (int)(i - p * ocg) * 4;
    const int oy = (int)(p / out.w), ox = (int)(p - (size_t)oy * out.w);
    float acc[4] = {bias[o4], bias[o4 + 1], bias[o4 + 2], bias[o4 + 3]};
    for (int ky = 0; ky < k; ky++) {
        const int iy = oy * stride + ky - pad;
        if (iy < 0 || iy >= in.h) continue;
        for (int kx = 0; kx < k; kx++) {
            const int ix = ox * stride + kx - pad;
            if (ix < 0 || ix >= in.w) continue;
            const float* xp = in.p + ((size_t)iy * in.w + ix) * in.ld;
            const float* wp = w + ((size_t)(ky * k + kx) * in.c) * out.c + o4;
            for (int ic = 0; ic < in.c; ic++) {
                const float xv = xp[ic];
                const float4 wv = *reinterpret_cast<const float4*>(wp + (size_t)ic * out.c);
                acc[0] += xv * wv.x; acc[1] += xv * wv.y; acc[2] += xv * wv.z; acc[3] += xv * wv.w;
            }
        }
    }
    float* o = out.p + p * out.ld + o4;
#pragma unroll
    for (int e = 0; e < 4; e++) { float v = acc[e]; if (v < 0.f) v = v * slope[o4 + e]; o[e] = v; }
}

// 3-channel NHWC blob (the FusionNet "output", already clipped to [0, 1]) -> u8 HWC RGB, cropped (rife.cpp:2434-2456)
__global__ void kg_to_u8(GView in, uint8_t* __restrict__ out, int w, int h) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w || y >= h) return;
    const float* p = in.p + ((size_t)y * in.w + x) * in.ld;
    uint8_t* o = out + ((size_t)y * w + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) o[c] = (uint8_t)min(max((int)(p[c] * 255.f + 0.5f), 0), 255);
}

// 3-channel NHWC blob -> float4 per padded pixel (the TTA averaging kernel's input)
__global__ void kg_to_float4(GView in, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)in.h * in.w) return;
    const float* p = in.p + i * in.ld;
    out[i] = make_float4(p[0], p[1], p[2], 0.f);
}

// v1-family TTA consensus on 2-channel flow blobs (rife.cpp:1525-1538 / 2304-2316 temporal, 1669-1716 spatial)
__global__ void kg_v1_temporal_merge(GView f, GView r) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)f.h * f.w) return;
    float* a = f.p + i * f.ld; float* b = r.p + i * r.ld;
    const float x = (a[0] - b[0]) * 0.5f, y = (a[1] - b[1]) * 0.5f;
    a[0] = x; a[1] = y; b[0] = -x; b[1] = -y;
}

__global__ void kg_v1_spatial_avg(Ptr8 fl, int ld, int W, int H) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y * blockDim.y + threadIdx.y;      // 2-D thread tiles (tta_block)
    if (j >= W || i >= H) return;
    float* q[8];
#pragma unroll
    for (int ti = 0; ti < 8; ti++) q[ti] = reinterpret_cast<float*>(fl.p[ti]) + tta_index(ti, i, j, W, H) * ld;
    const float x = (q[0][0] + -q[1][0] + -q[2][0] + q[3][0] + q[4][1] + q[5][1] + -q[6][1] + -q[7][1]) * 0.125f;
    const float y = (q[0][1] + q[1][1] + -q[2][1] + -q[3][1] + q[4][0] + -q[5][0] + -q[6][0] + q[7][0]) * 0.125f;
    q[0][0] = x;  q[0][1] = y;
    q[1][0] = -x; q[1][1] = y;
    q[2][0] = -x; q[2][1] = -y;
    q[3][0] = x;  q[3][1] = -y;
    q[4][0] = y;  q[4][1] = x;
    q[5][0] = -y; q[5][1] = x;
    q[6][0] = -y; q[6][1] = -x;
    q[7][0] = y;  q[7][1] = -x;
}

}  // namespace rife
