// ORACLE — TEST INFRASTRUCTURE ONLY (see ncnn_graph.h header).  PARITY UNPINNED.
//
// Direct fp32 Convolution / Deconvolution with ncnn's layer semantics.
// Summation order per output element: bias, then input channel (outer), ky, kx (inner) — the order
// of ncnn's generic convolution.  The loops are vectorised across x only (each output element still
// sees the same sequential sum), OpenMP over output channels like ncnn (`opt.num_threads`).
// This TU is compiled with FMA contraction allowed: ncnn's x86 kernels use FMA too, and the exact
// rounding of the reference's conv (Winograd / packed sgemm inside ncnn) is not reproducible anyway
//  — the +-1 LSB output tolerance absorbs it.
#include "ncnn_graph.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace oracle {

static inline float activate(float v, int act_type, const float* p) {
    switch (act_type) {
        case 0: return v;
        case 1: return v > 0 ? v : 0.f;
        case 2: return v > 0 ? v : v * p[0];             // leaky, slope = activation_params[0]
        case 3: return std::min(std::max(v, p[0]), p[1]);
        case 4: return 1.f / (1.f + std::exp(-v));       // sigmoid (fusionnet.param:62)
    }
    return v;
}

void conv2d(const Mat& in, Mat& out, const float* weight, const float* bias, int outc, int k, int stride, int pad,
            int act_type, const float* act_params, int num_threads) {
    const int inc = in.c, w = in.w, h = in.h;
    const int pw = w + 2 * pad, ph = h + 2 * pad;
    const int outw = (pw - k) / stride + 1, outh = (ph - k) / stride + 1;
    // zero border (ncnn copy_make_border, BORDER_CONSTANT 0)
    Mat padded(pw, ph, inc);
    std::memset(padded.data, 0, padded.total() * sizeof(float));
    for (int q = 0; q < inc; q++)
        for (int y = 0; y < h; y++)
            std::memcpy(padded.channel(q) + (size_t)(y + pad) * pw + pad, in.channel(q) + (size_t)y * w, (size_t)w * sizeof(float));
    out.create(outw, outh, outc);
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 1)
    for (int p = 0; p < outc; p++) {
        float* o = out.channel(p);
        const float b = bias ? bias[p] : 0.f;
        for (size_t i = 0; i < (size_t)outw * outh; i++) o[i] = b;
        for (int q = 0; q < inc; q++) {
            const float* kptr = weight + ((size_t)p * inc + q) * k * k;
            const float* src = padded.channel(q);
            if (k == 3 && stride == 1) {
                const float k0 = kptr[0], k1 = kptr[1], k2 = kptr[2], k3 = kptr[3], k4 = kptr[4], k5 = kptr[5], k6 = kptr[6], k7 = kptr[7], k8 = kptr[8];
                for (int y = 0; y < outh; y++) {
                    const float* r0 = src + (size_t)y * pw;
                    const float* r1 = r0 + pw;
                    const float* r2 = r1 + pw;
                    float* orow = o + (size_t)y * outw;
#pragma omp simd
                    for (int x = 0; x < outw; x++) {
                        float s = orow[x];
                        s += k0 * r0[x]; s += k1 * r0[x + 1]; s += k2 * r0[x + 2];
                        s += k3 * r1[x]; s += k4 * r1[x + 1]; s += k5 * r1[x + 2];
                        s += k6 * r2[x]; s += k7 * r2[x + 1]; s += k8 * r2[x + 2];
                        orow[x] = s;
                    }
                }
            } else {
                for (int y = 0; y < outh; y++) {
                    float* orow = o + (size_t)y * outw;
                    for (int ky = 0; ky < k; ky++) {
                        const float* r = src + (size_t)(y * stride + ky) * pw;
                        for (int kx = 0; kx < k; kx++) {
                            const float kv = kptr[ky * k + kx];
#pragma omp simd
                            for (int x = 0; x < outw; x++) orow[x] += kv * r[x * stride + kx];
                        }
                    }
                }
            }
        }
        if (act_type)
            for (size_t i = 0; i < (size_t)outw * outh; i++) o[i] = activate(o[i], act_type, act_params);
    }
}

// ncnn Deconvolution: scatter-accumulate onto a ((in-1)*stride + k) canvas, crop `pad` on every side,
// weights [oc][ic][ky][kx], no kernel flip.
void deconv2d(const Mat& in, Mat& out, const float* weight, const float* bias, int outc, int k, int stride, int pad,
              int act_type, const float* act_params, int num_threads) {
    const int inc = in.c, w = in.w, h = in.h;
    const int fw = (w - 1) * stride + k, fh = (h - 1) * stride + k;
    const int outw = fw - 2 * pad, outh = fh - 2 * pad;
    out.create(outw, outh, outc);
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 1)
    for (int p = 0; p < outc; p++) {
        std::vector<float> canvas((size_t)fw * fh, bias ? bias[p] : 0.f);
        for (int i = 0; i < h; i++) {
            for (int q = 0; q < inc; q++) {
                const float* kptr = weight + ((size_t)p * inc + q) * k * k;
                const float* srow = in.channel(q) + (size_t)i * w;
                for (int ky = 0; ky < k; ky++) {
                    float* crow = canvas.data() + (size_t)(i * stride + ky) * fw;
                    for (int kx = 0; kx < k; kx++) {
                        const float kv = kptr[ky * k + kx];
                        for (int j = 0; j < w; j++) crow[j * stride + kx] += srow[j] * kv;
                    }
                }
            }
        }
        float* o = out.channel(p);
        for (int y = 0; y < outh; y++)
            for (int x = 0; x < outw; x++)
                o[(size_t)y * outw + x] = activate(canvas[(size_t)(y + pad) * fw + x + pad], act_type, act_params);
    }
}

}  // namespace oracle
