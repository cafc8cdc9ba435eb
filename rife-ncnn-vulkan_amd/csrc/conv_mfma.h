// Implicit-GEMM convolution on the CDNA4 matrix cores, fp32 in / fp32 accumulate
// (v_mfma_f32_32x32x2_f32: exact f32 fma chain, 157 TFLOP/s dense peak on MI355X).
//
// Replaces the arithmetic the reference delegates to ncnn's Convolution / Deconvolution layers
// (reference models/rife-v4.6/flownet.param:11-45, models/rife-v2.3/*.param):
//   * 3x3 pad-1 convolution, stride 1 or 2, + bias, optional residual add, per-channel negative slope
//     (covers "no activation", fused LeakyReLU(0.2), ncnn ReLU(slope) and PReLU);
//   * 4x4 stride-2 pad-1 transposed convolution, decomposed into its 4 output parities, each a 2x2-tap
//     convolution over the input grid; optional fused PixelShuffle(2) scatter (v4.6 heads) or sigmoid.
//
// GEMM view: M = output pixels (conv) / input pixels (deconv parity), N = output channels, K = taps x Cin.
//   workgroup = 256 threads = 4 waves; tile = (4*MS) rows x 32 columns of pixels x NT = 32*NS channels
//   wave w owns rows [w*MS, w*MS+MS) of the tile: MS x NS accumulators of 32x32 (16 VGPRs each)
//   K loop  = Cin chunks of CC channels (staged to LDS: input halo tile + the chunk's weight slab)
//             x taps x 8-channel groups; one ds_read_b128 per lane feeds 4 consecutive MFMA k-steps:
//             lanes 0-31 hold channels g*8+0..3, lanes 32-63 channels g*8+4..7 (A: pixel rows, B: weights).
// Layouts: activations NHWC fp32 with explicit channel stride/offset (so producers can write straight into
// concat buffers); weights pre-packed on the host (see pack_conv_weights in engine.hip) in exactly the order the
// B-fragment ds_read_b128 wants: [ntile][chunk][tap][g][half][n][4].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rife {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum ConvEpilogue {
    EPI_STORE = 0,       // y = slope(acc + bias [+ residual])                    -> NHWC store
    EPI_DECONV = 1,      // same, output pixel (2y+py, 2x+px)                       -> NHWC store (v2.3 fusionnet)
    EPI_DECONV_PS = 2,   // acc + bias, deconv(24) + PixelShuffle(2) scatter        -> flow tensor [H*4][W*4][8] (v4.6 heads)
    EPI_DECONV_SIG = 3,  // sigmoid(acc + bias), output pixel (2y+py, 2x+px)        -> NHWC store (v2.3 fusionnet head)
};

struct ConvArgs {
    const float* in;       // NHWC, pixel stride in_ld floats, first channel at in_coff
    float* out;            // NHWC, pixel stride out_ld, first channel at out_coff
    const float* wpk;      // packed weights
    const float* bias;     // [Cout_padded]
    const float* slope;    // [Cout_padded] negative-side slope per channel
    const float* res;      // residual (same geometry as out) or nullptr
    int H, W;              // input height/width (pixels)
    int in_ld, in_coff;
    int Ho, Wo;            // GEMM-M grid (output pixels for conv; input pixels for deconv parities)
    int out_ld, out_coff;
    int res_ld, res_coff;
    int Cout;              // real output channels (stores masked beyond)
    int nchunks;           // Cin_padded / CC
    int ntaps;             // 9 (conv) or 4 (deconv parity)
    int npar;              // 1 (conv) or 4 (deconv)
    int tiles_x;
    int8_t tdy[4][9], tdx[4][9];   // tap offsets in input pixels, per parity
};

// geometry helpers (compile-time)
template <int STRIDE, int MS> struct ConvGeom {
    static constexpr int TH = 4 * MS, TW = 32;
    static constexpr int IH = (TH - 1) * STRIDE + 3, IW = (TW - 1) * STRIDE + 3;
};

template <int STRIDE, int MS, int NS, int CC>
constexpr int conv_lds_bytes() {
    return (ConvGeom<STRIDE, MS>::IH * ConvGeom<STRIDE, MS>::IW * (CC + 4) + 9 * CC * NS * 32) * 4;
}

template <int STRIDE, int MS, int NS, int CC, int EPI>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    using G = ConvGeom<STRIDE, MS>;
    constexpr int S = CC + 4;                 // LDS pixel stride (floats): S/4 odd -> conflict-free b128 column reads
    constexpr int NT = NS * 32;
    constexpr int NG = CC / 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_in = lds;                                   // [IH][IW][S]
    float* lds_w = lds + G::IH * G::IW * S;                // [ntaps][NG][2][NT][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, li = lane & 31;
    const int tile = blockIdx.x;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int par = blockIdx.z % a.npar, ntile = blockIdx.z / a.npar;
    const int oy0 = ty * G::TH, ox0 = tx * G::TW;
    const int iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;   // top-left of the staged halo tile in input pixels

    f32x16 acc[MS][NS];
#pragma unroll
    for (int m = 0; m < MS; m++)
#pragma unroll
        for (int n = 0; n < NS; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    const int wchunk = a.ntaps * CC * NT;                       // floats per weight chunk
    const float* wbase = a.wpk + (size_t)(ntile * a.npar + par) * a.nchunks * wchunk;

    for (int ch = 0; ch < a.nchunks; ch++) {
        __syncthreads();
        // ---- stage the input halo tile, channels [ch*CC, ch*CC+CC) ----
        {
            constexpr int NQ = CC / 4;
            constexpr int TOTAL = G::IH * G::IW * NQ;
            const int cbase = a.in_coff + ch * CC;
            for (int idx = tid; idx < TOTAL; idx += 256) {
                const int p = idx / NQ, q = idx - p * NQ;
                const int py = p / G::IW, px = p - py * G::IW;
                const int gy = iy0 + py, gx = ix0 + px;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W)
                    v = *reinterpret_cast<const float4*>(a.in + ((size_t)gy * a.W + gx) * a.in_ld + cbase + q * 4);
                *reinterpret_cast<float4*>(lds_in + p * S + q * 4) = v;
            }
        }
        // ---- stage the weight slab of this chunk (contiguous, pre-packed) ----
        {
            const float4* src = reinterpret_cast<const float4*>(wbase + (size_t)ch * wchunk);
            const int total = wchunk / 4;
            for (int idx = tid; idx < total; idx += 256) reinterpret_cast<float4*>(lds_w)[idx] = src[idx];
        }
        __syncthreads();
        // ---- MFMA over taps x 8-channel groups ----
        for (int t = 0; t < a.ntaps; t++) {
            const int dy = a.tdy[par][t] + 1, dx = a.tdx[par][t] + 1;
#pragma unroll
            for (int g = 0; g < NG; g++) {
                f32x4 af[MS], bf[NS];
#pragma unroll
                for (int m = 0; m < MS; m++) {
                    const int r = wv * MS + m;
                    const int pix = (r * STRIDE + dy) * G::IW + li * STRIDE + dx;
                    af[m] = *reinterpret_cast<const f32x4*>(lds_in + pix * S + g * 8 + half * 4);
                }
#pragma unroll
                for (int n = 0; n < NS; n++)
                    bf[n] = *reinterpret_cast<const f32x4*>(lds_w + (((t * NG + g) * 2 + half) * NT + n * 32 + li) * 4);
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int m = 0; m < MS; m++)
#pragma unroll
                        for (int n = 0; n < NS; n++)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m][s], bf[n][s], acc[m][n], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: D[i][j], j = lane&31 (channel), i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel column) ----
#pragma unroll
    for (int n = 0; n < NS; n++) {
        const int co = ntile * NT + n * 32 + li;
        const bool cok = co < a.Cout;
        const float bias = cok ? a.bias[co] : 0.f;
        const float slope = cok ? a.slope[co] : 1.f;
#pragma unroll
        for (int m = 0; m < MS; m++) {
            const int oy = oy0 + wv * MS + m;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (!cok || oy >= a.Ho || ox >= a.Wo) continue;
                float v = acc[m][n][r] + bias;
                if (EPI == EPI_STORE) {
                    if (a.res) v += a.res[((size_t)oy * a.Wo + ox) * a.res_ld + a.res_coff + co];
                    v = v < 0.f ? v * slope : v;
                    a.out[((size_t)oy * a.Wo + ox) * a.out_ld + a.out_coff + co] = v;
                } else if (EPI == EPI_DECONV || EPI == EPI_DECONV_SIG) {
                    const int py = par >> 1, px = par & 1;
                    if (EPI == EPI_DECONV_SIG) v = 1.f / (1.f + expf(-v));
                    else v = v < 0.f ? v * slope : v;
                    a.out[((size_t)(2 * oy + py) * (2 * a.Wo) + 2 * ox + px) * a.out_ld + a.out_coff + co] = v;
                } else {   // EPI_DECONV_PS: deconv pixel (2oy+py, 2ox+px), channel co -> flow[c = co>>2][.. *2 + i][.. *2 + j]
                    const int py = par >> 1, px = par & 1;
                    const int c = co >> 2, si = (co >> 1) & 1, sj = co & 1;
                    const int fy = 2 * (2 * oy + py) + si, fx = 2 * (2 * ox + px) + sj;
                    a.out[((size_t)fy * (4 * a.Wo) + fx) * a.out_ld + a.out_coff + c] = v;
                }
            }
        }
    }
}

}  // namespace rife
