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
//             lanes 0-31 hold channels g*8+0..3, lanes 32-63 channels g*8+4..7.  The weights are the A operand
//             and the pixels the B operand, so D[row = channel][col = pixel]: a lane ends up with 4 consecutive
//             output channels of one pixel per register quad and the epilogue stores 16 B per lane.
//   pipeline: the global loads of chunk k+1 are issued into registers before the MFMAs of chunk k and
//             written to LDS after them (issue-early / write-late), so HBM/L2 latency hides under the
//             matrix pipe even at one workgroup per CU.
//   block id -> (tile, n-tile/parity): n-tile fastest, and the tile index is remapped so that each XCD
//             (block b runs on XCD b % 8) walks a contiguous band of tiles and re-reads halos from its own L2.
// Layouts: activations NHWC fp32 with explicit channel stride/offset (so producers can write straight into
// concat buffers); weights pre-packed on the host (pack_weights in engine.hip) in exactly the order the
// B-fragment ds_read_b128 wants: [ntile][par][chunk][tap][g][half][n][4].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Bench-only ablation switches: template bits that strip or alter a phase of a kernel for timing experiments (tools/*_bench.py, through
// librife_hip_bench.so = these sources + bench_hooks.h with -DRIFE_HIP_BENCH_BUILD).  In the product build the test is the constant false
// whatever the template argument, so no product kernel carries an ablation path.
#ifdef RIFE_HIP_BENCH_BUILD
#define RIFE_ABL(bits) ((bits) != 0)
#else
#define RIFE_ABL(bits) (false)
#endif

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
    int nz;                // n-tiles x parities
    int tiles_x, ntiles_xy;
    int nsplit = 1;        // split-K: the K chunks are divided among nsplit workgroups per (tile, n-tile) ...
    float* partial = nullptr;   // ... which write raw accumulators to partial[split][pixel][Cpad] (reduced by k_splitk_reduce)
    int cpad = 0;
    // "S16" tensors (conv_t64.h): planes [16-channel chunk][hi | lo] of 32 bytes per pixel, each a zero-bordered image of
    // s16_pitch pixels per row (pixel (y, x) at (y + 1, x + 1)), s16_plane bytes per plane.
    // conv_h2s2_kernel<NS, true> writes one, head_h2_kernel<EPI, true> reads one.
    int s16_pitch = 0;
    unsigned s16_plane = 0;
    // two tensors through one launch (gridDim.y = 2; conv_h2b_kernel, conv_h2s2_kernel): the workgroups with blockIdx.y = 1 read in1 and write out1 (same geometry,
    // strides and weights) - the two ContextNet passes of rife-v2.x (contextnet.param run on (img0, flow01) and (img1, flow10), src/rife.cpp:1027-1060)
    const float* in1 = nullptr;
    float* out1 = nullptr;
    // second destination of the output (conv_h2b_kernel, direct-store epilogue): the U-Net skip connections of the rife-v2.x FusionNet - s0 / s1 / s2 are the input
    // of the next encoder level AND a channel slice of a decoder concat buffer (fusionnet.param:53, 56, 59) - are written twice by the producer instead of copied
    float* out2 = nullptr;
    int out2_ld = 0, out2_coff = 0;
};

template <int STRIDE, int MS, int KS = 3> struct ConvGeom {
    static constexpr int TH = 4 * MS, TW = 32;
    static constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
};

// TAG 5 selects the 5 x 5 (pad 2) geometry of the rife-HD IFNet; every other TAG is a 3 x 3 (pad 1) convolution
template <int TAG> constexpr int conv_ks() { return TAG == 5 ? 5 : 3; }
template <int EPI, int KS = 3> constexpr int conv_ntaps() { return EPI == EPI_STORE ? KS * KS : 4; }

template <int STRIDE, int MS, int NS, int CC, int EPI, int KS = 3>
constexpr int conv_lds_bytes() {
    return (ConvGeom<STRIDE, MS, KS>::IH * ConvGeom<STRIDE, MS, KS>::IW * (CC + 4) + conv_ntaps<EPI, KS>() * CC * NS * 32) * 4;
}

// TAG (other than 5) only gives a layer class its own kernel symbol (so rocprofv3 --stats reports it separately).
template <int STRIDE, int MS, int NS, int CC, int EPI, int TAG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_mfma_kernel(ConvArgs a) {
    constexpr int KS = conv_ks<TAG>();
    using G = ConvGeom<STRIDE, MS, KS>;
    constexpr int S = CC + 4;                 // LDS pixel stride (floats): S/4 odd -> conflict-free b128 column reads
    constexpr int NT = NS * 32;
    constexpr int NG = CC / 8;
    constexpr int NTAPS = conv_ntaps<EPI, KS>();
    constexpr int NPAR = EPI == EPI_STORE ? 1 : 4;
    constexpr int NQ = CC / 4;
    constexpr int IN_F4 = G::IH * G::IW * NQ;              // float4s of one input chunk tile
    constexpr int W_F4 = NTAPS * CC * NT / 4;              // float4s of one weight chunk slab
    constexpr int NIN = (IN_F4 + 255) / 256, NW = (W_F4 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lds_in = lds;                                   // [IH][IW][S]
    float* lds_w = lds + G::IH * G::IW * S;                // [NTAPS][NG][2][NT][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, li = lane & 31;

    // ---- block id -> (tile, z), XCD-aware (bijective for any grid size) ----
    int L;
    {
        const int n = gridDim.x, b = blockIdx.x;
        const int q = n >> 3, r = n & 7, xcd = b & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile = L / a.nz, z = L - tile * a.nz;
    const int par = z % NPAR, ntile = z / NPAR;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * G::TH, ox0 = tx * G::TW;
    const int iy0 = oy0 * STRIDE - KS / 2, ix0 = ox0 * STRIDE - KS / 2;   // top-left of the staged halo tile in input pixels

    // ---- per-thread staging slots (chunk-independent part of the addresses, hoisted out of the K loop) ----
    int goff[NIN];      // global float offset of this thread's k-th input float4 (channel chunk 0); 0 when out of image
    unsigned inside = 0;  // bit k: the k-th float4 lies inside the image (else it is the conv zero padding)
#pragma unroll
    for (int k = 0; k < NIN; k++) {
        const int idx = tid + k * 256;
        const int p = idx / NQ, q = idx - p * NQ;
        const int py = p / G::IW, px = p - py * G::IW;
        const int gy = iy0 + py, gx = ix0 + px;
        const bool ok = idx < IN_F4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[k] = ok ? (gy * a.W + gx) * a.in_ld + a.in_coff + q * 4 : a.in_coff;
        inside |= ok ? (1u << k) : 0u;
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk) + (size_t)z * a.nchunks * W_F4;

    f32x4 rin[NIN], rw[NW];   // native vector type: keeps the staging registers out of scratch
#define RIFE_ISSUE_LOADS(CH)                                                                                  \
    {                                                                                                         \
        _Pragma("unroll") for (int k = 0; k < NIN; k++)                                                       \
            rin[k] = *reinterpret_cast<const f32x4*>(a.in + goff[k] + (CH) * CC);                             \
        _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                      \
            const int idx = tid + k * 256;                                                                    \
            rw[k] = wsrc[(size_t)(CH) * W_F4 + ((W_F4 % 256 == 0 || idx < W_F4) ? idx : 0)];                  \
        }                                                                                                     \
    }
#define RIFE_WRITE_LDS()                                                                                      \
    {                                                                                                         \
        _Pragma("unroll") for (int k = 0; k < NIN; k++) {                                                     \
            const int idx = tid + k * 256;                                                                    \
            const int p = idx / NQ, q = idx - p * NQ;                                                         \
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};                                                         \
            const f32x4 v = ((inside >> k) & 1u) ? rin[k] : zero4;                                            \
            if (idx < IN_F4) *reinterpret_cast<f32x4*>(lds_in + p * S + q * 4) = v;                           \
        }                                                                                                     \
        _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                      \
            const int idx = tid + k * 256;                                                                    \
            if (W_F4 % 256 == 0 || idx < W_F4) reinterpret_cast<f32x4*>(lds_w)[idx] = rw[k];                  \
        }                                                                                                     \
    }

    f32x16 acc[MS][NS];
#pragma unroll
    for (int m = 0; m < MS; m++)
#pragma unroll
        for (int n = 0; n < NS; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[m][n][r] = 0.f;

    // A-fragment base (tap (0,0), group 0) and B-fragment base for this lane
    const float* abase = lds_in + ((wv * MS * STRIDE) * G::IW + li * STRIDE) * S + half * 4;
    const float* bbase = lds_w + (half * NT + li) * 4;

    RIFE_ISSUE_LOADS(0)
    RIFE_WRITE_LDS()
    __syncthreads();

    for (int ch = 0; ch < a.nchunks; ch++) {
        const bool more = ch + 1 < a.nchunks;
        if (more) RIFE_ISSUE_LOADS(ch + 1)                // in flight while the matrix pipe works on chunk ch
#pragma unroll
        for (int t = 0; t < NTAPS; t++) {
            int dy, dx;                                   // tap offset + 1 (halo origin)
            if (EPI == EPI_STORE) { dy = t / KS; dx = t % KS; }
            else {   // deconv parity p, tap bit: 0 -> d = 0; 1 -> d = (p ? +1 : -1)   (see configure() in engine.hip)
                dy = 1 + ((t >> 1) ? ((par >> 1) ? 1 : -1) : 0);
                dx = 1 + ((t & 1) ? ((par & 1) ? 1 : -1) : 0);
            }
#pragma unroll
            for (int g = 0; g < NG; g++) {
                f32x4 af[MS], bf[NS];
#pragma unroll
                for (int m = 0; m < MS; m++)
                    af[m] = *reinterpret_cast<const f32x4*>(abase + ((m * STRIDE + dy) * G::IW + dx) * S + g * 8);
#pragma unroll
                for (int n = 0; n < NS; n++)
                    bf[n] = *reinterpret_cast<const f32x4*>(bbase + ((t * NG + g) * 2 * NT + n * 32) * 4);
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int m = 0; m < MS; m++)
#pragma unroll
                        for (int n = 0; n < NS; n++)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[n][s], af[m][s], acc[m][n], 0, 0, 0);
            }
        }
        if (more) {
            __syncthreads();                               // every wave is done reading chunk ch
            RIFE_WRITE_LDS()
            __syncthreads();
        }
    }
#undef RIFE_ISSUE_LOADS
#undef RIFE_WRITE_LDS

    // ---- epilogue: D[i][j], j = lane&31 = pixel column, i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = channel within the 32-wide
    // sub-tile: register quad q holds channels 8q + 4*half + 0..3 of pixel (oy, ox0 + li) -> one 16-byte store per quad ----
    const int py = par >> 1, px = par & 1;
    const int ox = ox0 + li;
#pragma unroll
    for (int n = 0; n < NS; n++) {
#pragma unroll
        for (int m = 0; m < MS; m++) {
            const int oy = oy0 + wv * MS + m;
            const bool pok = oy < a.Ho && ox < a.Wo;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c0 = ntile * NT + n * 32 + 8 * q + 4 * half;
                const bool ok = pok && c0 < a.Cout;
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c0);      // bias / slope arrays are padded to the N-tile
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(a.slope + c0);
                f32x4 v;
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = acc[m][n][4 * q + k] + b4[k];
                if (EPI == EPI_STORE) {
                    if (a.res != nullptr) {
                        const f32x4 r4 = *reinterpret_cast<const f32x4*>(a.res + (ok ? ((size_t)oy * a.Wo + ox) * a.res_ld + a.res_coff + c0 : 0));
#pragma unroll
                        for (int k = 0; k < 4; k++) v[k] += r4[k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] = v[k] < 0.f ? v[k] * s4[k] : v[k];
                    if (ok) *reinterpret_cast<f32x4*>(a.out + ((size_t)oy * a.Wo + ox) * a.out_ld + a.out_coff + c0) = v;
                } else if (EPI == EPI_DECONV || EPI == EPI_DECONV_SIG) {
#pragma unroll
                    for (int k = 0; k < 4; k++) v[k] = EPI == EPI_DECONV_SIG ? 1.f / (1.f + expf(-v[k])) : (v[k] < 0.f ? v[k] * s4[k] : v[k]);
                    if (ok) *reinterpret_cast<f32x4*>(a.out + ((size_t)(2 * oy + py) * (2 * a.Wo) + 2 * ox + px) * a.out_ld + a.out_coff + c0) = v;
                } else {   // EPI_DECONV_PS: deconv pixel (2oy+py, 2ox+px), channels c0..c0+3 = PixelShuffle group c0>>2 -> a 2x2 block of the flow tensor
                    const int c = c0 >> 2;
                    const int fy = 2 * (2 * oy + py), fx = 2 * (2 * ox + px);
                    if (ok) {
#pragma unroll
                        for (int k = 0; k < 4; k++) a.out[((size_t)(fy + (k >> 1)) * (4 * a.Wo) + fx + (k & 1)) * a.out_ld + a.out_coff + c] = v[k];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// conv_mfma8_kernel: the trunk variant (3x3, stride 1, EPI_STORE) for layers with plenty of tiles.
//   512 threads = 8 waves, tile = 8 rows x 32 columns x NT channels, wave w owns row w (NS accumulators);
//   two LDS buffers: while the matrix pipe works on chunk k from buffer k&1, the registers holding chunk k+1 are
//   written to the other buffer in the middle of the tap loop and the global loads of chunk k+2 are issued right
//   after — one barrier per chunk, no MFMA-idle staging phase, loads get a full chunk of MFMA time to land.
//   CC = 8 keeps both buffers of a 64-wide N-tile at 69.5 KB, so two workgroups (16 waves) share a CU and cover
//   each other's prologue / epilogue.
// ------------------------------------------------------------------------------------------------------------
template <int NS, int CC>
constexpr int conv8_lds_bytes() { return 2 * (10 * 34 * (CC + 4) + 9 * CC * NS * 32) * 4; }

template <int NS, int CC, int WPE, int TAG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void conv_mfma8_kernel(ConvArgs a) {
    constexpr int IH = 10, IW = 34;
    constexpr int S = CC + 4, NT = NS * 32, NG = CC / 8, NQ = CC / 4;
    constexpr int IN_F4 = IH * IW * NQ, W_F4 = 9 * CC * NT / 4;
    constexpr int NIN = (IN_F4 + 511) / 512, NW = (W_F4 + 511) / 512;
    constexpr int BUF = IH * IW * S + 9 * CC * NT;          // floats per LDS buffer
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, li = lane & 31;
    int L;
    {
        const int n = gridDim.x, b = blockIdx.x;
        const int q = n >> 3, r = n & 7, xcd = b & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile = L / a.nz, ntile = L - tile * a.nz;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * 8, ox0 = tx * 32;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;

    int goff[NIN];
    unsigned inside = 0;
#pragma unroll
    for (int k = 0; k < NIN; k++) {
        const int idx = tid + k * 512;
        const int p = idx / NQ, q = idx - p * NQ;
        const int py = p / IW, px = p - py * IW;
        const int gy = iy0 + py, gx = ix0 + px;
        const bool ok = idx < IN_F4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[k] = ok ? (gy * a.W + gx) * a.in_ld + a.in_coff + q * 4 : a.in_coff;
        inside |= ok ? (1u << k) : 0u;
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk) + (size_t)ntile * a.nchunks * W_F4;

    f32x4 rin[NIN], rw[NW];
#define RIFE8_ISSUE(CH)                                                                                     \
    {                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < NIN; k++)                                                     \
            rin[k] = *reinterpret_cast<const f32x4*>(a.in + goff[k] + (CH) * CC);                           \
        _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                    \
            const int idx = tid + k * 512;                                                                  \
            rw[k] = wsrc[(size_t)(CH) * W_F4 + ((W_F4 % 512 == 0 || idx < W_F4) ? idx : 0)];                \
        }                                                                                                   \
    }
#define RIFE8_WRITE(BUFP)                                                                                   \
    {                                                                                                       \
        float* lin_ = (BUFP); float* lw_ = (BUFP) + IH * IW * S;                                            \
        _Pragma("unroll") for (int k = 0; k < NIN; k++) {                                                   \
            const int idx = tid + k * 512;                                                                  \
            const int p = idx / NQ, q = idx - p * NQ;                                                       \
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};                                                       \
            const f32x4 v = ((inside >> k) & 1u) ? rin[k] : zero4;                                          \
            if (IN_F4 % 512 == 0 || idx < IN_F4) *reinterpret_cast<f32x4*>(lin_ + p * S + q * 4) = v;       \
        }                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                    \
            const int idx = tid + k * 512;                                                                  \
            if (W_F4 % 512 == 0 || idx < W_F4) reinterpret_cast<f32x4*>(lw_)[idx] = rw[k];                  \
        }                                                                                                   \
    }
#define RIFE8_TAPS(BUFP, T0, T1)                                                                            \
    {                                                                                                       \
        const float* ab_ = (BUFP) + (wv * IW + li) * S + half * 4;                                          \
        const float* bb_ = (BUFP) + IH * IW * S + (half * NT + li) * 4;                                     \
        _Pragma("unroll") for (int t = (T0); t < (T1); t++) {                                               \
            const int dy = t / 3, dx = t % 3;                                                               \
            _Pragma("unroll") for (int g = 0; g < NG; g++) {                                                \
                const f32x4 af = *reinterpret_cast<const f32x4*>(ab_ + (dy * IW + dx) * S + g * 8);         \
                f32x4 bf[NS];                                                                               \
                _Pragma("unroll") for (int n = 0; n < NS; n++)                                              \
                    bf[n] = *reinterpret_cast<const f32x4*>(bb_ + ((t * NG + g) * 2 * NT + n * 32) * 4);    \
                _Pragma("unroll") for (int s4 = 0; s4 < 4; s4++)                                            \
                    _Pragma("unroll") for (int n = 0; n < NS; n++)                                          \
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[n][s4], af[s4], acc[n], 0, 0, 0);  \
            }                                                                                               \
        }                                                                                                   \
    }

    f32x16 acc[NS];
#pragma unroll
    for (int n = 0; n < NS; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;

    RIFE8_ISSUE(0)
    RIFE8_WRITE(lds)
    if (a.nchunks > 1) RIFE8_ISSUE(1)
    __syncthreads();

    for (int ch = 0; ch < a.nchunks; ch++) {
        float* cur = lds + (ch & 1) * BUF;
        float* oth = lds + ((ch & 1) ^ 1) * BUF;
        RIFE8_TAPS(cur, 0, 4)
        if (ch + 1 < a.nchunks) RIFE8_WRITE(oth)            // chunk ch+1: loaded during the previous chunk
        if (!RIFE_ABL(TAG & 512) && ch + 2 < a.nchunks) RIFE8_ISSUE(ch + 2)
        RIFE8_TAPS(cur, 4, 9)
        if (!RIFE_ABL(TAG & 1024)) __syncthreads();                 // chunk ch+1 visible; everyone is done with `cur`
    }
#undef RIFE8_ISSUE
#undef RIFE8_WRITE
#undef RIFE8_TAPS

    const int oy = oy0 + wv, ox = ox0 + li;
    const bool pok = oy < a.Ho && ox < a.Wo;
#pragma unroll
    for (int n = 0; n < NS; n++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c0 = ntile * NT + n * 32 + 8 * q + 4 * half;
            const bool ok = pok && c0 < a.Cout;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c0);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(a.slope + c0);
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = acc[n][4 * q + k] + b4[k];
            if (a.res != nullptr) {
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(a.res + (ok ? ((size_t)oy * a.Wo + ox) * a.res_ld + a.res_coff + c0 : 0));
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] += r4[k];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = v[k] < 0.f ? v[k] * s4[k] : v[k];
            if RIFE_ABL(TAG & 256) { if (v[0] == 123.456f) a.out[0] = v[1]; }   // ablation: keep the value alive, store (almost) never
            else if (ok) *reinterpret_cast<f32x4*>(a.out + ((size_t)oy * a.Wo + ox) * a.out_ld + a.out_coff + c0) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// conv_h2_kernel: trunk 3x3 stride-1 convolution on the f16 matrix pipe at fp32 accuracy ("split-f16").
//   * weights are fp16 on disk (ncnn tag 0x01306B47), so they enter the MFMA exactly;
//   * every fp32 activation a is split while it is staged into LDS:  hi = f16(a),  lo = f16(a - hi)
//     => a == hi + lo up to max(2^-22 |a|, 3e-8): lo is allowed to be an f16 subnormal — the gfx950 matrix pipe keeps f16
//     subnormal inputs (hipcc's default denorm mode; asserted by tests/test_gpu_kernels.py::test_f16_mfma_keeps_subnormals);
//   * two v_mfma_f32_32x32x16_f16 per 16-channel k-step accumulate W*hi and W*lo into the same fp32 accumulator.
//     Products are exact in fp32 (11 x 11 bit significands), so the result differs from the fp32 kernel only by
//     summation order and the 2^-22 split error.
//   * optional tap 9 = centre pixel with identity weights: the residual skip (x + conv(x)) inside the GEMM.
//   One k-step covers 16 channels: lanes 0-31 supply channels 0-7, lanes 32-63 channels 8-15 (8 f16 = 16 B per lane).
//   LDS pixel record = 32 B hi + 32 B lo + 16 B pad (80 B = 5 slots: conflict-free ds_read_b128 columns).
//   8 waves, tile 8 rows x 32 columns x 32*NS channels, chunk = 16 channels, two LDS buffers, loads issued two
//   chunks ahead into two alternating register sets (the matrix work per chunk is only ~1 us).
// ------------------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int NS, int NTAPS>
constexpr int convh2_lds_bytes() { return 2 * (10 * 34 * 80 + NTAPS * 2 * NS * 32 * 16) + 2 * NS * 32 * 4; }      // + bias and slopes (see conv_h2b_kernel)

template <int NS, int NTAPS, int TAG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_h2_kernel(ConvArgs a) {
    constexpr int IH = 10, IW = 34, CC = 16, NT = NS * 32;
    constexpr int PIXB = 80;                                   // bytes per pixel record in LDS
    constexpr int IN_F4 = IH * IW * 4;                         // float4 (4-channel) slots of one input chunk tile
    constexpr int W_16 = NTAPS * 2 * NT;                       // 16-byte units of one weight chunk slab
    constexpr int NIN = (IN_F4 + 511) / 512, NW = (W_16 + 511) / 512;
    constexpr int BUFB = IH * IW * PIXB + W_16 * 16;           // bytes per LDS buffer
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, li = lane & 31;
    const float* const tin = blockIdx.y ? a.in1 : a.in;                // gridDim.y = 2: the second tensor pair of the launch
    float* const tout = blockIdx.y ? a.out1 : a.out;
    int L;
    {
        const int n = gridDim.x, b = blockIdx.x;
        const int q = n >> 3, r = n & 7, xcd = b & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile = L / a.nz, ntile = L - tile * a.nz;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * 8, ox0 = tx * 32;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;

    int goff[NIN];
    unsigned inside = 0;
#pragma unroll
    for (int k = 0; k < NIN; k++) {
        const int idx = tid + k * 512;
        const int p = idx >> 2, q = idx & 3;
        const int py = p / IW, px = p - py * IW;
        const int gy = iy0 + py, gx = ix0 + px;
        const bool ok = idx < IN_F4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[k] = ok ? (gy * a.W + gx) * a.in_ld + a.in_coff + q * 4 : a.in_coff;
        inside |= ok ? (1u << k) : 0u;
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk) + (size_t)ntile * a.nchunks * W_16;   // 16-byte units (8 f16)

    f32x4 rinA[NIN], rinB[NIN], rwA[NW], rwB[NW];
#define H2_ISSUE(RIN, RW, CH)                                                                               \
    {                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < NIN; k++)                                                     \
            RIN[k] = *reinterpret_cast<const f32x4*>(tin + goff[k] + (CH) * CC);                           \
        _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                    \
            const int idx = tid + k * 512;                                                                  \
            RW[k] = wsrc[(size_t)(CH) * W_16 + ((W_16 % 512 == 0 || idx < W_16) ? idx : 0)];                \
        }                                                                                                   \
    }
#define H2_WRITE(RIN, RW, BUFP)                                                                             \
    {                                                                                                       \
        unsigned char* lin_ = (BUFP); unsigned char* lw_ = (BUFP) + IH * IW * PIXB;                         \
        _Pragma("unroll") for (int k = 0; k < NIN; k++) {                                                   \
            const int idx = tid + k * 512;                                                                  \
            const int p = idx >> 2, q = idx & 3;                                                            \
            f16x4 hi4, lo4;                                                                                 \
            _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                 \
                const float v = ((inside >> k) & 1u) ? RIN[k][e] : 0.f;                                     \
                const _Float16 h = (_Float16)v;                                                             \
                hi4[e] = h;                                                                                 \
                lo4[e] = (_Float16)(v - (float)h);                                                          \
            }                                                                                               \
            if (IN_F4 % 512 == 0 || idx < IN_F4) {                                                          \
                *reinterpret_cast<f16x4*>(lin_ + p * PIXB + q * 8) = hi4;                                   \
                *reinterpret_cast<f16x4*>(lin_ + p * PIXB + 32 + q * 8) = lo4;                              \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                    \
            const int idx = tid + k * 512;                                                                  \
            if (W_16 % 512 == 0 || idx < W_16) reinterpret_cast<f32x4*>(lw_)[idx] = RW[k];                  \
        }                                                                                                   \
    }
#define H2_TAPS(BUFP, T0, T1)                                                                               \
    {                                                                                                       \
        const unsigned char* ab_ = (BUFP) + (wv * IW + li) * PIXB + half * 16;                              \
        const unsigned char* bb_ = (BUFP) + IH * IW * PIXB + (half * NT + li) * 16;                         \
        _Pragma("unroll") for (int t = (T0); t < (T1); t++) {                                               \
            const int dy = t == 9 ? 1 : t / 3, dx = t == 9 ? 1 : t % 3;                                     \
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ab_ + (dy * IW + dx) * PIXB);                  \
            const f16x8 al = *reinterpret_cast<const f16x8*>(ab_ + (dy * IW + dx) * PIXB + 32);             \
            f16x8 bw[NS];                                                                                   \
            _Pragma("unroll") for (int n = 0; n < NS; n++)                                                  \
                bw[n] = *reinterpret_cast<const f16x8*>(bb_ + (t * 2 * NT + n * 32) * 16);                  \
            _Pragma("unroll") for (int n = 0; n < NS; n++)                                                  \
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[n], ah, acc[n], 0, 0, 0);                \
            _Pragma("unroll") for (int n = 0; n < NS; n++)                                                  \
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[n], al, acc[n], 0, 0, 0);                \
        }                                                                                                   \
    }

    f32x16 acc[NS];
#pragma unroll
    for (int n = 0; n < NS; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;

    const int nch = a.nchunks;
    unsigned char* buf0 = ldsb; unsigned char* buf1 = ldsb + BUFB;
    H2_ISSUE(rinA, rwA, 0)
    if (nch > 1) H2_ISSUE(rinB, rwB, 1)
    float* const lbs = reinterpret_cast<float*>(ldsb + 2 * BUFB);      // bias and slopes of the N-tile, fetched while the first chunks are in flight
    if (tid < NT) { lbs[tid] = a.bias[ntile * NT + tid]; lbs[NT + tid] = a.slope[ntile * NT + tid]; }
    H2_WRITE(rinA, rwA, buf0)
    if (nch > 2) H2_ISSUE(rinA, rwA, 2)
    __syncthreads();
    // two chunks per trip so that the register sets are named statically: even chunks live in A / buf0, odd in B / buf1
    for (int ch = 0; ch < nch; ch += 2) {
        // chunk ch (buf0); registers B hold chunk ch+1, registers A hold chunk ch+2 (in flight)
        H2_TAPS(buf0, 0, NTAPS / 2)
        if (ch + 1 < nch) H2_WRITE(rinB, rwB, buf1)
        if (ch + 3 < nch) H2_ISSUE(rinB, rwB, ch + 3)
        H2_TAPS(buf0, NTAPS / 2, NTAPS)
        __syncthreads();
        if (ch + 1 < nch) {
            // chunk ch+1 (buf1); registers A hold chunk ch+2, registers B hold chunk ch+3 (in flight)
            H2_TAPS(buf1, 0, NTAPS / 2)
            if (ch + 2 < nch) H2_WRITE(rinA, rwA, buf0)
            if (ch + 4 < nch) H2_ISSUE(rinA, rwA, ch + 4)
            H2_TAPS(buf1, NTAPS / 2, NTAPS)
            __syncthreads();
        }
    }
#undef H2_ISSUE
#undef H2_WRITE
#undef H2_TAPS

    // epilogue: like conv_h2b_kernel, the wave's 32 x NT tile goes through LDS (both staging buffers are free after the last
    // barrier of the K loop) so that consecutive lanes store consecutive 16-byte chunks of a pixel; direct stores otherwise
    const int oy = oy0 + wv, ox = ox0 + li;
    const bool pok = oy < a.Ho && ox < a.Wo;
    constexpr int ROWF = NT + 4;
    const bool via_lds = a.res == nullptr && a.out_ld == NT && a.Cout == NT && a.nz == 1;
    float* const tl = reinterpret_cast<float*>(ldsb) + wv * 32 * ROWF;
#pragma unroll
    for (int n = 0; n < NS; n++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c0 = ntile * NT + n * 32 + 8 * q + 4 * half;
            const bool ok = pok && c0 < a.Cout;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbs + n * 32 + 8 * q + 4 * half);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(lbs + NT + n * 32 + 8 * q + 4 * half);
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = acc[n][4 * q + k] + b4[k];
            if (a.res != nullptr) {
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(a.res + (ok ? ((size_t)oy * a.Wo + ox) * a.res_ld + a.res_coff + c0 : 0));
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] += r4[k];
            }
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = v[k] < 0.f ? v[k] * s4[k] : v[k];
            if (via_lds) *reinterpret_cast<f32x4*>(tl + li * ROWF + n * 32 + 8 * q + 4 * half) = v;
            else if (ok) *reinterpret_cast<f32x4*>(tout + ((size_t)oy * a.Wo + ox) * a.out_ld + a.out_coff + c0) = v;
        }
    }
    if (via_lds) {
        // 8 lanes per pixel and 32-channel sub-tile: every store instruction writes eight full 128-byte segments
        const int pl = lane >> 3, chunk = lane & 7;
        float* const orow = tout + ((size_t)oy * a.Wo + ox0) * a.out_ld + a.out_coff;
#pragma unroll
        for (int n = 0; n < NS; n++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int px = j * 8 + pl;
                const f32x4 v = *reinterpret_cast<const f32x4*>(tl + px * ROWF + n * 32 + chunk * 4);
                if (oy < a.Ho && ox0 + px < a.Wo) *reinterpret_cast<f32x4*>(orow + (size_t)px * a.out_ld + n * 32 + chunk * 4) = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
// conv_h2b_kernel: conv_h2_kernel re-balanced for two workgroups per CU (16 waves; ROWS = 4: three workgroups of 4 waves, for
// layers whose 8-row tiles would leave most CUs empty): the input tile is double
// buffered, the weight slab is not (2 x 27.2 KB + 20.5 KB = 75 KB), so a second workgroup's matrix work covers this
// one's load latency, barriers, prologue and epilogue.  One register set; the input prefetch of chunk k+2 is issued
// as soon as chunk k+1 has been written to LDS (mid-chunk), the weight prefetch right after the weight slab swap.
// ------------------------------------------------------------------------------------------------------------
template <int NS, int NTAPS, int ROWS = 8>
constexpr int convh2b_lds_bytes() { return 2 * (ROWS + 2) * 34 * 80 + NTAPS * 2 * NS * 32 * 16 + 2 * NS * 32 * 4; }      // + bias and slopes of the N-tile

template <int NS, int NTAPS, int TAG, int ROWS = 8>
__global__ __launch_bounds__(ROWS * 64) __attribute__((amdgpu_waves_per_eu(NS == 3 ? 2 : 4, NS == 3 ? 2 : 4))) void conv_h2b_kernel(ConvArgs a) {   // NS = 3: LDS allows 8 waves per CU anyway
    constexpr int IH = ROWS + 2, IW = 34, CC = 16, NT = NS * 32;
    constexpr int NTHR = ROWS * 64;                            // one wave per output row of the tile
    constexpr int PIXB = 80;
    constexpr int IN_F4 = IH * IW * 4;
    constexpr int W_16 = NTAPS * 2 * NT;
    constexpr int NIN = (IN_F4 + NTHR - 1) / NTHR, NW = (W_16 + NTHR - 1) / NTHR;
    constexpr int INB = IH * IW * PIXB;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* const lw = ldsb + 2 * INB;
    // bias and activation slopes of this N-tile wait in LDS from the prologue on: fetched from global memory in the epilogue they cost
    // every wave a full memory round trip (~4 us of a 22 us workgroup under load, tools/h2b_phase_trace.py) right when nothing else runs
    float* const lbs = reinterpret_cast<float*>(lw + W_16 * 16);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const float* const tin = blockIdx.y ? a.in1 : a.in;                // gridDim.y = 2: the second tensor pair of the launch
    float* const tout = blockIdx.y ? a.out1 : a.out;
    // bench-only RIFE_ABL(TAG & 32768): wave 0 of every workgroup records the shader clock at its phase boundaries into a.partial
    // ([workgroup][16] 64-bit slots: 0 start, 8 index math done, 9 first loads issued, 10 first chunk in LDS, 1 prologue barrier passed,
    // 2..5 one per chunk, 6 epilogue barrier passed, 7 tile in LDS, 13 stores issued, 14 stores done, 15 HW_ID | XCC_ID << 32)
#define H2B_STAMP(SLOT)                                                                                      \
    if (RIFE_ABL(TAG & 32768) && tid == 0) reinterpret_cast<long long*>(a.partial)[(size_t)blockIdx.x * 16 + (SLOT)] = (long long)__builtin_readcyclecounter();
    H2B_STAMP(0)
    if (RIFE_ABL(TAG & 32768) && tid == 0)
        reinterpret_cast<long long*>(a.partial)[(size_t)blockIdx.x * 16 + 15] =
            (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
    const int half = lane >> 5, li = lane & 31;
    int L;
    {
        const int n = gridDim.x, b = blockIdx.x;
        const int q = n >> 3, r = n & 7, xcd = b & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int split = L % a.nsplit;                    // split-K slice of this workgroup (fastest index: the slices of a tile share an XCD)
    const int L2 = L / a.nsplit;
    const int tile = L2 / a.nz, ntile = L2 - tile * a.nz;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * ROWS, ox0 = tx * 32;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const int chunks_per = (a.nchunks + a.nsplit - 1) / a.nsplit;
    const int chb = split * chunks_per;                // first chunk of this slice
    const int nch = min(chunks_per, a.nchunks - chb);

    int goff[NIN];
    unsigned inside = 0;
#pragma unroll
    for (int k = 0; k < NIN; k++) {
        const int idx = tid + k * NTHR;
        const int p = idx >> 2, q = idx & 3;
        const int py = p / IW, px = p - py * IW;
        const int gy = iy0 + py, gx = ix0 + px;
        const bool ok = idx < IN_F4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[k] = (ok ? (gy * a.W + gx) * a.in_ld + a.in_coff + q * 4 : a.in_coff) + chb * CC;
        inside |= ok ? (1u << k) : 0u;
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk) + ((size_t)ntile * a.nchunks + chb) * W_16;

    // two input register sets, one per chunk parity: the loads of chunk k + 3 are issued as soon as chunk k + 1 has left its set
    // (mid-chunk k), i.e. two chunks before they are needed - one chunk ahead was not enough for the memory latency under load
    // (tools/h2b_phase_trace.py: chunks 0..2 took 4.3 / 3.2 / 2.5 us against 1.6 us for the last one, which waits for nothing)
    f32x4 rinA[NIN], rinB[NIN], rw[NW];
#define H2B_ISSUE_IN(CH, RIN)                                                                               \
    _Pragma("unroll") for (int k = 0; k < NIN; k++) RIN[k] = *reinterpret_cast<const f32x4*>(tin + goff[k] + (CH) * CC);
#define H2B_ISSUE_W(CH)                                                                                     \
    _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                        \
        const int idx = tid + k * NTHR;                                                                      \
        rw[k] = wsrc[(size_t)(CH) * W_16 + ((W_16 % NTHR == 0 || idx < W_16) ? idx : 0)];                    \
    }
#define H2B_WRITE_IN(BUFP, RIN)                                                                                \
    _Pragma("unroll") for (int k = 0; k < NIN; k++) {                                                       \
        const int idx = tid + k * NTHR;                                                                      \
        const int p = idx >> 2, q = idx & 3;                                                                \
        f16x4 hi4, lo4;                                                                                     \
        _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                     \
            const float v = ((inside >> k) & 1u) ? RIN[k][e] : 0.f;                                         \
            const _Float16 h = (_Float16)v;                                                                 \
            hi4[e] = h;                                                                                     \
            lo4[e] = (_Float16)(v - (float)h);                                                              \
        }                                                                                                   \
        if (IN_F4 % NTHR == 0 || idx < IN_F4) {                                                              \
            *reinterpret_cast<f16x4*>((BUFP) + p * PIXB + q * 8) = hi4;                                     \
            *reinterpret_cast<f16x4*>((BUFP) + p * PIXB + 32 + q * 8) = lo4;                                \
        }                                                                                                   \
    }
#define H2B_WRITE_W()                                                                                       \
    _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                        \
        const int idx = tid + k * NTHR;                                                                      \
        if (W_16 % NTHR == 0 || idx < W_16) reinterpret_cast<f32x4*>(lw)[idx] = rw[k];                       \
    }
#define H2B_TAPS(BUFP, T0, T1)                                                                              \
    {                                                                                                       \
        const unsigned char* ab_ = (BUFP) + (wv * IW + li) * PIXB + half * 16;                              \
        const unsigned char* bb_ = lw + (half * NT + li) * 16;                                              \
        _Pragma("unroll") for (int t = (T0); t < (T1); t++) {                                               \
            const int dy = t == 9 ? 1 : t / 3, dx = t == 9 ? 1 : t % 3;                                     \
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ab_ + (dy * IW + dx) * PIXB);                  \
            const f16x8 al = *reinterpret_cast<const f16x8*>(ab_ + (dy * IW + dx) * PIXB + 32);             \
            f16x8 bw[NS];                                                                                   \
            _Pragma("unroll") for (int n = 0; n < NS; n++)                                                  \
                bw[n] = *reinterpret_cast<const f16x8*>(bb_ + (t * 2 * NT + n * 32) * 16);                  \
            _Pragma("unroll") for (int n = 0; n < NS; n++)                                                  \
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[n], ah, acc[n], 0, 0, 0);                \
            _Pragma("unroll") for (int n = 0; n < NS; n++)                                                  \
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[n], al, acc[n], 0, 0, 0);                \
        }                                                                                                   \
    }

    f32x16 acc[NS];
#pragma unroll
    for (int n = 0; n < NS; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;

    H2B_STAMP(8)
    H2B_ISSUE_IN(0, rinA)
    H2B_ISSUE_W(0)
    if (nch > 1) { H2B_ISSUE_IN(1, rinB) }
    H2B_STAMP(9)
    if (tid < NT) { lbs[tid] = a.bias[ntile * NT + tid]; lbs[NT + tid] = a.slope[ntile * NT + tid]; }
    H2B_WRITE_IN(ldsb, rinA)
    H2B_WRITE_W()
    H2B_STAMP(10)
    if (nch > 1) { H2B_ISSUE_W(1) }
    if (!RIFE_ABL(TAG & 512) && nch > 2) { H2B_ISSUE_IN(2, rinA) }
    __syncthreads();
    H2B_STAMP(1)
    // one chunk: RIN is the register set that holds chunk CH + 1 on entry and receives chunk CH + 3
#define H2B_CHUNK(CH, RIN)                                                                                  \
    {                                                                                                       \
        unsigned char* cur = ldsb + ((CH) & 1) * INB;                                                       \
        unsigned char* oth = ldsb + (((CH) & 1) ^ 1) * INB;                                                 \
        H2B_TAPS(cur, 0, NTAPS / 2)                                                                         \
        if (!RIFE_ABL(TAG & 2048) && (CH) + 1 < nch) { H2B_WRITE_IN(oth, RIN) }                                     \
        if (!RIFE_ABL(TAG & 512) && (CH) + 3 < nch) { H2B_ISSUE_IN((CH) + 3, RIN) }                                 \
        H2B_TAPS(cur, NTAPS / 2, NTAPS)                                                                     \
        if ((CH) + 1 < nch) {                                                                               \
            if (!RIFE_ABL(TAG & 1024)) __syncthreads();                 /* everyone is done with the weight slab of chunk CH */ \
            if (!RIFE_ABL(TAG & 2048)) { H2B_WRITE_W() }                                                            \
            if (!RIFE_ABL(TAG & 512) && (CH) + 2 < nch) { H2B_ISSUE_W((CH) + 2) }                                   \
            if (!RIFE_ABL(TAG & 1024)) __syncthreads();                                                             \
        }                                                                                                   \
        if ((CH) < 11) { H2B_STAMP(2 + (CH)) }                                                              \
    }
    for (int ch = 0; ch < nch; ch += 2) {
        H2B_CHUNK(ch, rinB)
        if (ch + 1 < nch) H2B_CHUNK(ch + 1, rinA)
    }
#undef H2B_CHUNK
#undef H2B_ISSUE_IN
#undef H2B_ISSUE_W
#undef H2B_WRITE_IN
#undef H2B_WRITE_W
#undef H2B_TAPS

    // ---- epilogue.  Direct path (split-K partials, ablations, ragged channel counts): one 16-byte store per register quad,
    // i.e. 32 B per pixel per instruction.  Normal path: each wave transposes its 32 x NT tile through LDS (the staging buffers
    // are free by now) so that consecutive lanes store consecutive 16-byte chunks: 1 KB of contiguous memory per instruction.
    const int oy = oy0 + wv, ox = ox0 + li;
    const bool pok = oy < a.Ho && ox < a.Wo;
    constexpr int ROWF = NT + 4;                               // floats per pixel row of the transpose tile (odd number of 16-B slots)
    const bool via_lds = a.nsplit == 1 && !RIFE_ABL(TAG & 256) && a.out_ld == NT && a.Cout == NT && a.nz == 1 && a.out2 == nullptr;
    if (via_lds) __syncthreads();                              // every wave is done reading the staging buffers
    H2B_STAMP(6)
    float* const tl = reinterpret_cast<float*>(ldsb) + wv * 32 * ROWF;
#pragma unroll
    for (int n = 0; n < NS; n++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c0 = ntile * NT + n * 32 + 8 * q + 4 * half;
            const bool ok = pok && c0 < a.Cout;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbs + n * 32 + 8 * q + 4 * half);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(lbs + NT + n * 32 + 8 * q + 4 * half);
            f32x4 v;
            if (a.nsplit > 1) {                             // raw partial sums; bias / activation happen in k_splitk_reduce
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = acc[n][4 * q + k];
                if (ok) *reinterpret_cast<f32x4*>(a.partial + ((size_t)split * a.Ho * a.Wo + (size_t)oy * a.Wo + ox) * a.cpad + c0) = v;
                continue;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = acc[n][4 * q + k] + b4[k];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = v[k] < 0.f ? v[k] * s4[k] : v[k];
            if (via_lds) *reinterpret_cast<f32x4*>(tl + li * ROWF + n * 32 + 8 * q + 4 * half) = v;
            else if RIFE_ABL(TAG & 256) { if (v[0] == 123.456f) tout[0] = v[1]; }   // ablation: keep alive, (almost) never store
            else if (ok) {
                *reinterpret_cast<f32x4*>(tout + ((size_t)oy * a.Wo + ox) * a.out_ld + a.out_coff + c0) = v;
                if (a.out2) *reinterpret_cast<f32x4*>(a.out2 + ((size_t)oy * a.Wo + ox) * a.out2_ld + a.out2_coff + c0) = v;
            }
        }
    }
    H2B_STAMP(7)
    if (via_lds) {
        constexpr int LPP = NT / 4;                            // lanes (16-byte chunks) per pixel
        constexpr int PPI = 64 / LPP;                          // pixels per store instruction
        const int pl = lane / LPP, chunk = lane - pl * LPP;
        float* const orow = tout + ((size_t)oy * a.Wo + ox0) * a.out_ld + a.out_coff + chunk * 4;
#pragma unroll
        for (int j = 0; j < 32 / PPI; j++) {
            const int px = j * PPI + pl;
            const f32x4 v = *reinterpret_cast<const f32x4*>(tl + px * ROWF + chunk * 4);
            if ((64 % LPP == 0 || pl < PPI) && oy < a.Ho && ox0 + px < a.Wo) *reinterpret_cast<f32x4*>(orow + (size_t)px * a.out_ld) = v;      // NT = 96: lanes 48-63 idle
        }
    }
    H2B_STAMP(13)
    if RIFE_ABL(TAG & 32768) { __builtin_amdgcn_s_waitcnt(0); H2B_STAMP(14) }
#undef H2B_STAMP
}


// ---- S16 tensor stores (conv_t64.h) ---------------------------------------------------------------------------------------
// v[16]: the 16 consecutive channels (one chunk) of this lane's pixel, fp32, activation applied.  hi_entry: address of the
// pixel's 32-byte entry in the chunk's hi plane; the lo plane follows `plane` bytes later.  Consecutive lanes hold consecutive
// pixels, so a store instruction covers 32 x 16 bytes at a 32-byte stride (the other half of every line follows in the next one).
// (History: with all chunks of a pixel interleaved as 64-byte records the stores ran at a 256-byte lane stride - 45 us for the
// 134 MB of a 4K trunk tensor, store-issue bound; a 4 x 4 DPP transpose inside lane quads brought that to 21 us, and the planar
// layout needs neither.)
__device__ __forceinline__ void s16_store_chunk(const float (&v)[16], unsigned char* hi_entry, unsigned plane, bool ok) {
    f16x8 hv[2], lv[2];
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const _Float16 hh = (_Float16)v[e];
        hv[e >> 3][e & 7] = hh;
        lv[e >> 3][e & 7] = (_Float16)(v[e] - (float)hh);
    }
    if (ok) {
        f16x8* const dh = reinterpret_cast<f16x8*>(hi_entry);
        f16x8* const dl = reinterpret_cast<f16x8*>(hi_entry + plane);
        dh[0] = hv[0]; dh[1] = hv[1]; dl[0] = lv[0]; dl[1] = lv[1];
    }
}

// ------------------------------------------------------------------------------------------------------------
// conv_h2s2_kernel: 3x3 stride-2 convolution (the second stem conv of every IFBlock, c/2 -> c) on the split-f16 pipe.
//   256 threads = 4 waves, tile = 4 x 32 outputs <- 9 x 65 input pixels per 16-channel chunk (46.8 KB) + weight slab;
//   single LDS buffer, next chunk prefetched into registers (issue early / write late); two workgroups per CU.
// ------------------------------------------------------------------------------------------------------------
template <int NS>
constexpr int convh2s2_lds_bytes() { return 9 * 65 * 80 + 9 * 2 * NS * 32 * 16 + 2 * NS * 32 * 4; }      // + bias and slopes

// S16OUT: the weights were packed with the row permutation s16_row_channel() (conv_t64.h), a lane then holds 16 consecutive
// channels per accumulator and the epilogue writes whole S16 records into the zero-bordered tensor a.out (a.s16_pitch).
template <int NS, bool S16OUT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_h2s2_kernel(ConvArgs a) {
    constexpr int IH = 9, IW = 65, CC = 16, NT = NS * 32, PIXB = 80;
    constexpr int IN_F4 = IH * IW * 4;
    constexpr int W_16 = 9 * 2 * NT;
    constexpr int NIN = (IN_F4 + 255) / 256, NW = (W_16 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char* const lw = ldsb + IH * IW * PIXB;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int half = lane >> 5, li = lane & 31;
    const float* const tin = blockIdx.y ? a.in1 : a.in;                // gridDim.y = 2: the second tensor pair of the launch
    float* const tout = blockIdx.y ? a.out1 : a.out;
    int L;
    {
        const int n = gridDim.x, b = blockIdx.x;
        const int q = n >> 3, r = n & 7, xcd = b & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int tile = L / a.nz, ntile = L - tile * a.nz;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * 4, ox0 = tx * 32;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;

    int goff[NIN];
    unsigned inside = 0;
#pragma unroll
    for (int k = 0; k < NIN; k++) {
        const int idx = tid + k * 256;
        const int p = idx >> 2, q = idx & 3;
        const int py = p / IW, px = p - py * IW;
        const int gy = iy0 + py, gx = ix0 + px;
        const bool ok = idx < IN_F4 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[k] = ok ? (gy * a.W + gx) * a.in_ld + a.in_coff + q * 4 : a.in_coff;
        inside |= ok ? (1u << k) : 0u;
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(a.wpk) + (size_t)ntile * a.nchunks * W_16;

    f32x4 rin[NIN], rw[NW];
#define S2_ISSUE(CH)                                                                                        \
    {                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < NIN; k++) rin[k] = *reinterpret_cast<const f32x4*>(tin + goff[k] + (CH) * CC); \
        _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                    \
            const int idx = tid + k * 256;                                                                  \
            rw[k] = wsrc[(size_t)(CH) * W_16 + ((W_16 % 256 == 0 || idx < W_16) ? idx : 0)];                \
        }                                                                                                   \
    }
#define S2_WRITE()                                                                                          \
    {                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < NIN; k++) {                                                   \
            const int idx = tid + k * 256;                                                                  \
            const int p = idx >> 2, q = idx & 3;                                                            \
            f16x4 hi4, lo4;                                                                                 \
            _Pragma("unroll") for (int e = 0; e < 4; e++) {                                                 \
                const float v = ((inside >> k) & 1u) ? rin[k][e] : 0.f;                                     \
                const _Float16 h = (_Float16)v;                                                             \
                hi4[e] = h;                                                                                 \
                lo4[e] = (_Float16)(v - (float)h);                                                          \
            }                                                                                               \
            if (IN_F4 % 256 == 0 || idx < IN_F4) {                                                          \
                *reinterpret_cast<f16x4*>(ldsb + p * PIXB + q * 8) = hi4;                                   \
                *reinterpret_cast<f16x4*>(ldsb + p * PIXB + 32 + q * 8) = lo4;                              \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < NW; k++) {                                                    \
            const int idx = tid + k * 256;                                                                  \
            if (W_16 % 256 == 0 || idx < W_16) reinterpret_cast<f32x4*>(lw)[idx] = rw[k];                   \
        }                                                                                                   \
    }

    f32x16 acc[NS];
#pragma unroll
    for (int n = 0; n < NS; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    const unsigned char* ab = ldsb + ((2 * wv) * IW + 2 * li) * PIXB + half * 16;
    const unsigned char* bb = lw + (half * NT + li) * 16;

    S2_ISSUE(0)
    float* const lbs = reinterpret_cast<float*>(ldsb + 9 * 65 * 80 + 9 * 2 * NT * 16);
    if (tid < NT) { lbs[tid] = a.bias[ntile * NT + tid]; lbs[NT + tid] = a.slope[ntile * NT + tid]; }
    S2_WRITE()
    __syncthreads();
    for (int ch = 0; ch < a.nchunks; ch++) {
        const bool more = ch + 1 < a.nchunks;
        if (more) S2_ISSUE(ch + 1)
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int dy = t / 3, dx = t % 3;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(ab + (dy * IW + dx) * PIXB);
            const f16x8 al = *reinterpret_cast<const f16x8*>(ab + (dy * IW + dx) * PIXB + 32);
            f16x8 bw[NS];
#pragma unroll
            for (int n = 0; n < NS; n++) bw[n] = *reinterpret_cast<const f16x8*>(bb + (t * 2 * NT + n * 32) * 16);
#pragma unroll
            for (int n = 0; n < NS; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[n], ah, acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NS; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[n], al, acc[n], 0, 0, 0);
        }
        if (more) {
            __syncthreads();
            S2_WRITE()
            __syncthreads();
        }
    }
#undef S2_ISSUE
#undef S2_WRITE

    const int oy = oy0 + wv, ox = ox0 + li;
    const bool pok = oy < a.Ho && ox < a.Wo;
    if (S16OUT) {
        unsigned char* const o = reinterpret_cast<unsigned char*>(tout) + ((size_t)(2 * half + 4 * NS * ntile) * a.s16_plane + ((size_t)(oy + 1) * a.s16_pitch + ox + 1) * 32);
#pragma unroll
        for (int n = 0; n < NS; n++) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbs + n * 32 + 16 * half + 4 * q);
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(lbs + NT + n * 32 + 16 * half + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float y = acc[n][4 * q + k] + b4[k];
                    v[4 * q + k] = y < 0.f ? y * s4[k] : y;
                }
            }
            s16_store_chunk(v, o + (size_t)(4 * n) * a.s16_plane, a.s16_plane, pok);
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NS; n++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int c0 = ntile * NT + n * 32 + 8 * q + 4 * half;
            const bool ok = pok && c0 < a.Cout;
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(lbs + n * 32 + 8 * q + 4 * half);
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(lbs + NT + n * 32 + 8 * q + 4 * half);
            f32x4 v;
#pragma unroll
            for (int k = 0; k < 4; k++) { v[k] = acc[n][4 * q + k] + b4[k]; v[k] = v[k] < 0.f ? v[k] * s4[k] : v[k]; }
            if (ok) *reinterpret_cast<f32x4*>(tout + ((size_t)oy * a.Wo + ox) * a.out_ld + a.out_coff + c0) = v;
        }
    }
}

// split-K reduction: out = slope(sum over splits (fixed order) + bias), 4 channels per thread
__global__ void k_splitk_reduce(const float* __restrict__ partial, int nsplit, size_t npix, int cpad, int cout, const float* __restrict__ bias,
                                const float* __restrict__ slope, float* __restrict__ out, int out_ld, int out_coff) {
    const int nq = cout / 4;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * nq) return;
    const size_t p = i / nq; const int c0 = (int)(i - p * nq) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(partial + p * cpad + c0);
    for (int s = 1; s < nsplit; s++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(partial + ((size_t)s * npix + p) * cpad + c0);
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] += t[k];
    }
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + c0), s4 = *reinterpret_cast<const f32x4*>(slope + c0);
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k] += b4[k]; v[k] = v[k] < 0.f ? v[k] * s4[k] : v[k]; }
    *reinterpret_cast<f32x4*>(out + p * out_ld + out_coff + c0) = v;
}

}  // namespace rife
