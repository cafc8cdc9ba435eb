// librife_hip: host side of the MI355X-native RIFE engine + the C-ABI declared in include/rife_hip.h.
//
// What of the reference this file replaces (all under /root/reference/src):
//   RIFE::RIFE / ~RIFE        rife.cpp:27-78     -> rife_hip_create / rife_hip_destroy
//   RIFE::load                rife.cpp:127-379   -> rife_hip_load  (ncnn::Net::load_param/load_model -> NcnnModel,
//                                                  pipeline creation -> kernels are compiled ahead of time)
//   RIFE::process_v4          rife.cpp:2462-3202 -> Engine::run_v4 (one fixed schedule instead of ncnn's graph walk)
//   ncnn VkCompute record/submit/wait (rife.cpp:2522-2530, 3176-3186) -> one HIP stream per in-flight pair
// There is deliberately no CPU path in this library: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <array>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// Three builds of these sources (csrc/Makefile): librife_hip.so = the PRODUCT (exports include/rife_hip.h, nothing else, one schedule);
// librife_hip_test.so (-DRIFE_HIP_TEST_BUILD) = + the parity taps / single-kernel entry points of include/rife_hip_test.h and the kernel-selection
// switches the kernel-vs-kernel tests flip; librife_hip_bench.so (-DRIFE_HIP_BENCH_BUILD) = + csrc/bench_hooks.h.
#if defined(RIFE_HIP_BENCH_BUILD) && !defined(RIFE_HIP_TEST_BUILD)
#define RIFE_HIP_TEST_BUILD 1
#endif
#include "../../include/rife_hip.h"
#ifdef RIFE_HIP_TEST_BUILD
#include "../../include/rife_hip_test.h"      // the parity taps and single-kernel entry points the test build also exports
#endif
#include "conv_mfma.h"
#include "conv_img.h"
#include "elementwise.h"
#include "elementwise_v2.h"
#include "stem_fused.h"
#include "stem_fused_v2.h"
#include "stem_rs.h"
#include "tail_rs.h"
#include "head_h2.h"
#include "conv_t64.h"
#include "conv_row.h"
#include "conv_rs.h"
#include "conv_rs2.h"
#ifdef RIFE_HIP_TEST_BUILD
#include "conv_ks.h"      // round-4 K-split trunk kernel: opt-in (RIFE_HIP_KS), measured slower with pairs in flight; not compiled into the product
#endif
#include "graph_kernels.h"
#include "model_hashes.h"
#include "ncnn_model.h"

namespace rife {

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return -code; }

// Kernel-selection and A/B switches (RIFE_HIP_T64 / RS / KS / STEM_RS / TAIL_RS / FUSE_FLOW / ...): read only by the TEST build (librife_hip_test.so, the same sources
// with -DRIFE_HIP_TEST_BUILD, which also exports include/rife_hip_test.h) and the bench build (librife_hip_bench.so).  The PRODUCT ignores them: it runs one schedule,
// and the only environment variables it reads are RIFE_HIP_TRUNK=f32 (fp32 matrix path), RIFE_HIP_GRAPH=1 (hipGraph replay), RIFE_HIP_BATCH_WORKERS (process_batch
// worker threads) and RIFE_HIP_PROFILE_FINE=1 (per-layer profile classes).
// Values of the switches are parsed by these NAMED helpers, never by immediately-invoked lambdas in static initialisers.  Round 5 found why: hipcc numbers the
// closure types of namespace-scope lambdas per `namespace rife { }` block, and this file re-opens the namespace four times, so the initialiser lambda of
// g_use_graph (first of its block) carried the mangled name of g_trunk_h2's (first of the first block) and the linker-visible internal symbol of the one was
// the code of the other: _GLOBAL__sub_I_engine.hip read RIFE_HIP_TRUNK into g_use_graph (hipGraph replay silently ON for every plain v4 pass <= 1080p since the
// second block appeared, RIFE_HIP_GRAPH itself never read) and RIFE_HIP_T64_LW into g_v2_fused_stem (objdump of the static initialiser; profiles/r5/README.md).
static inline bool env_on(const char* e) { return e && e[0] == '1'; }               // default off, "=1" switches on
static inline bool env_not_off(const char* e) { return !(e && e[0] == '0'); }       // default on, "=0" switches off
static inline bool env_is(const char* e, const char* value) { return e && std::strcmp(e, value) == 0; }
static inline int env_int(const char* e, int dflt, int lo, int hi) { if (!e) return dflt; const int v = atoi(e); return v >= lo && v <= hi ? v : dflt; }
static inline const char* ab_getenv(const char* name) {
#ifdef RIFE_HIP_TEST_BUILD
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(RIFE_HIP_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// conv layer: host-side description + packed device weights
// ------------------------------------------------------------------------------------------------
struct ConvLayer {
    int cin = 0, cin_p = 0, cout = 0, stride = 1;
    int ks = 3;                           // 3 (pad 1) or 5 (pad 2; fp32 MFMA kernel only)
    bool deconv = false;
    int epi = EPI_STORE;
    int MS = 2, NS = 2, CC = 16;          // kernel configuration
    int ntiles = 1, nchunks = 1, ntaps = 9, npar = 1;
    float *d_w = nullptr, *d_bias = nullptr, *d_slope = nullptr;
    float* d_w8 = nullptr;                // weights packed with CC = 8 for conv_mfma8_kernel (stride-1 3x3 layers only)
    int nchunks8 = 0;
    uint16_t* d_wh = nullptr;             // fp16 weights packed for conv_h2_kernel (split-f16 trunk path)
    int nchunksh = 0;
    bool skip = false;                    // layer is x + conv(x): identity folded into the GEMM
    // S16 trunk path (conv_t64.h): static LDS image of the persistent 64 -> 64 trunk kernel / row-permuted fp16 weights of the
    // stride-2 stem that writes the first S16 tensor
    bool want_t64 = false, want_s16out = false;
    unsigned char* d_t64 = nullptr;
    unsigned char* d_row = nullptr;      // 96 channels: the conv_row image next to the conv_t64 one (small grids)
    uint16_t* d_whp = nullptr;
    uint16_t* d_wimg = nullptr;           // 3 -> 32 stride-2 layer on the RGBX u8 frame (conv_img.h): f16 [K-step 3][k half 2][32][8]
    double flops_per_pixel = 0;           // algorithmic: 2 * MAC per GEMM-M pixel
    std::string cls;                      // profile class
    int tag = 0;                          // distinct kernel symbol for the profiled layer class
};

static void free_layer(ConvLayer& L) {
    if (L.d_w) (void)hipFree(L.d_w);
    if (L.d_bias) (void)hipFree(L.d_bias);
    if (L.d_slope) (void)hipFree(L.d_slope);
    if (L.d_w8) (void)hipFree(L.d_w8);
    if (L.d_wh) (void)hipFree(L.d_wh);
    if (L.d_t64) (void)hipFree(L.d_t64);
    if (L.d_row) (void)hipFree(L.d_row);
    if (L.d_whp) (void)hipFree(L.d_whp);
    if (L.d_wimg) (void)hipFree(L.d_wimg);
    L.d_wimg = nullptr;
    L.d_w = L.d_bias = L.d_slope = L.d_w8 = nullptr; L.d_wh = nullptr; L.d_t64 = nullptr; L.d_row = nullptr; L.d_whp = nullptr;
}

// Choose the kernel configuration for a layer (see conv_mfma.h for the meaning of MS / NS / CC).
static void configure(ConvLayer& L) {
    const int NT = L.cout <= 32 ? 32 : (L.cout % 64 == 0 ? 64 : (L.cout % 96 == 0 ? 96 : 64));      // 96-wide N tiles beat 3 x 32 (round-1 A/B); 192 = 3 x 64, not 2 x 96 (round-5 A/B)
    L.NS = NT / 32;
    L.ntiles = (L.cout + NT - 1) / NT;
    if (L.stride == 2) { L.MS = 1; L.CC = 8; }
    else if (L.NS == 3) { L.MS = 2; L.CC = 8; }
    else { L.MS = 2; L.CC = 16; }
    L.cin_p = (L.cin + L.CC - 1) / L.CC * L.CC;
    L.nchunks = L.cin_p / L.CC;
    L.ntaps = L.deconv ? 4 : 9;
    L.npar = L.deconv ? 4 : 1;
    if (L.ks == 5) {   // 25 taps: 32-channel N-tiles keep the weight slab of a chunk at 25.6 KB
        L.NS = 1; L.ntiles = (L.cout + 31) / 32; L.CC = 8; L.MS = L.stride == 2 ? 1 : 2;
        L.cin_p = (L.cin + 7) / 8 * 8; L.nchunks = L.cin_p / 8; L.ntaps = 25; L.tag = 5;
    }
}

// ncnn weight order [oc][ic][kh][kw] (also for Deconvolution) -> MFMA B-fragment order
// [ntile][par][chunk][tap][g][half][n][4], channel = chunk*CC + g*8 + half*4 + s.
static std::vector<float> pack_weights(const ConvLayer& L, const float* w) {
    const int NT = L.NS * 32, NG = L.CC / 8, K = L.deconv ? 4 : L.ks;
    // deconv: out(2y+p) gathers input y+d through kernel row k with 2(y+d) + k - 1 = 2y + p
    //   p=0: tap bit 0 -> (d=0,k=1), bit 1 -> (d=-1,k=3);  p=1: bit 0 -> (d=0,k=2), bit 1 -> (d=+1,k=0)   (offsets: conv_mfma.h)
    static const int KD[2][2] = {{1, 3}, {2, 0}};
    std::vector<float> out((size_t)L.ntiles * L.npar * L.nchunks * L.ntaps * L.CC * NT, 0.f);
    size_t o = 0;
    for (int nt = 0; nt < L.ntiles; nt++)
        for (int par = 0; par < L.npar; par++)
            for (int ch = 0; ch < L.nchunks; ch++)
                for (int t = 0; t < L.ntaps; t++) {
                    int ky, kx;
                    if (L.deconv) { ky = KD[par >> 1][t >> 1]; kx = KD[par & 1][t & 1]; }
                    else { ky = t / K; kx = t % K; }
                    for (int g = 0; g < NG; g++)
                        for (int half = 0; half < 2; half++)
                            for (int n = 0; n < NT; n++)
                                for (int s = 0; s < 4; s++, o++) {
                                    const int c = ch * L.CC + g * 8 + half * 4 + s, oc = nt * NT + n;
                                    if (c < L.cin && oc < L.cout) out[o] = w[(((size_t)oc * L.cin + c) * K + ky) * K + kx];
                                }
                }
    return out;
}

static uint16_t f2h(float f) {
    uint32_t x; std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t e = (int32_t)((x >> 23) & 0xff) - 127 + 15;
    uint32_t m = x & 0x7fffffu;
    if (((x >> 23) & 0xff) == 0xff) return (uint16_t)(sign | 0x7c00u | (m ? 0x200u : 0));
    if (e >= 31) return (uint16_t)(sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int sh = 14 - e;
        uint32_t r = m >> sh;
        const uint32_t rem = m & ((1u << sh) - 1), halfway = 1u << (sh - 1);
        if (rem > halfway || (rem == halfway && (r & 1))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(sign | r);
}

// fp16 B... A-fragment order of conv_h2_kernel: [ntile][chunk of 16 ch][tap][half][n][8], channel = chunk*16 + half*8 + e;
// tap 9 (only when the layer carries a skip connection) = identity on the centre pixel.
static std::vector<uint16_t> pack_weights_h2(const ConvLayer& L, const float* w, int ntaps) {
    const int NT = L.NS * 32, nch = (L.cin + 15) / 16;
    std::vector<uint16_t> out((size_t)L.ntiles * nch * ntaps * 2 * NT * 8, 0);
    size_t o = 0;
    for (int nt = 0; nt < L.ntiles; nt++)
        for (int ch = 0; ch < nch; ch++)
            for (int t = 0; t < ntaps; t++)
                for (int half = 0; half < 2; half++)
                    for (int n = 0; n < NT; n++)
                        for (int e = 0; e < 8; e++, o++) {
                            const int c = ch * 16 + half * 8 + e, oc = nt * NT + n;
                            if (oc >= L.cout || c >= L.cin) continue;
                            float v;
                            if (t == 9) v = (c == oc) ? 1.f : 0.f;
                            else v = w[((size_t)oc * L.cin + c) * 9 + t];
                            out[o] = f2h(v);
                        }
    return out;
}

// pack_weights_h2 with the output rows of every 32-row MFMA block permuted by s16_row_channel(): conv_h2s2_kernel<NS, true>
static std::vector<uint16_t> pack_weights_h2_perm(const ConvLayer& L, const float* w) {
    const int NT = L.NS * 32, nch = (L.cin + 15) / 16;
    std::vector<uint16_t> out((size_t)L.ntiles * nch * 9 * 2 * NT * 8, 0);
    size_t o = 0;
    for (int nt = 0; nt < L.ntiles; nt++)
        for (int ch = 0; ch < nch; ch++)
            for (int t = 0; t < 9; t++)
                for (int half = 0; half < 2; half++)
                    for (int n = 0; n < NT; n++)
                        for (int e = 0; e < 8; e++, o++) {
                            const int c = ch * 16 + half * 8 + e, oc = nt * NT + (n & ~31) + s16_row_channel(n & 31);
                            if (oc < L.cout && c < L.cin) out[o] = f2h(w[((size_t)oc * L.cin + c) * 9 + t]);
                        }
    return out;
}

// Weight image of conv_t64_kernel (conv_t64.h) for a C -> C layer: N-tiles of NT = 32 NS output channels (C = 64, 96: one N-tile of C;
// C = 128, 192: N-tiles of 64); per N-tile fp16 weights [chunk C/16][tap 9][k half 2][row NT][8] with the rows of each 32-row block
// permuted by s16_row_channel() (every chunk is one contiguous LDS-DMA source), then that N-tile's bias[NT] and slope[NT] as fp32.
static int t64_ns(int C) { return C == 96 ? 3 : 2; }
// NSf > 0 forces the N-tile width (conv_row_kernel: NSf = 1, one 32-channel output block per wave)
static std::vector<unsigned char> pack_t64_image(const float* w, const float* bias, float slope, int C = 64, int NSf = 0, const float* slopes = nullptr) {
    const int NS = NSf > 0 ? NSf : t64_ns(C), NT = 32 * NS, nnt = C / NT, nch = C / 16;
    const size_t stride = t64_img_nt(NS, nch);
    std::vector<unsigned char> img(stride * nnt, 0);
    for (int nt = 0; nt < nnt; nt++) {
        uint16_t* wh = reinterpret_cast<uint16_t*>(img.data() + nt * stride);
        for (int c = 0; c < nch; c++)
            for (int t = 0; t < 9; t++)
                for (int kh = 0; kh < 2; kh++)
                    for (int row = 0; row < NT; row++)
                        for (int e = 0; e < 8; e++) {
                            const int oc = nt * NT + (row & ~31) + s16_row_channel(row & 31), ic = 16 * c + 8 * kh + e;
                            wh[((((size_t)c * 9 + t) * 2 + kh) * NT + row) * 8 + e] = f2h(w[((size_t)oc * C + ic) * 9 + t]);
                        }
        float* bs = reinterpret_cast<float*>(img.data() + nt * stride + (size_t)nch * t64_wch(NS));
        for (int i = 0; i < NT; i++) { bs[i] = bias ? bias[nt * NT + i] : 0.f; bs[NT + i] = slopes ? slopes[nt * NT + i] : slope; }
    }
    return img;
}

// host mirror of head_uses() / the pair order of head_h2.h
static bool head_uses_h(int t, int par) {
    const int dy = t / 3 - 1, dx = t % 3 - 1, py = par >> 1, px = par & 1;
    return (dy == 0 || dy == (py ? 1 : -1)) && (dx == 0 || dx == (px ? 1 : -1));
}

// Deconvolution (k4 s2 p1) for head_h2_kernel: fp16 [ntile of 32 channels][chunk][(tap, parity) pair 16][half][n 32][8].
// Kernel row for (parity p, offset d): p=0: d=0 -> k=1, d=-1 -> k=3;  p=1: d=0 -> k=2, d=+1 -> k=0.
static std::vector<uint16_t> pack_weights_head_h2(const ConvLayer& L, const float* w) {
    const int nch = L.cin / 16, nt32 = (L.cout + 31) / 32;
    std::vector<uint16_t> out((size_t)nt32 * nch * 16 * 2 * 32 * 8, 0);
    size_t o = 0;
    for (int nt = 0; nt < nt32; nt++)
        for (int ch = 0; ch < nch; ch++)
            for (int t = 0; t < 9; t++)
                for (int par = 0; par < 4; par++) {
                    if (!head_uses_h(t, par)) continue;
                    const int dy = t / 3 - 1, dx = t % 3 - 1, py = par >> 1, px = par & 1;
                    const int ky = dy == 0 ? (py ? 2 : 1) : (py ? 0 : 3), kx = dx == 0 ? (px ? 2 : 1) : (px ? 0 : 3);
                    for (int half = 0; half < 2; half++)
                        for (int n = 0; n < 32; n++)
                            for (int e = 0; e < 8; e++, o++) {
                                const int c = ch * 16 + half * 8 + e, oc = nt * 32 + n;
                                if (oc < L.cout) out[o] = f2h(w[(((size_t)oc * L.cin + c) * 4 + ky) * 4 + kx]);
                            }
                }
    return out;
}

static int upload_layer(ConvLayer& L, const float* w, const float* bias, const float* slope /*per-channel or null*/, float uniform_slope) {
    configure(L);
    std::vector<float> wskip;
    const float* w_orig = w;
    if (L.skip) {
        // x + conv(x) == conv'(x) with W'[o][o][1][1] = W[o][o][1][1] + 1: the skip connection of the residual block
        // (flownet.param:13-15 "Split, Convolution, BinaryOp add") rides the centre tap of the fp32 GEMM instead of a second
        // read of x in the epilogue.  fp16-stored weights + 1.0f are exact in fp32 down to 2^-23.
        wskip.assign(w, w + (size_t)L.cin * L.cout * 9);
        for (int o = 0; o < L.cout; o++) wskip[((size_t)o * L.cin + o) * 9 + 4] += 1.0f;
        w = wskip.data();
    }
    std::vector<float> pk = pack_weights(L, w);
    const int cp = L.ntiles * L.NS * 32;
    std::vector<float> b(cp, 0.f), s(cp, 1.f);
    for (int i = 0; i < L.cout; i++) { b[i] = bias ? bias[i] : 0.f; s[i] = slope ? slope[i] : uniform_slope; }
    HIPCHK(hipMalloc(&L.d_w, pk.size() * 4));
    HIPCHK(hipMalloc(&L.d_bias, cp * 4));
    HIPCHK(hipMalloc(&L.d_slope, cp * 4));
    HIPCHK(hipMemcpy(L.d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(L.d_bias, b.data(), cp * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(L.d_slope, s.data(), cp * 4, hipMemcpyHostToDevice));
    const int K = L.deconv ? 16 : L.ks * L.ks;
    L.flops_per_pixel = 2.0 * L.cin * L.cout * K;
    if (L.ks != 3) return 0;          // the 8-wave and split-f16 variants below are 3 x 3 kernels
    if (!L.deconv && L.stride == 1 && L.epi == EPI_STORE && L.NS >= 2 && L.cin % 8 == 0) {
        if (L.CC == 8) { L.d_w8 = nullptr; L.nchunks8 = L.nchunks; }     // the normal packing already is CC = 8
        else {
            ConvLayer T = L; T.CC = 8; T.cin_p = L.cin; T.nchunks = L.cin / 8;
            std::vector<float> pk8 = pack_weights(T, w);
            HIPCHK(hipMalloc(&L.d_w8, pk8.size() * 4));
            HIPCHK(hipMemcpy(L.d_w8, pk8.data(), pk8.size() * 4, hipMemcpyHostToDevice));
            L.nchunks8 = T.nchunks;
        }
    }
    if (L.deconv && L.cin % 16 == 0 && L.cout % 4 == 0) {   // transposed convs: split-f16 kernel, 4 parities per workgroup, 32-channel N-tiles
        bool exact = true;
        for (size_t i = 0; i < (size_t)L.cin * L.cout * 16 && exact; i++) exact = (float)(_Float16)w_orig[i] == w_orig[i];
        if (exact) {
            std::vector<uint16_t> ph = pack_weights_head_h2(L, w_orig);
            HIPCHK(hipMalloc(&L.d_wh, ph.size() * 2));
            HIPCHK(hipMemcpy(L.d_wh, ph.data(), ph.size() * 2, hipMemcpyHostToDevice));
            L.nchunksh = L.cin / 16;
        }
    }
    if (!L.deconv && L.stride == 2 && L.cin == 12 && L.ntiles == 1) {      // v4 stem-0 of blocks 1..3: fused assemble + conv kernel
        std::vector<uint16_t> ph = pack_weights_h2(L, w_orig, 9);
        bool exact = true;
        for (size_t i = 0; i < (size_t)L.cin * L.cout * 9 && exact; i++) exact = (float)(_Float16)w_orig[i] == w_orig[i];
        if (exact) {
            HIPCHK(hipMalloc(&L.d_wh, ph.size() * 2));
            HIPCHK(hipMemcpy(L.d_wh, ph.data(), ph.size() * 2, hipMemcpyHostToDevice));
            L.nchunksh = 1;
        }
    }
    if (!L.deconv && L.stride == 2 && L.cin == 3 && L.cout == 32) {      // ContextNet's first convolution, read straight from the RGBX u8 frame (conv_img.h)
        bool exact = true;
        for (size_t i = 0; i < (size_t)L.cin * L.cout * 9 && exact; i++) exact = (float)(_Float16)w_orig[i] == w_orig[i];
        if (exact) {
            std::vector<uint16_t> pk(3 * 2 * 32 * 8, 0);
            for (int j = 0; j < 3; j++)
                for (int hh = 0; hh < 2; hh++)
                    for (int oc = 0; oc < 32; oc++)
                        for (int e = 0; e < 8; e++) {
                            const int t = 4 * j + 2 * hh + (e >> 2), c = e & 3;
                            if (t < 9 && c < 3) pk[((size_t)(j * 2 + hh) * 32 + oc) * 8 + e] = f2h(w_orig[((size_t)oc * 3 + c) * 9 + t]);
                        }
            HIPCHK(hipMalloc(&L.d_wimg, pk.size() * 2));
            HIPCHK(hipMemcpy(L.d_wimg, pk.data(), pk.size() * 2, hipMemcpyHostToDevice));
        }
    }
    // stem-1 class: split-f16 stride-2 kernel.  cin = 10 (rife-v2.x / v3.x: the first convolution of IFNet blocks 1.. and of the FusionNet, whose 10-channel
    // input is assembled as NHWC16 with six zero channels, elementwise_v2.h) rides the same kernel as one zero-padded 16-channel chunk instead of the fp32
    // matrix path (round 5: 125 -> us for the 1920x1088 -> 48-channel layer)
    static const bool stem16 = env_not_off(ab_getenv("RIFE_HIP_V2_STEM16"));
    if (!L.deconv && L.stride == 2 && L.epi == EPI_STORE && ((L.cin % 16 == 0 && L.cin >= 16) || (L.cin == 10 && stem16))) {
        bool exact = true;
        for (size_t i = 0; i < (size_t)L.cin * L.cout * 9 && exact; i++) exact = (float)(_Float16)w_orig[i] == w_orig[i];
        if (exact) {
            std::vector<uint16_t> ph = pack_weights_h2(L, w_orig, 9);
            HIPCHK(hipMalloc(&L.d_wh, ph.size() * 2));
            HIPCHK(hipMemcpy(L.d_wh, ph.data(), ph.size() * 2, hipMemcpyHostToDevice));
            L.nchunksh = (L.cin + 15) / 16;
            if (L.want_s16out && L.cin % 16 == 0 && L.cout % (L.NS * 32) == 0) {
                std::vector<uint16_t> pp = pack_weights_h2_perm(L, w_orig);
                HIPCHK(hipMalloc(&L.d_whp, pp.size() * 2));
                HIPCHK(hipMemcpy(L.d_whp, pp.data(), pp.size() * 2, hipMemcpyHostToDevice));
            }
        }
    }
    if (!L.deconv && L.stride == 1 && L.epi == EPI_STORE && L.cin % 16 == 0) {
        bool exact = true;   // the split-f16 path needs weights that are exactly fp16 (true for ncnn fp16-stored models)
        for (size_t i = 0; i < (size_t)L.cin * L.cout * 9 && exact; i++) {
            const uint16_t h = f2h(w_orig[i]);
            const uint32_t sgn = (uint32_t)(h & 0x8000u) << 16, ex = (h >> 10) & 0x1f, mn = h & 0x3ffu;
            float back;
            if (ex == 0) back = std::ldexp((float)mn, -24) * (sgn ? -1.f : 1.f);
            else { const uint32_t bits = sgn | ((ex + 112) << 23) | (mn << 13); std::memcpy(&back, &bits, 4); }
            exact = back == w_orig[i];
        }
        if (exact) {
            std::vector<uint16_t> ph = pack_weights_h2(L, w_orig, L.skip ? 10 : 9);
            HIPCHK(hipMalloc(&L.d_wh, ph.size() * 2));
            HIPCHK(hipMemcpy(L.d_wh, ph.data(), ph.size() * 2, hipMemcpyHostToDevice));
            L.nchunksh = L.cin / 16;
            if (L.want_t64 && L.skip && L.cin == L.cout && (L.cout == 64 || L.cout == 96 || L.cout == 128 || L.cout == 192) && !slope) {
                std::vector<unsigned char> img = pack_t64_image(w_orig, bias, uniform_slope, L.cout, L.cout >= 128 ? 1 : 0);
                HIPCHK(hipMalloc(&L.d_t64, img.size()));
                HIPCHK(hipMemcpy(L.d_t64, img.data(), img.size(), hipMemcpyHostToDevice));
                if (L.cout == 96) {
                    std::vector<unsigned char> ri = pack_t64_image(w_orig, bias, uniform_slope, L.cout, 1);
                    HIPCHK(hipMalloc(&L.d_row, ri.size()));
                    HIPCHK(hipMemcpy(L.d_row, ri.data(), ri.size(), hipMemcpyHostToDevice));
                }
            }
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// kernel dispatch
// ------------------------------------------------------------------------------------------------
template <int STRIDE, int MS, int NS, int CC, int EPI, int TAG>
static hipError_t launch_cfg(const ConvArgs& a, int nblocks, hipStream_t st) {
    auto kfn = conv_mfma_kernel<STRIDE, MS, NS, CC, EPI, TAG>;
    constexpr int lds = conv_lds_bytes<STRIDE, MS, NS, CC, EPI, conv_ks<TAG>()>();
    static_assert(lds <= 64 * 1024, "tile does not fit the default dynamic LDS limit");
    hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

struct TensorView { float* p; int ld, coff; };

// split-K partial-sum workspace: one per (device, stream), grown on demand (used only by small layers)
static float* splitk_workspace(hipStream_t st, size_t floats) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<float*, size_t>> ws;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    auto& e = ws[{dev, st}];
    if (e.second < floats) {
        if (e.first) { (void)hipStreamSynchronize(st); (void)hipFree(e.first); }
        if (hipMalloc(&e.first, floats * 4) != hipSuccess) { e.first = nullptr; e.second = 0; return nullptr; }
        e.second = floats;
    }
    return e.first;
}

// RIFE_HIP_TRUNK=f32 keeps the trunk convolutions on the fp32 matrix path (default: split-f16, see conv_h2_kernel): the documented
// numerics fallback, and bench.py's fp32 reference mode.  The round-1 A/B switches with a settled winner (fused stem, split-f16
// heads and stride-2 stems, split-K for tiny grids, fused tail, two-workgroup trunk kernel, 8-wave fp32 kernel, 96-wide N tiles,
// 4-row tiles below 400 workgroups) are constants now; the measurements behind them are in DESIGN.md and profiles/r1.
static const bool g_trunk_h2 = !env_is(getenv("RIFE_HIP_TRUNK"), "f32");
static constexpr bool g_fuse_stem = true, g_head_h2 = true, g_s2_h2 = true, g_splitk = true, g_fuse_tail = true, g_h2b = true, g_use_conv8 = true;

// x: NHWC input (H x W), y: output; for deconv layers y has 2H x 2W pixels (or the 4H x 4W flow tensor with EPI_DECONV_PS).
// s16_pitch > 0: the stride-2 stem writes / the head reads an S16 tensor (conv_t64.h) of that row pitch instead of NHWC fp32
static int launch_conv(const ConvLayer& L, TensorView x, int H, int W, TensorView y, const TensorView* res, hipStream_t st, const FinalArgs* fin = nullptr,
                       int s16_pitch = 0, unsigned s16_plane = 0, const float* in1 = nullptr, float* out1 = nullptr, const TensorView* y2 = nullptr) {
    // in1 / out1: a second tensor pair of the same geometry through the same launch (gridDim.y = 2; the stride-2 and stride-1 split-f16 kernels)
    ConvArgs a;
    a.in1 = in1; a.out1 = out1;
    const unsigned gy = in1 ? 2 : 1;
    // y2: the output goes to a second view as well (conv_h2b_kernel only: stride-1 split-f16 layers without split-K)
    if (y2) { a.out2 = y2->p; a.out2_ld = y2->ld; a.out2_coff = y2->coff; }
    if (y2 && !(L.nchunksh > 0 && L.stride == 1 && !L.deconv && g_trunk_h2 && res == nullptr && g_h2b && L.NS <= 2))
        return fail(RIFE_HIP_EINVAL, "no two-destination form of this convolution kernel");
    a.s16_pitch = s16_pitch; a.s16_plane = s16_plane;
    a.in = x.p; a.in_ld = x.ld; a.in_coff = x.coff; a.H = H; a.W = W;
    a.out = y.p; a.out_ld = y.ld; a.out_coff = y.coff;
    a.wpk = L.d_w; a.bias = L.d_bias; a.slope = L.d_slope;
    a.res = res ? res->p : nullptr; a.res_ld = res ? res->ld : 0; a.res_coff = res ? res->coff : 0;
    a.Ho = L.deconv ? H : (H + 2 - 3) / L.stride + 1;
    a.Wo = L.deconv ? W : (W + 2 - 3) / L.stride + 1;
    a.Cout = L.cout; a.nchunks = L.nchunks; a.nz = L.ntiles * L.npar;
    if (x.ld % 4 || x.coff % 4 || x.ld - x.coff < L.cin_p) return fail(RIFE_HIP_EINVAL, "conv input view is not padded to the channel chunk");
    if (L.epi != EPI_DECONV_PS && (y.ld % 4 || y.coff % 4 || L.cout % 4 || (res && (res->ld % 4 || res->coff % 4))))
        return fail(RIFE_HIP_EINVAL, "conv output / residual views must be 16-byte aligned per pixel (channel counts multiples of 4)");
    a.tiles_x = (a.Wo + 31) / 32;
    // rows per wave: 2 when that still gives every CU >= 1.5 workgroups, else 1 (more, smaller workgroups for the coarse blocks)
    int MS = L.MS;
    if (L.stride == 1) {
        const long wg2 = (long)a.tiles_x * ((a.Ho + 7) / 8) * a.nz;
        MS = wg2 >= 384 ? 2 : 1;
    }
    if (L.nchunksh > 0 && !L.deconv && L.stride == 2 && (L.cin >= 16 || L.cin == 10) && g_trunk_h2 && g_s2_h2 && res == nullptr) {
        if (x.ld - x.coff < 16 * L.nchunksh) return fail(RIFE_HIP_EINVAL, "conv input view is not padded to whole 16-channel chunks");
        a.ntiles_xy = a.tiles_x * ((a.Ho + 3) / 4);
        a.nchunks = L.nchunksh;
        a.wpk = reinterpret_cast<const float*>(L.d_wh);
        const int nb = a.ntiles_xy * a.nz;
        constexpr int ls1 = convh2s2_lds_bytes<1>(), ls2 = convh2s2_lds_bytes<2>(), ls3 = convh2s2_lds_bytes<3>();
        {
            static std::mutex smu; static std::map<int, bool> sdone;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(smu);
            if (!sdone[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2s2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, ls2));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2s2_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ls2));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2s2_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ls3));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2s2_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, ls3));
                sdone[dev] = true;
            }
        }
        if (s16_pitch > 0) {
            if ((L.NS != 2 && L.NS != 3) || !L.d_whp) return fail(RIFE_HIP_EINVAL, "no S16 variant of this stride-2 layer");
            a.wpk = reinterpret_cast<const float*>(L.d_whp);
            if (L.NS == 2) hipLaunchKernelGGL((conv_h2s2_kernel<2, true>), dim3(nb, gy), dim3(256), ls2, st, a);
            else hipLaunchKernelGGL((conv_h2s2_kernel<3, true>), dim3(nb, gy), dim3(256), ls3, st, a);
        }
        else if (L.NS == 1) hipLaunchKernelGGL(conv_h2s2_kernel<1>, dim3(nb, gy), dim3(256), ls1, st, a);
        else if (L.NS == 2) hipLaunchKernelGGL(conv_h2s2_kernel<2>, dim3(nb, gy), dim3(256), ls2, st, a);
        else hipLaunchKernelGGL(conv_h2s2_kernel<3>, dim3(nb, gy), dim3(256), ls3, st, a);
        hipError_t eh = hipGetLastError();
        if (eh != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_h2s2 launch: ") + hipGetErrorString(eh));
        return 0;
    }
    if (fin && !(L.nchunksh > 0 && L.deconv && g_trunk_h2 && g_head_h2)) return fail(RIFE_HIP_EINVAL, "fused tail needs the split-f16 head kernel");
    if (L.nchunksh > 0 && L.deconv && g_trunk_h2 && g_head_h2) {
        if (in1) return fail(RIFE_HIP_EINVAL, "no two-tensor form of the head kernel");
        a.ntiles_xy = a.tiles_x * ((a.Ho + 7) / 8);
        a.nchunks = L.nchunksh;
        a.nz = (L.cout + 31) / 32;
        a.wpk = reinterpret_cast<const float*>(L.d_wh);
        {
            static std::mutex hmu; static std::map<int, bool> hdone;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(hmu);
            if (!hdone[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_DECONV_PS>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_DECONV>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_DECONV_SIG>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_FINAL>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_FINAL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_DECONV_PS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                hdone[dev] = true;
            }
        }
        const int nb = a.ntiles_xy * a.nz;
        if (L.epi == EPI_DECONV_PS && (L.cout != 24 || y.ld != 8 || y.coff != 0))
            return fail(RIFE_HIP_EINVAL, "the PixelShuffle head kernel writes the 6-channel flow tensor [4H][4W][8] only");
        if (s16_pitch > 0 && L.epi != EPI_DECONV_PS) return fail(RIFE_HIP_EINVAL, "no S16 variant of this head");
        if (s16_pitch > 0 && fin) hipLaunchKernelGGL((head_h2_kernel<EPI_FINAL, true>), dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, *fin);
        else if (s16_pitch > 0) hipLaunchKernelGGL((head_h2_kernel<EPI_DECONV_PS, true>), dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, FinalArgs{});
        else if (fin && L.epi == EPI_DECONV_PS) hipLaunchKernelGGL(head_h2_kernel<EPI_FINAL>, dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, *fin);
        else if (L.epi == EPI_DECONV_PS) hipLaunchKernelGGL(head_h2_kernel<EPI_DECONV_PS>, dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, FinalArgs{});
        else if (L.epi == EPI_DECONV) hipLaunchKernelGGL(head_h2_kernel<EPI_DECONV>, dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, FinalArgs{});
        else hipLaunchKernelGGL(head_h2_kernel<EPI_DECONV_SIG>, dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, FinalArgs{});
        hipError_t eh = hipGetLastError();
        if (eh != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("head_h2 launch: ") + hipGetErrorString(eh));
        return 0;
    }
    // trunk layers: split-f16 matrix path (fp32-grade accuracy at 8x the fp32 MFMA rate) unless RIFE_HIP_TRUNK=f32
    if (L.nchunksh > 0 && L.stride == 1 && g_trunk_h2 && res == nullptr) {
        a.ntiles_xy = a.tiles_x * ((a.Ho + 7) / 8);
        a.nchunks = L.nchunksh;
        a.wpk = reinterpret_cast<const float*>(L.d_wh);
        const int nb = a.ntiles_xy * a.nz;
        constexpr int l29 = convh2_lds_bytes<2, 9>(), l210 = convh2_lds_bytes<2, 10>(), l39 = convh2_lds_bytes<3, 9>(), l310 = convh2_lds_bytes<3, 10>();
        {
            static std::mutex amu; static std::map<int, bool> done;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(amu);
            if (!done[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<2, 9, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, l29));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<2, 10, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, l210));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<2, 10, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, l210));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<3, 9, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, l39));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<3, 10, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, l310));
                done[dev] = true;
            }
        }
        constexpr int lb9 = convh2b_lds_bytes<2, 9>(), lb10 = convh2b_lds_bytes<2, 10>();
        {
            static std::mutex bmu; static std::map<int, bool> bdone;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(bmu);
            if (!bdone[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 9, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lb9));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<1, 9, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, convh2b_lds_bytes<1, 9>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<1, 10, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, convh2b_lds_bytes<1, 10>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 10, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lb10));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 10, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lb10));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 10, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (convh2b_lds_bytes<2, 10, 4>())));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 9, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (convh2b_lds_bytes<2, 9, 4>())));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<3, 10, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (convh2b_lds_bytes<3, 10, 4>())));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<3, 9, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (convh2b_lds_bytes<3, 9, 4>())));
                bdone[dev] = true;
            }
        }
        constexpr int lb19 = convh2b_lds_bytes<1, 9>();
        // split-K only for layers with a handful of workgroups (<= 64, i.e. under a quarter of the CUs): measured +35 % on the
        // 1080p block-0 trunk (30 workgroups); above that the partial-sum traffic and the extra launch eat the gain
        int nsplit = 1;
        if (g_splitk && g_h2b && L.NS == 2 && nb <= 64 && a.nchunks >= 4 && !in1 && !y2) nsplit = std::min(4, a.nchunks / 2);      // round-5 A/B of 2 / 8 slices and of the 64-workgroup limit: no change
        int nbl = nb;
        if (nsplit > 1) {
            a.nsplit = nsplit; a.cpad = L.ntiles * L.NS * 32;
            a.partial = splitk_workspace(st, (size_t)nsplit * a.Ho * a.Wo * a.cpad);
            if (!a.partial) return fail(RIFE_HIP_EHIP, "split-K workspace allocation failed");
            nbl = nb * nsplit;
        }
        // 4-row tiles (4 waves, three workgroups per CU) for layers whose 8-row tiles would occupy only part of the chip: twice the
        // workgroups, half the latency of each (below 400 8-row workgroups: round-1 A/B)
        constexpr int rows4_max = 400;
        static const int ns3_rows4 = env_int(ab_getenv("RIFE_HIP_NS3_ROWS4"), 0, INT_MIN, INT_MAX);      // A/B (round 5)
        static const int rows4_lim = env_int(ab_getenv("RIFE_HIP_ROWS4_MAX"), rows4_max, INT_MIN, INT_MAX);
        const bool rows4 = g_h2b && (L.NS == 2 || L.NS == 3) && nsplit == 1 && (nb < rows4_lim || (L.NS == 3 && ns3_rows4));
        if (rows4) {
            constexpr int l4_9 = convh2b_lds_bytes<2, 9, 4>(), l4_10 = convh2b_lds_bytes<2, 10, 4>();
            constexpr int l43_9 = convh2b_lds_bytes<3, 9, 4>(), l43_10 = convh2b_lds_bytes<3, 10, 4>();      // 96-wide N-tiles: 63 KB, two workgroups per CU
            a.ntiles_xy = a.tiles_x * ((a.Ho + 3) / 4);
            const int nb4 = a.ntiles_xy * a.nz;
            if (L.NS == 3 && L.skip) hipLaunchKernelGGL((conv_h2b_kernel<3, 10, 0, 4>), dim3(nb4, gy), dim3(256), l43_10, st, a);
            else if (L.NS == 3) hipLaunchKernelGGL((conv_h2b_kernel<3, 9, 0, 4>), dim3(nb4, gy), dim3(256), l43_9, st, a);
            else if (L.skip) hipLaunchKernelGGL((conv_h2b_kernel<2, 10, 0, 4>), dim3(nb4, gy), dim3(256), l4_10, st, a);
            else hipLaunchKernelGGL((conv_h2b_kernel<2, 9, 0, 4>), dim3(nb4, gy), dim3(256), l4_9, st, a);
            hipError_t e4 = hipGetLastError();
            if (e4 != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_h2b (4-row) launch: ") + hipGetErrorString(e4));
            return 0;
        }
        const int nb_saved = nb; (void)nb_saved;
#define nb nbl
        constexpr int lb110 = convh2b_lds_bytes<1, 10>();
        if (L.NS == 1 && L.skip) hipLaunchKernelGGL((conv_h2b_kernel<1, 10, 0>), dim3(nb, gy), dim3(512), lb110, st, a);
        else if (L.NS == 1) hipLaunchKernelGGL((conv_h2b_kernel<1, 9, 0>), dim3(nb, gy), dim3(512), lb19, st, a);
        else if (g_h2b && L.NS == 2 && L.skip && L.tag == 3) hipLaunchKernelGGL((conv_h2b_kernel<2, 10, 3>), dim3(nb, gy), dim3(512), lb10, st, a);
        else if (g_h2b && L.NS == 2 && L.skip) hipLaunchKernelGGL((conv_h2b_kernel<2, 10, 0>), dim3(nb, gy), dim3(512), lb10, st, a);
        else if (g_h2b && L.NS == 2) hipLaunchKernelGGL((conv_h2b_kernel<2, 9, 0>), dim3(nb, gy), dim3(512), lb9, st, a);
        else if (L.NS == 2 && L.skip && L.tag == 3) hipLaunchKernelGGL((conv_h2_kernel<2, 10, 3>), dim3(nb, gy), dim3(512), l210, st, a);
        else if (L.NS == 2 && L.skip) hipLaunchKernelGGL((conv_h2_kernel<2, 10, 0>), dim3(nb, gy), dim3(512), l210, st, a);
        else if (L.NS == 2) hipLaunchKernelGGL((conv_h2_kernel<2, 9, 0>), dim3(nb, gy), dim3(512), l29, st, a);
        else if (L.skip) hipLaunchKernelGGL((conv_h2_kernel<3, 10, 0>), dim3(nb, gy), dim3(512), l310, st, a);
        else hipLaunchKernelGGL((conv_h2_kernel<3, 9, 0>), dim3(nb, gy), dim3(512), l39, st, a);
#undef nb
        hipError_t eh = hipGetLastError();
        if (eh != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_h2 launch: ") + hipGetErrorString(eh));
        if (nsplit > 1) {
            const size_t npix = (size_t)a.Ho * a.Wo, n = npix * (L.cout / 4);
            hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.partial, nsplit, npix, a.cpad, L.cout, L.d_bias, L.d_slope,
                               y.p, y.ld, y.coff);
            eh = hipGetLastError();
            if (eh != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("splitk reduce launch: ") + hipGetErrorString(eh));
        }
        return 0;
    }
    if (in1) return fail(RIFE_HIP_EINVAL, "no two-tensor form of this convolution kernel");
    // layers with >= 2 full waves of 8-row tiles take the double-buffered 8-wave kernel
    if (L.nchunks8 > 0 && g_use_conv8) {
        const long wg8 = (long)a.tiles_x * ((a.Ho + 7) / 8) * a.nz;
        if (wg8 >= 448) {
            a.ntiles_xy = a.tiles_x * ((a.Ho + 7) / 8);
            a.nchunks = L.nchunks8;
            if (L.d_w8) a.wpk = L.d_w8;
            const int nb = a.ntiles_xy * a.nz;
            constexpr int lds28 = conv8_lds_bytes<2, 8>(), lds38 = conv8_lds_bytes<3, 8>();
            static_assert(lds28 <= 80 * 1024 && lds38 <= 160 * 1024, "LDS budget");
            {   // > 64 KB of dynamic LDS needs an opt-in per kernel and device
                static std::mutex amu; static std::map<int, bool> done;
                int dev = 0; (void)hipGetDevice(&dev);
                std::lock_guard<std::mutex> g(amu);
                if (!done[dev]) {
                    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma8_kernel<2, 8, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds28));
                    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma8_kernel<2, 8, 4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds28));
                    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma8_kernel<3, 8, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds38));
                    done[dev] = true;
                }
            }
            if (L.NS == 2 && L.tag == 3) hipLaunchKernelGGL((conv_mfma8_kernel<2, 8, 4, 3>), dim3(nb, gy), dim3(512), lds28, st, a);
            else if (L.NS == 2) hipLaunchKernelGGL((conv_mfma8_kernel<2, 8, 4, 0>), dim3(nb, gy), dim3(512), lds28, st, a);
            else hipLaunchKernelGGL((conv_mfma8_kernel<3, 8, 2, 0>), dim3(nb, gy), dim3(512), lds38, st, a);
            hipError_t e8 = hipGetLastError();
            if (e8 != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv8 launch: ") + hipGetErrorString(e8));
            return 0;
        }
    }
    a.ntiles_xy = a.tiles_x * ((a.Ho + 4 * MS - 1) / (4 * MS));
    const int nblocks = a.ntiles_xy * a.nz;
    hipError_t e = hipErrorInvalidValue;
#define RIFE_CFG(S_, MS_, NS_, CC_, E_, T_) \
    if (L.stride == S_ && MS == MS_ && L.NS == NS_ && L.CC == CC_ && L.epi == E_ && L.tag == T_) e = launch_cfg<S_, MS_, NS_, CC_, E_, T_>(a, nblocks, st); else
    RIFE_CFG(2, 1, 1, 8, EPI_STORE, 5)
    RIFE_CFG(1, 2, 1, 8, EPI_STORE, 5)
    RIFE_CFG(1, 1, 1, 8, EPI_STORE, 5)
    RIFE_CFG(2, 1, 1, 8, EPI_STORE, 0)
    RIFE_CFG(2, 1, 2, 8, EPI_STORE, 0)
    RIFE_CFG(2, 1, 3, 8, EPI_STORE, 0)
    RIFE_CFG(1, 2, 2, 16, EPI_STORE, 3)
    RIFE_CFG(1, 1, 2, 16, EPI_STORE, 3)
    RIFE_CFG(1, 2, 1, 16, EPI_STORE, 0)
    RIFE_CFG(1, 2, 2, 16, EPI_STORE, 0)
    RIFE_CFG(1, 2, 3, 8, EPI_STORE, 0)
    RIFE_CFG(1, 1, 1, 16, EPI_STORE, 0)
    RIFE_CFG(1, 1, 2, 16, EPI_STORE, 0)
    RIFE_CFG(1, 1, 3, 8, EPI_STORE, 0)
    RIFE_CFG(1, 2, 1, 16, EPI_DECONV_PS, 0)
    RIFE_CFG(1, 1, 1, 16, EPI_DECONV_PS, 0)
    RIFE_CFG(1, 2, 1, 16, EPI_DECONV, 0)
    RIFE_CFG(1, 2, 2, 16, EPI_DECONV, 0)
    RIFE_CFG(1, 2, 3, 8, EPI_DECONV, 0)
    RIFE_CFG(1, 1, 1, 16, EPI_DECONV, 0)
    RIFE_CFG(1, 1, 2, 16, EPI_DECONV, 0)
    RIFE_CFG(1, 1, 3, 8, EPI_DECONV, 0)
    RIFE_CFG(1, 2, 1, 16, EPI_DECONV_SIG, 0)
    RIFE_CFG(1, 1, 1, 16, EPI_DECONV_SIG, 0)
    { return fail(RIFE_HIP_ENOSYS, "no conv kernel instantiation for this layer shape"); }
#undef RIFE_CFG
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv launch: ") + hipGetErrorString(e));
    return 0;
}

// S16 tensor geometry for an H x W pixel grid (conv_t64.h): T64_TH x 32 tiles, one pixel of zero border on every side of every plane
struct S16Geom {
    int tiles_x, tiles_y, pitch, rows;
    S16Geom(int H, int W) : tiles_x((W + 31) / 32), tiles_y((H + T64_TH - 1) / T64_TH), pitch(tiles_x * 32 + 2), rows(tiles_y * T64_TH + 2) {}
    unsigned plane() const { return (unsigned)rows * pitch * 32u; }             // one [chunk][hi | lo] plane
    size_t bytes(int C) const { return (size_t)plane() * (C / 8); }
};

// compute units of the current device (cached): grid sizes of the persistent kernels and the kernel-selection thresholds below.
// tl_cu_budget > 0: the calling thread is enqueueing on a stream that owns only a PART of the chip (CU-masked stream, rife_hip_stream_create):
// persistent grids are sized for that part.
static thread_local int tl_cu_budget = 0;
static int device_cus(bool physical = false) {
    if (tl_cu_budget > 0 && !physical) return tl_cu_budget;
    int dev = 0; (void)hipGetDevice(&dev);
    static std::mutex mu; static std::map<int, int> ncu;
    std::lock_guard<std::mutex> g(mu);
    auto it = ncu.find(dev);
    if (it == ncu.end()) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        it = ncu.emplace(dev, n).first;
    }
    return it->second;
}

// one C -> C (C = 64, 96) residual trunk convolution, S16 in / S16 out, persistent workgroups (two / one per CU)
// reverse: walk the tiles last to first.  Consecutive trunk layers alternate, so that a layer starts on what its predecessor wrote
// last - still in the L2 / Infinity Cache (134 MB in + 134 MB out per 4K layer against 256 MB of cache: in one direction only the
// first rows of a layer's input were written more than a cache-full of traffic ago by the time they are read).
// RIFE_HIP_T64_LW=1: the 96-channel trunk with two loader waves (conv_t64.h, template parameter LW).  Off by default: measured equal or 2 % slower
// (4K, same call: trunk_b2 0.401 vs 0.392 - 0.396 ms per pair) - unlike in conv_rs_kernel, whose consumers also lose the weight stream and the stores
static const bool g_t64_loader_waves = env_on(ab_getenv("RIFE_HIP_T64_LW"));
static int launch_t64(const ConvLayer& L, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, bool reverse = false) {
    if (!L.d_t64) return fail(RIFE_HIP_EINVAL, "layer has no conv_t64 image");
    const int NS = t64_ns(L.cout);
    int dev = 0; (void)hipGetDevice(&dev);
    static std::mutex mu; static std::map<int, int> ncu;
    int cus;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = ncu.find(dev);
        if (it == ncu.end()) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_t64_kernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, t64_lds(2)));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_t64_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, t64_lds(3)));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_t64_kernel<2, 3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, t64_lds(3)));
            int n = 0;
            HIPCHK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
            it = ncu.emplace(dev, std::max(8, n / 8 * 8)).first;
        }
        cus = it->second;
    }
    if (tl_cu_budget > 0) cus = std::max(8, std::min(cus, tl_cu_budget) / 8 * 8);
    const S16Geom G(H, W);
    T64Args a;
    a.in = in; a.out = out; a.img = L.d_t64; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane(); a.tiles_x = G.tiles_x; a.ntiles = G.tiles_x * G.tiles_y; a.reverse = reverse ? 1 : 0;
    a.nchunks = L.cout / 16; a.nnt = L.cout / (32 * NS);
    const int nwg = std::min(t64_wg_per_cu(NS) * cus, (a.ntiles * a.nnt + 7) / 8 * 8);      // all workgroups resident at once
    if (L.cout == 64) hipLaunchKernelGGL((conv_t64_kernel<3, 2>), dim3(nwg), dim3(T64_NTHR), t64_lds(2), st, a);       // TAG: the profile class (trunk_b3 .. trunk_b0)
    else if (L.cout == 96 && g_t64_loader_waves) hipLaunchKernelGGL((conv_t64_kernel<2, 3, 2>), dim3(nwg), dim3(T64_NTHR + 128), t64_lds(3), st, a);      // two loader waves
    else if (L.cout == 96) hipLaunchKernelGGL((conv_t64_kernel<2, 3>), dim3(nwg), dim3(T64_NTHR), t64_lds(3), st, a);
    else return fail(RIFE_HIP_EINVAL, "conv_t64 serves 64 and 96 channels");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_t64 launch: ") + hipGetErrorString(e));
    return 0;
}

// the same 64 -> 64 layer on the row-streaming kernel (conv_rs.h): one workgroup per CU, specialised waves.  descend: walk every
// workgroup's range bottom-up; consecutive layers alternate so that a layer starts on the rows its predecessor wrote last.
static const bool g_rs_split = env_on(ab_getenv("RIFE_HIP_RS_SPLIT"));      // A/B: epilogue shared by all four io waves (conv_rs.h, SPLIT)
static int launch_rs(const ConvLayer& L, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, bool descend = false) {
    if (!L.d_t64 || L.cout != 64) return fail(RIFE_HIP_EINVAL, "layer has no 64-channel conv_t64 image");
    if ((H + 1) / 2 < RS_MIN_PAIRS) return fail(RIFE_HIP_EINVAL, "conv_rs needs at least " + std::to_string(2 * RS_MIN_PAIRS - 1) + " rows");
    int dev = 0; (void)hipGetDevice(&dev);
    static std::mutex mu; static std::map<int, int> ncu;
    int cus;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = ncu.find(dev);
        if (it == ncu.end()) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS));
            int n = 0;
            HIPCHK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
            it = ncu.emplace(dev, std::max(1, n)).first;
        }
        cus = it->second;
    }
    if (tl_cu_budget > 0) cus = std::min(cus, tl_cu_budget);
    const S16Geom G(H, W);
    RsArgs a;
    a.in = in; a.out = out; a.img = L.d_t64; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane();
    a.npairs = (H + 1) / 2; a.nunits = G.tiles_x * a.npairs; a.descend = descend ? 1 : 0;
    const int nwg = std::min(cus, a.nunits);                             // one workgroup per CU (154 KB of LDS), all resident
    if (g_rs_split) hipLaunchKernelGGL((conv_rs_kernel<0, 1>), dim3(nwg), dim3(RS_NTHR), RS_LDS, st, a);
    else hipLaunchKernelGGL((conv_rs_kernel<0>), dim3(nwg), dim3(RS_NTHR), RS_LDS, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_rs launch: ") + hipGetErrorString(e));
    return 0;
}

// TWO consecutive 64 -> 64 layers in one launch of the depth-fused row-streaming kernel (conv_rs2.h): layer A's rows stay in LDS.  Strips of 30 columns,
// every strip cut into kparts equal row ranges so that every CU of the (part of the) chip has one segment.  rs2_applies: false where the fused form
// does not apply - fewer than RS2_MIN_ROWS rows per segment, or a tensor of 2 GB and more (signed 32-bit DMA offsets) - and the caller runs two
// conv_rs launches instead: the bytes are the same either way.
static int rs2_plan(int H, int W, int cus, int& kparts, int& nstrips) {
    nstrips = (W + RS2_SW - 1) / RS2_SW;
    kparts = std::max(1, cus / nstrips);
    kparts = std::min(kparts, std::max(1, H / RS2_MIN_ROWS));
    return H / kparts;                                                  // rows of the shortest segment
}
static int rs2_cus() {
    int dev = 0; (void)hipGetDevice(&dev);
    static std::mutex mu; static std::map<int, int> ncu;
    int cus;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = ncu.find(dev);
        if (it == ncu.end()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, RS2_LDS) != hipSuccess) return 0;
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
            it = ncu.emplace(dev, std::max(1, n)).first;
        }
        cus = it->second;
    }
    if (tl_cu_budget > 0) cus = std::min(cus, tl_cu_budget);
    return cus;
}
static bool rs2_applies(int H, int W) {
    int kparts, nstrips;
    const int cus = rs2_cus();
    return cus > 0 && S16Geom(H, W).bytes(64) < (1ull << 31) && rs2_plan(H, W, cus, kparts, nstrips) >= RS2_MIN_ROWS;
}
static int launch_rs2(const ConvLayer& LA, const ConvLayer& LB, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, bool descend = false) {
    if (!LA.d_t64 || LA.cout != 64 || !LB.d_t64 || LB.cout != 64) return fail(RIFE_HIP_EINVAL, "layer has no 64-channel conv_t64 image");
    const S16Geom G(H, W);
    const int cus = rs2_cus();
    int kparts, nstrips;
    if (cus <= 0 || G.bytes(64) >= (1ull << 31) || rs2_plan(H, W, cus, kparts, nstrips) < RS2_MIN_ROWS) return fail(RIFE_HIP_EINVAL, "conv_rs2 does not apply to this tensor");
    Rs2Args a;
    a.in = in; a.out = out; a.imgA = LA.d_t64; a.imgB = LB.d_t64; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane(); a.rowmax = G.pitch - 2;
    a.kparts = kparts; a.nseg = nstrips * kparts; a.descend = descend ? 1 : 0; a.limit = (int)(G.bytes(64) - 16);
    const int nwg = std::min(cus, a.nseg);                               // one workgroup per CU (all of its LDS), all resident
    hipLaunchKernelGGL((conv_rs2_kernel<0>), dim3(nwg), dim3(RS2_NTHR), RS2_LDS, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_rs2 launch: ") + hipGetErrorString(e));
    return 0;
}

// one C -> C (C = 128, 192) residual trunk convolution of a coarse block, S16 in / S16 out: one workgroup per ROWS x 32 pixels (conv_row.h)
// nb > 0: one launch for the tensors inb[k] -> outb[k] of nb pairs in flight (gridDim.y = nb)
static int launch_row(const ConvLayer& L, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, int nb = 0,
                      const unsigned char* const* inb = nullptr, unsigned char* const* outb = nullptr) {
    const unsigned char* const rimg = L.cout == 96 ? L.d_row : L.d_t64;
    if (!rimg) return fail(RIFE_HIP_EINVAL, "layer has no conv_row image");
    {
        int dev = 0; (void)hipGetDevice(&dev);
        static std::mutex mu; static std::map<int, bool> done;
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_row_kernel<192, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (convrow_lds_bytes<192, 1>())));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_row_kernel<128, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (convrow_lds_bytes<128, 2>())));
            done[dev] = true;
        }
    }
    const S16Geom G(H, W);
    RowArgs a;
    a.in = in; a.out = out; a.img = rimg; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane(); a.tiles_x = G.tiles_x;
    if (nb > 4) return fail(RIFE_HIP_EINVAL, "conv_row batches at most four pairs");
    a.nb = nb;
    for (int k = 0; k < nb; k++) { a.inb[k] = inb[k]; a.outb[k] = outb[k]; }
    const unsigned gy = nb > 0 ? nb : 1;
    if (L.cout == 192) { a.ntiles = a.tiles_x * H; hipLaunchKernelGGL((conv_row_kernel<192, 1, 0>), dim3(a.ntiles, gy), dim3(384), (convrow_lds_bytes<192, 1>()), st, a); }
    else if (L.cout == 128) {
        a.ntiles = a.tiles_x * ((H + 1) / 2);
        hipLaunchKernelGGL((conv_row_kernel<128, 2, 1>), dim3(a.ntiles, gy), dim3(256), (convrow_lds_bytes<128, 2>()), st, a);
    }
    else if (L.cout == 96) { a.ntiles = a.tiles_x * ((H + 1) / 2); hipLaunchKernelGGL((conv_row_kernel<96, 2, 2>), dim3(a.ntiles, gy), dim3(192), (convrow_lds_bytes<96, 2, 2>()), st, a); }
    else return fail(RIFE_HIP_EINVAL, "conv_row serves 96, 128 and 192 channels");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_row launch: ") + hipGetErrorString(e));
    return 0;
}

#ifdef RIFE_HIP_TEST_BUILD
// the same coarse-block layers on the weight-stationary K-split kernel (conv_ks.h; round 4): C = 128 (block 1) and C = 96 (block 2).
// RIFE_HIP_KS (create time) = bit mask: 1 = 128 channels, 2 = 96 channels where conv_row served them (small grids), 4 = 96 channels at every
// size (instead of conv_t64), 0 = conv_row / conv_t64 as in round 3 (A/B, tests/test_gpu_ks.py).
template <int C, int NB, int CPW>
static int launch_ks_cfg(const unsigned char* img, const KsArgs& a0, int tiles_x, int gy, hipStream_t st) {
    using K = KsCfg<C, NB, CPW>;
    {
        int dev = 0; (void)hipGetDevice(&dev);
        static std::mutex mu; static std::map<int, bool> done;
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ks_kernel<C, NB, CPW, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS));
            done[dev] = true;
        }
    }
    KsArgs a = a0;
    a.img = img;
    // ranges per N group: one workgroup per CU (150 KB of LDS), all resident at once also when gy pairs share the launch; a multiple of the
    // strip count where possible, so that no range crosses a strip (a crossing costs a pipeline drain and refill)
    static const int div = env_int(ab_getenv("RIFE_HIP_KS_DIV"), 1, 1, 8);      // A/B: part of the chip only
    int G = std::max(1, device_cus() / (K::NG * gy * div));
    G = std::min(G, a.nunits);
    if (G >= tiles_x) G = G / tiles_x * tiles_x;
    hipLaunchKernelGGL((conv_ks_kernel<C, NB, CPW, 0>), dim3(G * K::NG, gy), dim3(K::NTHR), K::LDS, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_ks launch: ") + hipGetErrorString(e));
    return 0;
}
static bool ks_serves(int ks_mask, int C) { return (C == 128 && (ks_mask & 1)) || (C == 96 && (ks_mask & 6)); }
static int launch_ks(const ConvLayer& L, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, int nb = 0,
                     const unsigned char* const* inb = nullptr, unsigned char* const* outb = nullptr) {
    const unsigned char* const rimg = L.cout == 96 ? L.d_row : L.d_t64;
    if (!rimg) return fail(RIFE_HIP_EINVAL, "layer has no conv_row image");
    if (nb > 4) return fail(RIFE_HIP_EINVAL, "conv_ks batches at most four pairs");
    const S16Geom G(H, W);
    KsArgs a;
    a.in = in; a.out = out; a.img = rimg; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane(); a.nunits = G.tiles_x * H; a.skip = 1;
    a.nb = nb;
    for (int k = 0; k < nb; k++) { a.inb[k] = inb[k]; a.outb[k] = outb[k]; }
    const int gy = nb > 0 ? nb : 1;
    if (L.cout == 128) return launch_ks_cfg<128, 2, 2>(rimg, a, G.tiles_x, gy, st);
    if (L.cout == 96) return launch_ks_cfg<96, 3, 2>(rimg, a, G.tiles_x, gy, st);
    return fail(RIFE_HIP_EINVAL, "conv_ks serves 96 and 128 channels");
}
#else      // product: no conv_ks
static inline bool ks_serves(int, int) { return false; }
static inline int launch_ks(const ConvLayer&, const unsigned char*, unsigned char*, int, int, hipStream_t, int = 0, const unsigned char* const* = nullptr, unsigned char* const* = nullptr) {
    return fail(RIFE_HIP_ENOSYS, "conv_ks is not part of the product build");
}
#endif

// ------------------------------------------------------------------------------------------------
// profiler (rife_hip_profile_*): HIP events on the launch stream around every kernel
// ------------------------------------------------------------------------------------------------
}  // namespace rife
#include "graph_exec.h"
namespace rife {

struct Profiler {
    bool on = false;
    std::mutex mu;
    struct Rec { int cls; hipEvent_t e0, e1; double flops; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    std::vector<std::string> names;
    std::map<std::string, int> ids;
    std::vector<double> ms, flops;
    std::vector<long long> launches;
    int cls_id(const std::string& n) {
        auto it = ids.find(n);
        if (it != ids.end()) return it->second;
        int id = (int)names.size();
        names.push_back(n); ids[n] = id; ms.push_back(0); flops.push_back(0); launches.push_back(0);
        return id;
    }
    void begin(const std::string& cls, double fl, hipStream_t st, size_t& token) {
        token = (size_t)-1;
        if (!on) return;
        std::lock_guard<std::mutex> g(mu);
        Rec r; r.cls = cls_id(cls); r.flops = fl;
        if (pool.size() >= 2) { r.e0 = pool.back(); pool.pop_back(); r.e1 = pool.back(); pool.pop_back(); }   // events are recycled
        else if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        (void)hipEventRecord(r.e0, st);
        recs.push_back(r); token = recs.size() - 1;
    }
    void end(size_t token, hipStream_t st) {
        if (token == (size_t)-1) return;
        std::lock_guard<std::mutex> g(mu);
        (void)hipEventRecord(recs[token].e1, st);
    }
    void collect() {
        std::lock_guard<std::mutex> g(mu);
        for (Rec& r : recs) {
            (void)hipEventSynchronize(r.e1);
            float t = 0.f;
            if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) { ms[r.cls] += t; flops[r.cls] += r.flops; launches[r.cls]++; }
            pool.push_back(r.e0); pool.push_back(r.e1);
        }
        recs.clear();
    }
};

// ------------------------------------------------------------------------------------------------
// per-pair workspace ("context"): everything one in-flight frame pair needs, sized for one padded resolution
// ------------------------------------------------------------------------------------------------
struct Ctx {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_budget = 0;                                                // > 0: `stream` owns this many compute units only (pool partition, RIFE_HIP_POOL_PARTS)
    std::mutex use;                                                   // rife_hip_process_device: one caller at a time per stream workspace
    int w = 0, h = 0, wp = 0, hp = 0;
    uint8_t *d_in0 = nullptr, *d_in1 = nullptr, *d_out = nullptr;   // staging for the host-buffer entry point
    uint32_t *img0 = nullptr, *img1 = nullptr;                       // padded RGBX u8
    float *X = nullptr, *S1 = nullptr, *T0 = nullptr, *T1 = nullptr; // block input, stem-1 output, trunk ping/pong
    float *T2 = nullptr;                                             // rife-v4 (4.0): stem-1 output kept for the block's residual add
    unsigned char* P[4][2] = {};                                     // per block: trunk ping / pong as S16 tensors (conv_t64.h), zero borders
    float* flow[4] = {nullptr, nullptr, nullptr, nullptr};           // [hp/s][wp/s][8]
    float4* F = nullptr; float* M = nullptr;                         // full-resolution flow (4ch) and mask logit
    float4* F2 = nullptr; float* M2 = nullptr;                       // the other pair of buffers for a flow update fused into the next stem (stem_fused.h UPD); F, M are swapped with them
    float4* outf = nullptr;                                          // TTA only: out0 as float, padded
    // hipGraph replay of the plain v4 schedule for launch-bound frame sizes: fixed staging buffers (d_in0 / d_in1 / d_out), the
    // timestep in device memory, one warm-up pass (lazy allocations, kernel attributes), then capture once and replay
    float* d_ts = nullptr;
    hipEvent_t ev_group = nullptr;                                    // rife_hip_process_batch: cross-stream hand-off around a batched coarse trunk
    hipGraphExec_t gexec = nullptr;
    bool g_warm = false;
    // rife-v2.x only
    bool v2 = false;
    float4 *acc = nullptr, *D = nullptr, *head = nullptr;           // running half-res flow, deconv output, fusion head
    float4 *h0 = nullptr, *h1 = nullptr, *acc_s = nullptr;          // UHD: half-resolution fp32 frames and their (quarter-res) flow
    float *I8 = nullptr, *ca = nullptr, *cb = nullptr, *cc = nullptr, *feat[4] = {nullptr, nullptr, nullptr, nullptr}, *ctmp[3] = {nullptr, nullptr, nullptr};
    float2* fl[4] = {nullptr, nullptr, nullptr, nullptr};           // ContextNet flow pyramid
    // the second ContextNet pass (img1, flow10): its own activations, so that both passes ride one launch per layer (gridDim.y = 2)
    float *ca2 = nullptr, *cb2 = nullptr, *cc2 = nullptr, *feat2[4] = {nullptr, nullptr, nullptr, nullptr}, *ctmp2[3] = {nullptr, nullptr, nullptr};
    float2* fl2[4] = {nullptr, nullptr, nullptr, nullptr};
    float *e0a = nullptr, *e0b = nullptr, *e0c = nullptr, *B1 = nullptr, *e1a = nullptr, *B2 = nullptr, *e2a = nullptr, *B3 = nullptr, *e3a = nullptr, *B4 = nullptr;
    float *U0 = nullptr, *U1 = nullptr, *U2 = nullptr, *U3 = nullptr;
    // rife-v2.x TTA: per orientation RGBX frames, half-res flows [direction][orientation], float outputs [direction][orientation]
    uint32_t *timg0[8] = {}, *timg1[8] = {};
    float4 *tflow[2][8] = {}, *toutf[2][8] = {};
    // v1 family (generic graph executor): blob storage per net instance, one set per frame orientation (w x h / h x w for TTA);
    // [.][0] flownet, [1] / [2] contextnet of frame 0 / 1, [3] fusionnet, [4] tensors outside the nets (frames, UHD resizes, TTA flows)
    std::unique_ptr<GraphInst> ginst[2][5];
    std::vector<void*> allocs;
    ~Ctx() {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (ev_group) (void)hipEventDestroy(ev_group);
        for (void* p : allocs) (void)hipFree(p);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

template <typename T>
static int dalloc(Ctx& c, T*& p, size_t n) {
    void* v = nullptr;
    HIPCHK(hipMalloc(&v, n * sizeof(T)));
    c.allocs.push_back(v);
    p = (T*)v;
    return 0;
}

}  // namespace rife

using namespace rife;

// ------------------------------------------------------------------------------------------------
// the engine object behind rife_hip_t
// ------------------------------------------------------------------------------------------------
// Device buffers of released resident frames (rife_hip_frame_*), reused by the next upload of the same size: hipFree waits for
// the whole device, which would stall the pairs in flight every time a frame of a sequence retires.
struct FramePool {
    int gpuid = 0;
    std::mutex mu;
    std::vector<std::pair<size_t, uint8_t*>> idle;
    uint8_t* take(size_t nbytes) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].first == nbytes) { uint8_t* p = idle[i].second; idle.erase(idle.begin() + i); return p; }
        }
        uint8_t* p = nullptr;
        return hipMalloc((void**)&p, nbytes) == hipSuccess ? p : nullptr;
    }
    void give(uint8_t* p, size_t nbytes) {
        uint8_t* evict = nullptr;
        {
            std::lock_guard<std::mutex> g(mu);
            if (idle.size() >= 16) { evict = idle.front().second; idle.erase(idle.begin()); }      // oldest out: sizes may change over time
            idle.emplace_back(nbytes, p);
        }
        if (evict && hipSetDevice(gpuid) == hipSuccess) (void)hipFree(evict);
    }
    ~FramePool() {
        if (!idle.empty() && hipSetDevice(gpuid) == hipSuccess) for (auto& e : idle) (void)hipFree(e.second);
    }
};

struct rife_hip {
    int gpuid = 0;
    bool tta = false, tta_temporal = false, uhd = false, v2 = false, v4 = false;
    int num_threads = 1;
    bool loaded = false;
    // v4.x schedule: per block {stem0, stem1, res x8, head}
    struct Block { ConvLayer stem0, stem1, res[8], head; int c = 0, scale = 1; } blk[4];
    // rife-v4 (4.0) variant of the schedule: PReLU, plain trunk + one residual add, 5-channel deconv head at half the block
    // resolution (flow{b} is [hp/2s][wp/2s][8] instead of [hp/s][wp/s][8])
    bool v40 = false;
    // finest-block trunk on S16 tensors + the persistent conv_t64 kernel (RIFE_HIP_T64=0 at create time keeps conv_h2b: A/B and
    // the bit-equality test of the two trunk implementations)
    bool t64 = true;
    // block-3 trunk on the row-streaming kernel (conv_rs.h) instead of conv_t64 (RIFE_HIP_RS=0 at create time: A/B, bit-equality test)
    bool rs = true;
    // ... two layers per launch, layer A's rows LDS-resident (conv_rs2.h; RIFE_HIP_RS2=0 at create time: A/B, bit-equality test)
    bool rs2 = true;
    // coarse-block trunks on the weight-stationary K-split kernel (conv_ks.h): bit mask by channel count, see launch_ks (RIFE_HIP_KS at create time)
    int ks_mask = 0;
    // block 3: block-input assembly + both stem convolutions in one row-streaming kernel (stem_rs.h) instead of stem0_fused_kernel + conv_h2s2_kernel
    // (RIFE_HIP_STEM_RS=0 at create time: A/B, the comparison test)
    bool stem_rs = true;
    // block 3's head + the tail of the graph + postproc in one row-streaming kernel (tail_rs.h) instead of head_h2_kernel<EPI_FINAL, true>
    // (RIFE_HIP_TAIL_RS=0 at create time: A/B, the comparison test)
    bool tail_rs = true;
    bool tail_rs_always = false;      // RIFE_HIP_TAIL_RS=2: at every frame size (tests)
    // -x -z: temporal + spatial flow consensus of a block in one kernel (k_v4_consensus); RIFE_HIP_TTA_CONSENSUS=0 at create time: the two steps as
    // separate kernels (8 + 2 launches per block; A/B, the bit-identity test)
    bool tta_consensus = true;
    // RIFE_HIP_FUSE_FLOW=1 (A/B, parity taps): the flow updates after blocks 1 and 2 inside the fused stems of blocks 2 and 3 (stem_fused.h UPD)
    // instead of two k_flow_update launches.  Bit-identical, and measured SLOWER at 4K (432 vs 442 frames/s, same call): the update kernels
    // run at 6 - 7 TB/s, the stems are bound by gather latency and VALU issue and every load added to them costs more than the pass it removes
    // (stem0_b3 0.210 -> 0.285, stem0_b2 0.161 -> 0.272, flow_update 0.191 -> 0.034 ms per pair).  Off in the product.
    bool fuse_flow = false;
    int flow_div(int b) const { return v40 ? 2 * blk[b].scale : blk[b].scale; }
    // rife-v2.x schedule (IFNet + ContextNet + FusionNet)
    struct V2Block { ConvLayer stem0, stem1, conv[6], head; int c = 0, scale = 1; } fblk[4];
    // rife-v3.x: same ContextNet / FusionNet, IFNet of 3 blocks (scales 4, 2, 1; 160 channels; trunk = 3 x [conv, conv, + skip])
    bool v3 = false;
    bool prof_fine = false;                                              // RIFE_HIP_PROFILE_FINE=1: per-layer profile classes (load_v2)
    int n_fblk = 4;
    // v1 family (rife, rife-HD, rife-UHD, rife-anime): executed layer by layer from the .param (graph_exec.h)
    bool v1 = false;
    std::unique_ptr<GraphNet> gflow, gctx, gfus;
    ConvLayer ctxc[10];          // ContextNet convs in graph order
    ConvLayer fus[15];           // FusionNet: 10 down convs, 4 up deconvs, sigmoid head
    mutable Profiler prof;
    mutable std::mutex mu;
    mutable std::vector<std::unique_ptr<Ctx>> free_ctx;                  // pool for the host-buffer entry point
    mutable std::vector<hipEvent_t> batch_fork;                          // rife_hip_process_device_batch: recycled fork events
    mutable std::map<void*, int> part_streams;                           // rife_hip_stream_create: CU-masked streams of this engine -> compute units they own
    mutable std::map<void*, std::unique_ptr<Ctx>> stream_ctx;            // one workspace per caller stream
    mutable std::mutex tta_mu;                                           // TTA passes share one set of workspaces
    std::shared_ptr<FramePool> frame_pool;                               // shared with the frames: they may outlive the engine
    mutable std::vector<hipStream_t> upload_streams;                     // rife_hip_frame_upload: one copy stream per concurrent uploader
    mutable std::unique_ptr<Ctx> tta_ctx[2][8];                          // [direction][orientation]
    static constexpr int NLANE = 4;                                      // spatial TTA: orientations run on 4 worker streams
    mutable hipStream_t tta_lane[NLANE] = {nullptr, nullptr, nullptr, nullptr};
    mutable hipEvent_t tta_fork[6] = {}, tta_join[6][NLANE] = {};

    ~rife_hip() {
        (void)hipSetDevice(gpuid);
        free_ctx.clear(); stream_ctx.clear();
        for (auto& d : tta_ctx) for (auto& c : d) c.reset();
        for (auto& l : tta_lane) if (l) (void)hipStreamDestroy(l);
        for (auto& u : upload_streams) (void)hipStreamDestroy(u);
        for (auto& e : tta_fork) if (e) (void)hipEventDestroy(e);
        for (auto& e : batch_fork) if (e) (void)hipEventDestroy(e);
        for (auto& kv : part_streams) (void)hipStreamDestroy((hipStream_t)kv.first);
        for (auto& r : tta_join) for (auto& e : r) if (e) (void)hipEventDestroy(e);
        for (auto& b : blk) { free_layer(b.stem0); free_layer(b.stem1); for (auto& r : b.res) free_layer(r); free_layer(b.head); }
        for (auto& b : fblk) { free_layer(b.stem0); free_layer(b.stem1); for (auto& r : b.conv) free_layer(r); free_layer(b.head); }
        for (auto& l : ctxc) free_layer(l);
        for (auto& l : fus) free_layer(l);
    }
};

namespace rife {

// structural hashes of the graphs the schedules below were written for (= the reference's
// models/rife-v4.6/flownet.param; tests/test_models.py proves the equivalence whenever /root/reference exists)
static const uint64_t V46_HASH_OUT0 = RIFE_V46_HASH_OUT0;

static std::atomic<bool> g_fuse_flow_buffers{false};                 // some engine of the process asked for RIFE_HIP_FUSE_FLOW=1: workspaces carry F2, M2

// (Re)allocate a workspace for frames of w x h (padded wp x hp).  `scratch` != null: borrow the big per-layer
// scratch tensors (block input, stem output, trunk ping/pong) from another context of the same pixel count —
// the TTA passes run one after another on one stream, only flows / F / M / images must persist per pass.
static int ensure_ctx_dims_impl(Ctx& c, int w, int h, int wp, int hp, const Ctx* scratch, bool own_images, bool want_outf) {
    if (!c.v2 && c.wp == wp && c.hp == hp && c.w == w && c.h == h && (!want_outf || c.outf)) return 0;
    c.v2 = false;
    if (c.gexec) { (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; }
    c.g_warm = false; c.d_ts = nullptr;
    for (void* p : c.allocs) (void)hipFree(p);
    c.allocs.clear();
    c.outf = nullptr; c.F2 = nullptr; c.M2 = nullptr;
    for (auto& pb : c.P) pb[0] = pb[1] = nullptr;                       // S16 trunk tensors: allocated by the first block that runs on them (ensure_s16)
    c.w = w; c.h = h; c.wp = wp; c.hp = hp;
    const size_t P = (size_t)wp * hp;
    int rc;
    if (own_images) {
        if ((rc = dalloc(c, c.img0, P))) return rc;
        if ((rc = dalloc(c, c.img1, P))) return rc;
    }
    if (scratch) { c.X = scratch->X; c.S1 = scratch->S1; c.T0 = scratch->T0; c.T1 = scratch->T1; c.T2 = scratch->T2; }
    else {
        if ((rc = dalloc(c, c.d_in0, (size_t)w * h * 3))) return rc;
        if ((rc = dalloc(c, c.d_in1, (size_t)w * h * 3))) return rc;
        if ((rc = dalloc(c, c.d_out, (size_t)w * h * 3))) return rc;
        if ((rc = dalloc(c, c.X, P * 16))) return rc;                  // block 3: full res x 16 ch
        if ((rc = dalloc(c, c.S1, P / 4 * 32))) return rc;             // block 3 stem-0 output: (hp/2 x wp/2) x 32
        if ((rc = dalloc(c, c.T0, P / 16 * 64))) return rc;            // block 3 trunk: (hp/4 x wp/4) x 64 (the largest trunk)
        if ((rc = dalloc(c, c.T1, P / 16 * 64))) return rc;
        if ((rc = dalloc(c, c.T2, P / 16 * 64))) return rc;
    }
    static const int sc[4] = {8, 4, 2, 1};
    for (int b = 0; b < 4; b++) {
        const size_t n = P / (sc[b] * sc[b]) * 8;
        if ((rc = dalloc(c, c.flow[b], n))) return rc;
        // on the workspace's own stream, not the legacy stream: a synchronous hipMemset from one caller thread while others create
        // streams / launch on theirs makes the runtime fail intermittently ("legacy stream depend on a capturing blocking stream", then
        // every later call of the process reports a capture error) - tools/reentrancy_stress.py, ~1 in 100 concurrent calls
        if (c.stream) HIPCHK(hipMemsetAsync(c.flow[b], 0, n * 4, c.stream));
        else HIPCHK(hipMemset(c.flow[b], 0, n * 4));
    }
    if ((rc = dalloc(c, c.F, P))) return rc;
    if ((rc = dalloc(c, c.M, P))) return rc;
    if (!want_outf && !scratch && g_fuse_flow_buffers) {               // the plain pass of an engine created with RIFE_HIP_FUSE_FLOW=1 (not the TTA workspaces, whose updates go through the consensus kernels)
        if ((rc = dalloc(c, c.F2, P))) return rc;
        if ((rc = dalloc(c, c.M2, P))) return rc;
    }
    if (!scratch && (rc = dalloc(c, c.d_ts, 4))) return rc;
    if (want_outf && (rc = dalloc(c, c.outf, P))) return rc;
    return 0;
}

// A workspace whose (re)allocation failed half way is emptied, so that the next call reports the error again instead of taking the
// "already sized" early return and running on freed memory.
static void reset_ctx(Ctx& c) {
    if (c.gexec) { (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; }
    for (void* p : c.allocs) (void)hipFree(p);
    c.allocs.clear();
    c.w = c.h = c.wp = c.hp = 0; c.v2 = false; c.outf = nullptr; c.F2 = nullptr; c.M2 = nullptr; c.d_ts = nullptr; c.g_warm = false; for (auto& pb : c.P) pb[0] = pb[1] = nullptr;
}
static int ensure_ctx_dims(Ctx& c, int w, int h, int wp, int hp, const Ctx* scratch = nullptr, bool own_images = true, bool want_outf = false) {
    const int rc = ensure_ctx_dims_impl(c, w, h, wp, hp, scratch, own_images, want_outf);
    if (rc) reset_ctx(c);
    return rc;
}

static int ensure_ctx(Ctx& c, int w, int h) {
    return ensure_ctx_dims(c, w, h, (w + 31) / 32 * 32, (h + 31) / 32 * 32);   // pad to 32n, rife.cpp:2499-2500
}

struct Timed {
    Profiler& p; hipStream_t st; size_t tok;
    Timed(Profiler& p_, const std::string& cls, double fl, hipStream_t s) : p(p_), st(s) { p.begin(cls, fl, st, tok); }
    ~Timed() { p.end(tok, st); }
};

static inline dim3 grid2d(int w, int h) { return dim3((w + 255) / 256, h); }
// tiles of the kernels that touch all eight TTA orientations of a plane: a wave = 8 columns x 8 rows (32-byte and 16-byte elements: 256- / 128-byte
// runs in the straight AND in the transposed buffers) or 16 x 4 rows of a 16 x 16 block (4-byte elements: 64-byte runs both ways)
static inline dim3 tta_block(int elem_bytes) { return elem_bytes >= 16 ? dim3(8, 32) : dim3(16, 16); }
static inline dim3 tta_grid(int w, int h, int elem_bytes) { const dim3 b = tta_block(elem_bytes); return dim3((w + b.x - 1) / b.x, (h + b.y - 1) / b.y); }
// rife_preproc.comp: u8 HWC RGB -> zero-padded RGBX; four pixels per lane when the frame allows 4-byte loads
static inline void launch_preproc(hipStream_t st, const uint8_t* rgb, int w, int h, uint32_t* out, int wp, int hp) {
    if ((w & 3) == 0 && (reinterpret_cast<uintptr_t>(rgb) & 3) == 0) hipLaunchKernelGGL(k_preproc4, dim3((wp / 4 + 255) / 256, hp), dim3(256), 0, st, rgb, w, h, out, wp, hp);
    else hipLaunchKernelGGL(k_preproc, grid2d(wp, hp), dim3(256), 0, st, rgb, w, h, out, wp, hp);
}

}  // namespace rife
#include "graph_run.h"
namespace rife {

static int run_assemble(const rife_hip& E, Ctx& c, int b, float timestep, const float* tsp = nullptr) {
    hipStream_t st = c.stream;
    Timed t(E.prof, "assemble", 0, st);
    const int s = E.blk[b].scale;
    dim3 g = grid2d(c.wp / s, c.hp / s);
    if (b == 0) hipLaunchKernelGGL(k_assemble0, g, dim3(256), 0, st, c.img0, c.img1, timestep, tsp, c.X, c.wp, c.hp);
    else if (s == 4) hipLaunchKernelGGL(k_assemble<4>, g, dim3(256), 0, st, c.img0, c.img1, timestep, tsp, c.F, c.M, c.X, c.wp, c.hp);
    else if (s == 2) hipLaunchKernelGGL(k_assemble<2>, g, dim3(256), 0, st, c.img0, c.img1, timestep, tsp, c.F, c.M, c.X, c.wp, c.hp);
    else hipLaunchKernelGGL(k_assemble<1>, g, dim3(256), 0, st, c.img0, c.img1, timestep, tsp, c.F, c.M, c.X, c.wp, c.hp);
    HIPCHK(hipGetLastError());
    return 0;
}

// block 3 of rife-v4.6: frames + F, M -> the first S16 trunk tensor (stem_rs.h); two workgroups per CU, all resident
static int launch_stem_rs(const rife_hip& E, Ctx& c, const rife_hip::Block& B, unsigned char* out, int Hq, int Wq, float timestep, const float* tsp) {
    {
        int dev = 0; (void)hipGetDevice(&dev);
        static std::mutex mu; static std::map<int, bool> done;
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_rs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, SRS_LDS));
            done[dev] = true;
        }
    }
    const S16Geom G(Hq, Wq);
    StemRsArgs a;
    a.img0 = c.img0; a.img1 = c.img1; a.F = c.F; a.M = c.M;
    a.w0 = B.stem0.d_wh; a.bias0 = B.stem0.d_bias; a.slope0 = B.stem0.d_slope;
    a.w1 = B.stem1.d_whp; a.bias1 = B.stem1.d_bias; a.slope1 = B.stem1.d_slope;
    a.out = out; a.timestep = timestep; a.tsp = tsp; a.wp = c.wp; a.hp = c.hp; a.Hq = Hq; a.Wq = Wq; a.pitch = G.pitch; a.plane = G.plane();
    a.nunits = ((Wq + SRS_SW - 1) / SRS_SW) * Hq;
    const int nwg = std::min(2 * device_cus(), a.nunits);
    hipLaunchKernelGGL((stem_rs_kernel<0>), dim3(nwg), dim3(SRS_NTHR), SRS_LDS, c.stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("stem_rs launch: ") + hipGetErrorString(e));
    return 0;
}
// block 3 of rife-v4.6: last S16 trunk tensor + F, M + frames -> u8 frame (tail_rs.h); two workgroups per CU, all resident
static int launch_tail_rs(const rife_hip::Block& B, const unsigned char* in, int Hq, int Wq, const FinalArgs& fin, hipStream_t st) {
    {
        int dev = 0; (void)hipGetDevice(&dev);
        static std::mutex mu; static std::map<int, bool> done;
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(tail_rs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, TRS_LDS));
            done[dev] = true;
        }
    }
    const S16Geom G(Hq, Wq);
    TailRsArgs a;
    a.in = in; a.w = B.head.d_wh; a.bias = B.head.d_bias; a.img0 = fin.img0; a.img1 = fin.img1; a.F = fin.F; a.M = fin.M; a.out = fin.out;
    a.w_ = fin.w; a.h_ = fin.h; a.wp = fin.wp; a.hp = fin.hp; a.Hq = Hq; a.Wq = Wq; a.pitch = G.pitch; a.plane = G.plane();
    a.nunits = ((Wq + 31) / 32) * Hq;
    const int nwg = std::min(2 * device_cus(), a.nunits);
    hipLaunchKernelGGL((tail_rs_kernel<0>), dim3(nwg), dim3(TRS_NTHR), TRS_LDS, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("tail_rs launch: ") + hipGetErrorString(e));
    return 0;
}

// can block b's two stems run as one stem_rs launch?  (64-channel block 3 at scale 1 on the S16 trunk, 12 -> 32 -> 64 channels, uniform shapes)
static bool block_on_stem_rs(const rife_hip& E, const Ctx& c, int b) {
    const rife_hip::Block& B = E.blk[b];
    return E.stem_rs && b == 3 && B.scale == 1 && B.c == 64 && B.stem0.d_wh && B.stem0.cout == 32 && B.stem1.d_whp && B.stem1.cout == 64 &&
           g_trunk_h2 && g_fuse_stem && (c.hp % 4) == 0 && (c.wp % 4) == 0 &&
           (long long)c.wp * c.hp <= (1ll << 27);                       // the kernel addresses F (16 B per pixel) with 32-bit byte offsets; larger frames take the tile stems
}

// One IFBlock: stems, 8 residual convs, head -> flow[b]   (flownet.param:11-46, 63-98, 116-151, 166-201)
// which trunk kernel serves block b at this frame size: 0 = conv_t64 / conv_rs (fine blocks), 1 = conv_row (coarse blocks, small grids)
static bool block_on_row_kernel(const rife_hip& E, const Ctx& c, int b) {
    const rife_hip::Block& B = E.blk[b];
    const int s = B.scale, Ht = c.hp / s / 4, Wt = c.wp / s / 4;
    const int ptiles = ((Ht + 7) / 8) * ((Wt + 31) / 32), cus = device_cus(true);      // kernel selection never depends on a CU partition: same bytes on every stream
    const bool row_small = b == 2 && B.c == 96 && (ptiles <= cus || (E.ks_mask & 4));          // fewer 8 x 32 tiles than the chip has CUs (or conv_ks at every size)
    return (b == 1 && B.c == 128) || (b == 0 && B.c == 192 && ((Wt + 31) / 32) * Ht <= cus * 5 / 8) || row_small;      // MI355X: 160 of 256
}

// Does block b run on S16 trunk tensors (conv_rs / conv_t64 / conv_row) at this frame size?  Blocks 3 / 2 on the persistent kernels, the coarse
// blocks on the row kernel where block_on_row_kernel says so; never for rife-v4 (4.0), RIFE_HIP_T64=0, or a tensor of 4 GB and more (the
// kernels address S16 tensors with 32-bit byte offsets).
static bool block_on_s16(const rife_hip& E, const Ctx& c, int b) {
    const rife_hip::Block& B = E.blk[b];
    const int s = B.scale, Ht = c.hp / s / 4, Wt = c.wp / s / 4;
    const bool rowk = block_on_row_kernel(E, c, b);
    bool s16 = E.t64 && !E.v40 && g_trunk_h2 && B.stem1.d_whp && B.head.d_wh && B.head.epi == EPI_DECONV_PS &&
               ((b == 3 && B.c == 64) || (b == 2 && B.c == 96) || rowk);
    for (int i = 0; i < 8 && s16; i++) s16 = B.res[i].d_t64 != nullptr && (!(rowk && B.c == 96) || B.res[i].d_row != nullptr);
    return s16 && (unsigned long long)S16Geom(Ht, Wt).bytes(B.c) < (1ull << 32);
}
// the block's two S16 tensors (trunk ping / pong), allocated on first use with their zero borders: a workspace only carries the tensors of
// the blocks that really run on the S16 kernels (TTA: 16 workspaces)
static int ensure_s16(Ctx& c, int b, int Ht, int Wt, int C) {
    if (c.P[b][0] && c.P[b][1]) return 0;
    const size_t nb = S16Geom(Ht, Wt).bytes(C);
    int rc;
    if (c.stream) {      // a lazy hipMalloc inside a hipGraph capture would be illegal: the warm-up pass before a capture allocates everything (run_v4_replay)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c.stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(RIFE_HIP_EHIP, "S16 trunk tensors requested while the stream is capturing");
    }
    for (int k = 0; k < 2; k++) {
        if ((rc = dalloc(c, c.P[b][k], nb))) {
            if (k == 1) {                                                // the first buffer goes back: it is the newest entry of the workspace's allocation list
                if (!c.allocs.empty() && c.allocs.back() == (void*)c.P[b][0]) c.allocs.pop_back();
                (void)hipFree(c.P[b][0]);
            }
            c.P[b][0] = c.P[b][1] = nullptr;
            return rc;
        }
        if (c.stream) HIPCHK(hipMemsetAsync(c.P[b][k], 0, nb, c.stream));
        else HIPCHK(hipMemset(c.P[b][k], 0, nb));
    }
    return 0;
}

enum { PH_STEMS = 1, PH_TRUNK = 2, PH_HEAD = 4, PH_ALL = 7 };
// phases != PH_ALL (rife_hip_process_batch): the S16 path only; PH_TRUNK is then the caller's batched launch
// Can the flow update after block b - 1 be left to block b's fused stem (stem_fused.h UPD)?  Blocks 2 and 3 of rife-v4.6 only: their stems
// visit every full-resolution pixel.
static bool flow_update_fused_into(const rife_hip& E, const Ctx& c, int b) {
    return E.fuse_flow && !E.v40 && (b == 2 || b == 3) && c.F2 && E.blk[b].stem0.d_wh && g_trunk_h2 && g_fuse_stem;
}

// upd_flow != null: the flow of block b - 1, whose update of F, M this block's stem applies itself (flow_update_fused_into); F, M swap with F2, M2
static int run_block_convs(const rife_hip& E, Ctx& c, int b, float timestep, const FinalArgs* fin = nullptr, const float* tsp = nullptr, int phases = PH_ALL,
                           const float* upd_flow = nullptr, const float* first_flow = nullptr) {
    const rife_hip::Block& B = E.blk[b];
    hipStream_t st = c.stream;
    const int s = B.scale, Hb = c.hp / s, Wb = c.wp / s;
    const int xin_ld = b == 0 ? 8 : 16;
    int rc;
    // block 3: one row-streaming kernel for the assembly and both stems (stem_rs.h), launched where stem 1 used to be
    const bool srs = !upd_flow && block_on_stem_rs(E, c, b) && block_on_s16(E, c, b);
    if (!(phases & PH_STEMS) || srs) goto after_stem0;
    if (b == 0 && (rc = run_assemble(E, c, 0, timestep, tsp))) return rc;
    if (b > 0 && B.stem0.d_wh && g_trunk_h2 && g_fuse_stem) {
        // assemble + stem-0 in one kernel (stem_fused.h): the block input never goes to HBM
        Timed t(E.prof, B.stem0.cls, B.stem0.flops_per_pixel * (Hb / 2) * (Wb / 2), st);
        StemFusedArgs fa;
        fa.img0 = c.img0; fa.img1 = c.img1; fa.F = c.F; fa.M = c.M; fa.wpk = B.stem0.d_wh; fa.bias = B.stem0.d_bias; fa.slope = B.stem0.d_slope;
        fa.out = c.S1; fa.timestep = timestep; fa.tsp = tsp; fa.wp = c.wp; fa.hp = c.hp; fa.Ho = Hb / 2; fa.Wo = Wb / 2; fa.out_ld = B.c / 2; fa.Cout = B.c / 2;
        fa.tiles_x = (fa.Wo + 31) / 32;
        const int nb = fa.tiles_x * ((fa.Ho + 3) / 4);
        {
            static std::mutex fmu; static std::map<int, bool> fdone;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(fmu);
            if (!fdone[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<2, 2, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<4, 2, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
                fdone[dev] = true;
            }
        }
        if (upd_flow) {
            if (s > 2 || !c.F2) return fail(RIFE_HIP_EINVAL, "no fused flow update for this block");
            fa.pend.flow = upd_flow; fa.pend.Fw = c.F2; fa.pend.Mw = c.M2;
            if (s == 2) hipLaunchKernelGGL((stem0_fused_kernel<2, 2, 0, true>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
            else hipLaunchKernelGGL((stem0_fused_kernel<1, 1, 256, true>), dim3(nb), dim3(512), (stemf_lds_bytes<1, 256>()), st, fa);
            std::swap(c.F, c.F2); std::swap(c.M, c.M2);
        } else if (first_flow) {      // block 1 right after block 0: F, M are not materialised yet, the stem samples the first update itself (first_flow_merged)
            if (s != 4) return fail(RIFE_HIP_EINVAL, "the first flow update is sampled by the scale-4 stem only");
            fa.pend.flow = first_flow;
            hipLaunchKernelGGL((stem0_fused_kernel<4, 2, 0, 2>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
        } else if (s == 4) hipLaunchKernelGGL((stem0_fused_kernel<4, 2>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
        else if (s == 2) hipLaunchKernelGGL((stem0_fused_kernel<2, 2>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
        else hipLaunchKernelGGL((stem0_fused_kernel<1, 1, 256>), dim3(nb), dim3(512), (stemf_lds_bytes<1, 256>()), st, fa);      // 64-byte swizzled records, three workgroups per CU
        HIPCHK(hipGetLastError());
    } else {
        if (upd_flow || first_flow) return fail(RIFE_HIP_EINVAL, "fused flow update without the fused stem");
        if (b > 0 && (rc = run_assemble(E, c, b, timestep, tsp))) return rc;
        Timed t(E.prof, B.stem0.cls, B.stem0.flops_per_pixel * (Hb / 2) * (Wb / 2), st);
        if ((rc = launch_conv(B.stem0, {c.X, xin_ld, 0}, Hb, Wb, {c.S1, B.c / 2, 0}, nullptr, st))) return rc;
    }
after_stem0:
    FinalArgs fin_now;                                                   // the fused tail reads the F, M that are current AFTER this block's stem
    if (fin) { fin_now = *fin; fin_now.F = c.F; fin_now.M = c.M; fin = &fin_now; }
    const int Ht = Hb / 4, Wt = Wb / 4;
    // S16 trunk tensors: blocks 3 / 2 on the persistent LDS-DMA kernel (conv_t64.h; block 2 only when its grid fills a good part of the
    // chip), the coarse blocks 1 / 0 on the one-pass row kernel (conv_row.h).  (Blocks 1 / 0 as N-tiles of 64 output channels on the
    // persistent kernel were measured too: 4K trunk_b1 0.300 vs 0.285 ms per pair, trunk_b0 0.228 vs 0.206 - a chain of 8 - 12 dependent
    // steps whose fixed cost exceeds a step's matrix work at these sizes.)
    // block 0 on the row kernel only while its grid is small: at 4K all 272 workgroups stream the same 663 KB of weights through the L2 at
    // once (0.239 vs 0.208 ms per pair for the per-tile kernel), at 1080p (68 workgroups) it wins (0.133 vs 0.152); block 1 wins at both
    // block 2 on small grids (<= 256 tiles of 8 x 32: fewer tiles than CUs): the persistent kernel (one workgroup per CU for 96 channels) has at most one
    // tile per workgroup there and fills only part of the chip: 1080p (136 tiles) trunk_b2 0.229 -> 0.179 ms per pair on the row kernel, 4K (510 tiles)
    // 0.387 -> 0.401; block 3 (64 channels, two workgroups per CU) stays on the persistent kernel at every size (1080p 0.225 vs 0.233)
    const bool rowk = block_on_row_kernel(E, c, b);
    if (block_on_s16(E, c, b)) {
        if ((rc = ensure_s16(c, b, Ht, Wt, B.c))) return rc;
        unsigned char* const PA = c.P[b][0];
        unsigned char* const PB = c.P[b][1];
        // stem-1 writes the first S16 tensor, eight persistent trunk launches ping-pong between the two, the head reads the last one
        const S16Geom G(Ht, Wt);
        if ((phases & PH_STEMS) && srs) {
            Timed t(E.prof, "stems_b3", B.stem0.flops_per_pixel * (Hb / 2) * (Wb / 2) + B.stem1.flops_per_pixel * Ht * Wt, st);
            if ((rc = launch_stem_rs(E, c, B, PA, Ht, Wt, timestep, tsp))) return rc;
        } else if (phases & PH_STEMS) {
            Timed t(E.prof, B.stem1.cls, B.stem1.flops_per_pixel * Ht * Wt, st);
            if ((rc = launch_conv(B.stem1, {c.S1, B.c / 2, 0}, Hb / 2, Wb / 2, {reinterpret_cast<float*>(PA), B.c, 0}, nullptr, st, nullptr, G.pitch, G.plane()))) return rc;
        }
        unsigned char *pc = PA, *pn = PB;
        if (phases & PH_TRUNK) for (int i = 0; i < 8; i++) {
            if (!rowk && E.rs && E.rs2 && B.c == 64 && !(i & 1) && rs2_applies(Ht, Wt)) {      // layers i, i + 1 in one launch (conv_rs2.h)
                Timed t(E.prof, B.res[i].cls, (B.res[i].flops_per_pixel + B.res[i + 1].flops_per_pixel) * Ht * Wt, st);
                if ((rc = launch_rs2(B.res[i], B.res[i + 1], pc, pn, Ht, Wt, st, (i & 2) != 0))) return rc;
                std::swap(pc, pn); i++;
                continue;
            }
            Timed t(E.prof, B.res[i].cls, B.res[i].flops_per_pixel * Ht * Wt, st);
            if (rowk && ks_serves(E.ks_mask, B.c)) rc = launch_ks(B.res[i], pc, pn, Ht, Wt, st);
            else if (rowk) rc = launch_row(B.res[i], pc, pn, Ht, Wt, st);
            else if (E.rs && B.c == 64 && (Ht + 1) / 2 >= RS_MIN_PAIRS) rc = launch_rs(B.res[i], pc, pn, Ht, Wt, st, (i & 1) != 0);      // tiny tensors: conv_t64
            else rc = launch_t64(B.res[i], pc, pn, Ht, Wt, st, (i & 1) == 0);
            if (rc) return rc;
            std::swap(pc, pn);
        }
        if (!(phases & PH_HEAD)) return 0;
        Timed t(E.prof, B.head.cls, B.head.flops_per_pixel * Ht * Wt, st);      // eight layers: the trunk output is back in PA
        // the row-streaming tail where every workgroup has at least 16 steps to amortise its prologue over (4K: 32; 1080p: 8 - there the tile kernel
        // is as fast or faster: head_b3 0.037 vs 0.039 ms per pair, same call)
        if (fin && E.tail_rs && b == 3 && B.c == 64 && B.head.cout == 24 && B.head.d_wh && Ht * 4 == c.hp && Wt * 4 == c.wp &&
            (E.tail_rs_always || ((Wt + 31) / 32) * Ht >= 32 * device_cus(true)))
            return launch_tail_rs(B, PA, Ht, Wt, *fin, st);
        return launch_conv(B.head, {reinterpret_cast<float*>(PA), B.c, 0}, Ht, Wt, {c.flow[b], 8, 0}, nullptr, st, fin, G.pitch, G.plane());
    }
    if (phases != PH_ALL) return fail(RIFE_HIP_EINVAL, "phased block execution needs the S16 trunk path");
    float* const stem_out = E.v40 ? c.T2 : c.T0;
    {
        Timed t(E.prof, B.stem1.cls, B.stem1.flops_per_pixel * (Hb / 4) * (Wb / 4), st);
        if ((rc = launch_conv(B.stem1, {c.S1, B.c / 2, 0}, Hb / 2, Wb / 2, {stem_out, B.c, 0}, nullptr, st))) return rc;
    }
    float* cur = stem_out; float* nxt = E.v40 ? c.T0 : c.T1;
    for (int i = 0; i < 8; i++) {
        Timed t(E.prof, B.res[i].cls, B.res[i].flops_per_pixel * Ht * Wt, st);
        if ((rc = launch_conv(B.res[i], {cur, B.c, 0}, Ht, Wt, {nxt, B.c, 0}, nullptr, st))) return rc;   // v4.6: skip folded into the weights
        if (E.v40 && i == 0) { cur = c.T0; nxt = c.T1; }
        else std::swap(cur, nxt);
    }
    if (E.v40) {   // add_0 / add_3 / add_8 / add_12 (models/rife-v4/flownet.param): trunk output + stem output, no activation
        Timed t(E.prof, "v40_block_add", 0, st);
        const size_t n4 = (size_t)Ht * Wt * B.c / 4;
        hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<float4*>(cur), reinterpret_cast<const float4*>(c.T2), n4);
        HIPCHK(hipGetLastError());
    }
    {
        Timed t(E.prof, B.head.cls, B.head.flops_per_pixel * Ht * Wt, st);
        if ((rc = launch_conv(B.head, {cur, B.c, 0}, Ht, Wt, {c.flow[b], 8, 0}, nullptr, st, fin))) return rc;
    }
    return 0;
}

static int run_flow_update(const rife_hip& E, Ctx& c, int b) {
    hipStream_t st = c.stream;
    Timed t(E.prof, "flow_update", 0, st);
    dim3 g = grid2d(c.wp, c.hp);
    if (E.v40) {   // Interp x(2 x scale) of the 5-channel head output, then F (+)= u[0:4] * (2 x scale), M (+)= u[4]
        if (b == 0) hipLaunchKernelGGL((k_flow_update<16, true>), g, dim3(256), 0, st, c.flow[0], c.F, c.M, c.wp, c.hp);
        else if (b == 1) hipLaunchKernelGGL((k_flow_update<8, false>), g, dim3(256), 0, st, c.flow[1], c.F, c.M, c.wp, c.hp);
        else if (b == 2) hipLaunchKernelGGL((k_flow_update<4, false>), g, dim3(256), 0, st, c.flow[2], c.F, c.M, c.wp, c.hp);
        else hipLaunchKernelGGL((k_flow_update<2, false>), g, dim3(256), 0, st, c.flow[3], c.F, c.M, c.wp, c.hp);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (b == 0) hipLaunchKernelGGL((k_flow_update<8, true>), g, dim3(256), 0, st, c.flow[0], c.F, c.M, c.wp, c.hp);
    else if (b == 1) hipLaunchKernelGGL((k_flow_update<4, false>), g, dim3(256), 0, st, c.flow[1], c.F, c.M, c.wp, c.hp);
    else hipLaunchKernelGGL((k_flow_update<2, false>), g, dim3(256), 0, st, c.flow[2], c.F, c.M, c.wp, c.hp);
    HIPCHK(hipGetLastError());
    return 0;
}

// RIFE::process_v4, non-TTA branch (rife.cpp:2931-3173) on device-resident frames.
static int run_v4(const rife_hip& E, Ctx& c, const uint8_t* d_in0, const uint8_t* d_in1, float timestep, uint8_t* d_out, const float* tsp = nullptr) {
    hipStream_t st = c.stream;
    int rc;
    {
        Timed t(E.prof, "preproc", 0, st);
        launch_preproc(st, d_in0, c.w, c.h, c.img0, c.wp, c.hp);
        launch_preproc(st, d_in1, c.w, c.h, c.img1, c.wp, c.hp);
        HIPCHK(hipGetLastError());
    }
    const bool fuse_tail = !E.v40 && g_trunk_h2 && g_head_h2 && g_fuse_tail && E.blk[3].head.d_wh != nullptr;
    FinalArgs fin{c.img0, c.img1, c.F, c.M, d_out, c.w, c.h, c.wp, c.hp};
    const float* pending = nullptr;                                      // flow whose update of F, M the next block's stem applies
    // The update after block 0 never reaches HBM on its own (round 5): block 1's scale-4 stem samples it from flow0 (assemble_pixel UPD = 2) and ONE pass after
    // block 1 writes F, M with both updates applied (k_flow_update2) - bit for bit the tensors of the two-kernel sequence, one launch and 20 B / pixel of writes +
    // 20 B / pixel of reads less.  RIFE_HIP_MERGE_FLOW0=0 (A/B, test build): the three separate updates.
    const bool merge_env = env_not_off(ab_getenv("RIFE_HIP_MERGE_FLOW0"));      // per call (a null constant in the product)
    const bool merge0 = merge_env && !E.v40 && g_trunk_h2 && g_fuse_stem && E.blk[1].stem0.d_wh != nullptr && E.blk[1].scale == 4 && !flow_update_fused_into(E, c, 1) &&
                        !flow_update_fused_into(E, c, 2);
    for (int b = 0; b < 4; b++) {
        if ((rc = run_block_convs(E, c, b, timestep, (b == 3 && fuse_tail) ? &fin : nullptr, tsp, PH_ALL, pending, (merge0 && b == 1) ? c.flow[0] : nullptr))) return rc;
        pending = nullptr;
        if (merge0 && b == 0) continue;
        if (merge0 && b == 1) {
            Timed t(E.prof, "flow_update", 0, st);
            hipLaunchKernelGGL((k_flow_update2<8, 4>), grid2d(c.wp, c.hp), dim3(256), 0, st, c.flow[0], c.flow[1], c.F, c.M, c.wp, c.hp);
            HIPCHK(hipGetLastError());
            continue;
        }
        if (b < 3 && flow_update_fused_into(E, c, b + 1)) pending = c.flow[b];
        else if ((b < 3 || E.v40) && (rc = run_flow_update(E, c, b))) return rc;
    }
    if (E.v40) {
        Timed t(E.prof, "final", 0, st);
        hipLaunchKernelGGL(k_blend_final, grid2d(c.w, c.h), dim3(256), 0, st, c.img0, c.img1, c.F, c.M, d_out, c.w, c.h, c.wp, c.hp);
        HIPCHK(hipGetLastError());
    } else if (!fuse_tail) {
        Timed t(E.prof, "final", 0, st);
        hipLaunchKernelGGL(k_final, grid2d(c.w, c.h), dim3(256), 0, st, c.img0, c.img1, c.F, c.M, c.flow[3], d_out, c.w, c.h, c.wp, c.hp);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// RIFE::process_v4 for G (2..4) pairs in LOCKSTEP (rife_hip_process_batch, SURVEY 8f-2 "batch >= 2 pairs per launch for the coarse blocks"):
// every pair keeps its own workspace and stream, so the fine blocks of different pairs overlap as before; the eight trunk layers of a block that
// runs on conv_row_kernel (the coarse blocks 1 / 0, block 2 on small grids; flownet.param:14-42, 66-94) are ONE launch per layer for all pairs
// (gridDim.y = G) on the first pair's stream, between two event hand-offs.  The workgroups of all pairs stream the layer's weights from the L2
// together, and the coarse grids - 68 / 255 workgroups per pair at 1080p - fill the chip in one round instead of G.
// Same kernels, same arguments per tensor: the frames are bit-identical to G single calls.
static int run_v4_group(const rife_hip& E, Ctx* const* cs, int G, const uint8_t* const* d_in0, const uint8_t* const* d_in1, const float* ts, uint8_t* const* d_out) {
    int rc;
    for (int g = 0; g < G; g++) {
        Ctx& c = *cs[g];
        if (!c.ev_group) HIPCHK(hipEventCreateWithFlags(&c.ev_group, hipEventDisableTiming));
        Timed t(E.prof, "preproc", 0, c.stream);
        launch_preproc(c.stream, d_in0[g], c.w, c.h, c.img0, c.wp, c.hp);
        launch_preproc(c.stream, d_in1[g], c.w, c.h, c.img1, c.wp, c.hp);
        HIPCHK(hipGetLastError());
    }
    const bool fuse_tail = g_trunk_h2 && g_head_h2 && g_fuse_tail && E.blk[3].head.d_wh != nullptr;
    const float* pend[4] = {nullptr, nullptr, nullptr, nullptr};         // per pair: flow whose update the next block's stem applies (run_v4)
    auto after_block = [&](Ctx& c, int g, int b) -> int {
        if (b < 3 && flow_update_fused_into(E, c, b + 1)) { pend[g] = c.flow[b]; return 0; }
        return b < 3 ? run_flow_update(E, c, b) : 0;
    };
    for (int b = 0; b < 4; b++) {
        const rife_hip::Block& B = E.blk[b];
        const bool batched = G >= 2 && block_on_row_kernel(E, *cs[0], b) && block_on_s16(E, *cs[0], b);
        for (int g = 0; g < G; g++) {
            Ctx& c = *cs[g];
            FinalArgs fin{c.img0, c.img1, c.F, c.M, d_out[g], c.w, c.h, c.wp, c.hp};
            if (!batched) {
                if ((rc = run_block_convs(E, c, b, ts[g], (b == 3 && fuse_tail) ? &fin : nullptr, nullptr, PH_ALL, pend[g]))) return rc;
                pend[g] = nullptr;
                if ((rc = after_block(c, g, b))) return rc;
            } else {
                if ((rc = run_block_convs(E, c, b, ts[g], nullptr, nullptr, PH_STEMS, pend[g]))) return rc;
                pend[g] = nullptr;
                if (g > 0) HIPCHK(hipEventRecord(c.ev_group, c.stream));
            }
        }
        if (!batched) continue;
        hipStream_t lead = cs[0]->stream;
        for (int g = 1; g < G; g++) HIPCHK(hipStreamWaitEvent(lead, cs[g]->ev_group, 0));
        {
            const int s = B.scale, Ht = cs[0]->hp / s / 4, Wt = cs[0]->wp / s / 4;
            const unsigned char* pin[4]; unsigned char* pout[4];
            for (int i = 0; i < 8; i++) {
                for (int g = 0; g < G; g++) { pin[g] = cs[g]->P[b][i & 1]; pout[g] = cs[g]->P[b][(i & 1) ^ 1]; }
                Timed t(E.prof, B.res[i].cls, B.res[i].flops_per_pixel * Ht * Wt * G, lead);
                if (ks_serves(E.ks_mask, B.c)) rc = launch_ks(B.res[i], nullptr, nullptr, Ht, Wt, lead, G, pin, pout);
                else rc = launch_row(B.res[i], nullptr, nullptr, Ht, Wt, lead, G, pin, pout);
                if (rc) return rc;
            }
        }
        HIPCHK(hipEventRecord(cs[0]->ev_group, lead));
        for (int g = 0; g < G; g++) {
            Ctx& c = *cs[g];
            if (g > 0) HIPCHK(hipStreamWaitEvent(c.stream, cs[0]->ev_group, 0));
            if ((rc = run_block_convs(E, c, b, ts[g], nullptr, nullptr, PH_HEAD))) return rc;
            if ((rc = after_block(c, g, b))) return rc;
        }
    }
    for (int g = 0; g < G; g++)
        if (!fuse_tail) {
            Ctx& c = *cs[g];
            hipLaunchKernelGGL(k_final, grid2d(c.w, c.h), dim3(256), 0, c.stream, c.img0, c.img1, c.F, c.M, c.flow[3], d_out[g], c.w, c.h, c.wp, c.hp);
            HIPCHK(hipGetLastError());
        }
    return 0;
}

// Plain v4 pass replayed from a hipGraph for small frames (<= 1920 x 1088 padded), opt-in with RIFE_HIP_GRAPH=1: one graph launch
// instead of ~50 kernel launches (+ two small device copies into the fixed staging buffers).  Measured on MI355X
// (tools/graph_bench.py, profiler off): 1080p 1.116 vs 1.119 ms per pair, 720p 0.782 vs 0.782, 360p 0.673 vs 0.674 - no gain: the
// chain of ~50 dependent kernels (fill / drain of each launch), not host launch overhead, sets the floor, and a replayed graph
// executes the same chain.  Kept off by default; the profiler (events around every launch) bypasses it.
static const bool g_use_graph = env_on(getenv("RIFE_HIP_GRAPH"));

static int run_v4_replay(const rife_hip& E, Ctx& c, const uint8_t* d_in0, const uint8_t* d_in1, float timestep, uint8_t* d_out) {
    const bool eligible = g_use_graph && !E.prof.on && c.d_ts && (size_t)c.wp * c.hp <= (size_t)1920 * 1088;
    if (!eligible) return run_v4(E, c, d_in0, d_in1, timestep, d_out);
    hipStream_t st = c.stream;
    const size_t nbytes = (size_t)c.w * c.h * 3;
    if (d_in0 != c.d_in0) HIPCHK(hipMemcpyAsync(c.d_in0, d_in0, nbytes, hipMemcpyDeviceToDevice, st));
    if (d_in1 != c.d_in1) HIPCHK(hipMemcpyAsync(c.d_in1, d_in1, nbytes, hipMemcpyDeviceToDevice, st));
    uint32_t bits; std::memcpy(&bits, &timestep, 4);
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)c.d_ts, (int)bits, 1, st));
    int rc = 0;
    if (c.gexec) HIPCHK(hipGraphLaunch(c.gexec, st));
    else if (!c.g_warm) {
        if ((rc = run_v4(E, c, c.d_in0, c.d_in1, timestep, c.d_out, c.d_ts))) return rc;
        c.g_warm = true;
    } else {
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        rc = run_v4(E, c, c.d_in0, c.d_in1, timestep, c.d_out, c.d_ts);
        const hipError_t e = hipStreamEndCapture(st, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph) return fail(RIFE_HIP_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        const hipError_t ei = hipGraphInstantiate(&c.gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) { c.gexec = nullptr; return fail(RIFE_HIP_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); }
        HIPCHK(hipGraphLaunch(c.gexec, st));
    }
    if (d_out != c.d_out) HIPCHK(hipMemcpyAsync(d_out, c.d_out, nbytes, hipMemcpyDeviceToDevice, st));
    return 0;
}

// RIFE::process_v4 with -x and/or -z (rife.cpp:2534-2930 spatial TTA, 3036-3135 temporal only; CPU twin 3246-4145):
// nori = 8 orientations or 1, ntemp = 2 directions (in0,in1,t) / (in1,in0,1-t) or 1.  Per IFBlock stage the flows of
// all passes are merged (temporal first, then spatial, like the reference) before any pass goes on.
static int run_v4_tta(const rife_hip& E, hipStream_t st, const uint8_t* d_in0, const uint8_t* d_in1, int w, int h, float timestep, uint8_t* d_out) {
    const int nori = E.tta ? 8 : 1, ntemp = E.tta_temporal ? 2 : 1;
    const int wp = (w + 31) / 32 * 32, hp = (h + 31) / 32 * 32;
    constexpr int NL = rife_hip::NLANE;
    const bool lanes = nori == 8;          // the 8 orientations are independent between consensus points: 4 worker streams
    int rc;
    if (lanes && !E.tta_lane[0]) {
        for (int l = 0; l < NL; l++) HIPCHK(hipStreamCreateWithFlags(&E.tta_lane[l], hipStreamNonBlocking));
        for (int i = 0; i < 6; i++) {
            HIPCHK(hipEventCreateWithFlags(&E.tta_fork[i], hipEventDisableTiming));
            for (int l = 0; l < NL; l++) HIPCHK(hipEventCreateWithFlags(&E.tta_join[i][l], hipEventDisableTiming));
        }
    }
    auto lane_of = [&](int ti) { return lanes ? E.tta_lane[ti % NL] : st; };
    for (int dir = 0; dir < ntemp; dir++)
        for (int ti = 0; ti < nori; ti++) {
            auto& up = E.tta_ctx[dir][ti];
            if (!up) up.reset(new Ctx);
            Ctx& c = *up;
            c.stream = lane_of(ti);
            const bool swap = ti >= 4;
            // per-layer scratch is shared by the passes of one lane (they run back to back on that lane's stream)
            const int owner = lanes ? ti % NL : 0;
            const Ctx* scratch = (dir == 0 && ti == owner) ? nullptr : E.tta_ctx[0][owner].get();
            if ((rc = ensure_ctx_dims(c, swap ? h : w, swap ? w : h, swap ? hp : wp, swap ? wp : hp, scratch, dir == 0, true))) return rc;
            if (dir == 1) { c.img0 = E.tta_ctx[0][ti]->img1; c.img1 = E.tta_ctx[0][ti]->img0; }   // reversed pass sees the frames swapped
        }
    int sync_id = 0;
    auto fork = [&]() -> int {             // lanes wait for everything enqueued on the caller's stream so far
        if (!lanes) return 0;
        HIPCHK(hipEventRecord(E.tta_fork[sync_id], st));
        for (int l = 0; l < NL; l++) HIPCHK(hipStreamWaitEvent(E.tta_lane[l], E.tta_fork[sync_id], 0));
        return 0;
    };
    auto join = [&]() -> int {             // the caller's stream waits for all lanes
        if (!lanes) return 0;
        for (int l = 0; l < NL; l++) {
            HIPCHK(hipEventRecord(E.tta_join[sync_id][l], E.tta_lane[l]));
            HIPCHK(hipStreamWaitEvent(st, E.tta_join[sync_id][l], 0));
        }
        sync_id++;
        return 0;
    };
    {
        Timed t(E.prof, "preproc", 0, st);
        if (nori == 8) {
            Ptr8 a, b;
            for (int ti = 0; ti < 8; ti++) { a.p[ti] = E.tta_ctx[0][ti]->img0; b.p[ti] = E.tta_ctx[0][ti]->img1; }
            hipLaunchKernelGGL(k_preproc_tta, tta_grid(wp, hp, 4), tta_block(4), 0, st, d_in0, w, h, a, wp, hp);
            hipLaunchKernelGGL(k_preproc_tta, tta_grid(wp, hp, 4), tta_block(4), 0, st, d_in1, w, h, b, wp, hp);
        } else {
            launch_preproc(st, d_in0, w, h, E.tta_ctx[0][0]->img0, wp, hp);
            launch_preproc(st, d_in1, w, h, E.tta_ctx[0][0]->img1, wp, hp);
        }
        HIPCHK(hipGetLastError());
    }
    const bool fused_consensus = ntemp == 2 && nori == 8 && E.tta_consensus;
    for (int fi = 0; fi < 4; fi++) {
        const int Wf = wp / E.flow_div(fi), Hf = hp / E.flow_div(fi);
        if ((rc = fork())) return rc;
        for (int ti = 0; ti < nori; ti++) {
            hipStream_t ls = lane_of(ti);
            for (int dir = 0; dir < ntemp; dir++) {
                Ctx& c = *E.tta_ctx[dir][ti];
                if ((rc = run_block_convs(E, c, fi, dir ? 1.f - timestep : timestep))) return rc;
            }
            if (ntemp == 2 && !fused_consensus) {
                Timed t(E.prof, "tta_merge", 0, ls);
                const size_t npix = (size_t)Wf * Hf;
                hipLaunchKernelGGL(k_v4_temporal_merge, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ls,
                                   E.tta_ctx[0][ti]->flow[fi], E.tta_ctx[1][ti]->flow[fi], npix);
                HIPCHK(hipGetLastError());
            }
        }
        if ((rc = join())) return rc;
        if (fused_consensus) {       // -x -z: temporal and spatial consensus of the sixteen flow tensors in one pass (k_v4_consensus)
            Timed t(E.prof, "tta_merge", 0, st);
            Ptr8x2 f;
            for (int ti = 0; ti < 8; ti++) { f.f[ti] = E.tta_ctx[0][ti]->flow[fi]; f.r[ti] = E.tta_ctx[1][ti]->flow[fi]; }
            hipLaunchKernelGGL(k_v4_consensus, tta_grid(Wf, Hf, 32), tta_block(32), 0, st, f, Wf, Hf);
            HIPCHK(hipGetLastError());
        } else if (nori == 8) {
            Timed t(E.prof, "tta_merge", 0, st);
            for (int dir = 0; dir < ntemp; dir++) {
                Ptr8 f;
                for (int ti = 0; ti < 8; ti++) f.p[ti] = E.tta_ctx[dir][ti]->flow[fi];
                hipLaunchKernelGGL(k_v4_spatial_avg, tta_grid(Wf, Hf, 32), tta_block(32), 0, st, f, Wf, Hf);
            }
            HIPCHK(hipGetLastError());
        }
        if (fi < 3 || E.v40) {
            if (lanes) {   // flow updates run on the lanes; they must see the consensus written on the caller's stream
                HIPCHK(hipEventRecord(E.tta_fork[5], st));
                for (int l = 0; l < NL; l++) HIPCHK(hipStreamWaitEvent(E.tta_lane[l], E.tta_fork[5], 0));
            }
            for (int ti = 0; ti < nori; ti++)
                for (int dir = 0; dir < ntemp; dir++)
                    if ((rc = run_flow_update(E, *E.tta_ctx[dir][ti], fi))) return rc;
        }
    }
    Ptr16 outs;
    for (int i = 0; i < 16; i++) outs.p[i] = nullptr;
    {
        if ((rc = fork())) return rc;
        for (int ti = 0; ti < nori; ti++)
            for (int dir = 0; dir < ntemp; dir++) {
                Ctx& c = *E.tta_ctx[dir][ti];
                Timed t(E.prof, "final", 0, c.stream);
                if (E.v40) hipLaunchKernelGGL(k_blend_final_float, grid2d(c.wp, c.hp), dim3(256), 0, c.stream, c.img0, c.img1, c.F, c.M, c.outf, c.wp, c.hp);
                else hipLaunchKernelGGL(k_final_float, grid2d(c.wp, c.hp), dim3(256), 0, c.stream, c.img0, c.img1, c.F, c.M, c.flow[3], c.outf, c.wp, c.hp);
                outs.p[dir * 8 + ti] = c.outf;
            }
        if ((rc = join())) return rc;
        Timed t(E.prof, "final", 0, st);
        hipLaunchKernelGGL(k_postproc_tta, tta_grid(w, h, 16), tta_block(16), 0, st, outs, nori, ntemp, d_out, w, h, wp, hp);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// rife-v2.x: RIFE::process, non-TTA branch (rife.cpp:878-1183) = flownet -> slice -> contextnet x2 -> fusionnet
// ------------------------------------------------------------------------------------------------
static int ensure_ctx_v2(Ctx& c, int w, int h, bool uhd, int nori = 1, int ntemp = 1, bool v3 = false) {
    const int wp = (w + 31) / 32 * 32, hp = (h + 31) / 32 * 32;     // rife.cpp:417-418
    const bool ens = nori * ntemp > 1;
    if (c.v2 && c.wp == wp && c.hp == hp && c.w == w && c.h == h && (!uhd || c.h0) && (!ens || (c.toutf[0][0] && c.toutf[ntemp - 1][nori - 1])) && (!v3 || c.T2)) return 0;
    c.h0 = c.h1 = c.acc_s = nullptr; c.T2 = nullptr;
    for (int d = 0; d < 2; d++) for (int t = 0; t < 8; t++) { c.tflow[d][t] = c.toutf[d][t] = nullptr; if (!d) c.timg0[t] = c.timg1[t] = nullptr; }
    for (void* p : c.allocs) (void)hipFree(p);
    c.allocs.clear();
    c.v2 = true; c.w = w; c.h = h; c.wp = wp; c.hp = hp;
    const size_t P = (size_t)wp * hp;
    int rc;
#define A_(ptr, n) if ((rc = dalloc(c, ptr, (size_t)(n)))) { reset_ctx(c); return rc; }
    A_(c.d_in0, (size_t)w * h * 3) A_(c.d_in1, (size_t)w * h * 3) A_(c.d_out, (size_t)w * h * 3)
    A_(c.img0, P) A_(c.img1, P)
    if (v3) { A_(c.X, P * 16) A_(c.S1, P / 4 * 80) A_(c.T0, P / 16 * 160) A_(c.T1, P / 16 * 160) A_(c.T2, P / 16 * 160) }   // rife-v3.x block 2: 80 / 160 ch at 1/2, 1/4 res
    else { A_(c.X, P * 16) A_(c.S1, P / 4 * 48) A_(c.T0, P / 16 * 96) A_(c.T1, P / 16 * 96) }
    A_(c.acc, P / 4) A_(c.D, P / 4) A_(c.head, P)
    A_(c.I8, P * 8) A_(c.ca, P / 4 * 32) A_(c.cb, P / 4 * 32) A_(c.cc, P / 16 * 32)
    A_(c.feat[0], P / 16 * 32) A_(c.feat[1], P / 64 * 64) A_(c.feat[2], P / 256 * 128) A_(c.feat[3], P / 1024 * 256)
    A_(c.ctmp[0], P / 64 * 64) A_(c.ctmp[1], P / 256 * 128) A_(c.ctmp[2], P / 1024 * 256)
    A_(c.fl[0], P / 16) A_(c.fl[1], P / 64) A_(c.fl[2], P / 256) A_(c.fl[3], P / 1024)
    A_(c.ca2, P / 4 * 32) A_(c.cb2, P / 4 * 32) A_(c.cc2, P / 16 * 32)
    A_(c.feat2[0], P / 16 * 32) A_(c.feat2[1], P / 64 * 64) A_(c.feat2[2], P / 256 * 128) A_(c.feat2[3], P / 1024 * 256)
    A_(c.ctmp2[0], P / 64 * 64) A_(c.ctmp2[1], P / 256 * 128) A_(c.ctmp2[2], P / 1024 * 256)
    A_(c.fl2[0], P / 16) A_(c.fl2[1], P / 64) A_(c.fl2[2], P / 256) A_(c.fl2[3], P / 1024)
    A_(c.e0a, P / 4 * 32) A_(c.e0b, P / 4 * 32) A_(c.e0c, P / 16 * 64) A_(c.B1, P / 16 * 128) A_(c.e1a, P / 64 * 128) A_(c.B2, P / 64 * 256)
    A_(c.e2a, P / 256 * 256) A_(c.B3, P / 256 * 512) A_(c.e3a, P / 1024 * 512) A_(c.B4, P / 1024 * 1024)
    A_(c.U0, P / 256 * 512) A_(c.U1, P / 64 * 256) A_(c.U2, P / 16 * 128) A_(c.U3, P / 4 * 32)
    if (uhd) { A_(c.h0, P / 4) A_(c.h1, P / 4) A_(c.acc_s, P / 16) }
    if (ens) {
        c.timg0[0] = c.img0; c.timg1[0] = c.img1;
        for (int t = 1; t < nori; t++) { A_(c.timg0[t], P) A_(c.timg1[t], P) }
        for (int d = 0; d < ntemp; d++) for (int t = 0; t < nori; t++) { A_(c.tflow[d][t], P / 4) A_(c.toutf[d][t], P) }
    }
#undef A_
    return 0;
}

static int conv_t(const rife_hip& E, const ConvLayer& L, TensorView x, int H, int W, TensorView y, hipStream_t st, const float* in1 = nullptr, float* out1 = nullptr,
                  const TensorView* y2 = nullptr) {
    const int mo_h = L.deconv ? H : (H - 1) / L.stride + 1, mo_w = L.deconv ? W : (W - 1) / L.stride + 1;
    Timed t(E.prof, L.cls, L.flops_per_pixel * mo_h * mo_w * (in1 ? 2 : 1), st);
    return launch_conv(L, x, H, W, y, nullptr, st, nullptr, 0, 0, in1, out1, y2);      // in1 / out1: a second tensor pair through the same launch; y2: a second destination
}

// stem2_fused_kernel (stem_fused_v2.h): block-input assembly at scale S (1 or 2) fused into the 10 -> cout stride-2 convolution that consumes it.
// RIFE_HIP_V2_FUSED_STEM=0 (A/B): the unfused pair k2_assemble + conv_h2s2_kernel
static const bool g_v2_fused_stem = env_not_off(ab_getenv("RIFE_HIP_V2_FUSED_STEM"));
static bool stem2_fusable(const ConvLayer& L, int S, int wp, int hp) {
    return g_v2_fused_stem && g_trunk_h2 && (S == 1 || S == 2) && L.d_wh && L.cin == 10 && L.nchunksh == 1 && L.stride == 2 && !L.deconv && L.cout <= 128 && L.cout % 4 == 0 &&
           (wp / S) % 2 == 0 && (hp / S) % 2 == 0;
}
template <int S, typename IMG, bool FSCALE, bool R64 = false>
static int launch_stem2_cfg(const Stem2Args<IMG>& a, int nwg, hipStream_t st) {
    auto kfn = stem2_fused_kernel<S, IMG, FSCALE, R64>;
    {
        static std::mutex mu; static std::map<int, bool> done;
        int dev = 0; (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, stem2_lds_bytes(4)));
            done[dev] = true;
        }
    }
    hipLaunchKernelGGL(kfn, dim3(nwg), dim3(512), stem2_lds_bytes(a.nsub, R64), st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("stem2_fused launch: ") + hipGetErrorString(e));
    return 0;
}
template <typename IMG>
static int launch_stem2_fused(const rife_hip& E, const ConvLayer& L, int S, bool fscale, IMG img0, IMG img1, const float4* acc, int wp, int hp, float* out, int out_ld,
                              hipStream_t st) {
    Stem2Args<IMG> a;
    a.img0 = img0; a.img1 = img1; a.acc = acc; a.wpk = reinterpret_cast<const unsigned char*>(L.d_wh); a.bias = L.d_bias; a.slope = L.d_slope; a.out = out;
    a.wp = wp; a.hp = hp; a.Ho = hp / S / 2; a.Wo = wp / S / 2; a.out_ld = out_ld; a.Cout = L.cout; a.tiles_x = (a.Wo + 31) / 32;
    a.NS = L.NS; a.nsub = (L.cout + 31) / 32;
    const int nwg = a.tiles_x * ((a.Ho + 3) / 4);
    Timed t(E.prof, L.cls, L.flops_per_pixel * a.Ho * a.Wo, st);
    // scale 1 on the u8 frames: 64-byte halo records, weights from the L2, three workgroups per CU (stem_fused_v2.h R64); RIFE_HIP_V2_STEM_R64=0 (A/B, test build): two
    static const bool r64 = env_not_off(ab_getenv("RIFE_HIP_V2_STEM_R64"));
    if (S == 1 && std::is_same<IMG, ImgU8>::value && r64 && a.nsub <= 2) return launch_stem2_cfg<1, IMG, false, true>(a, nwg, st);
    if (S == 1) return launch_stem2_cfg<1, IMG, false>(a, nwg, st);
    if (fscale) return launch_stem2_cfg<2, IMG, true>(a, nwg, st);
    return launch_stem2_cfg<2, IMG, false>(a, nwg, st);
}

// IFNet of rife-v2.x on frames of wp x hp (flownet.param): 4 blocks at scales 8,4,2,1; the flow is accumulated at
// half of that resolution into `acc` (float4 per pixel).
template <typename IMG>
static int run_v2_ifnet(const rife_hip& E, Ctx& c, IMG img0, IMG img1, int wp, int hp, float4* acc) {
    hipStream_t st = c.stream;
    const int wh = wp / 2, hh = hp / 2;
    int rc;
    for (int b = 0; b < E.n_fblk; b++) {
        const rife_hip::V2Block& B = E.fblk[b];
        const int s = B.scale, Hb = hp / s, Wb = wp / s;
        const bool fused_stem = b > 0 && stem2_fusable(B.stem0, s, wp, hp);
        if (fused_stem) {
            if ((rc = launch_stem2_fused(E, B.stem0, s, E.v3 && s == 2, img0, img1, acc, wp, hp, c.S1, B.c / 2, st))) return rc;
        } else {
            Timed t(E.prof, E.prof_fine ? "fb" + std::to_string(b) + "_assemble" : std::string("v2_assemble"), 0, st);
            dim3 g = grid2d(Wb, Hb);
            if (b == 0 && s == 8) hipLaunchKernelGGL((k2_assemble0<8, IMG>), g, dim3(256), 0, st, img0, img1, c.X, wp, hp);
            else if (b == 0) hipLaunchKernelGGL((k2_assemble0<4, IMG>), g, dim3(256), 0, st, img0, img1, c.X, wp, hp);
            else if (E.v3 && s == 2) hipLaunchKernelGGL((k2_assemble<2, IMG, true>), g, dim3(256), 0, st, img0, img1, acc, c.X, wp, hp);
            else if (s == 4) hipLaunchKernelGGL((k2_assemble<4, IMG>), g, dim3(256), 0, st, img0, img1, acc, c.X, wp, hp);
            else if (s == 2) hipLaunchKernelGGL((k2_assemble<2, IMG>), g, dim3(256), 0, st, img0, img1, acc, c.X, wp, hp);
            else hipLaunchKernelGGL((k2_assemble<1, IMG>), g, dim3(256), 0, st, img0, img1, acc, c.X, wp, hp);   // v3: x 1.0 (Mul_139) is the identity
            HIPCHK(hipGetLastError());
        }
        if (!fused_stem && (rc = conv_t(E, B.stem0, {c.X, b == 0 ? 8 : 16, 0}, Hb, Wb, {c.S1, B.c / 2, 0}, st))) return rc;
        if ((rc = conv_t(E, B.stem1, {c.S1, B.c / 2, 0}, Hb / 2, Wb / 2, {c.T0, B.c, 0}, st))) return rc;
        float* cur = c.T0;
        const int Ht = Hb / 4, Wt = Wb / 4;
        if (E.v3) {
            // 3 x [conv + PReLU, conv + PReLU, BinaryOp add with the block input] (rife-v3.1 flownet.param:12-29)
            float* tmp = c.T1; float* nxt = c.T2;
            const size_t n4 = (size_t)Ht * Wt * B.c / 4;
            for (int i = 0; i < 3; i++) {
                if ((rc = conv_t(E, B.conv[2 * i], {cur, B.c, 0}, Ht, Wt, {tmp, B.c, 0}, st))) return rc;
                if ((rc = conv_t(E, B.conv[2 * i + 1], {tmp, B.c, 0}, Ht, Wt, {nxt, B.c, 0}, st))) return rc;
                {
                    Timed t(E.prof, "v3_res_add", 0, st);
                    hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<float4*>(nxt), reinterpret_cast<const float4*>(cur), n4);
                    HIPCHK(hipGetLastError());
                }
                std::swap(cur, nxt);
            }
        } else {
            float* nxt = c.T1;
            for (int i = 0; i < 6; i++) {
                if ((rc = conv_t(E, B.conv[i], {cur, B.c, 0}, Ht, Wt, {nxt, B.c, 0}, st))) return rc;
                std::swap(cur, nxt);
            }
        }
        if ((rc = conv_t(E, B.head, {cur, B.c, 0}, Ht, Wt, {reinterpret_cast<float*>(c.D), 4, 0}, st))) return rc;
        {
            Timed t(E.prof, "v2_flow_accum", 0, st);
            dim3 g = grid2d(wh, hh);
            if (E.v3) {
                if (b == 0) hipLaunchKernelGGL((k2_flow_accum<4, true, true>), g, dim3(256), 0, st, c.D, acc, wh, hh);
                else if (b == 1) hipLaunchKernelGGL((k2_flow_accum<2, false, true>), g, dim3(256), 0, st, c.D, acc, wh, hh);
                else hipLaunchKernelGGL((k2_flow_accum<1, false>), g, dim3(256), 0, st, c.D, acc, wh, hh);
            } else if (b == 0) hipLaunchKernelGGL((k2_flow_accum<8, true>), g, dim3(256), 0, st, c.D, acc, wh, hh);
            else if (b == 1) hipLaunchKernelGGL((k2_flow_accum<4, false>), g, dim3(256), 0, st, c.D, acc, wh, hh);
            else if (b == 2) hipLaunchKernelGGL((k2_flow_accum<2, false>), g, dim3(256), 0, st, c.D, acc, wh, hh);
            else hipLaunchKernelGGL((k2_flow_accum<1, false>), g, dim3(256), 0, st, c.D, acc, wh, hh);
            HIPCHK(hipGetLastError());
        }
    }
    return 0;
}

// flow estimate of one (img0, img1) pair of padded RGBX frames of wp x hp -> acc (float4 field of wp/2 x hp/2).
// IFNet (flownet.param); UHD mode estimates the flow on half-resolution frames (rife.cpp:928-945)
static int run_v2_flow(const rife_hip& E, Ctx& c, const uint32_t* img0, const uint32_t* img1, int wp, int hp, float4* acc) {
    hipStream_t st = c.stream;
    int rc;
    const int wh = wp / 2, hh = hp / 2;
    if (E.uhd) {
        {
            Timed t(E.prof, "v2_uhd_resample", 0, st);
            hipLaunchKernelGGL(k2_image_half, grid2d(wh, hh), dim3(256), 0, st, img0, c.h0, wp, hp);
            hipLaunchKernelGGL(k2_image_half, grid2d(wh, hh), dim3(256), 0, st, img1, c.h1, wp, hp);
            HIPCHK(hipGetLastError());
        }
        if ((rc = run_v2_ifnet(E, c, ImgF4{c.h0}, ImgF4{c.h1}, wh, hh, c.acc_s))) return rc;
        {
            Timed t(E.prof, "v2_uhd_resample", 0, st);
            hipLaunchKernelGGL(k2_flow_up2_double, grid2d(wh, hh), dim3(256), 0, st, c.acc_s, acc, wh, hh);
            HIPCHK(hipGetLastError());
        }
    } else if ((rc = run_v2_ifnet(E, c, ImgU8{img0}, ImgU8{img1}, wp, hp, acc))) return rc;
    return 0;
}

// (img0, img1, flow) -> interpolated frame: slice -> ContextNet x2 -> FusionNet -> blend (rife.cpp:1008-1183).
// Writes the u8 w x h frame to d_out, or (outf != null) the clipped float frame of wp x hp for the TTA averaging.
static int run_v2_synth(const rife_hip& E, Ctx& c, const uint32_t* img0, const uint32_t* img1, const float4* acc, int wp, int hp,
                        uint8_t* d_out, float4* outf) {
    hipStream_t st = c.stream;
    int rc;
    const int wh = wp / 2, hh = hp / 2;
    // ---- ContextNet twice (contextnet.param): (img0, flow[0:2]) -> "3".."6", (img1, flow[2:4]) -> "7".."10",
    //      each warped level written straight into its slice of the FusionNet concat buffers ----
    static const bool img_env = env_not_off(ab_getenv("RIFE_HIP_CTX0_IMG"));      // A/B (round 5)
    const bool ctx0_img = E.ctxc[0].d_wimg != nullptr && g_trunk_h2 && img_env;      // RIFE_HIP_TRUNK=f32 keeps the fp32 matrix path
    float* cat_buf[4] = {c.B1, c.B2, c.B3, c.B4};
    const int cat_ld[4] = {128, 256, 512, 1024}, cat_off[4] = {64, 128, 256, 512}, lvl_c[4] = {32, 64, 128, 256};
    // both passes through ONE launch per layer (gridDim.y = 2: same weights, twice the workgroups - the deep levels are grids of 72 - 272 workgroups);
    // RIFE_HIP_V2_CTX_BATCH=0 (A/B, test build): one pass after the other
    static const bool ctx_batch_env = env_not_off(ab_getenv("RIFE_HIP_V2_CTX_BATCH"));
    const bool ctx_batch = ctx0_img && ctx_batch_env;
    if (ctx_batch) {
        float2* const* flp[2] = {c.fl, c.fl2};
        float* const* featp[2] = {c.feat, c.feat2};
        {
            Timed t(E.prof, "v2_ctx_misc", 0, st);
            for (int im = 0; im < 2; im++) {
                hipLaunchKernelGGL(k2_flow_half<true>, grid2d(wh / 2, hh / 2), dim3(256), 0, st, reinterpret_cast<const float*>(acc), im * 2, flp[im][0], wh, hh);
                for (int l = 1; l < 4; l++)
                    hipLaunchKernelGGL(k2_flow_half<false>, grid2d((wh >> l) / 2, (hh >> l) / 2), dim3(256), 0, st, reinterpret_cast<const float*>(flp[im][l - 1]), 0, flp[im][l],
                                       wh >> l, hh >> l);
            }
            HIPCHK(hipGetLastError());
        }
        {
            const ConvLayer& L0 = E.ctxc[0];
            Timed t(E.prof, L0.cls, 2 * L0.flops_per_pixel * (hp / 2) * (wp / 2), st);
            ImgConvArgs ia;
            ia.img = img0; ia.out = c.ca; ia.img1 = img1; ia.out1 = c.ca2; ia.wpk = L0.d_wimg; ia.bias = L0.d_bias; ia.slope = L0.d_slope;
            ia.wp = wp; ia.hp = hp; ia.Wo = wp / 2; ia.Ho = hp / 2; ia.tiles_x = (ia.Wo + 31) / 32; ia.ntiles = ia.tiles_x * ia.Ho;
            const int nwg = std::min((ia.ntiles + 3) / 4, 4 * device_cus(true));
            hipLaunchKernelGGL(conv_img_s2_kernel, dim3(nwg, 2), dim3(256), 0, st, ia);
            HIPCHK(hipGetLastError());
        }
        if ((rc = conv_t(E, E.ctxc[1], {c.ca, 32, 0}, hp / 2, wp / 2, {c.cb, 32, 0}, st, c.ca2, c.cb2))) return rc;
        if ((rc = conv_t(E, E.ctxc[2], {c.cb, 32, 0}, hp / 2, wp / 2, {c.cc, 32, 0}, st, c.cb2, c.cc2))) return rc;
        if ((rc = conv_t(E, E.ctxc[3], {c.cc, 32, 0}, hp / 4, wp / 4, {c.feat[0], 32, 0}, st, c.cc2, c.feat2[0]))) return rc;
        for (int l = 1; l < 4; l++) {
            const int Hl = hp >> (l + 1), Wl = wp >> (l + 1);      // input resolution of this level's strided conv
            if ((rc = conv_t(E, E.ctxc[2 + 2 * l], {c.feat[l - 1], lvl_c[l - 1], 0}, Hl, Wl, {c.ctmp[l - 1], lvl_c[l], 0}, st, c.feat2[l - 1], c.ctmp2[l - 1]))) return rc;
            if ((rc = conv_t(E, E.ctxc[3 + 2 * l], {c.ctmp[l - 1], lvl_c[l], 0}, Hl / 2, Wl / 2, {c.feat[l], lvl_c[l], 0}, st, c.ctmp2[l - 1], c.feat2[l]))) return rc;
        }
        {
            Timed t(E.prof, E.prof_fine ? "ctx_warps" : "v2_ctx_misc", 0, st);
            WarpBatch wb;
            for (int im = 0; im < 2; im++)
                for (int l = 0; l < 4; l++) {
                    const int z = 4 * im + l;
                    wb.feat[z] = featp[im][l]; wb.flow[z] = flp[im][l]; wb.out[z] = cat_buf[l]; wb.C[z] = lvl_c[l]; wb.out_ld[z] = cat_ld[l];
                    wb.out_coff[z] = cat_off[l] + im * lvl_c[l]; wb.w[z] = wp >> (l + 2); wb.h[z] = hp >> (l + 2);
                }
            const int W0 = wp >> 2, H0 = hp >> 2, ppb0 = 256 / (lvl_c[0] / 4);      // level 0: the largest pixel grid and the most pixels per block
            hipLaunchKernelGGL(k2_warp_nhwc_batch, dim3((W0 + ppb0 - 1) / ppb0, H0, 8), dim3(256), 0, st, wb);
            HIPCHK(hipGetLastError());
        }
    } else
    for (int im = 0; im < 2; im++) {
        {
            Timed t(E.prof, "v2_ctx_misc", 0, st);
            const size_t P = (size_t)wp * hp;
            if (!ctx0_img) hipLaunchKernelGGL(k2_image_nhwc8, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, im ? img1 : img0, c.I8, P);
            hipLaunchKernelGGL(k2_flow_half<true>, grid2d(wh / 2, hh / 2), dim3(256), 0, st, reinterpret_cast<const float*>(acc), im * 2, c.fl[0], wh, hh);
            for (int l = 1; l < 4; l++)
                hipLaunchKernelGGL(k2_flow_half<false>, grid2d((wh >> l) / 2, (hh >> l) / 2), dim3(256), 0, st, reinterpret_cast<const float*>(c.fl[l - 1]), 0, c.fl[l],
                                   wh >> l, hh >> l);
            HIPCHK(hipGetLastError());
        }
        if (ctx0_img) {
            const ConvLayer& L0 = E.ctxc[0];
            Timed t(E.prof, L0.cls, L0.flops_per_pixel * (hp / 2) * (wp / 2), st);
            ImgConvArgs ia;
            ia.img = im ? img1 : img0; ia.out = c.ca; ia.wpk = L0.d_wimg; ia.bias = L0.d_bias; ia.slope = L0.d_slope;
            ia.wp = wp; ia.hp = hp; ia.Wo = wp / 2; ia.Ho = hp / 2; ia.tiles_x = (ia.Wo + 31) / 32; ia.ntiles = ia.tiles_x * ia.Ho;
            const int nwg = std::min((ia.ntiles + 3) / 4, 8 * device_cus(true));
            hipLaunchKernelGGL(conv_img_s2_kernel, dim3(nwg), dim3(256), 0, st, ia);
            HIPCHK(hipGetLastError());
        } else if ((rc = conv_t(E, E.ctxc[0], {c.I8, 8, 0}, hp, wp, {c.ca, 32, 0}, st))) return rc;
        if ((rc = conv_t(E, E.ctxc[1], {c.ca, 32, 0}, hp / 2, wp / 2, {c.cb, 32, 0}, st))) return rc;
        if ((rc = conv_t(E, E.ctxc[2], {c.cb, 32, 0}, hp / 2, wp / 2, {c.cc, 32, 0}, st))) return rc;
        if ((rc = conv_t(E, E.ctxc[3], {c.cc, 32, 0}, hp / 4, wp / 4, {c.feat[0], 32, 0}, st))) return rc;
        for (int l = 1; l < 4; l++) {
            const int Hl = hp >> (l + 1), Wl = wp >> (l + 1);      // input resolution of this level's strided conv
            if ((rc = conv_t(E, E.ctxc[2 + 2 * l], {c.feat[l - 1], lvl_c[l - 1], 0}, Hl, Wl, {c.ctmp[l - 1], lvl_c[l], 0}, st))) return rc;
            if ((rc = conv_t(E, E.ctxc[3 + 2 * l], {c.ctmp[l - 1], lvl_c[l], 0}, Hl / 2, Wl / 2, {c.feat[l], lvl_c[l], 0}, st))) return rc;
        }
        {
            Timed t(E.prof, E.prof_fine ? "ctx_warps" : "v2_ctx_misc", 0, st);
            for (int l = 0; l < 4; l++) {
                const int Hl = hp >> (l + 2), Wl = wp >> (l + 2), nq = lvl_c[l] / 4, ppb = 256 / nq;
                hipLaunchKernelGGL(k2_warp_nhwc, dim3((Wl + ppb - 1) / ppb, Hl), dim3(256), 0, st, c.feat[l], lvl_c[l], c.fl[l], cat_buf[l], cat_ld[l],
                                   cat_off[l] + im * lvl_c[l], Wl, Hl);
            }
            HIPCHK(hipGetLastError());
        }
    }
    // ---- FusionNet (fusionnet.param) ----
    const bool fused_f0 = stem2_fusable(E.fus[0], 1, wp, hp);
    if (!fused_f0) {
        Timed t(E.prof, E.prof_fine ? "fus_assemble" : "v2_assemble", 0, st);
        hipLaunchKernelGGL((k2_assemble<1, ImgU8>), grid2d(wp, hp), dim3(256), 0, st, ImgU8{img0}, ImgU8{img1}, acc, c.X, wp, hp);
        HIPCHK(hipGetLastError());
    }
    auto copy_view = [&](const float* src, int sld, int soff, float* dst, int dld, int doff, int C, size_t npix) {
        const size_t n = npix * (C / 4);
        hipLaunchKernelGGL(k2_copy_view, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, sld, soff, dst, dld, doff, C, npix);
    };
    const ConvLayer* F = E.fus;
    if (fused_f0) { if ((rc = launch_stem2_fused(E, F[0], 1, false, ImgU8{img0}, ImgU8{img1}, acc, wp, hp, c.e0a, 32, st))) return rc; }
    else if ((rc = conv_t(E, F[0], {c.X, 16, 0}, hp, wp, {c.e0a, 32, 0}, st))) return rc;
    if ((rc = conv_t(E, F[1], {c.e0a, 32, 0}, hp / 2, wp / 2, {c.e0b, 32, 0}, st))) return rc;
    if ((rc = conv_t(E, F[2], {c.e0b, 32, 0}, hp / 2, wp / 2, {c.e0c, 64, 0}, st))) return rc;
    // s0 / s1 / s2 are written twice by their producers - into the next encoder level's concat buffer and into the decoder's (Concat(up, s), fusionnet.param:53, 56, 59) -
    // instead of being copied (RIFE_HIP_V2_SKIP_COPY=1, A/B in the test build: the three k2_copy_view launches); the split-K form (tiny grids) has no second destination
    static const bool skip_copy_env = env_on(ab_getenv("RIFE_HIP_V2_SKIP_COPY"));
    auto dual_ok = [&](const ConvLayer& L, int H, int W) {
        const long nb = (long)((W + 31) / 32) * ((H + 7) / 8) * L.ntiles;
        return !skip_copy_env && g_trunk_h2 && L.nchunksh > 0 && L.NS <= 2 && !(L.NS == 2 && nb <= 64 && L.nchunksh >= 4);
    };
    const bool dual = dual_ok(F[3], hp / 4, wp / 4) && dual_ok(F[5], hp / 8, wp / 8) && dual_ok(F[7], hp / 16, wp / 16);
    const TensorView u2v{c.U2, 128, 64}, u1v{c.U1, 256, 128}, u0v{c.U0, 512, 256};
    if ((rc = conv_t(E, F[3], {c.e0c, 64, 0}, hp / 4, wp / 4, {c.B1, 128, 0}, st, nullptr, nullptr, dual ? &u2v : nullptr))) return rc;            // s0 -> B1[0:64] (+ U2[64:128])
    if ((rc = conv_t(E, F[4], {c.B1, 128, 0}, hp / 4, wp / 4, {c.e1a, 128, 0}, st))) return rc;
    if ((rc = conv_t(E, F[5], {c.e1a, 128, 0}, hp / 8, wp / 8, {c.B2, 256, 0}, st, nullptr, nullptr, dual ? &u1v : nullptr))) return rc;           // s1 -> B2[0:128] (+ U1[128:256])
    if ((rc = conv_t(E, F[6], {c.B2, 256, 0}, hp / 8, wp / 8, {c.e2a, 256, 0}, st))) return rc;
    if ((rc = conv_t(E, F[7], {c.e2a, 256, 0}, hp / 16, wp / 16, {c.B3, 512, 0}, st, nullptr, nullptr, dual ? &u0v : nullptr))) return rc;         // s2 -> B3[0:256] (+ U0[256:512])
    if ((rc = conv_t(E, F[8], {c.B3, 512, 0}, hp / 16, wp / 16, {c.e3a, 512, 0}, st))) return rc;
    if ((rc = conv_t(E, F[9], {c.e3a, 512, 0}, hp / 32, wp / 32, {c.B4, 1024, 0}, st))) return rc;        // s3 -> B4[0:512]
    if (!dual) {
        Timed t(E.prof, "v2_skip_copy", 0, st);
        copy_view(c.B3, 512, 0, c.U0, 512, 256, 256, (size_t)(hp / 16) * (wp / 16));                      // Concat(up0, s2)
        copy_view(c.B2, 256, 0, c.U1, 256, 128, 128, (size_t)(hp / 8) * (wp / 8));                        // Concat(up1, s1)
        copy_view(c.B1, 128, 0, c.U2, 128, 64, 64, (size_t)(hp / 4) * (wp / 4));                          // Concat(up2, s0)
        HIPCHK(hipGetLastError());
    }
    if ((rc = conv_t(E, F[10], {c.B4, 1024, 0}, hp / 32, wp / 32, {c.U0, 512, 0}, st))) return rc;
    if ((rc = conv_t(E, F[11], {c.U0, 512, 0}, hp / 16, wp / 16, {c.U1, 256, 0}, st))) return rc;
    if ((rc = conv_t(E, F[12], {c.U1, 256, 0}, hp / 8, wp / 8, {c.U2, 128, 0}, st))) return rc;
    if ((rc = conv_t(E, F[13], {c.U2, 128, 0}, hp / 4, wp / 4, {c.U3, 32, 0}, st))) return rc;
    if ((rc = conv_t(E, F[14], {c.U3, 32, 0}, hp / 2, wp / 2, {reinterpret_cast<float*>(c.head), 4, 0}, st))) return rc;
    {
        Timed t(E.prof, "final", 0, st);
        if (outf) hipLaunchKernelGGL(k2_final_float, grid2d(wp, hp), dim3(256), 0, st, img0, img1, acc, c.head, outf, wp, hp);
        else hipLaunchKernelGGL(k2_final, grid2d(c.w, c.h), dim3(256), 0, st, img0, img1, acc, c.head, d_out, c.w, c.h, wp, hp);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// RIFE::process for the v2 family: plain branch rife.cpp:878-1183; TTA branches 459-877 (CPU twin 1256-2138) with
// nori = 8 orientations (-x) and / or ntemp = 2 time directions (-z); SURVEY App. G.
static int run_v2(const rife_hip& E, Ctx& c, const uint8_t* d_in0, const uint8_t* d_in1, uint8_t* d_out) {
    hipStream_t st = c.stream;
    const int wp = c.wp, hp = c.hp;
    const int nori = E.tta ? 8 : 1, ntemp = E.tta_temporal ? 2 : 1;
    int rc;
    if (nori * ntemp == 1) {
        {
            Timed t(E.prof, "preproc", 0, st);
            launch_preproc(st, d_in0, c.w, c.h, c.img0, wp, hp);
            launch_preproc(st, d_in1, c.w, c.h, c.img1, wp, hp);
            HIPCHK(hipGetLastError());
        }
        if ((rc = run_v2_flow(E, c, c.img0, c.img1, wp, hp, c.acc))) return rc;
        return run_v2_synth(E, c, c.img0, c.img1, c.acc, wp, hp, d_out, nullptr);
    }
    {
        Timed t(E.prof, "preproc", 0, st);
        if (nori == 8) {
            Ptr8 a, b;
            for (int ti = 0; ti < 8; ti++) { a.p[ti] = c.timg0[ti]; b.p[ti] = c.timg1[ti]; }
            hipLaunchKernelGGL(k_preproc_tta, tta_grid(wp, hp, 4), tta_block(4), 0, st, d_in0, c.w, c.h, a, wp, hp);
            hipLaunchKernelGGL(k_preproc_tta, tta_grid(wp, hp, 4), tta_block(4), 0, st, d_in1, c.w, c.h, b, wp, hp);
        } else {
            launch_preproc(st, d_in0, c.w, c.h, c.timg0[0], wp, hp);
            launch_preproc(st, d_in1, c.w, c.h, c.timg1[0], wp, hp);
        }
        HIPCHK(hipGetLastError());
    }
    const size_t nflow = (size_t)(wp / 2) * (hp / 2);
    const unsigned gflow = (unsigned)((nflow + 255) / 256);
    auto ow = [&](int ti) { return ti < 4 ? wp : hp; };
    auto oh = [&](int ti) { return ti < 4 ? hp : wp; };
    for (int ti = 0; ti < nori; ti++) {
        if ((rc = run_v2_flow(E, c, c.timg0[ti], c.timg1[ti], ow(ti), oh(ti), c.tflow[0][ti]))) return rc;
        if (ntemp == 2) {
            if ((rc = run_v2_flow(E, c, c.timg1[ti], c.timg0[ti], ow(ti), oh(ti), c.tflow[1][ti]))) return rc;
            Timed t(E.prof, "tta_merge", 0, st);
            hipLaunchKernelGGL(k2_temporal_merge, dim3(gflow), dim3(256), 0, st, c.tflow[0][ti], c.tflow[1][ti], nflow);
            HIPCHK(hipGetLastError());
        }
    }
    if (nori == 8) {
        Timed t(E.prof, "tta_merge", 0, st);
        for (int d = 0; d < ntemp; d++) {
            Ptr8 f;
            for (int ti = 0; ti < 8; ti++) f.p[ti] = c.tflow[d][ti];
            hipLaunchKernelGGL(k2_spatial_avg, tta_grid(wp / 2, hp / 2, 16), tta_block(16), 0, st, f, wp / 2, hp / 2);
        }
        if (ntemp == 2)
            for (int ti = 0; ti < 8; ti++) hipLaunchKernelGGL(k2_temporal_merge, dim3(gflow), dim3(256), 0, st, c.tflow[0][ti], c.tflow[1][ti], nflow);
        HIPCHK(hipGetLastError());
    }
    // the reference's reversed FusionNet pass re-uses the forward contexts swapped (rife.cpp:2026-2047); flow_reversed is
    // (z, w, x, y) of flow after the merge, so recomputing ContextNet(img1, flow_reversed[0:2]) is the identical computation
    for (int ti = 0; ti < nori; ti++) {
        if ((rc = run_v2_synth(E, c, c.timg0[ti], c.timg1[ti], c.tflow[0][ti], ow(ti), oh(ti), nullptr, c.toutf[0][ti]))) return rc;
        if (ntemp == 2 && (rc = run_v2_synth(E, c, c.timg1[ti], c.timg0[ti], c.tflow[1][ti], ow(ti), oh(ti), nullptr, c.toutf[1][ti]))) return rc;
    }
    {
        Timed t(E.prof, "final", 0, st);
        Ptr16 outs;
        for (int d = 0; d < 2; d++) for (int ti = 0; ti < 8; ti++) outs.p[d * 8 + ti] = c.toutf[d][ti];
        hipLaunchKernelGGL(k_postproc_tta, tta_grid(c.w, c.h, 16), tta_block(16), 0, st, outs, nori, ntemp, d_out, c.w, c.h, wp, hp);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// v1 family (models/rife, rife-HD, rife-UHD, rife-anime): RIFE::process with rife_v2 = rife_v4 = false (rife.cpp:381-1212, CPU twin
// 1214-2460) on the generic graph executor.  One 2-channel flow; frame 0's ContextNet binds it to "flow.0", frame 1's to "flow.1"
// (the graph negates it, contextnet.param:4-5; rife.cpp:1027-1060).  -u, -x and -z like the v2 family, with the 2-channel algebra.
// ------------------------------------------------------------------------------------------------
static int ensure_ctx_v1(Ctx& c, int w, int h, int nori, int ntemp) {
    const int wp = (w + 31) / 32 * 32, hp = (h + 31) / 32 * 32;
    const bool ens = nori * ntemp > 1;
    if (c.v2 && c.wp == wp && c.hp == hp && c.w == w && c.h == h && c.img0 && (!ens || (c.toutf[0][0] && c.toutf[ntemp - 1][nori - 1]))) return 0;
    for (int d = 0; d < 2; d++) for (int t = 0; t < 8; t++) { c.tflow[d][t] = c.toutf[d][t] = nullptr; if (!d) c.timg0[t] = c.timg1[t] = nullptr; }
    for (void* p : c.allocs) (void)hipFree(p);
    c.allocs.clear();
    for (auto& o : c.ginst) for (auto& g : o) g.reset();
    c.v2 = true; c.w = w; c.h = h; c.wp = wp; c.hp = hp;
    c.h0 = c.h1 = c.acc_s = nullptr; c.T2 = nullptr;
    const size_t P = (size_t)wp * hp;
    int rc;
    if ((rc = dalloc(c, c.d_in0, (size_t)w * h * 3))) { reset_ctx(c); return rc; }
    if ((rc = dalloc(c, c.d_in1, (size_t)w * h * 3))) { reset_ctx(c); return rc; }
    if ((rc = dalloc(c, c.d_out, (size_t)w * h * 3))) { reset_ctx(c); return rc; }
    if ((rc = dalloc(c, c.img0, P))) { reset_ctx(c); return rc; }
    if ((rc = dalloc(c, c.img1, P))) { reset_ctx(c); return rc; }
    c.timg0[0] = c.img0; c.timg1[0] = c.img1;
    for (int t = 1; t < nori; t++) { if ((rc = dalloc(c, c.timg0[t], P))) { reset_ctx(c); return rc; } if ((rc = dalloc(c, c.timg1[t], P))) { reset_ctx(c); return rc; } }
    if (ens) for (int d = 0; d < ntemp; d++) for (int t = 0; t < nori; t++) if ((rc = dalloc(c, c.toutf[d][t], P))) { reset_ctx(c); return rc; }
    return 0;
}

static int run_v1(const rife_hip& E, Ctx& c, const uint8_t* d_in0, const uint8_t* d_in1, uint8_t* d_out) {
    hipStream_t st = c.stream;
    const int wp = c.wp, hp = c.hp;
    const int nori = E.tta ? 8 : 1, ntemp = E.tta_temporal ? 2 : 1;
    int rc;
    {
        Timed t(E.prof, "preproc", 0, st);
        if (nori == 8) {
            Ptr8 a, b;
            for (int ti = 0; ti < 8; ti++) { a.p[ti] = c.timg0[ti]; b.p[ti] = c.timg1[ti]; }
            hipLaunchKernelGGL(k_preproc_tta, tta_grid(wp, hp, 4), tta_block(4), 0, st, d_in0, c.w, c.h, a, wp, hp);
            hipLaunchKernelGGL(k_preproc_tta, tta_grid(wp, hp, 4), tta_block(4), 0, st, d_in1, c.w, c.h, b, wp, hp);
        } else {
            launch_preproc(st, d_in0, c.w, c.h, c.timg0[0], wp, hp);
            launch_preproc(st, d_in1, c.w, c.h, c.timg1[0], wp, hp);
        }
        HIPCHK(hipGetLastError());
    }
    // tensors outside the three nets live in the "aux" instance of their orientation; slot numbers are fixed:
    //   0 / 1 frames as 3-channel blobs, 2 / 3 their half-size versions (-u), 4 upscaled flow, 5 doubled flow,
    //   8 + dir * 8 + ti: the flow of pass (dir, ti)
    auto inst = [&](int o, int k) -> GraphInst& {
        if (!c.ginst[o][k]) c.ginst[o][k].reset(new GraphInst);
        return *c.ginst[o][k];
    };
    auto aux = [&](int o) -> GraphInst& {
        GraphInst& A = inst(o, 4);
        if (A.v.size() != 32) { A.v.assign(32, GView{nullptr, 0, 0, 0, 0}); A.owned.assign(32, nullptr); A.cap.assign(32, 0); }
        return A;
    };
    auto frames = [&](int ti, int o, int W, int H) -> int {      // RGBX -> "input0" / "input1" style blobs (slots 0, 1)
        GraphInst& A = aux(o);
        int r;
        if ((r = g_alloc(A, 0, 3, H, W, false, st))) return r;
        if ((r = g_alloc(A, 1, 3, H, W, false, st))) return r;
        const size_t P = (size_t)W * H;
        hipLaunchKernelGGL(kg_from_rgbx, dim3(g_blocks(P)), dim3(256), 0, st, (const uint32_t*)c.timg0[ti], A.v[0].p, A.v[0].ld, P);
        hipLaunchKernelGGL(kg_from_rgbx, dim3(g_blocks(P)), dim3(256), 0, st, (const uint32_t*)c.timg1[ti], A.v[1].p, A.v[1].ld, P);
        HIPCHK(hipGetLastError());
        return 0;
    };
    // flow of (first, second) -> slot `dst` of the aux instance (2 channels, half resolution); rife.cpp:912-950
    auto flow_of = [&](int o, int first, int second, int dst) -> int {
        GraphInst& A = aux(o);
        GraphInst& F = inst(o, 0);
        int r;
        GView fl;
        if (E.uhd) {
            const GView a = A.v[first], b = A.v[second];
            if ((r = g_alloc(A, 2, 3, a.h / 2, a.w / 2, false, st))) return r;
            if ((r = g_alloc(A, 3, 3, a.h / 2, a.w / 2, false, st))) return r;
            {
                Timed t(E.prof, "g_interp", 0, st);
                hipLaunchKernelGGL(kg_interp, grid2d(a.w / 2, a.h / 2), dim3(256), 0, st, a, A.v[2]);     // rife_uhd_downscale_image (rife.cpp:294-305)
                hipLaunchKernelGGL(kg_interp, grid2d(a.w / 2, a.h / 2), dim3(256), 0, st, b, A.v[3]);
            }
            if ((r = graph_run(E, *E.gflow, F, st, {{"input0", A.v[2]}, {"input1", A.v[3]}}, {"flow"}))) return r;
            const GView fd = F.v[E.gflow->blob("flow")];
            if ((r = g_alloc(A, 4, fd.c, fd.h * 2, fd.w * 2, false, st))) return r;
            if ((r = g_alloc(A, 5, fd.c, fd.h * 2, fd.w * 2, false, st))) return r;
            Timed t(E.prof, "g_interp", 0, st);
            hipLaunchKernelGGL(kg_interp, grid2d(fd.w * 2, fd.h * 2), dim3(256), 0, st, fd, A.v[4]);          // rife_uhd_upscale_flow (306-318)
            hipLaunchKernelGGL(kg_binary_scalar, dim3(g_blocks((size_t)A.v[4].h * A.v[4].w * fd.c)), dim3(256), 0, st, A.v[4], A.v[5], 2, 2.0f);   // rife_uhd_double_flow (319-332)
            fl = A.v[5];
        } else {
            if ((r = graph_run(E, *E.gflow, F, st, {{"input0", A.v[first]}, {"input1", A.v[second]}}, {"flow"}))) return r;
            fl = F.v[E.gflow->blob("flow")];
        }
        if (fl.c != 2) return fail(RIFE_HIP_EMODEL, "the v1-family flownet must produce a 2-channel flow");
        if ((r = g_alloc(A, dst, 2, fl.h, fl.w, false, st))) return r;
        hipLaunchKernelGGL(kg_copy_channels, dim3(g_blocks((size_t)fl.h * fl.w * 2)), dim3(256), 0, st, (const float*)fl.p, fl.ld, 0, A.v[dst].p, A.v[dst].ld, 0, 2, (size_t)fl.h * fl.w);
        HIPCHK(hipGetLastError());
        return 0;
    };
    // (first, second, flow) -> FusionNet "output" view
    auto synth = [&](int o, int first, int second, int flow_slot, GView& out) -> int {
        GraphInst& A = aux(o);
        int r;
        static const char* const fn[4] = {"f1", "f2", "f3", "f4"};
        if ((r = graph_run(E, *E.gctx, inst(o, 1), st, {{"input.1", A.v[first]}, {"flow.0", A.v[flow_slot]}}, {"f1", "f2", "f3", "f4"}))) return r;
        if ((r = graph_run(E, *E.gctx, inst(o, 2), st, {{"input.1", A.v[second]}, {"flow.1", A.v[flow_slot]}}, {"f1", "f2", "f3", "f4"}))) return r;
        std::vector<std::pair<std::string, GView>> in = {{"img0", A.v[first]}, {"img1", A.v[second]}, {"flow", A.v[flow_slot]}};
        static const char* const n0[4] = {"3", "4", "5", "6"};
        static const char* const n1[4] = {"7", "8", "9", "10"};
        for (int k = 0; k < 4; k++) {
            in.push_back({n0[k], inst(o, 1).v[E.gctx->blob(fn[k])]});
            in.push_back({n1[k], inst(o, 2).v[E.gctx->blob(fn[k])]});
        }
        if ((r = graph_run(E, *E.gfus, inst(o, 3), st, in, {"output"}))) return r;
        out = inst(o, 3).v[E.gfus->blob("output")];
        if (out.c != 3) return fail(RIFE_HIP_EMODEL, "the FusionNet output must have 3 channels");
        return 0;
    };
    auto ow = [&](int ti) { return ti < 4 ? wp : hp; };
    auto oh = [&](int ti) { return ti < 4 ? hp : wp; };
    if (nori * ntemp == 1) {
        GView out;
        if ((rc = frames(0, 0, wp, hp))) return rc;
        if ((rc = flow_of(0, 0, 1, 8))) return rc;
        if ((rc = synth(0, 0, 1, 8, out))) return rc;
        Timed t(E.prof, "final", 0, st);
        hipLaunchKernelGGL(kg_to_u8, grid2d(c.w, c.h), dim3(256), 0, st, out, d_out, c.w, c.h);
        HIPCHK(hipGetLastError());
        return 0;
    }
    // ---- ensembles: all flows first (they are merged across passes), then one synthesis per pass.  The frames of an orientation are
    // converted again for the synthesis stage because orientations of the same shape share the aux slots 0 / 1.
    for (int ti = 0; ti < nori; ti++) {
        const int o = ti < 4 ? 0 : 1;
        if ((rc = frames(ti, o, ow(ti), oh(ti)))) return rc;
        if ((rc = flow_of(o, 0, 1, 8 + ti))) return rc;
        if (ntemp == 2) {
            if ((rc = flow_of(o, 1, 0, 16 + ti))) return rc;
            GraphInst& A = aux(o);
            Timed t(E.prof, "tta_merge", 0, st);
            hipLaunchKernelGGL(kg_v1_temporal_merge, dim3(g_blocks((size_t)A.v[8 + ti].h * A.v[8 + ti].w)), dim3(256), 0, st, A.v[8 + ti], A.v[16 + ti]);
            HIPCHK(hipGetLastError());
        }
    }
    if (nori == 8) {
        Timed t(E.prof, "tta_merge", 0, st);
        for (int d = 0; d < ntemp; d++) {
            Ptr8 f;
            for (int ti = 0; ti < 8; ti++) f.p[ti] = aux(ti < 4 ? 0 : 1).v[8 + d * 8 + ti].p;
            const GView f0 = aux(0).v[8 + d * 8];
            hipLaunchKernelGGL(kg_v1_spatial_avg, tta_grid(f0.w, f0.h, 16), tta_block(16), 0, st, f, f0.ld, f0.w, f0.h);
        }
        if (ntemp == 2)
            for (int ti = 0; ti < 8; ti++) {
                GraphInst& A = aux(ti < 4 ? 0 : 1);
                hipLaunchKernelGGL(kg_v1_temporal_merge, dim3(g_blocks((size_t)A.v[8 + ti].h * A.v[8 + ti].w)), dim3(256), 0, st, A.v[8 + ti], A.v[16 + ti]);
            }
        HIPCHK(hipGetLastError());
    }
    for (int ti = 0; ti < nori; ti++) {
        const int o = ti < 4 ? 0 : 1;
        if ((rc = frames(ti, o, ow(ti), oh(ti)))) return rc;
        for (int d = 0; d < ntemp; d++) {
            GView out;
            // reversed pass: frames swapped, flow_reversed = -flow after the merge; the contexts the reference re-uses swapped
            // (rife.cpp:1099-1131) are the same computation
            if ((rc = synth(o, d ? 1 : 0, d ? 0 : 1, 8 + d * 8 + ti, out))) return rc;
            hipLaunchKernelGGL(kg_to_float4, dim3(g_blocks((size_t)out.h * out.w)), dim3(256), 0, st, out, c.toutf[d][ti]);
            HIPCHK(hipGetLastError());
        }
    }
    {
        Timed t(E.prof, "final", 0, st);
        Ptr16 outs;
        for (int d = 0; d < 2; d++) for (int ti = 0; ti < 8; ti++) outs.p[d * 8 + ti] = c.toutf[d][ti];
        hipLaunchKernelGGL(k_postproc_tta, tta_grid(c.w, c.h, 16), tta_block(16), 0, st, outs, nori, ntemp, d_out, c.w, c.h, wp, hp);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

static int load_v1(rife_hip* E, const std::string& dir) {
    E->gflow.reset(new GraphNet); E->gctx.reset(new GraphNet); E->gfus.reset(new GraphNet);
    int rc;
    if ((rc = graph_load(*E->gflow, dir + "/flownet"))) return rc;
    if ((rc = graph_load(*E->gctx, dir + "/contextnet"))) return rc;
    if ((rc = graph_load(*E->gfus, dir + "/fusionnet"))) return rc;
    // the blob-name contract RIFE::process relies on (rife.cpp:948-950, 1027-1060, 1070-1098)
    static const char* const need_f[] = {"input0", "input1", "flow"};
    static const char* const need_c[] = {"input.1", "flow.0", "flow.1", "f1", "f2", "f3", "f4"};
    static const char* const need_u[] = {"img0", "img1", "flow", "3", "4", "5", "6", "7", "8", "9", "10", "output"};
    for (const char* n : need_f) if (E->gflow->blob(n) < 0) return fail(RIFE_HIP_EMODEL, dir + "/flownet.param has no blob " + n);
    for (const char* n : need_c) if (E->gctx->blob(n) < 0) return fail(RIFE_HIP_EMODEL, dir + "/contextnet.param has no blob " + n + " (not a v1-family model?)");
    for (const char* n : need_u) if (E->gfus->blob(n) < 0) return fail(RIFE_HIP_EMODEL, dir + "/fusionnet.param has no blob " + n);
    E->v1 = true;
    return 0;
}

// weights of the three v2 nets -> ConvLayers (conv/deconv each optionally followed by its PReLU in the .bin stream)
static int load_v2(rife_hip* E, const std::string& dir) {
    NcnnModel mf, mc, mu;
    if (!mf.load_param(dir + "/flownet.param")) return fail(RIFE_HIP_EIO, mf.error);
    if (!mc.load_param(dir + "/contextnet.param")) return fail(RIFE_HIP_EIO, mc.error);
    if (!mu.load_param(dir + "/fusionnet.param")) return fail(RIFE_HIP_EIO, mu.error);
    const uint64_t fh = mf.structural_hash("flow");
    E->v3 = fh == RIFE_V3_HASH_FLOW;
    E->n_fblk = E->v3 ? 3 : 4;
    if ((fh != RIFE_V23_HASH_FLOW && fh != RIFE_V3_HASH_FLOW) || mc.structural_hash("f1") != RIFE_V23_HASH_F1 ||
        mc.structural_hash("f2") != RIFE_V23_HASH_F2 || mc.structural_hash("f3") != RIFE_V23_HASH_F3 ||
        mc.structural_hash("f4") != RIFE_V23_HASH_F4 || mu.structural_hash("output") != RIFE_V23_HASH_OUTPUT)
        return fail(RIFE_HIP_EMODEL, dir + " does not hold the rife-v2.x / rife-v3.x IFNet/ContextNet/FusionNet graphs this engine schedules");
    if (!mf.load_bin(dir + "/flownet.bin")) return fail(RIFE_HIP_EIO, mf.error);
    if (!mc.load_bin(dir + "/contextnet.bin")) return fail(RIFE_HIP_EIO, mc.error);
    if (!mu.load_bin(dir + "/fusionnet.bin")) return fail(RIFE_HIP_EIO, mu.error);
    int rc;
    // RIFE_HIP_PROFILE_FINE=1: one profile class per layer position (fb<b>_stem0 / _stem1 / _trunk / _head, ctx<i>, fus<i>) instead of the coarse classes
    // bench.py reports - what tools/part_profile.py reads on a CU-masked stream, where rocprofv3 cannot follow (its queue interception drops the mask)
    const bool fine = env_on(getenv("RIFE_HIP_PROFILE_FINE"));
    E->prof_fine = fine;
    std::string fine_name;
    auto take = [&](std::vector<const NcnnLayer*>& wl, size_t& k, ConvLayer& L, int cin, int cout, int stride, bool deconv, int epi, const char* cls0) -> int {
        const char* cls = fine && !fine_name.empty() ? fine_name.c_str() : cls0;
        if (k >= wl.size()) return fail(RIFE_HIP_EMODEL, "weight stream ended early");
        const NcnnLayer* nl = wl[k++];
        const int kk = deconv ? 16 : 9;
        if (nl->type != (deconv ? "Deconvolution" : "Convolution") || nl->geti(0, 0) != cout || (int)nl->weight.size() != cin * cout * kk ||
            nl->geti(3, 1) != stride)
            return fail(RIFE_HIP_EMODEL, "weighted layer " + nl->name + " does not match the rife-v2.x schedule");
        const float* slope = nullptr;
        if (k < wl.size() && wl[k]->type == "PReLU") {
            if ((int)wl[k]->slope.size() != cout) return fail(RIFE_HIP_EMODEL, "PReLU width mismatch after " + nl->name);
            slope = wl[k++]->slope.data();
        }
        free_layer(L);
        L.cin = cin; L.cout = cout; L.stride = deconv ? 1 : stride; L.deconv = deconv; L.epi = epi; L.cls = cls; L.tag = 0;
        return upload_layer(L, nl->weight.data(), nl->bias.data(), slope, 1.0f);
    };
    {
        std::vector<const NcnnLayer*> wl = mf.weighted(); size_t k = 0;
        static const int C2[4] = {384, 256, 192, 96}, SC2[4] = {8, 4, 2, 1}, C3[4] = {160, 160, 160, 0}, SC3[4] = {4, 2, 1, 1};
        const int* C = E->v3 ? C3 : C2; const int* SC = E->v3 ? SC3 : SC2;
        for (int b = 0; b < E->n_fblk; b++) {
            rife_hip::V2Block& B = E->fblk[b];
            B.c = C[b]; B.scale = SC[b];
            const std::string fb = "fb" + std::to_string(b);
            fine_name = fb + "_stem0";
            if ((rc = take(wl, k, B.stem0, b == 0 ? 6 : 10, C[b] / 2, 2, false, EPI_STORE, "v2_flow_stem"))) return rc;
            fine_name = fb + "_stem1";
            if ((rc = take(wl, k, B.stem1, C[b] / 2, C[b], 2, false, EPI_STORE, "v2_flow_stem"))) return rc;
            fine_name = fb + "_trunk";
            for (int i = 0; i < 6; i++)
                if ((rc = take(wl, k, B.conv[i], C[b], C[b], 1, false, EPI_STORE, b == 0 ? "v2_flow_trunk_b0" : b == 1 ? "v2_flow_trunk_b1" : b == 2 ? "v2_flow_trunk_b2" : "v2_flow_trunk_b3"))) return rc;
            fine_name = fb + "_head";
            if ((rc = take(wl, k, B.head, C[b], 4, 2, true, EPI_DECONV, "v2_flow_head"))) return rc;
        }
        if (k != wl.size()) return fail(RIFE_HIP_EMODEL, "flownet.bin has extra weighted layers");
    }
    {
        std::vector<const NcnnLayer*> wl = mc.weighted(); size_t k = 0;
        static const int CI[10] = {3, 32, 32, 32, 32, 64, 64, 128, 128, 256}, CO[10] = {32, 32, 32, 32, 64, 64, 128, 128, 256, 256};
        static const int ST[10] = {2, 1, 2, 1, 2, 1, 2, 1, 2, 1};
        for (int i = 0; i < 10; i++) {
            fine_name = "ctx" + std::to_string(i);
            if ((rc = take(wl, k, E->ctxc[i], CI[i], CO[i], ST[i], false, EPI_STORE, "v2_context"))) return rc;
        }
        if (k != wl.size()) return fail(RIFE_HIP_EMODEL, "contextnet.bin has extra weighted layers");
    }
    {
        std::vector<const NcnnLayer*> wl = mu.weighted(); size_t k = 0;
        static const int CI[10] = {10, 32, 32, 64, 128, 128, 256, 256, 512, 512}, CO[10] = {32, 32, 64, 64, 128, 128, 256, 256, 512, 512};
        static const int ST[10] = {2, 1, 2, 1, 2, 1, 2, 1, 2, 1};
        for (int i = 0; i < 10; i++) {
            fine_name = "fus" + std::to_string(i);
            if ((rc = take(wl, k, E->fus[i], CI[i], CO[i], ST[i], false, EPI_STORE, "v2_fusion_down"))) return rc;
        }
        static const int UI[4] = {1024, 512, 256, 128}, UO[4] = {256, 128, 64, 32};
        for (int i = 0; i < 4; i++) {
            fine_name = "fus" + std::to_string(10 + i);
            if ((rc = take(wl, k, E->fus[10 + i], UI[i], UO[i], 2, true, EPI_DECONV, "v2_fusion_up"))) return rc;
        }
        fine_name = "fus14";
        if ((rc = take(wl, k, E->fus[14], 32, 4, 2, true, EPI_DECONV_SIG, "v2_fusion_head"))) return rc;
        if (k != wl.size()) return fail(RIFE_HIP_EMODEL, "fusionnet.bin has extra weighted layers");
    }
    return 0;
}

static int check_device(int gpuid) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RIFE_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
    if (gpuid < 0 || gpuid >= n) return fail(RIFE_HIP_ENODEV, "invalid gpu device");
    HIPCHK(hipSetDevice(gpuid));
    return 0;
}

}  // namespace rife

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

const char* rife_hip_last_error(void) { return g_err.c_str(); }

int rife_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

rife_hip_t* rife_hip_create(int gpuid, int tta_mode, int tta_temporal_mode, int uhd_mode, int num_threads, int rife_v2, int rife_v4) {
    if (check_device(gpuid)) return nullptr;
    rife_hip* E = new rife_hip;
    E->gpuid = gpuid; E->tta = tta_mode; E->tta_temporal = tta_temporal_mode; E->uhd = uhd_mode;
    E->num_threads = num_threads; E->v2 = rife_v2; E->v4 = rife_v4;
    E->frame_pool = std::make_shared<FramePool>();
    E->frame_pool->gpuid = gpuid;
    { const char* e = ab_getenv("RIFE_HIP_T64"); E->t64 = !(e && e[0] == '0'); }
    { const char* e = ab_getenv("RIFE_HIP_RS"); E->rs = !(e && e[0] == '0'); }
    { const char* e = ab_getenv("RIFE_HIP_RS2"); E->rs2 = !(e && e[0] == '0'); }
    { const char* e = ab_getenv("RIFE_HIP_KS"); if (e && e[0] >= '0' && e[0] <= '9') E->ks_mask = atoi(e); }
    { const char* e = ab_getenv("RIFE_HIP_STEM_RS"); E->stem_rs = !(e && e[0] == '0'); }
    { const char* e = ab_getenv("RIFE_HIP_TTA_CONSENSUS"); E->tta_consensus = !(e && e[0] == '0'); }
    { const char* e = ab_getenv("RIFE_HIP_TAIL_RS"); E->tail_rs = !(e && e[0] == '0'); E->tail_rs_always = e && e[0] == '2'; }
    { const char* e = ab_getenv("RIFE_HIP_FUSE_FLOW"); E->fuse_flow = e && e[0] == '1'; if (E->fuse_flow) g_fuse_flow_buffers = true; }
    return E;
}

void rife_hip_destroy(rife_hip_t* r) { delete r; }

static int rife_hip_load_impl(rife_hip_t* E, const char* modeldir) {
    if (!E || !modeldir) return fail(RIFE_HIP_EINVAL, "null argument");
    int rc;
    if ((rc = check_device(E->gpuid))) return rc;
    if (E->v2 && !E->v4) {
        if ((rc = load_v2(E, modeldir))) return rc;
        E->loaded = true;
        return 0;
    }
    if (!E->v4) {
        if ((rc = load_v1(E, modeldir))) return rc;
        E->loaded = true;
        return 0;
    }
    NcnnModel m;
    const std::string base = std::string(modeldir) + "/flownet";
    if (!m.load_param(base + ".param")) return fail(RIFE_HIP_EIO, m.error);
    const uint64_t gh = m.structural_hash("out0");
    if (gh != V46_HASH_OUT0 && gh != RIFE_V40_HASH_OUT0)
        return fail(RIFE_HIP_EMODEL, base + ".param is neither the rife-v4.6 nor the rife-v4 IFNet graph this engine schedules");
    if (!m.load_bin(base + ".bin")) return fail(RIFE_HIP_EIO, m.error);
    std::vector<const NcnnLayer*> wl = m.weighted();
    E->v40 = gh == RIFE_V40_HASH_OUT0;
    if (E->v40) {
        // rife-v4 (4.0): every conv is followed by its PReLU in the weight stream; the 5-channel head is padded to 8 output channels
        // (zero weights / bias) so that flow{b} keeps the [.][.][8] = {x, y, z, w, mask, 0, 0, 0} layout of the v4.6 schedule
        static const int C[4] = {192, 128, 96, 64}, SC[4] = {8, 4, 2, 1};
        size_t k = 0;
        for (int b = 0; b < 4; b++) {
            rife_hip::Block& B = E->blk[b];
            B.c = C[b]; B.scale = SC[b];
            static const char* const SN0[4] = {"stem0_b0", "stem0_b1", "stem0_b2", "stem0_b3"};
            static const char* const SN1[4] = {"stem1_b0", "stem1_b1", "stem1_b2", "stem1_b3"};
            static const char* const TN[4] = {"trunk_b0", "trunk_b1", "trunk_b2", "trunk_b3"};
            static const char* const HN[4] = {"head_b0", "head_b1", "head_b2", "head_b3"};
            auto take = [&](ConvLayer& L, int cin, int cout, int stride, bool deconv, const char* cls) -> int {
                if (k >= wl.size()) return fail(RIFE_HIP_EMODEL, "weight stream ended early");
                const NcnnLayer* nl = wl[k++];
                const int kk = deconv ? 16 : 9;
                if (nl->type != (deconv ? "Deconvolution" : "Convolution") || nl->geti(0, 0) != cout || (int)nl->weight.size() != cin * cout * kk ||
                    nl->geti(3, 1) != stride)
                    return fail(RIFE_HIP_EMODEL, "weighted layer " + nl->name + " does not match the rife-v4 schedule");
                free_layer(L);
                L.cin = cin; L.stride = deconv ? 1 : stride; L.deconv = deconv; L.epi = deconv ? EPI_DECONV : EPI_STORE; L.cls = cls; L.tag = 0; L.skip = false;
                if (deconv) {
                    L.cout = 8;
                    std::vector<float> w((size_t)8 * cin * 16, 0.f), bias(8, 0.f);
                    std::copy(nl->weight.begin(), nl->weight.end(), w.begin());              // ncnn deconv weights are [oc][ic][ky][kx]
                    std::copy(nl->bias.begin(), nl->bias.end(), bias.begin());
                    return upload_layer(L, w.data(), bias.data(), nullptr, 1.0f);
                }
                L.cout = cout;
                if (k >= wl.size() || wl[k]->type != "PReLU" || (int)wl[k]->slope.size() != cout) return fail(RIFE_HIP_EMODEL, "PReLU expected after " + nl->name);
                return upload_layer(L, nl->weight.data(), nl->bias.data(), wl[k++]->slope.data(), 1.0f);
            };
            if ((rc = take(B.stem0, b == 0 ? 7 : 12, C[b] / 2, 2, false, SN0[b]))) return rc;
            if ((rc = take(B.stem1, C[b] / 2, C[b], 2, false, SN1[b]))) return rc;
            for (int i = 0; i < 8; i++) if ((rc = take(B.res[i], C[b], C[b], 1, false, TN[b]))) return rc;
            if ((rc = take(B.head, C[b], 5, 2, true, HN[b]))) return rc;
        }
        if (k != wl.size()) return fail(RIFE_HIP_EMODEL, "flownet.bin has extra weighted layers");
        E->loaded = true;
        return 0;
    }
    if (wl.size() != 44) return fail(RIFE_HIP_EMODEL, "unexpected number of weighted layers");
    static const int C[4] = {192, 128, 96, 64}, SC[4] = {8, 4, 2, 1};
    size_t k = 0;
    for (int b = 0; b < 4; b++) {
        rife_hip::Block& B = E->blk[b];
        B.c = C[b]; B.scale = SC[b];
        char name[64];
        auto setup = [&](ConvLayer& L, int cin, int cout, int stride, bool deconv, int epi, float slope, const char* cls, bool fold_skip = false) -> int {
            const NcnnLayer* nl = wl[k++];
            const int kk = deconv ? 16 : 9;
            if (nl->type != (deconv ? "Deconvolution" : "Convolution") || nl->geti(0, 0) != cout || (int)nl->weight.size() != cin * cout * kk ||
                nl->geti(3, 1) != stride)
                return fail(RIFE_HIP_EMODEL, "weighted layer " + nl->name + " does not match the rife-v4.6 schedule");
            free_layer(L);
            L.cin = cin; L.cout = cout; L.stride = deconv ? 1 : stride; L.deconv = deconv; L.epi = epi; L.cls = cls;
            L.tag = std::strcmp(cls, "trunk_b3") == 0 ? 3 : 0;
            L.skip = fold_skip;
            L.want_t64 = fold_skip && cin == cout;
            L.want_s16out = !deconv && stride == 2 && cout == C[b];
            return upload_layer(L, nl->weight.data(), nl->bias.data(), nullptr, slope);
        };
        std::snprintf(name, sizeof name, "stem0_b%d", b);
        if ((rc = setup(B.stem0, b == 0 ? 7 : 12, C[b] / 2, 2, false, EPI_STORE, 0.2f, name))) return rc;
        std::snprintf(name, sizeof name, "stem1_b%d", b);
        if ((rc = setup(B.stem1, C[b] / 2, C[b], 2, false, EPI_STORE, 0.2f, name))) return rc;
        std::snprintf(name, sizeof name, "trunk_b%d", b);
        for (int i = 0; i < 8; i++)
            if ((rc = setup(B.res[i], C[b], C[b], 1, false, EPI_STORE, 0.2f, name, true))) return rc;
        std::snprintf(name, sizeof name, "head_b%d", b);
        if ((rc = setup(B.head, C[b], 24, 2, true, EPI_DECONV_PS, 1.0f, name))) return rc;
    }
    E->loaded = true;
    return 0;
}
int rife_hip_load(rife_hip_t* E, const char* modeldir) {      // nothing may throw across the C boundary (malformed model files, std::bad_alloc)
    try { return rife_hip_load_impl(E, modeldir); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EIO, std::string("rife_hip_load: ") + e.what()); }
    catch (...) { return fail(RIFE_HIP_EIO, "rife_hip_load: unknown exception"); }
}

static int process_common(const rife_hip* E, int w, int h, float timestep) {
    tl_cu_budget = 0;                                                    // every entry point starts on the whole chip; rife_hip_process_device sets its stream's part
    if (!E) return fail(RIFE_HIP_EINVAL, "null engine");
    if (!E->loaded) return fail(RIFE_HIP_EINVAL, "process() before load()");
    if (w <= 0 || h <= 0) return fail(RIFE_HIP_EINVAL, "bad frame size");
    if ((long long)((w + 31) / 32 * 32) * ((h + 31) / 32 * 32) > (1ll << 27))      // element indices are ints and the widest full-resolution tensor has 16 channels; (stem_rs has its own gate, block_on_stem_rs)
        return fail(RIFE_HIP_EINVAL, "frame too large (more than 2^27 padded pixels)");
    (void)timestep;
    if (E->uhd && !E->v4 && (((w + 31) / 32 * 32 / 2) % 32 || ((h + 31) / 32 * 32 / 2) % 32))
        return fail(RIFE_HIP_EINVAL, "UHD mode needs a padded frame whose half size is a multiple of 32 (the reference's graph mis-sizes otherwise)");
    return 0;
}

// lease a workspace (+ its private stream) from the pool of the host-buffer entry points
static int lease_ctx(const rife_hip* E, std::unique_ptr<Ctx>& c, int w, int h) {
    {
        std::lock_guard<std::mutex> g(E->mu);
        if (!E->free_ctx.empty()) { c = std::move(E->free_ctx.back()); E->free_ctx.pop_back(); }
    }
    if (!c) {
        c.reset(new Ctx);
        // RIFE_HIP_POOL_PARTS=n (A/B; default 1): the pool's streams own 1 / n of the compute units each (CU index mod n), like rife_hip_stream_create
        static const int parts = env_int(ab_getenv("RIFE_HIP_POOL_PARTS"), 1, 1, 16);
        static std::atomic<int> next{0};
        if (parts > 1) {
            const int ncu = device_cus(true), part = next++ % parts;
            std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
            int mine = 0;
            for (int cu = 0; cu < ncu; cu++) if (cu % parts == part) { mask[cu / 32] |= 1u << (cu % 32); mine++; }
            if (hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()) != hipSuccess) return fail(RIFE_HIP_EHIP, "hipExtStreamCreateWithCUMask failed");
            c->cu_budget = mine;
        } else if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(RIFE_HIP_EHIP, "hipStreamCreate failed");
        c->own_stream = true;
    }
    tl_cu_budget = c->cu_budget;                                         // the caller enqueues on this workspace's stream next
    return E->v4 ? ensure_ctx(*c, w, h) : E->v1 ? ensure_ctx_v1(*c, w, h, E->tta ? 8 : 1, E->tta_temporal ? 2 : 1)
                                                : ensure_ctx_v2(*c, w, h, E->uhd, E->tta ? 8 : 1, E->tta_temporal ? 2 : 1, E->v3);
}

static void release_ctx(const rife_hip* E, std::unique_ptr<Ctx>& c) {
    std::lock_guard<std::mutex> g(E->mu);
    E->free_ctx.push_back(std::move(c));
}

// H2D of both frames, the whole pass and the D2H of the result, all enqueued on the workspace's stream (no host wait)
static int enqueue_host_pair(const rife_hip* E, Ctx& c, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep, uint8_t* out) {
    const size_t nbytes = (size_t)w * h * 3;
    hipError_t e = hipMemcpyAsync(c.d_in0, in0, nbytes, hipMemcpyHostToDevice, c.stream);
    if (e == hipSuccess) e = hipMemcpyAsync(c.d_in1, in1, nbytes, hipMemcpyHostToDevice, c.stream);
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("H2D: ") + hipGetErrorString(e));
    int rc;
    if (E->v1) rc = run_v1(*E, c, c.d_in0, c.d_in1, c.d_out);
    else if (!E->v4) rc = run_v2(*E, c, c.d_in0, c.d_in1, c.d_out);
    else if (E->tta || E->tta_temporal) {
        // the TTA workspaces are shared by all callers: serialise, and drain before the next caller may reuse them
        std::lock_guard<std::mutex> g(E->tta_mu);
        rc = run_v4_tta(*E, c.stream, c.d_in0, c.d_in1, w, h, timestep, c.d_out);
        if (!rc && hipStreamSynchronize(c.stream) != hipSuccess) rc = fail(RIFE_HIP_EHIP, "TTA stream sync failed");
    } else rc = run_v4_replay(*E, c, c.d_in0, c.d_in1, timestep, c.d_out);
    if (rc) return rc;
    e = hipMemcpyAsync(out, c.d_out, nbytes, hipMemcpyDeviceToHost, c.stream);
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("D2H: ") + hipGetErrorString(e));
    return 0;
}

static int rife_hip_process_impl(const rife_hip_t* E, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep, uint8_t* out) {
    int rc;
    if ((rc = process_common(E, w, h, timestep))) return rc;
    if (!in0 || !in1 || !out) return fail(RIFE_HIP_EINVAL, "null frame pointer");
    const size_t nbytes = (size_t)w * h * 3;
    // rife.cpp:2470-2480: timestep 0 / 1 return an input frame unchanged (the reference rebinds the Mat; a copy is pixel-identical)
    if (timestep == 0.f) { std::memmove(out, in0, nbytes); return 0; }
    if (timestep == 1.f) { std::memmove(out, in1, nbytes); return 0; }
    if ((rc = check_device(E->gpuid))) return rc;
    std::unique_ptr<Ctx> c;
    rc = lease_ctx(E, c, w, h);
    if (!rc) rc = enqueue_host_pair(E, *c, in0, in1, w, h, timestep, out);
    if (c && hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(RIFE_HIP_EHIP, "stream sync failed");
    if (c) release_ctx(E, c);
    return rc;
}
int rife_hip_process(const rife_hip_t* E, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep, uint8_t* out) {      // nothing may throw across the C boundary (malformed model files, std::bad_alloc)
    try { return rife_hip_process_impl(E, in0, in1, w, h, timestep, out); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EIO, std::string("rife_hip_process: ") + e.what()); }
    catch (...) { return fail(RIFE_HIP_EIO, "rife_hip_process: unknown exception"); }
}

// n independent frame pairs from host memory in one call.  Copies from / to pageable host memory block the thread that issues
// them, so overlap of one pair's copies with another pair's kernels needs several host threads - the reference's proc threads
// (src/main.cpp:849-866).  The batch call brings its own: 2 workers (the reference default), each a plain rife_hip_process() loop over its share of
// the pairs (every call leases its own workspace + stream).  Same pixels as n rife_hip_process() calls.
struct rife_hip_frame {
    uint8_t* d = nullptr;      // tight u8 HWC RGB, the layout every run_* entry takes
    int w = 0, h = 0, gpuid = 0;
    size_t nbytes = 0;
    std::shared_ptr<FramePool> pool;
};

int rife_hip_process_batch(const rife_hip_t* E, int n, const uint8_t* const* in0, const uint8_t* const* in1, const float* timestep,
                           uint8_t* const* out, int w, int h) {
    int rc;
    if (n < 0 || (n > 0 && (!in0 || !in1 || !timestep || !out))) return fail(RIFE_HIP_EINVAL, "bad batch arguments");
    if ((rc = process_common(E, w, h, 0.5f))) return rc;
    for (int i = 0; i < n; i++) if (!in0[i] || !in1[i] || !out[i]) return fail(RIFE_HIP_EINVAL, "null frame pointer");
    if (n == 0) return 0;
    if ((rc = check_device(E->gpuid))) return rc;
    // three workers = three pairs in flight: tools/host_path_bench2.py, 4K, 48 pairs: process() from 1 / 2 / 3 / 4 caller threads
    // 192 / 341 / 389 / 365 frames/s from pageable frames (resident frames: 395), 244 / 307 / 349 / 344 from page-locked ones
    static const int batch_workers = env_int(getenv("RIFE_HIP_BATCH_WORKERS"), 0, 1, 16);      // A/B
    const int K = std::min(n, batch_workers ? batch_workers : 4);      // round 4: four workers (measured against 3 / 5 / 6 / 8: 4K 439 vs 426 / 426 / 431 / 446, 1080p 1,334 vs 1,251 / 1,367 / 1,443 / 1,398 pageable; page-locked best at 4)
    // A host frame that serves several pairs of the batch (consecutive pairs of a sequence share one) crosses PCIe once: it becomes a
    // resident frame on first use and is released after its last (stream mode, below).  Batches without shared frames run as before.
    struct Shared { std::mutex mu; rife_hip_frame_t* f = nullptr; int left = 0; };
    std::map<const uint8_t*, std::unique_ptr<Shared>> shared;
    bool any_shared = false;
    for (int i = 0; i < n; i++) {
        if (timestep[i] == 0.f || timestep[i] == 1.f) continue;
        for (const uint8_t* p : {in0[i], in1[i]}) {
            auto& sl = shared[p];
            if (!sl) sl.reset(new Shared);
            any_shared |= ++sl->left > 1;
        }
    }
    auto resident = [&](const uint8_t* p, rife_hip_frame_t*& f) -> int {
        Shared& sl = *shared.find(p)->second;
        std::lock_guard<std::mutex> g(sl.mu);
        const int r = sl.f ? 0 : rife_hip_frame_upload(E, p, w, h, &sl.f);
        f = sl.f;
        return r;
    };
    auto retire = [&](const uint8_t* p) {
        Shared& sl = *shared.find(p)->second;
        std::lock_guard<std::mutex> g(sl.mu);
        if (--sl.left == 0) { rife_hip_frame_release(sl.f); sl.f = nullptr; }
    };
    std::vector<int> rcs(K, 0);
    std::vector<std::string> errs(K);
    // Lockstep groups (plain rife-v4.6 on the S16 trunks): three workers, each takes groups of two consecutive pairs through run_v4_group - the coarse
    // blocks of a group are batched launches - so up to six pairs are in flight and one worker's copies overlap the others' passes.  A trailing odd pair,
    // timestep 0 / 1 copies and every other model family take the per-pair path below.
    // (only where the coarse grids leave CUs idle - block 0 on the row kernel, frames up to ~1080p: at 3840x2160 a coarse layer of ONE pair already
    // fills the chip, measured 405 - 413 frames/s in groups against 400 - 425 per pair; RIFE_HIP_BATCH_GROUPS=1 / 0 forces / forbids the path)
    const char* genv = ab_getenv("RIFE_HIP_BATCH_GROUPS");
    const int Ht0 = (h + 31) / 32, Wt0 = (w + 31) / 32;
    const bool small_grid = ((Wt0 + 31) / 32) * Ht0 <= device_cus(true) * 5 / 8;      // MI355X: 160 workgroups, block 0 on the row kernel (block_on_row_kernel)
    const bool groups = E->v4 && !E->v40 && !E->v1 && !E->tta && !E->tta_temporal && E->t64 && n >= 2 && (genv ? genv[0] != '0' : small_grid);
    if (groups) {
        std::vector<std::array<int, 2>> grp;                 // pair indices of a group, -1 = none
        std::vector<int> singles;
        {
            int pend = -1;
            for (int i = 0; i < n; i++) {
                if (timestep[i] == 0.f || timestep[i] == 1.f) { singles.push_back(i); continue; }
                if (pend < 0) pend = i; else { grp.push_back({pend, i}); pend = -1; }
            }
            if (pend >= 0) singles.push_back(pend);
        }
        const int KG = std::min<int>(batch_workers ? batch_workers : 4, (int)grp.size() + (singles.empty() ? 0 : 1));      // four workers x two pairs in flight
        std::vector<int> grc(std::max(KG, 1), 0);
        std::vector<std::string> gerr(std::max(KG, 1));
        const size_t nbytes = (size_t)w * h * 3;
        auto one_pair = [&](int i) -> int {
            if (timestep[i] == 0.f || timestep[i] == 1.f) return rife_hip_process(E, in0[i], in1[i], w, h, timestep[i], out[i]);
            rife_hip_frame_t *f0 = nullptr, *f1 = nullptr;
            int r = resident(in0[i], f0);
            if (!r) r = resident(in1[i], f1);
            if (!r) r = rife_hip_process_frames(E, f0, f1, timestep[i], out[i]);
            retire(in0[i]); retire(in1[i]);
            return r;
        };
        auto gworker = [&](int k) {
            (void)hipSetDevice(E->gpuid);
            std::unique_ptr<Ctx> c[2];
            int r = 0;
            for (size_t q = k; q < grp.size() && !r; q += KG) {
                const int ia = grp[q][0], ib = grp[q][1];
                rife_hip_frame_t* f[4] = {nullptr, nullptr, nullptr, nullptr};
                const uint8_t* hp[4] = {in0[ia], in1[ia], in0[ib], in1[ib]};
                int nres = 0;
                for (; nres < 4 && !r; nres++) r = resident(hp[nres], f[nres]);
                if (r) nres--;
                for (int g = 0; g < 2 && !r; g++) if (!c[g]) r = lease_ctx(E, c[g], w, h);
                if (!r) {
                    Ctx* cs[2] = {c[0].get(), c[1].get()};
                    const uint8_t* d0[2] = {f[0]->d, f[2]->d}; const uint8_t* d1[2] = {f[1]->d, f[3]->d};
                    const float ts[2] = {timestep[ia], timestep[ib]};
                    uint8_t* dout[2] = {c[0]->d_out, c[1]->d_out};
                    r = run_v4_group(*E, cs, 2, d0, d1, ts, dout);
                    if (!r && hipMemcpyAsync(out[ia], c[0]->d_out, nbytes, hipMemcpyDeviceToHost, c[0]->stream) != hipSuccess) r = fail(RIFE_HIP_EHIP, "D2H failed");
                    if (!r && hipMemcpyAsync(out[ib], c[1]->d_out, nbytes, hipMemcpyDeviceToHost, c[1]->stream) != hipSuccess) r = fail(RIFE_HIP_EHIP, "D2H failed");
                }
                for (int g = 0; g < 2; g++) if (c[g] && hipStreamSynchronize(c[g]->stream) != hipSuccess && !r) r = fail(RIFE_HIP_EHIP, "stream sync failed");
                for (int j = 0; j < nres; j++) retire(hp[j]);
            }
            for (int g = 0; g < 2; g++) if (c[g]) release_ctx(E, c[g]);
            if (!r && k == KG - 1) for (int i : singles) if ((r = one_pair(i))) break;       // the leftovers ride on the last worker
            if (r) { grc[k] = r; gerr[k] = g_err; }
        };
        std::vector<std::thread> gth;
        for (int k = 1; k < KG; k++) gth.emplace_back(gworker, k);
        if (KG > 0) gworker(0);
        for (auto& t : gth) t.join();
        for (auto& kv : shared) if (kv.second->f) rife_hip_frame_release(kv.second->f);      // only after an error
        for (int k = 0; k < KG; k++) if (grc[k]) { g_err = gerr[k]; return grc[k]; }
        return 0;
    }
    auto worker = [&](int k) {
        (void)hipSetDevice(E->gpuid);
        for (int i = k; i < n; i += K) {
            int r;
            if (!any_shared || timestep[i] == 0.f || timestep[i] == 1.f) r = rife_hip_process(E, in0[i], in1[i], w, h, timestep[i], out[i]);
            else {
                rife_hip_frame_t *f0 = nullptr, *f1 = nullptr;
                r = resident(in0[i], f0);
                if (!r) r = resident(in1[i], f1);
                if (!r) r = rife_hip_process_frames(E, f0, f1, timestep[i], out[i]);
                retire(in0[i]); retire(in1[i]);
            }
            if (r) { rcs[k] = r; errs[k] = g_err; return; }     // g_err is thread-local: carry it back to the caller
        }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < K; k++) th.emplace_back(worker, k);
    worker(0);
    for (auto& t : th) t.join();
    for (auto& kv : shared) if (kv.second->f) rife_hip_frame_release(kv.second->f);      // only after an error
    for (int k = 0; k < K; k++) if (rcs[k]) { g_err = errs[k]; return rcs[k]; }
    return 0;
}

// ---- stream mode: frames resident in device memory across calls (include/rife_hip.h) ----

static int rife_hip_frame_upload_impl(const rife_hip_t* E, const uint8_t* rgb, int w, int h, rife_hip_frame_t** frame) {
    if (frame) *frame = nullptr;
    if (!E || !rgb || !frame) return fail(RIFE_HIP_EINVAL, "null argument");
    if (w <= 0 || h <= 0) return fail(RIFE_HIP_EINVAL, "bad frame size");
    int rc;
    if ((rc = check_device(E->gpuid))) return rc;
    std::unique_ptr<rife_hip_frame> f(new rife_hip_frame);
    f->w = w; f->h = h; f->gpuid = E->gpuid;
    const size_t nbytes = (size_t)w * h * 3;
    f->nbytes = nbytes; f->pool = E->frame_pool;
    if (!(f->d = f->pool->take(nbytes))) return fail(RIFE_HIP_EHIP, "hipMalloc of a resident frame failed");
    // a copy on its own stream, drained here: the frame is complete before any stream of any caller can see the handle
    hipStream_t st = nullptr;
    {
        std::lock_guard<std::mutex> g(E->mu);
        if (!E->upload_streams.empty()) { st = E->upload_streams.back(); E->upload_streams.pop_back(); }
    }
    hipError_t e = st ? hipSuccess : hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(f->d, rgb, nbytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (st) { std::lock_guard<std::mutex> g(E->mu); E->upload_streams.push_back(st); }
    if (e != hipSuccess) { f->pool->give(f->d, nbytes); return fail(RIFE_HIP_EHIP, std::string("frame upload: ") + hipGetErrorString(e)); }
    *frame = f.release();
    return 0;
}
int rife_hip_frame_upload(const rife_hip_t* E, const uint8_t* rgb, int w, int h, rife_hip_frame_t** frame) {      // nothing may throw across the C boundary (malformed model files, std::bad_alloc)
    try { return rife_hip_frame_upload_impl(E, rgb, w, h, frame); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EIO, std::string("rife_hip_frame_upload: ") + e.what()); }
    catch (...) { return fail(RIFE_HIP_EIO, "rife_hip_frame_upload: unknown exception"); }
}

void rife_hip_frame_release(rife_hip_frame_t* f) {
    if (!f) return;
    if (f->d) f->pool->give(f->d, f->nbytes);
    delete f;
}

static int rife_hip_process_frames_impl(const rife_hip_t* E, const rife_hip_frame_t* f0, const rife_hip_frame_t* f1, float timestep, uint8_t* out) {
    if (!f0 || !f1 || !out) return fail(RIFE_HIP_EINVAL, "null frame pointer");
    if (f0->w != f1->w || f0->h != f1->h) return fail(RIFE_HIP_EINVAL, "the two frames differ in size");
    const int w = f0->w, h = f0->h;
    int rc;
    if ((rc = process_common(E, w, h, timestep))) return rc;
    if (f0->gpuid != E->gpuid || f1->gpuid != E->gpuid) return fail(RIFE_HIP_EINVAL, "frame was uploaded to another device");
    if ((rc = check_device(E->gpuid))) return rc;
    const size_t nbytes = (size_t)w * h * 3;
    if (timestep == 0.f || timestep == 1.f) {                 // rife.cpp:2470-2480 (a copy stream of the pool, never the legacy stream)
        hipStream_t st = nullptr;
        {
            std::lock_guard<std::mutex> g(E->mu);
            if (!E->upload_streams.empty()) { st = E->upload_streams.back(); E->upload_streams.pop_back(); }
        }
        hipError_t e = st ? hipSuccess : hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMemcpyAsync(out, timestep == 0.f ? f0->d : f1->d, nbytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (st) { std::lock_guard<std::mutex> g(E->mu); E->upload_streams.push_back(st); }
        if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("frame download: ") + hipGetErrorString(e));
        return 0;
    }
    std::unique_ptr<Ctx> c;
    rc = lease_ctx(E, c, w, h);
    if (!rc) {
        Ctx& C = *c;
        if (E->v1) rc = run_v1(*E, C, f0->d, f1->d, C.d_out);
        else if (!E->v4) rc = run_v2(*E, C, f0->d, f1->d, C.d_out);
        else if (E->tta || E->tta_temporal) {
            std::lock_guard<std::mutex> g(E->tta_mu);
            rc = run_v4_tta(*E, C.stream, f0->d, f1->d, w, h, timestep, C.d_out);
            if (!rc && hipStreamSynchronize(C.stream) != hipSuccess) rc = fail(RIFE_HIP_EHIP, "TTA stream sync failed");
        } else rc = run_v4_replay(*E, C, f0->d, f1->d, timestep, C.d_out);
        if (!rc && hipMemcpyAsync(out, C.d_out, nbytes, hipMemcpyDeviceToHost, C.stream) != hipSuccess) rc = fail(RIFE_HIP_EHIP, "D2H failed");
    }
    if (c && hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(RIFE_HIP_EHIP, "stream sync failed");
    if (c) release_ctx(E, c);
    return rc;
}
int rife_hip_process_frames(const rife_hip_t* E, const rife_hip_frame_t* f0, const rife_hip_frame_t* f1, float timestep, uint8_t* out) {      // nothing may throw across the C boundary (malformed model files, std::bad_alloc)
    try { return rife_hip_process_frames_impl(E, f0, f1, timestep, out); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EIO, std::string("rife_hip_process_frames: ") + e.what()); }
    catch (...) { return fail(RIFE_HIP_EIO, "rife_hip_process_frames: unknown exception"); }
}

static int rife_hip_process_device_impl(const rife_hip_t* E, const void* d_in0, const void* d_in1, int w, int h, float timestep, void* d_out, void* hip_stream) {
    int rc;
    if ((rc = process_common(E, w, h, timestep))) return rc;
    if (!d_in0 || !d_in1 || !d_out) return fail(RIFE_HIP_EINVAL, "null frame pointer");
    if ((rc = check_device(E->gpuid))) return rc;
    const size_t nbytes = (size_t)w * h * 3;
    Ctx* c;
    {
        std::lock_guard<std::mutex> g(E->mu);
        auto ps = E->part_streams.find(hip_stream);
        if (ps != E->part_streams.end()) tl_cu_budget = ps->second;      // a stream of rife_hip_stream_create: persistent grids for its part of the chip
        auto& slot = E->stream_ctx[hip_stream];
        if (!slot) {
            slot.reset(new Ctx);
            if (hip_stream) slot->stream = (hipStream_t)hip_stream;
            else {
                if (hipStreamCreateWithFlags(&slot->stream, hipStreamNonBlocking) != hipSuccess) return fail(RIFE_HIP_EHIP, "hipStreamCreate failed");
                slot->own_stream = true;
            }
        }
        c = slot.get();
    }
    // two host threads on the same stream (in particular NULL = the engine's own) share one workspace: the second waits here instead of
    // racing on its (re)allocation and scratch tensors - work on one stream executes in order anyway
    std::lock_guard<std::mutex> use(c->use);
    if (timestep == 0.f || timestep == 1.f) {
        HIPCHK(hipMemcpyAsync(d_out, timestep == 0.f ? d_in0 : d_in1, nbytes, hipMemcpyDeviceToDevice, c->stream));
    } else {
        if (E->v1) {
            if ((rc = ensure_ctx_v1(*c, w, h, E->tta ? 8 : 1, E->tta_temporal ? 2 : 1))) return rc;
            if ((rc = run_v1(*E, *c, (const uint8_t*)d_in0, (const uint8_t*)d_in1, (uint8_t*)d_out))) return rc;
        } else if (!E->v4) {
            if ((rc = ensure_ctx_v2(*c, w, h, E->uhd, E->tta ? 8 : 1, E->tta_temporal ? 2 : 1, E->v3))) return rc;
            if ((rc = run_v2(*E, *c, (const uint8_t*)d_in0, (const uint8_t*)d_in1, (uint8_t*)d_out))) return rc;
        } else if (E->tta || E->tta_temporal) {
            // the TTA workspaces are shared: serialise, and drain before another stream may reuse them
            std::lock_guard<std::mutex> g(E->tta_mu);
            if ((rc = run_v4_tta(*E, c->stream, (const uint8_t*)d_in0, (const uint8_t*)d_in1, w, h, timestep, (uint8_t*)d_out))) return rc;
            HIPCHK(hipStreamSynchronize(c->stream));
        } else {
            if ((rc = ensure_ctx(*c, w, h))) return rc;
            if ((rc = run_v4_replay(*E, *c, (const uint8_t*)d_in0, (const uint8_t*)d_in1, timestep, (uint8_t*)d_out))) return rc;
        }
    }
    if (!hip_stream) HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}
int rife_hip_process_device(const rife_hip_t* E, const void* d_in0, const void* d_in1, int w, int h, float timestep, void* d_out, void* hip_stream) {      // nothing may throw across the C boundary (malformed model files, std::bad_alloc)
    try { return rife_hip_process_device_impl(E, d_in0, d_in1, w, h, timestep, d_out, hip_stream); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EIO, std::string("rife_hip_process_device: ") + e.what()); }
    catch (...) { return fail(RIFE_HIP_EIO, "rife_hip_process_device: unknown exception"); }
}

// n resident pairs in one call (include/rife_hip.h): lockstep groups of two pairs (run_v4_group: the coarse-block trunks of a group are one
// launch per layer) on leased workspaces and their streams, forked from and joined into `hip_stream` with events - no host wait.
static int rife_hip_process_device_batch_impl(const rife_hip_t* E, int n, const void* const* d_in0, const void* const* d_in1, const float* timestep,
                                              void* const* d_out, int w, int h, void* hip_stream) {
    int rc;
    if (n < 0 || (n > 0 && (!d_in0 || !d_in1 || !timestep || !d_out))) return fail(RIFE_HIP_EINVAL, "bad batch arguments");
    if ((rc = process_common(E, w, h, 0.5f))) return rc;
    for (int i = 0; i < n; i++) if (!d_in0[i] || !d_in1[i] || !d_out[i]) return fail(RIFE_HIP_EINVAL, "null frame pointer");
    if (n == 0) return 0;
    if ((rc = check_device(E->gpuid))) return rc;
    const bool groups = E->v4 && !E->v40 && !E->v1 && !E->tta && !E->tta_temporal && E->t64;
    if (!groups) {      // other families / TTA: the pairs one after the other on the caller's stream
        for (int i = 0; i < n; i++)
            if ((rc = rife_hip_process_device_impl(E, d_in0[i], d_in1[i], w, h, timestep[i], d_out[i], hip_stream))) return rc;
        return 0;
    }
    hipStream_t user = (hipStream_t)hip_stream;
    const size_t nbytes = (size_t)w * h * 3;
    // the fork event goes back to the pool on EVERY exit path (re-recorded by its next user; waits already enqueued keep their own snapshot of it)
    struct ForkLease {
        const rife_hip_t* E; hipEvent_t ev = nullptr;
        ~ForkLease() { if (ev) { std::lock_guard<std::mutex> g(E->mu); E->batch_fork.push_back(ev); } }
    } fk{E};
    {
        std::lock_guard<std::mutex> g(E->mu);
        if (!E->batch_fork.empty()) { fk.ev = E->batch_fork.back(); E->batch_fork.pop_back(); }
    }
    if (!fk.ev) HIPCHK(hipEventCreateWithFlags(&fk.ev, hipEventDisableTiming));
    hipEvent_t fork = fk.ev;
    if (user) HIPCHK(hipEventRecord(fork, user));      // NULL = "the engine's own streams": nothing to order against, the call synchronises before it returns
    // At most MAXG groups (2 MAXG workspaces) are in flight however many pairs the call carries: further groups re-use them round-robin - work on a
    // workspace's stream executes in order, so a re-used workspace simply queues behind its previous pair (device memory stays O(1) in n).
    constexpr int MAXG = 4;
    std::vector<std::unique_ptr<Ctx>> cs;
    std::unique_ptr<Ctx> copy_ctx;                       // timestep 0 / 1 with no caller stream: one internal stream for the D2D copies
    size_t next_slot = 0;
    auto lease_new = [&](std::unique_ptr<Ctx>& c) -> bool {
        if (lease_ctx(E, c, w, h)) return false;
        if (!c->ev_group && hipEventCreateWithFlags(&c->ev_group, hipEventDisableTiming) != hipSuccess) { release_ctx(E, c); return false; }
        if (user && hipStreamWaitEvent(c->stream, fork, 0) != hipSuccess) { release_ctx(E, c); return false; }
        return true;
    };
    auto lease = [&]() -> Ctx* {
        if (cs.size() < (size_t)(2 * MAXG)) {
            std::unique_ptr<Ctx> c;
            if (!lease_new(c)) return nullptr;
            cs.push_back(std::move(c));
            return cs.back().get();
        }
        Ctx* c = cs[next_slot++ % cs.size()].get();     // always taken in pairs from an even-sized pool: the two of a group are distinct
        tl_cu_budget = c->cu_budget;
        return c;
    };
    rc = 0;
    int pend = -1;
    for (int i = 0; i <= n && !rc; i++) {
        const bool copy = i < n && (timestep[i] == 0.f || timestep[i] == 1.f);
        if (i < n && copy) {             // rife.cpp:2470-2480: an input frame unchanged - a D2D copy, no workspace; on the caller's stream when there is one
            hipStream_t cst = user;
            if (!cst) {
                if (!copy_ctx && !lease_new(copy_ctx)) { rc = fail(RIFE_HIP_EHIP, "rife_hip_process_device_batch: no workspace (" + g_err + ")"); break; }
                cst = copy_ctx->stream;
            }
            if (hipMemcpyAsync(d_out[i], timestep[i] == 0.f ? d_in0[i] : d_in1[i], nbytes, hipMemcpyDeviceToDevice, cst) != hipSuccess) rc = fail(RIFE_HIP_EHIP, "copy failed");
            continue;
        }
        if (i < n && pend < 0) { pend = i; continue; }
        if (pend < 0) break;
        Ctx* a = lease(); Ctx* b = a ? lease() : nullptr;      // the odd pair left over takes (and leaves idle) the second workspace of its slot pair
        if (!a || !b) { rc = fail(RIFE_HIP_EHIP, "rife_hip_process_device_batch: no workspace (" + g_err + ")"); break; }
        if (i < n) {            // group (pend, i)
            Ctx* g2[2] = {a, b};
            const uint8_t* p0[2] = {(const uint8_t*)d_in0[pend], (const uint8_t*)d_in0[i]};
            const uint8_t* p1[2] = {(const uint8_t*)d_in1[pend], (const uint8_t*)d_in1[i]};
            const float ts[2] = {timestep[pend], timestep[i]};
            uint8_t* po[2] = {(uint8_t*)d_out[pend], (uint8_t*)d_out[i]};
            rc = run_v4_group(*E, g2, 2, p0, p1, ts, po);
        } else {
            tl_cu_budget = a->cu_budget;
            rc = run_v4_replay(*E, *a, (const uint8_t*)d_in0[pend], (const uint8_t*)d_in1[pend], timestep[pend], (uint8_t*)d_out[pend]);
        }
        pend = -1;
    }
    // join: the caller's stream continues after every internal stream (also after an error: nothing may still run on the frames when we return control of them)
    if (copy_ctx) cs.push_back(std::move(copy_ctx));
    for (auto& c : cs) {
        if (!user) { if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(RIFE_HIP_EHIP, "stream sync failed"); }
        else if (hipEventRecord(c->ev_group, c->stream) != hipSuccess || hipStreamWaitEvent(user, c->ev_group, 0) != hipSuccess) { (void)hipStreamSynchronize(c->stream); if (!rc) rc = fail(RIFE_HIP_EHIP, "join failed"); }
    }
    for (auto& c : cs) release_ctx(E, c);
    return rc;
}
int rife_hip_process_device_batch(const rife_hip_t* E, int n, const void* const* d_in0, const void* const* d_in1, const float* timestep,
                                  void* const* d_out, int w, int h, void* hip_stream) {
    try { return rife_hip_process_device_batch_impl(E, n, d_in0, d_in1, timestep, d_out, w, h, hip_stream); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EIO, std::string("rife_hip_process_device_batch: ") + e.what()); }
    catch (...) { return fail(RIFE_HIP_EIO, "rife_hip_process_device_batch: unknown exception"); }
}

// ---- streams that own a part of the chip (include/rife_hip.h) ----
int rife_hip_stream_create(const rife_hip_t* E, int part, int nparts, void** hip_stream) {
    if (hip_stream) *hip_stream = nullptr;
    if (!E || !hip_stream) return fail(RIFE_HIP_EINVAL, "null argument");
    int rc;
    if ((rc = check_device(E->gpuid))) return rc;
    const int ncu = device_cus(true);
    if (nparts < 1 || nparts > ncu || part < 0 || part >= nparts) return fail(RIFE_HIP_EINVAL, "bad partition");
    hipStream_t st = nullptr;
    int mine = 0;
    if (nparts == 1) {
        HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        mine = ncu;
    } else {
        std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
        for (int cu = 0; cu < ncu; cu++)
            if (cu % nparts == part) { mask[cu / 32] |= 1u << (cu % 32); mine++; }
        HIPCHK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    }
    std::lock_guard<std::mutex> g(E->mu);
    E->part_streams[(void*)st] = mine;
    *hip_stream = (void*)st;
    return 0;
}
int rife_hip_stream_destroy(const rife_hip_t* E, void* hip_stream) {
    if (!E || !hip_stream) return fail(RIFE_HIP_EINVAL, "null argument");
    int rc;
    if ((rc = check_device(E->gpuid))) return rc;
    {
        std::lock_guard<std::mutex> g(E->mu);
        auto it = E->part_streams.find(hip_stream);
        if (it == E->part_streams.end()) return fail(RIFE_HIP_EINVAL, "not a stream of rife_hip_stream_create");
        E->part_streams.erase(it);
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)hip_stream));
    {
        std::lock_guard<std::mutex> g(E->mu);
        E->stream_ctx.erase(hip_stream);                                 // its workspace
    }
    HIPCHK(hipStreamDestroy((hipStream_t)hip_stream));
    return 0;
}

// ---- page-locked host frames (include/rife_hip.h) ----
void* rife_hip_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { g_err = "hipHostMalloc failed"; return nullptr; }
    return p;
}
void rife_hip_host_free(void* p) { if (p) (void)hipHostFree(p); }
int rife_hip_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) return fail(RIFE_HIP_EINVAL, "null range");
    HIPCHK(hipHostRegister(p, bytes, hipHostRegisterPortable));
    return 0;
}
int rife_hip_host_unregister(void* p) {
    if (!p) return fail(RIFE_HIP_EINVAL, "null pointer");
    HIPCHK(hipHostUnregister(p));
    return 0;
}

int rife_hip_profile_enable(rife_hip_t* E, int on) {
    if (!E) return fail(RIFE_HIP_EINVAL, "null engine");
    E->prof.collect();
    E->prof.on = on != 0;
    if (on) {
        std::lock_guard<std::mutex> g(E->prof.mu);
        std::fill(E->prof.ms.begin(), E->prof.ms.end(), 0.0);
        std::fill(E->prof.flops.begin(), E->prof.flops.end(), 0.0);
        std::fill(E->prof.launches.begin(), E->prof.launches.end(), 0LL);
    }
    return 0;
}

int rife_hip_profile_read(rife_hip_t* E, char* names, size_t names_cap, double* total_ms, long long* launches, double* flops, int max_classes) {
    if (!E) return fail(RIFE_HIP_EINVAL, "null engine");
    E->prof.collect();
    std::lock_guard<std::mutex> g(E->prof.mu);
    std::string all;
    int n = std::min<int>(max_classes, (int)E->prof.names.size());
    for (int i = 0; i < n; i++) {
        all += E->prof.names[i]; all += '\n';
        total_ms[i] = E->prof.ms[i]; launches[i] = E->prof.launches[i]; flops[i] = E->prof.flops[i];
    }
    if (names && names_cap) { std::strncpy(names, all.c_str(), names_cap - 1); names[names_cap - 1] = 0; }
    return n;
}

#ifdef RIFE_HIP_TEST_BUILD      // ======== include/rife_hip_test.h: test and bench builds only ========
// ---- stage tap: flow{fi} with optional injection of flow0..flow{n_inject-1} (rife.cpp:2653-2669) -------------
int rife_hip_v4_extract_flow(const rife_hip_t* E, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep, int fi,
                             const float* const* inject, int n_inject, float* out6chw) {
    int rc;
    if ((rc = process_common(E, w, h, timestep))) return rc;
    if (!E->v4) return fail(RIFE_HIP_EINVAL, "stage taps exist for the rife-v4 family only");
    if (fi < 0 || fi > 3 || n_inject < 0 || n_inject > fi) return fail(RIFE_HIP_EINVAL, "bad stage index");
    if ((rc = check_device(E->gpuid))) return rc;
    Ctx c;
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) return fail(RIFE_HIP_EHIP, "hipStreamCreate failed");
    c.own_stream = true;
    if ((rc = ensure_ctx(c, w, h))) return rc;
    const size_t nbytes = (size_t)w * h * 3;
    HIPCHK(hipMemcpyAsync(c.d_in0, in0, nbytes, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(c.d_in1, in1, nbytes, hipMemcpyHostToDevice, c.stream));
    launch_preproc(c.stream, c.d_in0, c.w, c.h, c.img0, c.wp, c.hp);
    launch_preproc(c.stream, c.d_in1, c.w, c.h, c.img1, c.wp, c.hp);
    float* tmp = nullptr;
    if ((rc = dalloc(c, tmp, (size_t)c.wp * c.hp * 6))) return rc;
    const int nc = E->v40 ? 5 : 6;      // channels of blob flow{b}: rife-v4.6 PixelShuffle output 6, rife-v4 deconv output 5
    for (int b = 0; b <= fi; b++) {
        const int s = E->flow_div(b), Hb = c.hp / s, Wb = c.wp / s;
        if (b < n_inject) {
            HIPCHK(hipMemcpyAsync(tmp, inject[b], (size_t)Hb * Wb * nc * 4, hipMemcpyHostToDevice, c.stream));
            hipLaunchKernelGGL(k_chw_to_nhwc, grid2d(Wb, Hb), dim3(256), 0, c.stream, tmp, c.flow[b], nc, Hb, Wb, 8);
        } else {
            if ((rc = run_block_convs(*E, c, b, timestep))) return rc;
        }
        if (b < fi && (rc = run_flow_update(*E, c, b))) return rc;
    }
    const int s = E->flow_div(fi), Hb = c.hp / s, Wb = c.wp / s;
    hipLaunchKernelGGL(k_nhwc_to_chw, grid2d(Wb, Hb), dim3(256), 0, c.stream, c.flow[fi], tmp, nc, Hb, Wb, 8);
    HIPCHK(hipMemcpyAsync(out6chw, tmp, (size_t)Hb * Wb * nc * 4, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return 0;
}

// ---- parity taps of the gather code (round 3): the 12-channel block input and the tail of the graph, on injected flows --------------------
// Shared prologue: frames -> padded RGBX, then for every block k < n_inject the injected blob flow{k} goes through the hot path's own
// k_flow_update into F, M (flownet.param:47-58, 99-105, 152-158).
// `pending` != null: as in run_v4, the update of the LAST injected flow is left to the fused stem of the next block where the product does so
// (flow_update_fused_into); *pending is then that flow.
static int tap_prologue(const rife_hip_t* E, Ctx& c, const uint8_t* in0, const uint8_t* in1, int w, int h, const float* const* inject, int n_inject, float*& tmp,
                        const float** pending = nullptr) {
    int rc;
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) return fail(RIFE_HIP_EHIP, "hipStreamCreate failed");
    c.own_stream = true;
    if ((rc = ensure_ctx(c, w, h))) return rc;
    const size_t nbytes = (size_t)w * h * 3;
    HIPCHK(hipMemcpyAsync(c.d_in0, in0, nbytes, hipMemcpyHostToDevice, c.stream));
    HIPCHK(hipMemcpyAsync(c.d_in1, in1, nbytes, hipMemcpyHostToDevice, c.stream));
    launch_preproc(c.stream, c.d_in0, c.w, c.h, c.img0, c.wp, c.hp);
    launch_preproc(c.stream, c.d_in1, c.w, c.h, c.img1, c.wp, c.hp);
    if ((rc = dalloc(c, tmp, (size_t)c.wp * c.hp * 16))) return rc;
    for (int k = 0; k < n_inject; k++) {
        const int s = E->blk[k].scale, Hb = c.hp / s, Wb = c.wp / s;
        HIPCHK(hipMemcpyAsync(tmp, inject[k], (size_t)Hb * Wb * 6 * 4, hipMemcpyHostToDevice, c.stream));
        hipLaunchKernelGGL(k_chw_to_nhwc, grid2d(Wb, Hb), dim3(256), 0, c.stream, tmp, c.flow[k], 6, Hb, Wb, 8);
        if (pending && k == n_inject - 1 && k < 3 && flow_update_fused_into(*E, c, k + 1)) { *pending = c.flow[k]; continue; }
        if (k < 3 && (rc = run_flow_update(*E, c, k))) return rc;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// what = 0: block input of IFBlock b (1..3; blobs 99 / 199 / 262 of models/rife-v4.6/flownet.param:62, 115, 165) as k_assemble<S> computes it
//           (the unfused form of the same assemble_pixel<S> / warp_rgbx code);
// what = 1: the same tensor read back THROUGH THE PRODUCT'S FUSED STEM KERNEL stem0_fused_kernel<S, ...> (stem_fused.h), which keeps it in
//           LDS only: the kernel is run with one-hot weights (output channel 12 p + k = input channel k under tap (1 + p / 2, 1 + p % 2), bias 0,
//           slope 1), so that its stride-2 output holds the block input's four pixel parities; the split-f16 matrix path returns hi + lo of
//           every value, i.e. the value to 2^-22 relative;
// what = 2: blob out0 (flownet.param:217) before the postproc, from the unfused float tail k_final_float (b ignored; n_inject = 4);
// what = 4 / 3: F, M as block b's stem reads them: after k_flow_update / written by the stem that applies the update of flow{b-1} itself.
// what = 5: block 3's input through stem_rs_kernel, the product's kernel for that block (see below).
// out: planar CHW fp32, 12 x hp/S x wp/S (what 0, 1) or 3 x hp x wp (what 2).  n_inject must be b (what 0, 1) or 4 (what 2).
static int rife_hip_v4_tap_impl(const rife_hip_t* E, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep, int what, int b,
                                const float* const* inject, int n_inject, float* out) {
    int rc;
    if ((rc = process_common(E, w, h, timestep))) return rc;
    if (!E->v4 || E->v40) return fail(RIFE_HIP_EINVAL, "the gather taps exist for the rife-v4.6 graph only");
    if (what < 0 || what > 5) return fail(RIFE_HIP_EINVAL, "bad tap");
    if (what == 5 && b != 3) return fail(RIFE_HIP_EINVAL, "the row-streaming stem kernel serves block 3");
    if (what == 2 ? n_inject != 4 : (b < 1 || b > 3 || n_inject != b)) return fail(RIFE_HIP_EINVAL, "bad block / injection count");
    if ((rc = check_device(E->gpuid))) return rc;
    Ctx c; float* tmp = nullptr;
    const float* pending = nullptr;
    if ((rc = tap_prologue(E, c, in0, in1, w, h, inject, n_inject, tmp, (what == 1 || what == 3) ? &pending : nullptr))) return rc;
    hipStream_t st = c.stream;
    // what = 3 / 4: F (4 channels) and M as block b's stem finds them, [5][hp][wp]: 4 = after k_flow_update, 3 = as written by the stem that applies
    // the last update itself (only where the product fuses it: EINVAL otherwise)
    auto copy_fm = [&](const float4* F, const float* M) -> int {
        hipLaunchKernelGGL(k_nhwc_to_chw, grid2d(c.wp, c.hp), dim3(256), 0, st, reinterpret_cast<const float*>(F), tmp, 4, c.hp, c.wp, 4);
        HIPCHK(hipGetLastError());
        const size_t P = (size_t)c.wp * c.hp;
        HIPCHK(hipMemcpyAsync(out, tmp, P * 16, hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(out + 4 * P, M, P * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    };
    if (what == 4) return copy_fm(c.F, c.M);
    if (what == 5) {
        // Block 3's input THROUGH THE PRODUCT'S ROW-STREAMING STEM KERNEL stem_rs_kernel (stem_rs.h): both of its convolutions run with one-hot
        // weights.  Stem 0: output channel 12 j + k = input channel k under tap (1 + g, 1 + j) (pixel parity p = 2 g + j of the block input; two
        // parities per launch); stem 1: output channel = input channel under tap (ty, tx) in {1, 2}^2 (the four parities of the half-resolution
        // tensor).  Eight launches return every pixel of the 12-channel block input once; each value passed the split-f16 matrix path twice
        // (hi + lo of hi + lo: 2^-21 relative).  Bias 0, slope 1.
        const int Hq = c.hp / 4, Wq = c.wp / 4;
        const S16Geom G(Hq, Wq);
        const size_t nb = G.bytes(64), pl = G.plane();
        unsigned char* dout = nullptr; uint16_t *dw0 = nullptr, *dw1 = nullptr; float *dbias = nullptr, *dslope = nullptr;
        if ((rc = dalloc(c, dout, nb)) || (rc = dalloc(c, dw0, (size_t)9 * 2 * 32 * 8)) || (rc = dalloc(c, dw1, (size_t)2 * 9 * 2 * 64 * 8)) ||
            (rc = dalloc(c, dbias, 64)) || (rc = dalloc(c, dslope, 64))) return rc;
        std::vector<float> hz(64, 0.f), ho(64, 1.f);
        HIPCHK(hipMemcpyAsync(dbias, hz.data(), 256, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dslope, ho.data(), 256, hipMemcpyHostToDevice, st));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_rs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, SRS_LDS));
        std::vector<unsigned char> host(nb);
        int inv[32];                                                     // row of a 32-row block that holds channel ch (pack_weights_h2_perm)
        for (int i = 0; i < 32; i++) inv[s16_row_channel(i)] = i;
        for (int g = 0; g < 2; g++)
            for (int ty = 1; ty <= 2; ty++)
                for (int tx = 1; tx <= 2; tx++) {
                    std::vector<uint16_t> h0((size_t)9 * 2 * 32 * 8, 0), h1((size_t)2 * 9 * 2 * 64 * 8, 0);
                    for (int j = 0; j < 2; j++)
                        for (int k = 0; k < 12; k++) h0[(((size_t)((1 + g) * 3 + 1 + j) * 2 + k / 8) * 32 + 12 * j + k) * 8 + k % 8] = f2h(1.f);
                    for (int oc = 0; oc < 24; oc++)
                        h1[((((size_t)(oc / 16) * 9 + ty * 3 + tx) * 2 + (oc % 16) / 8) * 64 + inv[oc]) * 8 + oc % 8] = f2h(1.f);
                    HIPCHK(hipMemcpyAsync(dw0, h0.data(), h0.size() * 2, hipMemcpyHostToDevice, st));
                    HIPCHK(hipMemcpyAsync(dw1, h1.data(), h1.size() * 2, hipMemcpyHostToDevice, st));
                    HIPCHK(hipMemsetAsync(dout, 0, nb, st));
                    StemRsArgs a;
                    a.img0 = c.img0; a.img1 = c.img1; a.F = c.F; a.M = c.M; a.w0 = dw0; a.bias0 = dbias; a.slope0 = dslope; a.w1 = dw1; a.bias1 = dbias; a.slope1 = dslope;
                    a.out = dout; a.timestep = timestep; a.tsp = nullptr; a.wp = c.wp; a.hp = c.hp; a.Hq = Hq; a.Wq = Wq; a.pitch = G.pitch; a.plane = G.plane();
                    a.nunits = ((Wq + SRS_SW - 1) / SRS_SW) * Hq;
                    const int nwg = std::min(2 * device_cus(), a.nunits);
                    hipLaunchKernelGGL((stem_rs_kernel<0>), dim3(nwg), dim3(SRS_NTHR), SRS_LDS, st, a);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipMemcpyAsync(host.data(), dout, nb, hipMemcpyDeviceToHost, st));
                    HIPCHK(hipStreamSynchronize(st));
                    for (int j = 0; j < 2; j++)
                        for (int k = 0; k < 12; k++) {
                            const int oc = 12 * j + k;
                            const _Float16* hi = reinterpret_cast<const _Float16*>(host.data() + (size_t)(2 * (oc / 16)) * pl);
                            const _Float16* lo = reinterpret_cast<const _Float16*>(host.data() + (size_t)(2 * (oc / 16) + 1) * pl);
                            for (int q = 0; q < Hq; q++)
                                for (int x = 0; x < Wq; x++) {
                                    const size_t e = ((size_t)(q + 1) * G.pitch + x + 1) * 16 + oc % 16;
                                    out[((size_t)k * c.hp + 4 * q + 2 * (ty - 1) + g) * c.wp + 4 * x + 2 * (tx - 1) + j] = (float)hi[e] + (float)lo[e];
                                }
                        }
                }
        return 0;
    }
    if (what == 3 && !pending) return fail(RIFE_HIP_EINVAL, "the flow update before this block is not fused into its stem");
    if (what == 2) {
        float4* outf = nullptr;
        if ((rc = dalloc(c, outf, (size_t)c.wp * c.hp))) return rc;
        hipLaunchKernelGGL(k_final_float, grid2d(c.wp, c.hp), dim3(256), 0, st, c.img0, c.img1, c.F, c.M, c.flow[3], outf, c.wp, c.hp);
        hipLaunchKernelGGL(k_nhwc_to_chw, grid2d(c.wp, c.hp), dim3(256), 0, st, reinterpret_cast<const float*>(outf), tmp, 3, c.hp, c.wp, 4);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, tmp, (size_t)c.wp * c.hp * 3 * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
    const rife_hip::Block& B = E->blk[b];
    const int s = B.scale, Hb = c.hp / s, Wb = c.wp / s;
    if (what == 0) {
        if ((rc = run_assemble(*E, c, b, timestep))) return rc;
        hipLaunchKernelGGL(k_nhwc_to_chw, grid2d(Wb, Hb), dim3(256), 0, st, c.X, tmp, 12, Hb, Wb, 16);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out, tmp, (size_t)Hb * Wb * 12 * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        return 0;
    }
    // what == 1: the fused stem kernel of the product with one-hot weights
    const int NT = s == 1 ? 32 : 64, cout = B.c / 2, per = std::min(4, cout / 12), nlaunch = (4 + per - 1) / per;
    const int Ho = Hb / 2, Wo = Wb / 2;
    std::vector<float> hbias(64, 0.f), hslope(64, 1.f), host((size_t)Ho * Wo * cout);
    float *dbias = nullptr, *dslope = nullptr; uint16_t* dw = nullptr;
    if ((rc = dalloc(c, dbias, 64)) || (rc = dalloc(c, dslope, 64)) || (rc = dalloc(c, dw, (size_t)9 * 2 * NT * 8))) return rc;
    HIPCHK(hipMemcpyAsync(dbias, hbias.data(), 256, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(dslope, hslope.data(), 256, hipMemcpyHostToDevice, st));
    for (int l = 0; l < nlaunch; l++) {
        std::vector<uint16_t> hw((size_t)9 * 2 * NT * 8, 0);
        for (int q = 0; q < per && l * per + q < 4; q++) {
            const int p = l * per + q, t = (1 + p / 2) * 3 + 1 + p % 2;
            for (int k = 0; k < 12; k++) hw[(((size_t)t * 2 + k / 8) * NT + 12 * q + k) * 8 + k % 8] = f2h(1.f);
        }
        HIPCHK(hipMemcpyAsync(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice, st));
        StemFusedArgs fa;
        fa.img0 = c.img0; fa.img1 = c.img1; fa.F = c.F; fa.M = c.M; fa.wpk = dw; fa.bias = dbias; fa.slope = dslope;
        fa.out = c.S1; fa.timestep = timestep; fa.tsp = nullptr; fa.wp = c.wp; fa.hp = c.hp; fa.Ho = Ho; fa.Wo = Wo; fa.out_ld = cout; fa.Cout = cout;
        fa.tiles_x = (Wo + 31) / 32;
        const int nb = fa.tiles_x * ((Ho + 3) / 4);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<2, 2, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
        if (pending) {                                                   // every launch reads the old F, M and writes the same new ones
            fa.pend.flow = pending; fa.pend.Fw = c.F2; fa.pend.Mw = c.M2;
            if (s == 2) hipLaunchKernelGGL((stem0_fused_kernel<2, 2, 0, true>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
            else hipLaunchKernelGGL((stem0_fused_kernel<1, 1, 256, true>), dim3(nb), dim3(512), (stemf_lds_bytes<1, 256>()), st, fa);
            if (what == 3) { HIPCHK(hipGetLastError()); return copy_fm(c.F2, c.M2); }
        } else if (s == 4) hipLaunchKernelGGL((stem0_fused_kernel<4, 2>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
        else if (s == 2) hipLaunchKernelGGL((stem0_fused_kernel<2, 2>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
        else hipLaunchKernelGGL((stem0_fused_kernel<1, 1, 256>), dim3(nb), dim3(512), (stemf_lds_bytes<1, 256>()), st, fa);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(host.data(), c.S1, host.size() * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int q = 0; q < per && l * per + q < 4; q++) {
            const int p = l * per + q, py = p / 2, px = p % 2;
            for (int k = 0; k < 12; k++)
                for (int y = 0; y < Ho; y++)
                    for (int x = 0; x < Wo; x++)
                        out[((size_t)k * Hb + 2 * y + py) * Wb + 2 * x + px] = host[((size_t)y * Wo + x) * cout + 12 * q + k];
        }
    }
    return 0;
}
int rife_hip_v4_tap(const rife_hip_t* E, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep, int what, int b,
                    const float* const* inject, int n_inject, float* out) {      // nothing may throw across the C boundary
    if (!in0 || !in1 || !out) return fail(RIFE_HIP_EINVAL, "null frame / output pointer");
    if (n_inject > 0 && !inject) return fail(RIFE_HIP_EINVAL, "n_inject > 0 without blobs");
    for (int k = 0; k < n_inject && k < 4; k++) if (!inject[k]) return fail(RIFE_HIP_EINVAL, "null injected blob");
    try { return rife_hip_v4_tap_impl(E, in0, in1, w, h, timestep, what, b, inject, n_inject, out); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EINVAL, std::string("v4_tap: ") + e.what()); }
}

// The plain v4 pass with the first n_inject (0..3) blobs flow{k} injected instead of computed: the remaining blocks and the tail run on the
// product's own schedule (fused stems, fused tail of head_h2_kernel<EPI_FINAL>), so that flows which leave the frame by hundreds of pixels
// reach exactly the gather code a real pass runs.  out: w x h u8 RGB.
static int rife_hip_v4_process_injected_impl(const rife_hip_t* E, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep,
                                             const float* const* inject, int n_inject, uint8_t* out) {
    int rc;
    if ((rc = process_common(E, w, h, timestep))) return rc;
    if (!E->v4 || E->v40) return fail(RIFE_HIP_EINVAL, "flow injection into the plain pass exists for the rife-v4.6 graph only");
    if (n_inject < 0 || n_inject > 3) return fail(RIFE_HIP_EINVAL, "bad injection count");
    if ((rc = check_device(E->gpuid))) return rc;
    Ctx c; float* tmp = nullptr;
    const float* pending = nullptr;
    if ((rc = tap_prologue(E, c, in0, in1, w, h, inject, n_inject, tmp, &pending))) return rc;
    const bool fuse_tail = g_trunk_h2 && g_head_h2 && g_fuse_tail && E->blk[3].head.d_wh != nullptr;
    FinalArgs fin{c.img0, c.img1, c.F, c.M, c.d_out, c.w, c.h, c.wp, c.hp};
    for (int b = n_inject; b < 4; b++) {
        if ((rc = run_block_convs(*E, c, b, timestep, (b == 3 && fuse_tail) ? &fin : nullptr, nullptr, PH_ALL, pending))) return rc;
        pending = nullptr;
        if (b < 3 && flow_update_fused_into(*E, c, b + 1)) pending = c.flow[b];
        else if (b < 3 && (rc = run_flow_update(*E, c, b))) return rc;
    }
    if (!fuse_tail) hipLaunchKernelGGL(k_final, grid2d(c.w, c.h), dim3(256), 0, c.stream, c.img0, c.img1, c.F, c.M, c.flow[3], c.d_out, c.w, c.h, c.wp, c.hp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c.d_out, (size_t)w * h * 3, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    return 0;
}
int rife_hip_v4_process_injected(const rife_hip_t* E, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep,
                                 const float* const* inject, int n_inject, uint8_t* out) {      // nothing may throw across the C boundary
    if (!in0 || !in1 || !out) return fail(RIFE_HIP_EINVAL, "null frame / output pointer");
    if (n_inject > 0 && !inject) return fail(RIFE_HIP_EINVAL, "n_inject > 0 without blobs");
    for (int k = 0; k < n_inject && k < 4; k++) if (!inject[k]) return fail(RIFE_HIP_EINVAL, "null injected blob");
    try { return rife_hip_v4_process_injected_impl(E, in0, in1, w, h, timestep, inject, n_inject, out); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EINVAL, std::string("v4_process_injected: ") + e.what()); }
}

#endif  // RIFE_HIP_TEST_BUILD

static int rife_hip_graph_check_impl(const char* base) {
    if (!base) return fail(RIFE_HIP_EINVAL, "null argument");
    GraphNet n;
    return graph_load(n, base, true);
}
int rife_hip_graph_check(const char* base) {      // nothing may throw across the C boundary (malformed model files, std::bad_alloc)
    try { return rife_hip_graph_check_impl(base); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EIO, std::string("rife_hip_graph_check: ") + e.what()); }
    catch (...) { return fail(RIFE_HIP_EIO, "rife_hip_graph_check: unknown exception"); }
}

#ifdef RIFE_HIP_TEST_BUILD      // ======== include/rife_hip_test.h (continued) ========
int rife_hip_v4_flow_dims(const rife_hip_t* E, int w, int h, int fi, int* channels, int* fh, int* fw) {
    if (!E || !E->loaded || !E->v4) return fail(RIFE_HIP_EINVAL, "flow blobs exist for a loaded rife-v4 family engine only");
    if (fi < 0 || fi > 3 || w <= 0 || h <= 0 || !channels || !fh || !fw) return fail(RIFE_HIP_EINVAL, "bad argument");
    const int wp = (w + 31) / 32 * 32, hp = (h + 31) / 32 * 32;
    *channels = E->v40 ? 5 : 6; *fh = hp / E->flow_div(fi); *fw = wp / E->flow_div(fi);
    return 0;
}

// ---- single-kernel entry points ------------------------------------------------------------------------------
static int op_conv_common(int gpuid, const float* x, int c, int h, int w, const float* weight, const float* bias, int outc, int stride,
                          bool deconv, int epi, const float* residual, const float* slope, float* out) {
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    ConvLayer L;
    L.cin = c; L.cout = outc; L.stride = deconv ? 1 : stride; L.deconv = deconv; L.epi = epi;
    if ((rc = upload_layer(L, weight, bias, slope, 1.0f))) { free_layer(L); return rc; }
    const int ho = deconv ? 2 * h : (h + 2 - 3) / stride + 1, wo = deconv ? 2 * w : (w + 2 - 3) / stride + 1;
    const int ldi = L.cin_p;
    float *d_chw = nullptr, *d_x = nullptr, *d_y = nullptr, *d_r = nullptr, *d_o = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_chw); (void)hipFree(d_x); (void)hipFree(d_y); (void)hipFree(d_r); (void)hipFree(d_o); free_layer(L); };
    const size_t nin = (size_t)c * h * w, nout = (size_t)outc * ho * wo;
    hipError_t e = hipMalloc(&d_chw, std::max(nin, nout) * 4);
    if (e == hipSuccess) e = hipMalloc(&d_x, (size_t)h * w * ldi * 4);
    if (e == hipSuccess) e = hipMalloc(&d_y, (size_t)ho * wo * outc * 4);
    if (e == hipSuccess) e = hipMalloc(&d_o, nout * 4);
    if (e == hipSuccess && residual) e = hipMalloc(&d_r, (size_t)ho * wo * outc * 4);
    if (e != hipSuccess) { cleanup(); return fail(RIFE_HIP_EHIP, "hipMalloc failed"); }
    (void)hipMemcpy(d_chw, x, nin * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_chw_to_nhwc, grid2d(w, h), dim3(256), 0, 0, d_chw, d_x, c, h, w, ldi);
    TensorView rv{d_r, outc, 0};
    if (residual) {
        (void)hipMemcpy(d_chw, residual, nout * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_chw_to_nhwc, grid2d(wo, ho), dim3(256), 0, 0, d_chw, d_r, outc, ho, wo, outc);
    }
    rc = launch_conv(L, {d_x, ldi, 0}, h, w, {d_y, outc, 0}, residual ? &rv : nullptr, 0);
    if (!rc) {
        hipLaunchKernelGGL(k_nhwc_to_chw, grid2d(wo, ho), dim3(256), 0, 0, d_y, d_o, outc, ho, wo, outc);
        e = hipMemcpy(out, d_o, nout * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RIFE_HIP_EHIP, std::string("op: ") + hipGetErrorString(e));
    }
    cleanup();
    return rc;
}

int rife_hip_op_conv3x3(int gpuid, const float* x, int c, int h, int w, const float* weight, const float* bias, int outc, int stride,
                        const float* residual, const float* slope, float* out) {
    if (stride != 1 && stride != 2) return fail(RIFE_HIP_EINVAL, "stride must be 1 or 2");
    return op_conv_common(gpuid, x, c, h, w, weight, bias, outc, stride, false, EPI_STORE, residual, slope, out);
}

int rife_hip_op_deconv4x4(int gpuid, const float* x, int c, int h, int w, const float* weight, const float* bias, int outc, const float* slope, float* out) {
    return op_conv_common(gpuid, x, c, h, w, weight, bias, outc, 2, true, EPI_DECONV, nullptr, slope, out);
}

int rife_hip_op_warp(int gpuid, const float* image, const float* flow, int c, int h, int w, float* out) {
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    float *d_i = nullptr, *d_f = nullptr, *d_o = nullptr;
    const size_t n = (size_t)c * h * w;
    hipError_t e = hipMalloc(&d_i, n * 4);
    if (e == hipSuccess) e = hipMalloc(&d_f, (size_t)2 * h * w * 4);
    if (e == hipSuccess) e = hipMalloc(&d_o, n * 4);
    if (e == hipSuccess) e = hipMemcpy(d_i, image, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_f, flow, (size_t)2 * h * w * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_warp_chw, grid2d(w, h), dim3(256), 0, 0, d_i, d_f, d_o, c, h, w);
        e = hipMemcpy(out, d_o, n * 4, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_i); (void)hipFree(d_f); (void)hipFree(d_o);
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("op_warp: ") + hipGetErrorString(e));
    return 0;
}

#endif  // RIFE_HIP_TEST_BUILD

#ifdef RIFE_HIP_BENCH_BUILD
#include "bench_hooks.h"      // bench-only / probe entry points and ablation instantiations: librife_hip_bench.so (tools/*.py), never the product
#endif

// tooling: structural hash of a named blob of a .param file (used to derive / test the compiled-in constants)
static int rife_hip_param_hash_impl(const char* param_path, const char* blob, uint64_t* out) {
    NcnnModel m;
    if (!m.load_param(param_path)) return fail(RIFE_HIP_EIO, m.error);
    *out = m.structural_hash(blob);
    return *out ? 0 : fail(RIFE_HIP_EMODEL, "no such blob");
}
int rife_hip_param_hash(const char* param_path, const char* blob, uint64_t* out) {      // nothing may throw across the C boundary (malformed model files, std::bad_alloc)
    try { return rife_hip_param_hash_impl(param_path, blob, out); }
    catch (const std::exception& e) { return fail(RIFE_HIP_EIO, std::string("rife_hip_param_hash: ") + e.what()); }
    catch (...) { return fail(RIFE_HIP_EIO, "rife_hip_param_hash: unknown exception"); }
}

}  // extern "C"
