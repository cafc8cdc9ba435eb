// engine_layers.h: ConvLayer - host-side description of a convolution layer + packing of its weights for every kernel form
// One translation unit (engine.hip includes the engine_*.h sections in dependency order; every function here is file-local).
// No include guard on purpose: a section is included exactly once, by engine.hip.

namespace rife {

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
    const bool stem16 = process_switches().v2_stem16;
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

}  // namespace rife
