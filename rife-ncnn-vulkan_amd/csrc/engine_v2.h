// engine_v2.h: rife-v2.x / v3.x schedule: RIFE::process (rife.cpp:381-1212), flownet -> contextnet x 2 -> fusionnet, UHD and TTA modes
// One translation unit (engine.hip includes the engine_*.h sections in dependency order; every function here is file-local).
// No include guard on purpose: a section is included exactly once, by engine.hip.

namespace rife {

// ------------------------------------------------------------------------------------------------
// rife-v2.x: RIFE::process, non-TTA branch (rife.cpp:878-1183) = flownet -> slice -> contextnet x2 -> fusionnet
// ------------------------------------------------------------------------------------------------
// Both ContextNet passes per launch (run_v2_synth, gridDim.y = 2)?  Needs the RGBX form of the first convolution and the split-f16 kernels (the only ones with a
// two-tensor form) for EVERY other layer: one layer whose weights are not f16-exact would fail the whole call with EINVAL in launch_conv; the sequential loop
// serves such a model.  RIFE_HIP_CTX0_IMG / RIFE_HIP_V2_CTX_BATCH = 0 (A/B, test build).
static bool ctx_batch_serves(const rife_hip& E) {
    const bool img_env = process_switches().ctx0_img, batch_env = process_switches().v2_ctx_batch;
    bool ok = E.ctxc[0].d_wimg != nullptr && trunk_h2() && img_env && batch_env;
    for (int i = 1; i < 10 && ok; i++) ok = E.ctxc[i].nchunksh > 0;
    return ok;
}
static int ensure_ctx_v2(Ctx& c, int w, int h, bool uhd, int nori = 1, int ntemp = 1, bool v3 = false, bool ctx2 = true) {
    const int wp = (w + 31) / 32 * 32, hp = (h + 31) / 32 * 32;     // rife.cpp:417-418
    const bool ens = nori * ntemp > 1;
    if (c.v2 && c.wp == wp && c.hp == hp && c.w == w && c.h == h && (!uhd || c.h0) && (!ens || (c.toutf[0][0] && c.toutf[ntemp - 1][nori - 1])) && (!v3 || c.T2) && (!ctx2 || c.ca2)) return 0;
    c.h0 = c.h1 = c.acc_s = nullptr; c.T2 = nullptr; c.ca2 = nullptr;
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
    if (ctx2) {      // the second ContextNet pass's activations, only where both passes really share a launch (ctx_batch_serves): + 0.2 GB at 1080p, + 0.8 GB at 4K per workspace
        A_(c.ca2, P / 4 * 32) A_(c.cb2, P / 4 * 32) A_(c.cc2, P / 16 * 32)
        A_(c.feat2[0], P / 16 * 32) A_(c.feat2[1], P / 64 * 64) A_(c.feat2[2], P / 256 * 128) A_(c.feat2[3], P / 1024 * 256)
        A_(c.ctmp2[0], P / 64 * 64) A_(c.ctmp2[1], P / 256 * 128) A_(c.ctmp2[2], P / 1024 * 256)
        A_(c.fl2[0], P / 16) A_(c.fl2[1], P / 64) A_(c.fl2[2], P / 256) A_(c.fl2[3], P / 1024)
    }
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
static inline bool v2_fused_stem() { return process_switches().v2_fused_stem; }
static bool stem2_fusable(const ConvLayer& L, int S, int wp, int hp) {
    return v2_fused_stem() && trunk_h2() && (S == 1 || S == 2) && L.d_wh && L.cin == 10 && L.nchunksh == 1 && L.stride == 2 && !L.deconv && L.cout <= 96 && L.cout % 4 == 0 &&      // cout <= 96: three 32-channel subtiles, two workgroups per CU fit the LDS (a fourth: 169.5 KB)
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
    const bool r64 = process_switches().v2_stem_r64;
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
    const bool img_env = process_switches().ctx0_img;      // A/B (round 5)
    const bool ctx0_img = E.ctxc[0].d_wimg != nullptr && trunk_h2() && img_env;      // RIFE_HIP_TRUNK=f32 keeps the fp32 matrix path
    float* cat_buf[4] = {c.B1, c.B2, c.B3, c.B4};
    const int cat_ld[4] = {128, 256, 512, 1024}, cat_off[4] = {64, 128, 256, 512}, lvl_c[4] = {32, 64, 128, 256};
    // both passes through ONE launch per layer (gridDim.y = 2: same weights, twice the workgroups - the deep levels are grids of 72 - 272 workgroups);
    // RIFE_HIP_V2_CTX_BATCH=0 (A/B, test build): one pass after the other
    const bool ctx_batch = ctx0_img && ctx_batch_serves(E) && c.ca2 != nullptr;
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
    const bool skip_copy_env = process_switches().v2_skip_copy;
    auto dual_ok = [&](const ConvLayer& L, int H, int W) {
        const long nb = (long)((W + 31) / 32) * ((H + 7) / 8) * L.ntiles;
        return !skip_copy_env && trunk_h2() && L.nchunksh > 0 && L.NS <= 2 && !(L.NS == 2 && nb <= 64 && L.nchunksh >= 4);
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

}  // namespace rife
