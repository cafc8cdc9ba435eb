// engine_v1.h: v1 family on the graph executor (run_v1), model loading (load_v4 / load_v2 / load_v1), device checks
// One translation unit (engine.hip includes the engine_*.h sections in dependency order; every function here is file-local).
// No include guard on purpose: a section is included exactly once, by engine.hip.

namespace rife {

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
    const bool fine = read_switches().profile_fine;
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
