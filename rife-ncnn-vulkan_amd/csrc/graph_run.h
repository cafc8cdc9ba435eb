// graph_run(): executes the layers of a GraphNet that the wanted blobs need (included by engine.hip after rife_hip / Timed).
#pragma once

namespace rife {

static int graph_run(const rife_hip& E, const GraphNet& N, GraphInst& I, hipStream_t st, const std::vector<std::pair<std::string, GView>>& bound,
                     const std::vector<std::string>& wanted) {
    const size_t nb = N.blob_names.size();
    if (I.v.size() != nb) { I.v.assign(nb, GView{nullptr, 0, 0, 0, 0}); I.owned.assign(nb, nullptr); I.cap.assign(nb, 0); }
    std::vector<char> have(nb, 0);
    for (const auto& kv : bound) {
        const int b = N.blob(kv.first);
        if (b < 0) return fail(RIFE_HIP_EMODEL, N.name + ": no blob named " + kv.first);
        I.v[b] = kv.second; have[b] = 1;
    }
    std::vector<char> need(N.layers.size(), 0);
    std::vector<int> stack;
    for (const std::string& w : wanted) {
        const int b = N.blob(w);
        if (b < 0) return fail(RIFE_HIP_EMODEL, N.name + ": no blob named " + w);
        if (!have[b]) stack.push_back(b);
    }
    while (!stack.empty()) {
        const int b = stack.back(); stack.pop_back();
        const int li = N.producer[b];
        if (li < 0 || N.layers[li].kind == G_INPUT) return fail(RIFE_HIP_EINVAL, N.name + ": input blob " + N.blob_names[b] + " is not bound");
        if (need[li]) continue;
        need[li] = 1;
        for (int bb : N.layers[li].bottoms) if (!have[bb]) stack.push_back(bb);
    }
    int rc;
    for (size_t li = 0; li < N.layers.size(); li++) {
        if (!need[li]) continue;
        const GLayer& L = N.layers[li];
        const NcnnLayer& nl = *L.nl;
        if (L.folded) { have[L.tops[0]] = 1; continue; }                           // executed by its producer (PReLU) or its consumer (SE tail)
        if (L.kind == G_PRELU && L.se_y >= 0) {
            const GView y = I.v[L.se_y], sc = I.v[L.se_scale], sk = I.v[L.se_skip];
            if (!have[L.se_y] || !have[L.se_scale] || !have[L.se_skip]) return fail(RIFE_HIP_EMODEL, N.name + ": SE tail operands missing at " + nl.name);
            if (sc.c != y.c || sc.h != 1 || sc.w != 1 || sk.c != y.c || sk.h != y.h || sk.w != y.w || (int)nl.slope.size() != y.c)
                return fail(RIFE_HIP_EMODEL, N.name + ": SE tail shape mismatch at " + nl.name);
            if ((rc = g_alloc(I, L.out_blob, y.c, y.h, y.w, false, st))) return rc;
            Timed t(E.prof, "g_se_tail", 0, st);
            hipLaunchKernelGGL(kg_se_tail, dim3(g_blocks((size_t)y.h * y.w * y.c)), dim3(256), 0, st, y, (const float*)sc.p, sk, (const float*)L.d_slope, I.v[L.out_blob]);
            HIPCHK(hipGetLastError());
            have[L.out_blob] = 1;
            continue;
        }
        for (int b : L.bottoms)
            if (!have[b]) return fail(RIFE_HIP_EMODEL, N.name + ": blob " + N.blob_names[b] + " is not available for " + nl.name);
        const GView x = L.bottoms.empty() ? GView{nullptr, 0, 0, 0, 0} : I.v[L.bottoms[0]];
        const int ob = L.out_blob;
        auto out_alloc = [&](int c, int h, int w, bool vec = false) { return g_alloc(I, ob, c, h, w, vec, st); };
        switch (L.kind) {
            case G_INPUT: break;
            case G_SPLIT:
                for (int t : L.tops) { I.v[t] = x; have[t] = 1; }
                continue;
            case G_CONCAT: {
                int c = 0;
                for (int b : L.bottoms) c += I.v[b].c;
                if ((rc = out_alloc(c, x.h, x.w))) return rc;
                Timed t(E.prof, "g_concat", 0, st);
                int c0 = 0;
                const size_t npix = (size_t)x.h * x.w;
                for (int b : L.bottoms) {
                    const GView s = I.v[b];
                    if (s.h != x.h || s.w != x.w) return fail(RIFE_HIP_EMODEL, N.name + ": Concat of different sizes at " + nl.name);
                    hipLaunchKernelGGL(kg_copy_channels, dim3(g_blocks(npix * s.c)), dim3(256), 0, st, s.p, s.ld, 0, I.v[ob].p, I.v[ob].ld, c0, s.c, npix);
                    c0 += s.c;
                }
                break;
            }
            case G_CROP: {
                const int c0 = (int)nl.pa.at(9)[0];
                const double e = nl.pa.at(10)[0];
                int c1 = e >= 2147483647.0 ? x.c : (int)e;
                if (c1 > x.c) c1 = x.c;
                if (c1 < 0) c1 += x.c;
                if (c1 <= c0) return fail(RIFE_HIP_EMODEL, N.name + ": empty Crop at " + nl.name);
                if ((rc = out_alloc(c1 - c0, x.h, x.w))) return rc;
                Timed t(E.prof, "g_crop", 0, st);
                const size_t npix = (size_t)x.h * x.w;
                hipLaunchKernelGGL(kg_copy_channels, dim3(g_blocks(npix * (c1 - c0))), dim3(256), 0, st, x.p, x.ld, c0, I.v[ob].p, I.v[ob].ld, 0, c1 - c0, npix);
                break;
            }
            case G_INTERP: {
                auto hs = nl.p.find(1), ws = nl.p.find(2);
                const float fh = hs == nl.p.end() ? 1.f : (float)hs->second, fw = ws == nl.p.end() ? 1.f : (float)ws->second;
                const int oh = (int)(x.h * fh), ow = (int)(x.w * fw);
                if (oh == x.h && ow == x.w) { I.v[ob] = x; have[ob] = 1; continue; }      // ncnn returns the input blob itself
                if (oh <= 0 || ow <= 0) return fail(RIFE_HIP_EMODEL, N.name + ": Interp to an empty blob at " + nl.name);
                if ((rc = out_alloc(x.c, oh, ow))) return rc;
                Timed t(E.prof, "g_interp", 0, st);
                hipLaunchKernelGGL(kg_interp, grid2d(ow, oh), dim3(256), 0, st, x, I.v[ob]);
                break;
            }
            case G_CONV: case G_DECONV: {
                const bool dc = L.kind == G_DECONV;
                const int oh = dc ? 2 * x.h : (x.h - 1) / L.conv.stride + 1, ow = dc ? 2 * x.w : (x.w - 1) / L.conv.stride + 1;
                if (x.c != L.conv.cin) return fail(RIFE_HIP_EMODEL, N.name + ": channel mismatch at " + nl.name);
                if ((rc = out_alloc(nl.geti(0, 0), oh, ow))) return rc;
                {
                    Timed t(E.prof, L.conv.cls, L.conv.flops_per_pixel * (dc ? (double)x.h * x.w : (double)oh * ow), st);
                    if ((rc = launch_conv(L.conv, {x.p, x.ld, 0}, x.h, x.w, {I.v[ob].p, I.v[ob].ld, 0}, nullptr, st))) return rc;
                }
                if (L.post_act == 4) {
                    Timed t(E.prof, "g_pointwise", 0, st);
                    hipLaunchKernelGGL(kg_pointwise, dim3(g_blocks((size_t)oh * ow * I.v[ob].c)), dim3(256), 0, st, I.v[ob], I.v[ob], 1, 0.f, 0.f, (const float*)nullptr);
                }
                break;
            }
            case G_CONV_DIRECT: {
                const int k = nl.geti(1, 1), stride = nl.geti(3, 1), pad = nl.geti(4, 0), outc = nl.geti(0, 0);
                const int oh = (x.h + 2 * pad - k) / stride + 1, ow = (x.w + 2 * pad - k) / stride + 1;
                if (x.c * outc * k * k != nl.geti(6, 0)) return fail(RIFE_HIP_EMODEL, N.name + ": channel mismatch at " + nl.name);
                if ((rc = out_alloc(outc, oh, ow))) return rc;
                {
                    Timed t(E.prof, "g_conv_direct", 2.0 * x.c * outc * k * k * oh * ow, st);
                    hipLaunchKernelGGL(kg_conv_direct, dim3(g_blocks((size_t)oh * ow * (outc / 4))), dim3(256), 0, st, x, I.v[ob], L.d_w, L.d_bias, L.d_slope, k, stride, pad);
                }
                if (L.post_act == 4) hipLaunchKernelGGL(kg_pointwise, dim3(g_blocks((size_t)oh * ow * outc)), dim3(256), 0, st, I.v[ob], I.v[ob], 1, 0.f, 0.f, (const float*)nullptr);
                break;
            }
            case G_PIXELSHUFFLE: {
                const int r = nl.geti(0, 1);
                if (x.c % (r * r)) return fail(RIFE_HIP_EMODEL, N.name + ": PixelShuffle channel count at " + nl.name);
                if ((rc = out_alloc(x.c / (r * r), x.h * r, x.w * r))) return rc;
                Timed t(E.prof, "g_pixelshuffle", 0, st);
                hipLaunchKernelGGL(kg_pixelshuffle, dim3(g_blocks((size_t)I.v[ob].h * I.v[ob].w * I.v[ob].c)), dim3(256), 0, st, x, I.v[ob], r);
                break;
            }
            case G_RELU: case G_PRELU: case G_SIGMOID: case G_CLIP: case G_UNARY: {
                const bool vec = x.h == 1 && x.w == 1 && x.ld == x.c;
                if ((rc = out_alloc(x.c, x.h, x.w, vec))) return rc;
                int mode = 0; float p0 = 0.f, p1 = 0.f;
                if (L.kind == G_SIGMOID) mode = 1;
                else if (L.kind == G_CLIP) { mode = 2; auto lo = nl.p.find(0), hi = nl.p.find(1); p0 = lo == nl.p.end() ? -3.4e38f : (float)lo->second; p1 = hi == nl.p.end() ? 3.4e38f : (float)hi->second; }
                else if (L.kind == G_RELU) { mode = 3; auto sl = nl.p.find(0); p0 = sl == nl.p.end() ? 0.f : (float)sl->second; }
                else if (L.kind == G_PRELU) {
                    mode = 4;
                    if ((int)nl.slope.size() != x.c) return fail(RIFE_HIP_EMODEL, N.name + ": PReLU width mismatch at " + nl.name);
                }
                Timed t(E.prof, "g_pointwise", 0, st);
                hipLaunchKernelGGL(kg_pointwise, dim3(g_blocks((size_t)x.h * x.w * x.c)), dim3(256), 0, st, x, I.v[ob], mode, p0, p1, (const float*)L.d_slope);
                break;
            }
            case G_BINARY: {
                const int op = nl.geti(0, 0);
                const bool vec = x.h == 1 && x.w == 1 && x.ld == x.c;
                if ((rc = out_alloc(x.c, x.h, x.w, vec))) return rc;
                Timed t(E.prof, "g_binary", 0, st);
                const unsigned g = g_blocks((size_t)x.h * x.w * x.c);
                if (L.bottoms.size() == 1) {
                    if (!nl.geti(1, 0)) return fail(RIFE_HIP_EMODEL, N.name + ": unary BinaryOp without a scalar at " + nl.name);
                    auto sb = nl.p.find(2);
                    hipLaunchKernelGGL(kg_binary_scalar, dim3(g), dim3(256), 0, st, x, I.v[ob], op, sb == nl.p.end() ? 0.f : (float)sb->second);
                } else {
                    const GView y = I.v[L.bottoms[1]];
                    int bmode;
                    if (y.h == x.h && y.w == x.w && y.c == x.c) bmode = 0;
                    else if (y.h == 1 && y.w == 1 && y.c == x.c) bmode = 1;
                    else if (y.h == x.h && y.w == x.w && y.c == 1) bmode = 2;
                    else return fail(RIFE_HIP_EMODEL, N.name + ": unsupported BinaryOp broadcast at " + nl.name);
                    hipLaunchKernelGGL(kg_binary, dim3(g), dim3(256), 0, st, x, y, I.v[ob], op, bmode);
                }
                break;
            }
            case G_ELTWISE: {
                const GView y = I.v[L.bottoms[1]];
                if (y.h != x.h || y.w != x.w || y.c != x.c) return fail(RIFE_HIP_EMODEL, N.name + ": Eltwise shape mismatch at " + nl.name);
                if ((rc = out_alloc(x.c, x.h, x.w))) return rc;
                auto cf = nl.pa.find(1);
                const bool hc = cf != nl.pa.end() && cf->second.size() >= 2;
                Timed t(E.prof, "g_binary", 0, st);
                hipLaunchKernelGGL(kg_eltwise2, dim3(g_blocks((size_t)x.h * x.w * x.c)), dim3(256), 0, st, x, y, I.v[ob], hc ? (float)cf->second[0] : 1.f, hc ? (float)cf->second[1] : 1.f, hc ? 1 : 0);
                break;
            }
            case G_POOL: {
                if ((rc = out_alloc(x.c, 1, 1, true))) return rc;
                const size_t npix = (size_t)x.h * x.w;
                if (x.c % 4 || x.c > 1024) return fail(RIFE_HIP_EMODEL, N.name + ": global pooling needs a channel count that is a multiple of 4 (<= 1024) at " + nl.name);
                const int nchunks = (int)std::min<size_t>(512, (npix + 511) / 512);
                const size_t needp = (size_t)nchunks * x.c;
                if (I.partial_cap < needp) {
                    HIPCHK(hipStreamSynchronize(st));
                    if (I.partial) (void)hipFree(I.partial);
                    I.partial = nullptr;
                    HIPCHK(hipMalloc(&I.partial, needp * 8));
                    I.partial_cap = needp;
                }
                Timed t(E.prof, "g_pool", 0, st);
                hipLaunchKernelGGL(kg_pool_partial, dim3(nchunks), dim3(256), 0, st, x, I.partial, nchunks);
                hipLaunchKernelGGL(kg_pool_finish, dim3((x.c + 255) / 256), dim3(256), 0, st, (const double*)I.partial, nchunks, x.c, 1.0 / (double)npix, I.v[ob].p);
                break;
            }
            case G_INNER: {
                if (x.h != 1 || x.w != 1 || x.ld != x.c) return fail(RIFE_HIP_EMODEL, N.name + ": InnerProduct expects a pooled vector at " + nl.name);
                const int outc = nl.geti(0, 0);
                if (x.c * outc != nl.geti(2, 0)) return fail(RIFE_HIP_EMODEL, N.name + ": InnerProduct width mismatch at " + nl.name);
                if ((rc = out_alloc(outc, 1, 1, true))) return rc;
                auto ap = nl.pa.find(10);
                Timed t(E.prof, "g_inner", 0, st);
                hipLaunchKernelGGL(kg_inner, dim3((outc + 63) / 64), dim3(64), 0, st, (const float*)x.p, x.c, (const float*)L.d_w, (const float*)L.d_bias, outc, nl.geti(9, 0),
                                   ap != nl.pa.end() && !ap->second.empty() ? (float)ap->second[0] : 0.f, I.v[ob].p);
                break;
            }
            case G_WARP: {
                const GView f = I.v[L.bottoms[1]];
                if (f.h != x.h || f.w != x.w || f.c < 2) return fail(RIFE_HIP_EMODEL, N.name + ": Warp flow shape mismatch at " + nl.name);
                if ((rc = out_alloc(x.c, x.h, x.w))) return rc;
                Timed t(E.prof, "g_warp", 0, st);
                hipLaunchKernelGGL(kg_warp, grid2d(x.w, x.h), dim3(256), 0, st, x, f, I.v[ob]);
                break;
            }
        }
        HIPCHK(hipGetLastError());
        if (ob >= 0) have[ob] = 1;
    }
    for (const std::string& w : wanted) if (!have[N.blob(w)]) return fail(RIFE_HIP_EMODEL, N.name + ": blob " + w + " was not produced");
    return 0;
}

}  // namespace rife
