// Generic layer-wise executor for ncnn graphs on the GPU (included by engine.hip after ConvLayer / launch_conv).
//
// The rife-v4.x and rife-v2.x / v3.x families run on hand-written fused schedules.  The v1 family (models/rife, rife-HD,
// rife-UHD, rife-anime) is older, rarely used and structurally different (SE blocks, 5 x 5 convs), so it is
// executed the way the reference executes everything: layer by layer, following the .param in file order with the semantics
// of ncnn::Extractor (bound blobs cut the graph, only layers the requested outputs need are run, Split is an alias).
// Convolution 3x3 / Deconvolution 4x4 still go through the MFMA kernels of conv_mfma.h (a following PReLU is folded into
// the epilogue); everything else uses the plain kernels of graph_kernels.h.  There is no CPU path here either.
#pragma once

namespace rife {

enum GKind { G_INPUT, G_SPLIT, G_CONCAT, G_CROP, G_INTERP, G_CONV, G_DECONV, G_CONV_DIRECT, G_PIXELSHUFFLE, G_RELU, G_PRELU, G_SIGMOID, G_CLIP,
             G_BINARY, G_ELTWISE, G_UNARY, G_POOL, G_INNER, G_WARP };

struct GLayer {
    const NcnnLayer* nl = nullptr;
    GKind kind = G_INPUT;
    std::vector<int> bottoms, tops;
    ConvLayer conv;                                   // G_CONV / G_DECONV
    float *d_w = nullptr, *d_bias = nullptr, *d_slope = nullptr;   // G_CONV_DIRECT / G_INNER / G_PRELU
    int out_blob = -1;                                // where the result goes (the folded PReLU's top for conv + PReLU pairs)
    bool folded = false;                              // PReLU executed by its producer
    int post_act = 0;                                 // activation the conv epilogue cannot apply (4 = sigmoid): extra pointwise pass
    int se_y = -1, se_scale = -1, se_skip = -1;       // G_PRELU closing an SE block: y * scale + skip is computed here too (kg_se_tail)
};

struct GraphNet {
    NcnnModel m;
    std::vector<std::string> blob_names;
    std::map<std::string, int> blob_index;
    std::vector<int> producer;                        // blob -> layer
    std::vector<GLayer> layers;
    std::string name;
    ~GraphNet() {
        for (GLayer& L : layers) {
            free_layer(L.conv);
            if (L.d_w) (void)hipFree(L.d_w);
            if (L.d_bias) (void)hipFree(L.d_bias);
            if (L.d_slope) (void)hipFree(L.d_slope);
        }
    }
    int blob(const std::string& n) const { auto it = blob_index.find(n); return it == blob_index.end() ? -1 : it->second; }
};

static int g_upload(float*& d, const std::vector<float>& v) {
    HIPCHK(hipMalloc(&d, std::max<size_t>(v.size(), 1) * 4));
    if (!v.empty()) HIPCHK(hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return 0;
}

// Parse <base>.param / <base>.bin and prepare every layer.  check_only: no device work (the CPU-side "is every layer of this
// graph supported" probe behind rife_hip_graph_check).
static int graph_load(GraphNet& N, const std::string& base, bool check_only = false) {
    N.name = base;
    if (!N.m.load_param(base + ".param")) return fail(RIFE_HIP_EIO, N.m.error);
    if (!check_only && !N.m.load_bin(base + ".bin")) return fail(RIFE_HIP_EIO, N.m.error);
    auto bid = [&](const std::string& s) {
        auto it = N.blob_index.find(s);
        if (it != N.blob_index.end()) return it->second;
        const int id = (int)N.blob_names.size();
        N.blob_names.push_back(s); N.blob_index[s] = id;
        return id;
    };
    N.layers.resize(N.m.layers.size());
    for (size_t li = 0; li < N.m.layers.size(); li++) {
        const NcnnLayer& nl = N.m.layers[li];
        GLayer& L = N.layers[li];
        L.nl = &nl;
        for (const std::string& b : nl.bottoms) L.bottoms.push_back(bid(b));
        for (const std::string& t : nl.tops) L.tops.push_back(bid(t));
    }
    N.producer.assign(N.blob_names.size(), -1);
    std::vector<int> nuse(N.blob_names.size(), 0), consumer(N.blob_names.size(), -1);
    for (size_t li = 0; li < N.layers.size(); li++) {
        for (int t : N.layers[li].tops) N.producer[t] = (int)li;
        for (int b : N.layers[li].bottoms) { nuse[b]++; consumer[b] = (int)li; }
    }
    int rc;
    for (size_t li = 0; li < N.layers.size(); li++) {
        GLayer& L = N.layers[li];
        const NcnnLayer& nl = *L.nl;
        const std::string& t = nl.type;
        L.out_blob = L.tops.empty() ? -1 : L.tops[0];
        auto bad = [&](const std::string& why) { return fail(RIFE_HIP_EMODEL, base + ".param: layer " + nl.name + " (" + t + "): " + why); };
        {   // arity first: everything below indexes bottoms / tops by position
            const size_t nb = L.bottoms.size(), nt = L.tops.size();
            const bool ok = t == "Input" ? (nb == 0 && nt >= 1) : t == "Split" ? (nb == 1 && nt >= 1) : t == "Concat" ? (nb >= 1 && nt == 1)
                          : t == "BinaryOp" ? ((nb == 1 || nb == 2) && nt == 1) : (t == "Eltwise" || t == "rife.Warp") ? (nb == 2 && nt == 1) : (nb == 1 && nt == 1);
            if (!ok) return bad("unexpected number of inputs / outputs");
        }
        if (t == "Input") L.kind = G_INPUT;
        else if (t == "Split") L.kind = G_SPLIT;
        else if (t == "Concat") L.kind = G_CONCAT;
        else if (t == "Crop") {
            L.kind = G_CROP;
            auto a = nl.pa.find(11);
            if (!nl.pa.count(9) || !nl.pa.count(10) || a == nl.pa.end() || a->second.size() != 1 || (int)a->second[0] != 0) return bad("only channel-axis slices are supported");
        } else if (t == "Interp") { L.kind = G_INTERP; if (nl.geti(0, 0) != 2) return bad("only bilinear resize is supported"); }
        else if (t == "PixelShuffle") L.kind = G_PIXELSHUFFLE;
        else if (t == "ReLU") L.kind = G_RELU;
        else if (t == "Sigmoid") L.kind = G_SIGMOID;
        else if (t == "Clip") L.kind = G_CLIP;
        else if (t == "BinaryOp") { L.kind = G_BINARY; const int op = nl.geti(0, 0); if (op != 0 && op != 1 && op != 2 && op != 3 && op != 7) return bad("unsupported op type"); }
        else if (t == "Eltwise") { L.kind = G_ELTWISE; if (nl.geti(0, 0) != 1 || L.bottoms.size() != 2) return bad("only 2-input SUM is supported"); }
        else if (t == "UnaryOp") { L.kind = G_UNARY; if (nl.geti(0, 0) != 1) return bad("only neg is supported"); }
        else if (t == "Pooling") { L.kind = G_POOL; if (nl.geti(0, 0) != 1 || nl.geti(4, 0) != 1) return bad("only global average pooling is supported"); }
        else if (t == "rife.Warp") L.kind = G_WARP;
        else if (t == "PReLU") {
            L.kind = G_PRELU;
            if (!check_only && (rc = g_upload(L.d_slope, nl.slope))) return rc;
        } else if (t == "InnerProduct") {
            L.kind = G_INNER;
            const int act = nl.geti(9, 0);
            if (act != 0 && act != 1 && act != 2 && act != 4) return bad("unsupported activation");
            if (!check_only) {
                if ((rc = g_upload(L.d_w, nl.weight))) return rc;
                if (nl.geti(1, 0) && (rc = g_upload(L.d_bias, nl.bias))) return rc;
            }
        } else if (t == "Convolution" || t == "Deconvolution") {
            const bool deconv = t == "Deconvolution";
            const int outc = nl.geti(0, 0), k = nl.geti(1, 1), stride = nl.geti(3, 1), pad = nl.geti(4, 0), act = nl.geti(9, 0);
            if (nl.geti(2, 1) != 1) return bad("dilation is not supported");
            const int outc_p = (outc + 3) / 4 * 4;      // the kernels store 4 channels at a time: pad with zero filters (writes zeros into the blob's pad channels)
            if (act != 0 && act != 2 && act != 4) return bad("unsupported fused activation");
            const int cin = outc > 0 && k > 0 ? nl.geti(6, 0) / (outc * k * k) : 0;
            if (cin <= 0 || cin * outc * k * k != nl.geti(6, 0)) return bad("weight count does not factor");
            // conv (+ bias) -> PReLU pairs: the PReLU rides the conv epilogue when the conv output has no other reader
            std::vector<float> slope(outc, 1.0f);
            if (act == 2) { auto ap = nl.pa.find(10); const float sl = ap != nl.pa.end() && !ap->second.empty() ? (float)ap->second[0] : 0.f; slope.assign(outc, sl); }
            if (act == 4) L.post_act = 4;
            const int top = L.tops[0];
            if (act == 0 && nuse[top] == 1 && N.layers[consumer[top]].nl->type == "PReLU") {
                GLayer& P = N.layers[consumer[top]];
                if (!check_only && (int)P.nl->slope.size() == outc) slope = P.nl->slope;
                if (check_only || (int)P.nl->slope.size() == outc) { P.folded = true; L.out_blob = P.tops[0]; }
            }
            const bool mfma = deconv ? (k == 4 && stride == 2 && pad == 1) : ((k == 3 || k == 5) && pad == k / 2 && (stride == 1 || stride == 2));
            if (mfma) {
                L.kind = deconv ? G_DECONV : G_CONV;
                L.conv.cin = cin; L.conv.cout = outc; L.conv.stride = deconv ? 1 : stride; L.conv.deconv = deconv; L.conv.epi = deconv ? EPI_DECONV : EPI_STORE;
                L.conv.ks = deconv ? 3 : k;
                L.conv.cls = deconv ? "g_deconv4x4" : (k == 5 ? "g_conv5x5" : (stride == 2 ? "g_conv3x3_s2" : "g_conv3x3")); L.conv.tag = 0; L.conv.skip = false;
                if (!check_only) {
                    if (outc_p == outc) { if ((rc = upload_layer(L.conv, nl.weight.data(), nl.bias.data(), slope.data(), 1.0f))) return rc; }
                    else {
                        std::vector<float> wp((size_t)outc_p * cin * k * k, 0.f), bp(outc_p, 0.f), sp(outc_p, 1.f);
                        std::copy(nl.weight.begin(), nl.weight.end(), wp.begin());               // [oc][ic][ky][kx]: extra filters go last
                        std::copy(nl.bias.begin(), nl.bias.end(), bp.begin());
                        std::copy(slope.begin(), slope.end(), sp.begin());
                        L.conv.cout = outc_p;
                        if ((rc = upload_layer(L.conv, wp.data(), bp.data(), sp.data(), 1.0f))) return rc;
                    }
                }
            } else {
                if (deconv || pad != k / 2 || (stride != 1 && stride != 2)) return bad("no kernel for this convolution geometry");
                L.kind = G_CONV_DIRECT;
                if (outc % 4) return bad("the direct kernel needs an output channel count that is a multiple of 4");
                if (!check_only) {
                    std::vector<float> w((size_t)k * k * cin * outc);
                    for (int o = 0; o < outc; o++)
                        for (int i = 0; i < cin; i++)
                            for (int kk = 0; kk < k * k; kk++) w[((size_t)kk * cin + i) * outc + o] = nl.weight[((size_t)o * cin + i) * k * k + kk];
                    if ((rc = g_upload(L.d_w, w))) return rc;
                    if ((rc = g_upload(L.d_bias, nl.bias))) return rc;
                    if ((rc = g_upload(L.d_slope, slope))) return rc;
                }
            }
        } else return bad("unsupported layer type");
    }
    // squeeze-and-excitation tails: BinaryOp mul (y, per-channel vector) -> BinaryOp add (., skip) -> PReLU, each link read once
    for (size_t li = 0; li < N.layers.size(); li++) {
        GLayer& P = N.layers[li];
        if (P.kind != G_PRELU || P.folded) continue;
        const int pb = N.producer[P.bottoms[0]];
        if (pb < 0 || nuse[P.bottoms[0]] != 1) continue;
        GLayer& B = N.layers[pb];
        if (B.kind != G_BINARY || B.nl->geti(0, 0) != 0 || B.bottoms.size() != 2) continue;
        const int pa = N.producer[B.bottoms[0]];
        if (pa < 0 || nuse[B.bottoms[0]] != 1) continue;
        GLayer& A = N.layers[pa];
        if (A.kind != G_BINARY || A.nl->geti(0, 0) != 2 || A.bottoms.size() != 2) continue;
        const int pv = N.producer[A.bottoms[1]];
        if (pv < 0 || N.layers[pv].kind != G_INNER) continue;                      // the scale must be the pooled FC output
        P.se_y = A.bottoms[0]; P.se_scale = A.bottoms[1]; P.se_skip = B.bottoms[1];
        A.folded = true; B.folded = true;
    }
    return 0;
}

// blob storage of one running instance of a net (per context; re-used while the geometry stays the same)
struct GraphInst {
    std::vector<GView> v;
    std::vector<float*> owned;
    std::vector<size_t> cap;
    double* partial = nullptr; size_t partial_cap = 0;
    ~GraphInst() {
        for (float* p : owned) if (p) (void)hipFree(p);
        if (partial) (void)hipFree(partial);
    }
};

static int g_alloc(GraphInst& I, int b, int c, int h, int w, bool vec, hipStream_t st) {
    const int ld = vec ? c : (c + 15) / 16 * 16;
    const size_t need = (size_t)h * w * ld;
    if (!I.owned[b] || I.cap[b] < need) {
        HIPCHK(hipStreamSynchronize(st));
        if (I.owned[b]) (void)hipFree(I.owned[b]);
        I.owned[b] = nullptr;
        HIPCHK(hipMalloc(&I.owned[b], need * 4));
        I.cap[b] = need;
        HIPCHK(hipMemsetAsync(I.owned[b], 0, need * 4, st));     // pad channels stay zero for ever: no kernel writes them
    } else if (I.v[b].c != c || I.v[b].h != h || I.v[b].w != w || I.v[b].p != I.owned[b]) {
        HIPCHK(hipMemsetAsync(I.owned[b], 0, need * 4, st));     // same buffer, new geometry: stale values would sit in the pads
    }
    I.v[b] = GView{I.owned[b], c, h, w, ld};
    return 0;
}

static inline unsigned g_blocks(size_t n) { return (unsigned)((n + 255) / 256); }

// Extractor semantics: `bound` blobs are given; run what `wanted` needs; results stay in I.v[...]
static int graph_run(const rife_hip& E, const GraphNet& N, GraphInst& I, hipStream_t st, const std::vector<std::pair<std::string, GView>>& bound,
                     const std::vector<std::string>& wanted);

}  // namespace rife
