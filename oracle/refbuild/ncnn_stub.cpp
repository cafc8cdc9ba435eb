// ORACLE BUILD RECIPE - TEST INFRASTRUCTURE ONLY (see ncnn_stub/ncnn_stub.h).
//
// The part of the ncnn look-alike that executes something: ncnn::Net / ncnn::Extractor on top of the graph interpreter of
// oracle/ncnn_graph.cpp (same layer arithmetic as the restated oracle), custom layers (`rife.Warp` = the reference's own compiled
// Warp::forward, registered by the reference's RIFE::load, src/rife.cpp:146-148) and the three layer types RIFE::load creates by name
// (src/rife.cpp:294-351).
#include "ncnn_stub/ncnn_stub.h"

#include <map>

#include "../ncnn_graph.h"

namespace ncnn {

// ---- tensor conversion: ncnn::Mat (cstep-padded planes) <-> oracle::Mat (dense planes); values are copied bit for bit
static oracle::Mat to_oracle(const Mat& m) {
    const int c = m.dims == 3 ? m.c : 1;
    const int h = m.dims >= 2 ? m.h : 1;
    oracle::Mat o(m.w, h, c);
    const size_t plane = (size_t)m.w * h;
    for (int q = 0; q < c; q++) memcpy(o.channel(q), (const unsigned char*)m.data + m.cstep * q * m.elemsize, plane * sizeof(float));
    return o;
}
static Mat from_oracle(const oracle::Mat& o) {
    Mat m(o.w, o.h, o.c);
    const size_t plane = (size_t)o.w * o.h;
    for (int q = 0; q < o.c; q++) memcpy((unsigned char*)m.data + m.cstep * q * m.elemsize, o.channel(q), plane * sizeof(float));
    return m;
}

// ---- Layer defaults (ncnn: the vector form of a one-blob layer forwards to the single form and vice versa)
int Layer::forward(const std::vector<Mat>& bottom_blobs, std::vector<Mat>& top_blobs, const Option& opt) const {
    if (bottom_blobs.size() != 1 || top_blobs.size() != 1) return -1;
    return forward(bottom_blobs[0], top_blobs[0], opt);
}
int Layer::forward(const Mat&, Mat&, const Option&) const { return -1; }

namespace {

// Interp, bilinear with scale factors (params 0 = 2, 1 = height scale, 2 = width scale): oracle::interp_bilinear
class InterpLayer : public Layer {
public:
    InterpLayer() : resize_type(0), hs(1.f), ws(1.f) { one_blob_only = true; }
    int load_param(const ParamDict& pd) override { resize_type = pd.get(0, 0); hs = pd.get(1, 1.f); ws = pd.get(2, 1.f); return 0; }
    using Layer::forward;
    int forward(const Mat& in, Mat& out, const Option&) const override {
        if (resize_type != 2) return -1;
        oracle::Mat o;
        oracle::interp_bilinear(to_oracle(in), o, hs, ws);
        out = from_oracle(o);
        return 0;
    }
    int resize_type;
    float hs, ws;
};

// BinaryOp with a scalar operand (params 0 = op, 1 = with_scalar, 2 = b); the reference creates "mul by 2" (src/rife.cpp:321-331)
class BinaryOpLayer : public Layer {
public:
    BinaryOpLayer() : op(0), with_scalar(0), b(0.f) { one_blob_only = true; }
    int load_param(const ParamDict& pd) override { op = pd.get(0, 0); with_scalar = pd.get(1, 0); b = pd.get(2, 0.f); return 0; }
    using Layer::forward;
    int forward(const Mat& in, Mat& out, const Option&) const override {
        if (!with_scalar) return -1;
        out = in.clone();
        for (int q = 0; q < out.c; q++) {
            float* p = out.channel(q);
            const size_t n = (size_t)out.w * out.h;
            for (size_t i = 0; i < n; i++) {
                switch (op) {
                    case 0: p[i] = p[i] + b; break;
                    case 1: p[i] = p[i] - b; break;
                    case 2: p[i] = p[i] * b; break;
                    case 3: p[i] = p[i] / b; break;
                    default: return -1;
                }
            }
        }
        return 0;
    }
    int op, with_scalar;
    float b;
};

// Slice along axis 0 (channels) with slice points -233 = "equal parts" (src/rife.cpp:337-350)
class SliceLayer : public Layer {
public:
    SliceLayer() : axis(0) {}
    int load_param(const ParamDict& pd) override { slices = pd.get(0, Mat()); axis = pd.get(1, 0); return 0; }
    using Layer::forward;
    int forward(const std::vector<Mat>& bottoms, std::vector<Mat>& tops, const Option&) const override {
        if (bottoms.size() != 1 || axis != 0 || tops.empty()) return -1;
        const Mat& in = bottoms[0];
        const int* sp = (const int*)slices.data;
        for (size_t i = 0; i < tops.size(); i++)
            if (!sp || (size_t)slices.w != tops.size() || sp[i] != -233) return -1;
        const int per = in.c / (int)tops.size();
        for (size_t i = 0; i < tops.size(); i++) {
            tops[i].create(in.w, in.h, per);
            for (int q = 0; q < per; q++) memcpy((float*)tops[i].channel(q), (const float*)in.channel((int)i * per + q), (size_t)in.w * in.h * sizeof(float));
        }
        return 0;
    }
    Mat slices;
    int axis;
};

}  // namespace

Layer* create_layer(const char* type) {
    const std::string t(type);
    Layer* l = 0;
    if (t == "Interp") l = new InterpLayer;
    else if (t == "BinaryOp") l = new BinaryOpLayer;
    else if (t == "Slice") l = new SliceLayer;
    if (!l) { fprintf(stderr, "ncnn_stub: create_layer(%s) is not provided\n", type); abort(); }
    l->type = t;
    return l;
}

// ---- Net / Extractor
class NetImpl {
public:
    oracle::Net net;
    std::map<std::string, std::unique_ptr<Layer>> custom;       // one instance per registered type (rife.Warp has no parameters)
    const Option* opt = 0;
};

class ExtractorImpl {
public:
    explicit ExtractorImpl(const oracle::Net& n) : ex(n) {}
    oracle::Extractor ex;
};

Net::Net() : d(new NetImpl) { d->opt = &opt; }
Net::~Net() {
    for (auto& kv : d->custom) kv.second->destroy_pipeline(opt);
    delete d;
}

int Net::register_custom_layer(const char* type, layer_creator_func creator, layer_destroyer_func, void* userdata) {
    Layer* l = creator(userdata);
    if (!l) return -1;
    l->type = type;
    d->custom[type].reset(l);
    NetImpl* impl = d;
    d->net.custom[type] = [impl, l](const std::vector<oracle::Mat>& bottoms, std::vector<oracle::Mat>& tops) -> int {
        std::vector<Mat> b, t(tops.size());
        for (const oracle::Mat& m : bottoms) b.push_back(from_oracle(m));
        int r = l->forward(b, t, *impl->opt);
        if (r) return r;
        for (size_t i = 0; i < tops.size(); i++) tops[i] = to_oracle(t[i]);
        return 0;
    };
    return 0;
}

int Net::load_param(const char* path) {
    d->net.num_threads = opt.num_threads;
    int r = d->net.load_param(path);
    if (r) return r;
    // ncnn: every layer gets vkdev and create_pipeline(opt) after load_model; the custom layers have no weights
    for (auto& kv : d->custom) { kv.second->vkdev = 0; kv.second->create_pipeline(opt); }
    return 0;
}
int Net::load_model(const char* path) { return d->net.load_model(path); }

Extractor Net::create_extractor() const {
    d->net.num_threads = opt.num_threads;
    return Extractor(d);
}

Extractor::Extractor(const NetImpl* net) : d(new ExtractorImpl(net->net)) {}
Extractor::~Extractor() {}
Extractor::Extractor(const Extractor& o) : d(o.d) {}
Extractor& Extractor::operator=(const Extractor& o) { d = o.d; return *this; }

int Extractor::input(const char* blob_name, const Mat& in) { return d->ex.input(blob_name, to_oracle(in)); }
int Extractor::extract(const char* blob_name, Mat& feat) {
    oracle::Mat o;
    int r = d->ex.extract(blob_name, o);
    if (r) return r;
    feat = from_oracle(o);
    return 0;
}

}  // namespace ncnn
