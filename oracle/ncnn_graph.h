// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
//
// A minimal CPU interpreter for the ncnn `.param` / `.bin` graphs that
// nihui/rife-ncnn-vulkan feeds to Tencent/ncnn.  It restates, in plain C++
// and fp32, the semantics of the ~16 ncnn layer types the RIFE models use
//  plus the in-tree custom layer `rife.Warp`
// (reference src/warp.cpp:96-168).
//
// PARITY UNPINNED for the layer arithmetic in this file: Tencent/ncnn is an
// un-vendored submodule of the reference (`.gitmodules:1-3`, `src/ncnn/` is
// empty, pinned SHA unknown) and the reference ships no tests or golden
// vectors, so this restatement follows the *published* ncnn layer semantics
// and is pinned only against (a) an independent PyTorch-CPU execution of the
// same graphs (tests/test_oracle_vs_torch.py) and (b) self-generated fixtures
// under tests/golden/.  (The ORCHESTRATION around these layers and rife.Warp
// are pinned by the reference's own compiled code: oracle/refbuild/,
// tests/test_ref_build.py.)
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything in this directory.
#pragma once
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace oracle {

// Planar CHW fp32 tensor (the shape of ncnn::Mat with elempack=1; cstep padding
// is not modelled: channel q starts at q*w*h).
struct Mat {
    int w = 0, h = 0, c = 0;
    std::shared_ptr<std::vector<float>> buf;
    float* data = nullptr;

    Mat() {}
    Mat(int w_, int h_, int c_) { create(w_, h_, c_); }
    void create(int w_, int h_, int c_) {
        w = w_; h = h_; c = c_;
        buf = std::make_shared<std::vector<float>>((size_t)w * h * c);
        data = buf->data();
    }
    bool empty() const { return data == nullptr; }
    size_t total() const { return (size_t)w * h * c; }
    float* channel(int q) { return data + (size_t)q * w * h; }
    const float* channel(int q) const { return data + (size_t)q * w * h; }
    // zero-copy channel range view (ncnn Crop on axis 0 copies; values identical)
    Mat channel_range(int c0, int n) const {
        Mat m; m.w = w; m.h = h; m.c = n; m.buf = buf; m.data = data + (size_t)c0 * w * h; return m;
    }
    Mat clone() const {
        Mat m(w, h, c);
        std::copy(data, data + total(), m.data);
        return m;
    }
};

struct Layer {
    std::string type, name;
    std::vector<int> bottoms, tops;          // blob indices
    std::map<int, double> p;                 // scalar params (id -> value)
    std::map<int, std::vector<double>> pa;   // array params (id -> values)
    // weights (fp32 after widening)
    std::vector<float> weight, bias, slope;
    double getp(int id, double def) const { auto it = p.find(id); return it == p.end() ? def : it->second; }
    int geti(int id, int def) const { return (int)getp(id, def); }
};

struct Net {
    std::vector<Layer> layers;
    std::vector<std::string> blob_names;
    std::map<std::string, int> blob_index;
    std::vector<int> producer;               // blob -> layer index
    int num_threads = 1;
    size_t bin_bytes_consumed = 0, bin_bytes_total = 0;
    // Layer types executed OUTSIDE the interpreter (ncnn::Net::register_custom_layer).  Only oracle/refbuild uses it: there `rife.Warp`
    // is the reference's own compiled Warp::forward (src/warp.cpp:96-168) instead of warp() below.  Empty in liboracle.so.
    std::map<std::string, std::function<int(const std::vector<Mat>& bottoms, std::vector<Mat>& tops)>> custom;

    // returns 0 on success (reference: ncnn::Net::load_param / load_model, rife.cpp:112-121)
    int load_param(const std::string& path);
    int load_model(const std::string& path);
    int find_blob(const std::string& name) const;
};

// Demand-driven evaluation, the semantics of ncnn::Extractor:
// input() binds any blob, extract() lazily evaluates only the missing producers.
struct Extractor {
    const Net* net;
    std::vector<Mat> blobs;
    bool light = true;   // drop intermediates after their last use (ncnn light mode)
    explicit Extractor(const Net& n) : net(&n), blobs(n.blob_names.size()) {}
    int input(const std::string& name, const Mat& m);
    int extract(const std::string& name, Mat& out);
private:
    int forward_layer(int li);
};

// ---- individual op restatements (also exported for per-kernel parity tests) ----
void conv2d(const Mat& in, Mat& out, const float* weight, const float* bias, int outc, int k, int stride, int pad,
            int act_type, const float* act_params, int num_threads);
void deconv2d(const Mat& in, Mat& out, const float* weight, const float* bias, int outc, int k, int stride, int pad,
              int act_type, const float* act_params, int num_threads);
void interp_bilinear(const Mat& in, Mat& out, float hscale, float wscale);
void warp(const Mat& image, const Mat& flow, Mat& out, int num_threads);   // reference src/warp.cpp:96-168
void pixelshuffle(const Mat& in, Mat& out, int r);

}  // namespace oracle
