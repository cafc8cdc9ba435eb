// ORACLE — TEST INFRASTRUCTURE ONLY (see ncnn_graph.h header).  PARITY UNPINNED.
//
// Param/bin reader + demand-driven graph evaluation + the element-wise /
// resampling layers.  Compiled with -ffp-contract=off so that every a*b+c below
// is two roundings, which is what the reference's portable C++ (e.g. warp.cpp,
// built without -march flags, src/CMakeLists.txt) and ncnn's generic layer
// code do.
#include "ncnn_graph.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace oracle {

// ---------------------------------------------------------------------------------------------
// .param  (ncnn text format: magic 7767517, "<layers> <blobs>", one layer / line,
// "id=value" scalars and "-(23300+id)=n,v0,v1,.." arrays, e.g. rife-v4.6/flownet.param:11,49)
// ---------------------------------------------------------------------------------------------
int Net::find_blob(const std::string& name) const {
    auto it = blob_index.find(name);
    return it == blob_index.end() ? -1 : it->second;
}

int Net::load_param(const std::string& path) {
    std::ifstream f(path);
    if (!f) return -1;
    std::string line;
    if (!std::getline(f, line)) return -2;
    if (std::atoi(line.c_str()) != 7767517) return -3;
    if (!std::getline(f, line)) return -2;
    int nlayers = 0, nblobs = 0;
    { std::istringstream ss(line); ss >> nlayers >> nblobs; }
    layers.clear(); blob_names.clear(); blob_index.clear();
    auto blob_id = [&](const std::string& n) {
        auto it = blob_index.find(n);
        if (it != blob_index.end()) return it->second;
        int id = (int)blob_names.size();
        blob_names.push_back(n); blob_index[n] = id;
        return id;
    };
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        Layer L; int nin = 0, nout = 0;
        if (!(ss >> L.type >> L.name >> nin >> nout)) continue;
        for (int i = 0; i < nin; i++) { std::string b; ss >> b; L.bottoms.push_back(blob_id(b)); }
        for (int i = 0; i < nout; i++) { std::string b; ss >> b; L.tops.push_back(blob_id(b)); }
        std::string kv;
        while (ss >> kv) {
            size_t eq = kv.find('=');
            if (eq == std::string::npos) continue;
            int id = std::atoi(kv.substr(0, eq).c_str());
            std::string val = kv.substr(eq + 1);
            if (id <= -23300) {
                id = -id - 23300;
                std::vector<double> arr; std::istringstream vs(val); std::string tok; bool first = true;
                while (std::getline(vs, tok, ',')) { if (first) { first = false; continue; } arr.push_back(std::atof(tok.c_str())); }
                L.pa[id] = arr;
            } else {
                L.p[id] = std::atof(val.c_str());
            }
        }
        layers.push_back(L);
    }
    if ((int)layers.size() != nlayers) return -4;
    producer.assign(blob_names.size(), -1);
    for (size_t li = 0; li < layers.size(); li++)
        for (int t : layers[li].tops) producer[t] = (int)li;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// .bin  (SURVEY App. D): per Convolution/Deconvolution a u32 tag (0x01306B47 => fp16 payload padded
// to 4 B; 0 => raw fp32) + weights, then num_output fp32 biases; per PReLU num_slope fp32.
// ---------------------------------------------------------------------------------------------
static float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, f;
    if (exp == 0) {
        if (man == 0) f = sign;
        else { int e = -1; do { e++; man <<= 1; } while (!(man & 0x400)); man &= 0x3ff; f = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13); }
    } else if (exp == 31) f = sign | 0x7f800000u | (man << 13);
    else f = sign | ((exp + 112) << 23) | (man << 13);
    float out; std::memcpy(&out, &f, 4); return out;
}

int Net::load_model(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return -1;
    f.seekg(0, std::ios::end); bin_bytes_total = (size_t)f.tellg(); f.seekg(0);
    std::vector<uint8_t> raw(bin_bytes_total);
    f.read((char*)raw.data(), (std::streamsize)bin_bytes_total);
    size_t pos = 0;
    auto need = [&](size_t n) { return pos + n <= raw.size(); };
    for (Layer& L : layers) {
        if (L.type == "Convolution" || L.type == "Deconvolution") {
            int n = L.geti(6, 0), outc = L.geti(0, 0), has_bias = L.geti(5, 0);
            if (!need(4)) return -2;
            uint32_t tag; std::memcpy(&tag, &raw[pos], 4); pos += 4;
            L.weight.resize(n);
            if (tag == 0x01306B47u) {
                size_t bytes = ((size_t)n * 2 + 3) / 4 * 4;
                if (!need(bytes)) return -2;
                for (int i = 0; i < n; i++) { uint16_t h; std::memcpy(&h, &raw[pos + 2 * (size_t)i], 2); L.weight[i] = half_to_float(h); }
                pos += bytes;
            } else if (tag == 0) {
                if (!need((size_t)n * 4)) return -2;
                std::memcpy(L.weight.data(), &raw[pos], (size_t)n * 4); pos += (size_t)n * 4;
            } else return -5;   // quantized / int8 storage: not used by any RIFE model
            L.bias.assign(outc, 0.f);
            if (has_bias) {
                if (!need((size_t)outc * 4)) return -2;
                std::memcpy(L.bias.data(), &raw[pos], (size_t)outc * 4); pos += (size_t)outc * 4;
            }
        } else if (L.type == "PReLU") {
            int n = L.geti(0, 0);
            if (!need((size_t)n * 4)) return -2;
            L.slope.resize(n);
            std::memcpy(L.slope.data(), &raw[pos], (size_t)n * 4); pos += (size_t)n * 4;
        } else if (L.type == "InnerProduct") {
            // ncnn InnerProduct: 0 = num_output, 1 = bias_term, 2 = weight_data_size; same tagged weight block as Convolution
            // (the SE blocks of the rife / rife-HD / rife-UHD / rife-anime graphs, e.g. models/rife/flownet.param:15-16)
            int n = L.geti(2, 0), outc = L.geti(0, 0), has_bias = L.geti(1, 0);
            if (!need(4)) return -2;
            uint32_t tag; std::memcpy(&tag, &raw[pos], 4); pos += 4;
            L.weight.resize(n);
            if (tag == 0x01306B47u) {
                size_t bytes = ((size_t)n * 2 + 3) / 4 * 4;
                if (!need(bytes)) return -2;
                for (int i = 0; i < n; i++) { uint16_t h; std::memcpy(&h, &raw[pos + 2 * (size_t)i], 2); L.weight[i] = half_to_float(h); }
                pos += bytes;
            } else if (tag == 0) {
                if (!need((size_t)n * 4)) return -2;
                std::memcpy(L.weight.data(), &raw[pos], (size_t)n * 4); pos += (size_t)n * 4;
            } else return -5;
            L.bias.assign(outc, 0.f);
            if (has_bias) {
                if (!need((size_t)outc * 4)) return -2;
                std::memcpy(L.bias.data(), &raw[pos], (size_t)outc * 4); pos += (size_t)outc * 4;
            }
        }
    }
    bin_bytes_consumed = pos;
    return pos == raw.size() ? 0 : -7;
}

// ---------------------------------------------------------------------------------------------
// Layers
// ---------------------------------------------------------------------------------------------

// ncnn Interp, resize_type=2 (bilinear), align_corner=0: linear_coeffs + resize_bilinear_image
// (horizontal pass first, then vertical).  SURVEY App. C-6.
static void linear_coeffs(int w, int outw, std::vector<int>& xofs, std::vector<float>& alpha) {
    double scale = (double)w / outw;
    xofs.resize(outw); alpha.resize((size_t)outw * 2);
    for (int dx = 0; dx < outw; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { sx = 0; fx = 0.f; }
        if (sx >= w - 1) { sx = w - 2; fx = 1.f; }
        xofs[dx] = sx;
        alpha[dx * 2] = 1.f - fx;
        alpha[dx * 2 + 1] = fx;
    }
}

void interp_bilinear(const Mat& in, Mat& out, float hscale, float wscale) {
    int outw = (int)(in.w * wscale), outh = (int)(in.h * hscale);
    out.create(outw, outh, in.c);
    std::vector<int> xofs, yofs; std::vector<float> alpha, beta;
    linear_coeffs(in.w, outw, xofs, alpha);
    linear_coeffs(in.h, outh, yofs, beta);
    for (int q = 0; q < in.c; q++) {
        const float* src = in.channel(q);
        float* dst = out.channel(q);
        std::vector<float> rows0(outw), rows1(outw);
        for (int dy = 0; dy < outh; dy++) {
            int sy = yofs[dy];
            const float* S0 = src + (size_t)sy * in.w;
            const float* S1 = src + (size_t)(sy + 1) * in.w;
            for (int dx = 0; dx < outw; dx++) {
                int sx = xofs[dx];
                float a0 = alpha[dx * 2], a1 = alpha[dx * 2 + 1];
                rows0[dx] = S0[sx] * a0 + S0[sx + 1] * a1;
                rows1[dx] = S1[sx] * a0 + S1[sx + 1] * a1;
            }
            float b0 = beta[dy * 2], b1 = beta[dy * 2 + 1];
            float* D = dst + (size_t)dy * outw;
            for (int dx = 0; dx < outw; dx++) D[dx] = rows0[dx] * b0 + rows1[dx] * b1;
        }
    }
}

// reference src/warp.cpp:96-168 (CPU Warp::forward), restated literally: same operation order,
// alpha/beta taken against the *clamped* x0/y0.
void warp(const Mat& image, const Mat& flow, Mat& out, int num_threads) {
    int w = image.w, h = image.h, channels = image.c;
    out.create(w, h, channels);
#pragma omp parallel for num_threads(num_threads)
    for (int q = 0; q < channels; q++) {
        float* outptr = out.channel(q);
        const float* img = image.channel(q);
        const float* fxptr = flow.channel(0);
        const float* fyptr = flow.channel(1);
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                float flow_x = *fxptr++, flow_y = *fyptr++;
                float sample_x = x + flow_x;
                float sample_y = y + flow_y;
                int x0 = (int)std::floor(sample_x);
                int y0 = (int)std::floor(sample_y);
                int x1 = x0 + 1, y1 = y0 + 1;
                x0 = std::min(std::max(x0, 0), w - 1);
                y0 = std::min(std::max(y0, 0), h - 1);
                x1 = std::min(std::max(x1, 0), w - 1);
                y1 = std::min(std::max(y1, 0), h - 1);
                float alpha = sample_x - x0;
                float beta = sample_y - y0;
                float v0 = img[(size_t)y0 * w + x0];
                float v1 = img[(size_t)y0 * w + x1];
                float v2 = img[(size_t)y1 * w + x0];
                float v3 = img[(size_t)y1 * w + x1];
                float v4 = v0 * (1 - alpha) + v1 * alpha;
                float v5 = v2 * (1 - alpha) + v3 * alpha;
                *outptr++ = v4 * (1 - beta) + v5 * beta;
            }
        }
    }
}

// ncnn PixelShuffle mode 0
void pixelshuffle(const Mat& in, Mat& out, int r) {
    int outc = in.c / (r * r);
    out.create(in.w * r, in.h * r, outc);
    for (int p = 0; p < outc; p++)
        for (int sh = 0; sh < r; sh++)
            for (int sw = 0; sw < r; sw++) {
                const float* sptr = in.channel(p * r * r + sh * r + sw);
                float* o = out.channel(p);
                for (int i = 0; i < in.h; i++)
                    for (int j = 0; j < in.w; j++)
                        o[(size_t)(i * r + sh) * out.w + (j * r + sw)] = sptr[(size_t)i * in.w + j];
            }
}

static inline float binop(int op, float a, float b) {
    switch (op) {
        case 0: return a + b;
        case 1: return a - b;
        case 2: return a * b;
        case 3: return a / b;
        case 4: return std::max(a, b);
        case 5: return std::min(a, b);
        case 7: return b - a;   // RSUB
        case 8: return b / a;   // RDIV
    }
    std::fprintf(stderr, "oracle: unsupported BinaryOp %d\n", op); std::abort();
}

int Extractor::input(const std::string& name, const Mat& m) {
    int b = net->find_blob(name);
    if (b < 0) return -1;
    blobs[b] = m;
    return 0;
}

int Extractor::extract(const std::string& name, Mat& out) {
    int target = net->find_blob(name);
    if (target < 0) return -1;
    if (blobs[target].empty()) {
        // reverse reachability from the target, stopping at blobs that are already bound (ncnn's lazy
        // Extractor semantics), then run the needed layers in file (= topological) order,
        // dropping every intermediate as soon as its last needed consumer has run (ncnn "light mode").
        const size_t nl = net->layers.size();
        std::vector<char> need(nl, 0);
        std::vector<int> stack;
        if (net->producer[target] < 0) return -2;
        stack.push_back(net->producer[target]);
        while (!stack.empty()) {
            int li = stack.back(); stack.pop_back();
            if (need[li]) continue;
            need[li] = 1;
            for (int b : net->layers[li].bottoms) {
                if (!blobs[b].empty()) continue;
                int pl = net->producer[b];
                if (pl < 0 || net->layers[pl].type == "Input") {
                    std::fprintf(stderr, "oracle: input blob %s not bound\n", net->blob_names[b].c_str());
                    return -4;
                }
                stack.push_back(pl);
            }
        }
        std::vector<int> uses(blobs.size(), 0);
        for (size_t li = 0; li < nl; li++) if (need[li]) for (int b : net->layers[li].bottoms) uses[b]++;
        uses[target]++;
        std::vector<char> bound(blobs.size(), 0);
        for (size_t b = 0; b < blobs.size(); b++) bound[b] = !blobs[b].empty();
        for (size_t li = 0; li < nl; li++) {
            if (!need[li]) continue;
            int r = forward_layer((int)li);
            if (r) return r;
            if (!light) continue;
            for (int b : net->layers[li].bottoms) if (--uses[b] == 0 && !bound[b]) blobs[b] = Mat();
            for (int t : net->layers[li].tops) if (uses[t] == 0) blobs[t] = Mat();
        }
    }
    out = blobs[target];
    return 0;
}

int Extractor::forward_layer(int li) {
    const Layer& L = net->layers[li];
    for (int b : L.bottoms)
        if (blobs[b].empty()) { std::fprintf(stderr, "oracle: blob %s missing\n", net->blob_names[b].c_str()); return -3; }
    const int nt = net->num_threads;
    const std::string& t = L.type;
    if (!net->custom.empty()) {
        auto cu = net->custom.find(t);
        if (cu != net->custom.end()) {
            std::vector<Mat> bottoms, tops(L.tops.size());
            for (int b : L.bottoms) bottoms.push_back(blobs[b]);
            int r = cu->second(bottoms, tops);
            if (r) return r;
            for (size_t i = 0; i < L.tops.size(); i++) blobs[L.tops[i]] = tops[i];
            return 0;
        }
    }
    if (t == "Input") {
        std::fprintf(stderr, "oracle: input blob %s not bound\n", net->blob_names[L.tops[0]].c_str());
        return -4;
    } else if (t == "Split") {
        for (int o : L.tops) blobs[o] = blobs[L.bottoms[0]];
    } else if (t == "Concat") {
        int w = blobs[L.bottoms[0]].w, h = blobs[L.bottoms[0]].h, c = 0;
        for (int b : L.bottoms) {
            if (blobs[b].w != w || blobs[b].h != h) {   // e.g. -u on a padded size that is not a multiple of 64: the graph's pyramid does not close
                std::fprintf(stderr, "oracle: Concat %s: %dx%d vs %dx%d\n", L.name.c_str(), blobs[b].w, blobs[b].h, w, h);
                return -14;
            }
            c += blobs[b].c;
        }
        Mat out(w, h, c); int q = 0;
        for (int b : L.bottoms) {
            const Mat& m = blobs[b];
            std::copy(m.data, m.data + m.total(), out.channel(q)); q += m.c;
        }
        blobs[L.tops[0]] = out;
    } else if (t == "Crop") {
        // channel-axis slice only (-23309 starts, -23310 ends, -23311 axes = 0), SURVEY App. C-8
        const Mat& in = blobs[L.bottoms[0]];
        auto s = L.pa.find(9), e = L.pa.find(10), a = L.pa.find(11);
        if (s == L.pa.end() || e == L.pa.end() || a == L.pa.end() || a->second.size() != 1 || (int)a->second[0] != 0) return -5;
        int c0 = (int)s->second[0];
        double ed = e->second[0];
        int c1 = ed >= 2147483647.0 ? in.c : (int)ed;
        if (c1 > in.c) c1 = in.c;
        if (c1 < 0) c1 += in.c;
        blobs[L.tops[0]] = in.channel_range(c0, c1 - c0).clone();
    } else if (t == "Slice") {
        // equal split along channels (the {-233,-233} helper the reference creates in code, rife.cpp:334-351)
        const Mat& in = blobs[L.bottoms[0]];
        int n = (int)L.tops.size(), per = in.c / n;
        for (int i = 0; i < n; i++) blobs[L.tops[i]] = in.channel_range(i * per, per).clone();
    } else if (t == "Interp") {
        if (L.geti(0, 0) != 2) return -6;
        Mat out; interp_bilinear(blobs[L.bottoms[0]], out, (float)L.getp(1, 1.0), (float)L.getp(2, 1.0));
        blobs[L.tops[0]] = out;
    } else if (t == "Convolution" || t == "Deconvolution") {
        int outc = L.geti(0, 0), k = L.geti(1, 1), stride = L.geti(3, 1), pad = L.geti(4, 0), act = L.geti(9, 0);
        float actp[2] = {0.f, 0.f};
        auto ap = L.pa.find(10);
        if (ap != L.pa.end()) for (size_t i = 0; i < ap->second.size() && i < 2; i++) actp[i] = (float)ap->second[i];
        Mat out;
        if (t == "Convolution") conv2d(blobs[L.bottoms[0]], out, L.weight.data(), L.bias.data(), outc, k, stride, pad, act, actp, nt);
        else deconv2d(blobs[L.bottoms[0]], out, L.weight.data(), L.bias.data(), outc, k, stride, pad, act, actp, nt);
        blobs[L.tops[0]] = out;
    } else if (t == "PixelShuffle") {
        Mat out; pixelshuffle(blobs[L.bottoms[0]], out, L.geti(0, 1));
        blobs[L.tops[0]] = out;
    } else if (t == "ReLU") {
        float slope = (float)L.getp(0, 0.0);
        Mat out = blobs[L.bottoms[0]].clone();
        size_t n = out.total();
        if (slope == 0.f) { for (size_t i = 0; i < n; i++) if (out.data[i] < 0) out.data[i] = 0; }
        else { for (size_t i = 0; i < n; i++) if (out.data[i] < 0) out.data[i] *= slope; }
        blobs[L.tops[0]] = out;
    } else if (t == "PReLU") {
        Mat out = blobs[L.bottoms[0]].clone();
        size_t plane = (size_t)out.w * out.h;
        for (int q = 0; q < out.c; q++) {
            float s = L.slope.size() > 1 ? L.slope[q] : L.slope[0];
            float* p = out.channel(q);
            for (size_t i = 0; i < plane; i++) if (p[i] < 0) p[i] *= s;
        }
        blobs[L.tops[0]] = out;
    } else if (t == "Sigmoid") {
        Mat out = blobs[L.bottoms[0]].clone();
        size_t n = out.total();
        for (size_t i = 0; i < n; i++) out.data[i] = 1.f / (1.f + std::exp(-out.data[i]));
        blobs[L.tops[0]] = out;
    } else if (t == "Clip") {
        float lo = (float)L.getp(0, -3.4e38), hi = (float)L.getp(1, 3.4e38);
        Mat out = blobs[L.bottoms[0]].clone();
        size_t n = out.total();
        for (size_t i = 0; i < n; i++) { float v = out.data[i]; if (v < lo) v = lo; if (v > hi) v = hi; out.data[i] = v; }
        blobs[L.tops[0]] = out;
    } else if (t == "BinaryOp") {
        int op = L.geti(0, 0), with_scalar = L.geti(1, 0); float sb = (float)L.getp(2, 0.0);
        const Mat& a = blobs[L.bottoms[0]];
        Mat out(a.w, a.h, a.c);
        size_t plane = (size_t)a.w * a.h;
        if (L.bottoms.size() == 1) {
            if (!with_scalar) return -7;
            for (size_t i = 0; i < a.total(); i++) out.data[i] = binop(op, a.data[i], sb);
        } else {
            const Mat& b = blobs[L.bottoms[1]];
            if (b.w == 1 && b.h == 1 && b.c == a.c && (a.w != 1 || a.h != 1)) {   // per-channel scalar (SE scale; ncnn broadcasts a 1-D operand of length c)
                for (int q = 0; q < a.c; q++) { const float bv = b.channel(q)[0]; for (size_t i = 0; i < plane; i++) out.channel(q)[i] = binop(op, a.channel(q)[i], bv); }
                blobs[L.tops[0]] = out;
                return 0;
            }
            if (b.w != a.w || b.h != a.h) return -8;
            if (b.c == a.c) { for (size_t i = 0; i < a.total(); i++) out.data[i] = binop(op, a.data[i], b.data[i]); }
            else if (b.c == 1) {   // 1-channel operand broadcast over channels (flownet.param:213,216)
                for (int q = 0; q < a.c; q++) for (size_t i = 0; i < plane; i++) out.channel(q)[i] = binop(op, a.channel(q)[i], b.data[i]);
            } else return -8;
        }
        blobs[L.tops[0]] = out;
    } else if (t == "Eltwise") {
        if (L.geti(0, 0) != 1) return -9;   // SUM only
        const Mat& a = blobs[L.bottoms[0]];
        Mat out(a.w, a.h, a.c);
        auto cf = L.pa.find(1);
        size_t n = a.total();
        if (cf == L.pa.end() || cf->second.empty()) {
            const Mat& b = blobs[L.bottoms[1]];
            for (size_t i = 0; i < n; i++) out.data[i] = a.data[i] + b.data[i];
            for (size_t bi = 2; bi < L.bottoms.size(); bi++) { const Mat& m = blobs[L.bottoms[bi]]; for (size_t i = 0; i < n; i++) out.data[i] += m.data[i]; }
        } else {
            const Mat& b = blobs[L.bottoms[1]];
            float c0 = (float)cf->second[0], c1 = (float)cf->second[1];
            for (size_t i = 0; i < n; i++) out.data[i] = a.data[i] * c0 + b.data[i] * c1;
            for (size_t bi = 2; bi < L.bottoms.size(); bi++) { const Mat& m = blobs[L.bottoms[bi]]; float ck = (float)cf->second[bi]; for (size_t i = 0; i < n; i++) out.data[i] += m.data[i] * ck; }
        }
        blobs[L.tops[0]] = out;
    } else if (t == "Pooling") {
        // global average pooling only (0 = 1 avg, 4 = 1 global): sequential fp32 sum / size -> c values (ncnn Pooling::forward)
        if (L.geti(0, 0) != 1 || L.geti(4, 0) != 1) return -11;
        const Mat& in = blobs[L.bottoms[0]];
        Mat out(1, 1, in.c);
        const size_t plane = (size_t)in.w * in.h;
        for (int q = 0; q < in.c; q++) {
            const float* p = in.channel(q);
            float sum = 0.f;
            for (size_t i = 0; i < plane; i++) sum += p[i];
            out.channel(q)[0] = sum / (float)plane;
        }
        blobs[L.tops[0]] = out;
    } else if (t == "InnerProduct") {
        const Mat& in = blobs[L.bottoms[0]];
        const int outc = L.geti(0, 0), n = (int)in.total(), act = L.geti(9, 0);
        if ((int)L.weight.size() != outc * n) return -12;
        float actp[2] = {0.f, 0.f};
        auto ap = L.pa.find(10);
        if (ap != L.pa.end()) for (size_t i = 0; i < ap->second.size() && i < 2; i++) actp[i] = (float)ap->second[i];
        Mat out(1, 1, outc);
        for (int p = 0; p < outc; p++) {
            float sum = L.bias[p];
            const float* wv = L.weight.data() + (size_t)p * n;
            for (int i = 0; i < n; i++) sum += in.data[i] * wv[i];
            if (act == 1) sum = sum > 0 ? sum : 0.f;
            else if (act == 2) sum = sum > 0 ? sum : sum * actp[0];
            else if (act == 3) sum = std::min(std::max(sum, actp[0]), actp[1]);
            else if (act == 4) sum = 1.f / (1.f + std::exp(-sum));
            out.channel(p)[0] = sum;
        }
        blobs[L.tops[0]] = out;
    } else if (t == "UnaryOp") {
        if (L.geti(0, 0) != 1) return -13;   // 1 = neg (the only op the RIFE graphs use)
        Mat out = blobs[L.bottoms[0]].clone();
        size_t n = out.total();
        for (size_t i = 0; i < n; i++) out.data[i] = -out.data[i];
        blobs[L.tops[0]] = out;
    } else if (t == "rife.Warp") {
        Mat out; warp(blobs[L.bottoms[0]], blobs[L.bottoms[1]], out, nt);
        blobs[L.tops[0]] = out;
    } else {
        std::fprintf(stderr, "oracle: unsupported layer type %s\n", t.c_str());
        return -10;
    }
    return 0;
}

}  // namespace oracle
