#include "ncnn_model.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <set>
#include <sstream>

namespace rife {

bool NcnnModel::load_param(const std::string& path) {
    std::ifstream f(path);
    if (!f) { error = "cannot open " + path; return false; }
    std::string line;
    if (!std::getline(f, line) || std::atoi(line.c_str()) != 7767517) { error = path + ": not an ncnn param file"; return false; }
    int nlayers = 0, nblobs = 0;
    if (!std::getline(f, line)) { error = path + ": truncated"; return false; }
    std::sscanf(line.c_str(), "%d %d", &nlayers, &nblobs);
    layers.clear();
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        NcnnLayer L; int nin = 0, nout = 0;
        if (!(ss >> L.type >> L.name >> nin >> nout)) continue;
        if (nin < 0 || nout < 0 || nin > 4096 || nout > 4096) { error = path + ": implausible blob count in layer " + L.name; return false; }
        L.bottoms.resize(nin); L.tops.resize(nout);
        for (auto& b : L.bottoms) ss >> b;
        for (auto& t : L.tops) ss >> t;
        if (!ss) { error = path + ": layer line of " + L.name + " ends before its blob names"; return false; }
        std::string kv;
        while (ss >> kv) {
            const size_t eq = kv.find('=');
            if (eq == std::string::npos) continue;
            int id = std::atoi(kv.substr(0, eq).c_str());
            const std::string val = kv.substr(eq + 1);
            if (id <= -23300) {
                std::vector<double> arr; std::istringstream vs(val); std::string tok; bool first = true;
                while (std::getline(vs, tok, ',')) { if (first) { first = false; continue; } arr.push_back(std::atof(tok.c_str())); }
                L.pa[-id - 23300] = arr;
            } else {
                L.p[id] = std::atof(val.c_str());
            }
        }
        layers.push_back(std::move(L));
    }
    if ((int)layers.size() != nlayers) { error = path + ": layer count mismatch"; return false; }
    return true;
}

static float h2f(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ff, bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { int sh = 0; while (!(m & 0x400)) { m <<= 1; sh++; } m &= 0x3ff; bits = sign | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) bits = sign | 0x7f800000u | (m << 13);
    else bits = sign | ((e + 112) << 23) | (m << 13);
    float f; std::memcpy(&f, &bits, 4); return f;
}

bool NcnnModel::load_bin(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { error = "cannot open " + path; return false; }
    const size_t total = (size_t)f.tellg();
    f.seekg(0);
    std::vector<uint8_t> raw(total);
    f.read((char*)raw.data(), (std::streamsize)total);
    size_t pos = 0;
    auto have = [&](size_t n) { return pos + n <= total; };
    for (NcnnLayer& L : layers) {
        if (L.type == "Convolution" || L.type == "Deconvolution") {
            const int n = L.geti(6, 0), outc = L.geti(0, 0);
            if (n < 0 || outc < 0 || (size_t)n > total || (size_t)outc > total) { error = path + ": weight count of " + L.name + " exceeds the file"; return false; }
            if (!have(4)) { error = path + ": truncated"; return false; }
            uint32_t tag; std::memcpy(&tag, &raw[pos], 4); pos += 4;
            L.weight.resize(n);
            if (tag == 0x01306B47u) {            // fp16 payload, padded to 4 bytes
                const size_t bytes = ((size_t)n * 2 + 3) & ~(size_t)3;
                if (!have(bytes)) { error = path + ": truncated"; return false; }
                for (int i = 0; i < n; i++) { uint16_t h; std::memcpy(&h, &raw[pos + (size_t)i * 2], 2); L.weight[i] = h2f(h); }
                pos += bytes;
            } else if (tag == 0) {               // raw fp32
                if (!have((size_t)n * 4)) { error = path + ": truncated"; return false; }
                if (n) std::memcpy(L.weight.data(), &raw[pos], (size_t)n * 4);
                pos += (size_t)n * 4;
            } else { error = path + ": unsupported weight storage tag"; return false; }
            L.bias.assign(outc, 0.f);
            if (L.geti(5, 0)) {
                if (!have((size_t)outc * 4)) { error = path + ": truncated"; return false; }
                if (outc) std::memcpy(L.bias.data(), &raw[pos], (size_t)outc * 4);
                pos += (size_t)outc * 4;
            }
        } else if (L.type == "PReLU") {
            const int n = L.geti(0, 0);
            if (n < 0 || !have((size_t)n * 4)) { error = path + ": truncated"; return false; }
            L.slope.resize(n);
            if (n) std::memcpy(L.slope.data(), &raw[pos], (size_t)n * 4);
            pos += (size_t)n * 4;
        } else if (L.type == "InnerProduct") {
            // 0 = num_output, 1 = bias_term, 2 = weight_data_size (SE blocks of the v1 family, models/rife/flownet.param:15-16)
            const int n = L.geti(2, 0), outc = L.geti(0, 0);
            if (n < 0 || outc < 0 || (size_t)n > total || (size_t)outc > total) { error = path + ": weight count of " + L.name + " exceeds the file"; return false; }
            if (!have(4)) { error = path + ": truncated"; return false; }
            uint32_t tag; std::memcpy(&tag, &raw[pos], 4); pos += 4;
            L.weight.resize(n);
            if (tag == 0x01306B47u) {
                const size_t bytes = ((size_t)n * 2 + 3) & ~(size_t)3;
                if (!have(bytes)) { error = path + ": truncated"; return false; }
                for (int i = 0; i < n; i++) { uint16_t h; std::memcpy(&h, &raw[pos + (size_t)i * 2], 2); L.weight[i] = h2f(h); }
                pos += bytes;
            } else if (tag == 0) {
                if (!have((size_t)n * 4)) { error = path + ": truncated"; return false; }
                if (n) std::memcpy(L.weight.data(), &raw[pos], (size_t)n * 4);
                pos += (size_t)n * 4;
            } else { error = path + ": unsupported weight storage tag"; return false; }
            L.bias.assign(outc, 0.f);
            if (L.geti(1, 0)) {
                if (!have((size_t)outc * 4)) { error = path + ": truncated"; return false; }
                if (outc) std::memcpy(L.bias.data(), &raw[pos], (size_t)outc * 4);
                pos += (size_t)outc * 4;
            }
        }
    }
    if (pos != total) { error = path + ": trailing bytes after the last weighted layer"; return false; }
    return true;
}

std::vector<const NcnnLayer*> NcnnModel::weighted() const {
    std::vector<const NcnnLayer*> v;
    for (const NcnnLayer& L : layers)
        if (L.type == "Convolution" || L.type == "Deconvolution" || L.type == "PReLU") v.push_back(&L);
    return v;
}

static inline uint64_t fnv(uint64_t h, const void* d, size_t n) {
    const uint8_t* p = (const uint8_t*)d;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
static inline uint64_t fnv_s(uint64_t h, const std::string& s) { return fnv(fnv(h, s.data(), s.size()), "|", 1); }

uint64_t NcnnModel::structural_hash(const std::string& blob) const {
    std::map<std::string, std::pair<int, int>> producer;   // blob -> (layer, output index)
    std::vector<int> widx(layers.size(), -1);
    int wi = 0;
    for (size_t li = 0; li < layers.size(); li++) {
        const NcnnLayer& L = layers[li];
        if (L.type == "Convolution" || L.type == "Deconvolution" || L.type == "PReLU") widx[li] = wi++;
        for (size_t t = 0; t < L.tops.size(); t++) producer[L.tops[t]] = {(int)li, (int)t};
    }
    std::map<std::string, uint64_t> memo;
    std::set<std::string> open_blobs;                      // blobs on the current path: a malformed file may contain a cycle
    std::function<uint64_t(const std::string&)> H = [&](const std::string& b) -> uint64_t {
        auto mi = memo.find(b);
        if (mi != memo.end()) return mi->second;
        auto pi = producer.find(b);
        if (pi == producer.end()) return 0;
        if (!open_blobs.insert(b).second || open_blobs.size() > 4096) return 0;      // cycle, or deeper than any real graph: no valid hash
        const NcnnLayer& L = layers[pi->second.first];
        uint64_t h = 14695981039346656037ull;
        if (L.type == "Split") h = L.bottoms.empty() ? 0 : H(L.bottoms[0]);
        else if (L.type == "Input") h = fnv_s(fnv_s(h, "Input"), b);
        else {
            char tmp[64];
            h = fnv_s(h, L.type);
            std::snprintf(tmp, sizeof tmp, "o%d w%d", pi->second.second, widx[pi->second.first]);
            h = fnv_s(h, tmp);
            for (auto& kv : L.p) { std::snprintf(tmp, sizeof tmp, "%d=%.9g", kv.first, kv.second); h = fnv_s(h, tmp); }
            for (auto& kv : L.pa) {
                std::snprintf(tmp, sizeof tmp, "a%d", kv.first); h = fnv_s(h, tmp);
                for (double v : kv.second) { std::snprintf(tmp, sizeof tmp, "%.9g", v); h = fnv_s(h, tmp); }
            }
            for (const std::string& x : L.bottoms) { uint64_t hx = H(x); h = fnv(h, &hx, 8); }
        }
        open_blobs.erase(b);
        memo[b] = h;
        return h;
    };
    return H(blob);
}

}  // namespace rife
