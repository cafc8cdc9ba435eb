// Reader for the reference's on-disk model format (ncnn `.param` text + `.bin` weights) — the part of
// `RIFE::load()` -> `load_param_model()` (reference src/rife.cpp:112-121) that ncnn::Net::load_param /
// load_model perform.  The engine does not interpret the graph at run time (the schedules are compiled in);
// it parses the files to (a) prove the directory holds the topology a schedule was written for
// (structural hash of the named output blobs) and (b) pull the weights out of the `.bin` stream, which is
// ordered by the weighted layers of the `.param`.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace rife {

struct NcnnLayer {
    std::string type, name;
    std::vector<std::string> bottoms, tops;
    std::map<int, double> p;
    std::map<int, std::vector<double>> pa;
    std::vector<float> weight, bias, slope;   // filled by load_bin
    int geti(int id, int def) const { auto it = p.find(id); return it == p.end() ? def : (int)it->second; }
};

struct NcnnModel {
    std::vector<NcnnLayer> layers;
    std::string error;
    bool load_param(const std::string& path);
    bool load_bin(const std::string& path);
    // order-independent hash of the sub-graph producing `blob` (Split = alias; inputs hashed by name;
    // weighted layers carry their ordinal in the .bin stream). 0 if the blob does not exist.
    uint64_t structural_hash(const std::string& blob) const;
    // weighted layers (Convolution / Deconvolution / PReLU) in .bin order
    std::vector<const NcnnLayer*> weighted() const;
};

}  // namespace rife
