// Drives the RIFE class exactly like the reference's proc thread does (src/main.cpp:187, 332, 360):
//   shim_demo <modeldir> <w> <h> <timestep> <in0.rgb> <in1.rgb> <out.rgb>     (raw tightly packed RGB files)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rife.h"

static std::vector<unsigned char> slurp(const char* p, size_t n) {
    std::vector<unsigned char> v(n);
    FILE* f = fopen(p, "rb");
    if (!f || fread(v.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", p); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc != 8) { fprintf(stderr, "usage: shim_demo modeldir w h timestep in0 in1 out\n"); return 2; }
    const int w = atoi(argv[2]), h = atoi(argv[3]);
    const float t = (float)atof(argv[4]);
    std::vector<unsigned char> p0 = slurp(argv[5], (size_t)w * h * 3), p1 = slurp(argv[6], (size_t)w * h * 3);
    RIFE rife(0, false, false, false, 1, false, true);
    if (rife.load(argv[1]) != 0) return 3;
    ncnn::Mat in0(w, h, (void*)p0.data(), (size_t)3, 3);
    ncnn::Mat in1(w, h, (void*)p1.data(), (size_t)3, 3);
    ncnn::Mat out(w, h, (size_t)3, 3);
    if (rife.process(in0, in1, t, out) != 0) return 4;
    FILE* f = fopen(argv[7], "wb");
    fwrite(out.data, 1, (size_t)w * h * 3, f);
    fclose(f);
    // timestep 0 shares the input buffer, like the reference
    ncnn::Mat alias(w, h, (size_t)3, 3);
    rife.process(in0, in1, 0.f, alias);
    if (alias.data != in0.data) return 5;
    return 0;
}
