// Minimal stand-in for the slice of `ncnn::Mat` that crosses the reference's RIFE boundary
// (reference src/main.cpp:187 `ncnn::Mat(w, h, (void*)pixeldata, (size_t)3, 3)` wraps caller-owned pixels,
//  src/main.cpp:332 `ncnn::Mat(w, h, (size_t)3, 3)` allocates the output; src/rife.cpp:407-411, 2482-2486 read
//  `.data/.w/.h`; rife.cpp:2470-2480 assigns one Mat to another, sharing the buffer).
// Real ncnn is an un-vendored submodule of the reference, so a build against this engine uses this header; a build
// inside the reference tree keeps using ncnn's own Mat — the field names and semantics match.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdlib>

namespace ncnn {

class Mat {
public:
    Mat() {}
    // caller-owned pixels, not freed (src/main.cpp:187)
    Mat(int w_, int h_, void* data_, size_t elemsize_, int elempack_) : data(data_), elemsize(elemsize_), elempack(elempack_), w(w_), h(h_), c(1) {}
    // owning, reference counted (src/main.cpp:332)
    Mat(int w_, int h_, size_t elemsize_, int elempack_) : elemsize(elemsize_), elempack(elempack_), w(w_), h(h_), c(1) {
        data = std::malloc((size_t)w * h * elemsize);
        refcount = new std::atomic<int>(1);
    }
    Mat(const Mat& m) : data(m.data), refcount(m.refcount), elemsize(m.elemsize), elempack(m.elempack), w(m.w), h(m.h), c(m.c) {
        if (refcount) refcount->fetch_add(1);
    }
    Mat& operator=(const Mat& m) {
        if (this == &m) return *this;
        if (m.refcount) m.refcount->fetch_add(1);
        release();
        data = m.data; refcount = m.refcount; elemsize = m.elemsize; elempack = m.elempack; w = m.w; h = m.h; c = m.c;
        return *this;
    }
    ~Mat() { release(); }
    void release() {
        if (refcount && refcount->fetch_sub(1) == 1) { std::free(data); delete refcount; }
        data = nullptr; refcount = nullptr;
    }
    bool empty() const { return data == nullptr || (size_t)w * h == 0; }

    void* data = nullptr;
    std::atomic<int>* refcount = nullptr;
    size_t elemsize = 0;
    int elempack = 0;
    int w = 0, h = 0, c = 0;
};

}  // namespace ncnn
