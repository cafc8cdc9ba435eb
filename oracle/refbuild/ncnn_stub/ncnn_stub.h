// ORACLE BUILD RECIPE - TEST INFRASTRUCTURE ONLY.  Not part of the shipped product.
//
// An ncnn LOOK-ALIKE, just wide enough that the reference's own src/rife.cpp and src/warp.cpp compile UNMODIFIED, from where they lie under
// /root/reference, into oracle/_ref/libref_rife.so (oracle/refbuild/Makefile).  Tencent/ncnn itself is an un-vendored submodule of the
// reference (`.gitmodules:1-3`, src/ncnn/ is empty), so this file re-declares the slice of its public API that the two sources use:
//   * ncnn::Mat            - implemented here for real: ref-counted planar fp32 tensor with ncnn's cstep rule, from_pixels / to_pixels,
//                            channel() / row() / fill() / clone() (everything src/rife.cpp:1214-2460, 3204-4401 and src/warp.cpp:96-168 touch);
//   * ncnn::Net / Extractor / Layer / ParamDict / Option / create_layer / DEFINE_LAYER_CREATOR
//                          - implemented in ncnn_stub.cpp ON TOP OF the graph interpreter of oracle/ncnn_graph.cpp (the same layer arithmetic the
//                            restated oracle uses), with `rife.Warp` dispatched to the reference's OWN Warp::forward through register_custom_layer;
//   * the Vulkan side      - VulkanDevice, VkAllocator, VkMat, VkCompute, Pipeline, compile_spirv_module, get_gpu_device: declarations whose
//                            bodies abort().  With gpuid = -1 (`vkdev == 0`) the reference never executes one of them (src/rife.cpp:383-393).
// What this buys: the reference's 4,400 lines of orchestration (padding, the *255+0.5 flat crop, TTA index algebra, temporal merges, UHD path, slice,
// blob binding order) and its Warp::forward are THE REFERENCE'S OWN COMPILED CODE in tests/test_ref_build.py; what stays restated is the
// arithmetic of the ncnn built-in layers.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <string>
#include <vector>

namespace ncnn {

class Allocator;
class VkAllocator {};
class VkCompute;
class Pipeline;
class Option;

[[noreturn]] inline void stub_vulkan_called(const char* what) {
    fprintf(stderr, "ncnn_stub: %s called - the Vulkan side is declared only (gpuid must be -1)\n", what);
    abort();
}

// ------------------------------------------------------------------------------------------------ Mat
// ncnn::Mat semantics: dims 1..3, planar channels, channel stride cstep = alignSize(w * h * elemsize, 16) / elemsize,
// shared ownership through a reference count, non-owning when constructed around external memory.
class Mat {
public:
    enum PixelType {
        PIXEL_CONVERT_SHIFT = 16,
        PIXEL_FORMAT_MASK = 0x0000ffff,
        PIXEL_CONVERT_MASK = 0xffff0000,
        PIXEL_RGB = 1,
        PIXEL_BGR = 2,
        PIXEL_RGB2BGR = PIXEL_RGB | (PIXEL_BGR << PIXEL_CONVERT_SHIFT),
        PIXEL_BGR2RGB = PIXEL_BGR | (PIXEL_RGB << PIXEL_CONVERT_SHIFT),
    };

    Mat() { reset(); }
    Mat(int _w, size_t _elemsize = 4u, Allocator* = 0) { reset(); create(_w, _elemsize); }
    Mat(int _w, int _h, size_t _elemsize = 4u, Allocator* = 0) { reset(); create(_w, _h, _elemsize); }
    Mat(int _w, int _h, int _c, size_t _elemsize = 4u, Allocator* = 0) { reset(); create(_w, _h, _c, _elemsize); }
    // packed forms (the image carriers of src/main.cpp:187, 332: elemsize 3, elempack 3 = one RGB pixel per element)
    Mat(int _w, int _h, size_t _elemsize, int _elempack, Allocator* = 0) { reset(); create(_w, _h, _elemsize, _elempack); }
    Mat(int _w, int _h, void* _data, size_t _elemsize, int _elempack, Allocator* = 0) {
        reset();
        data = _data; elemsize = _elemsize; elempack = _elempack; dims = 2; w = _w; h = _h; d = 1; c = 1;
        cstep = (size_t)w * h;
    }
    Mat(int _w, int _h, void* _data, size_t _elemsize = 4u, Allocator* = 0) {
        reset();
        data = _data; elemsize = _elemsize; elempack = 1; dims = 2; w = _w; h = _h; d = 1; c = 1;
        cstep = (size_t)w * h;
    }
    Mat(const Mat& m) { copy_fields(m); addref(); }
    ~Mat() { release(); }
    Mat& operator=(const Mat& m) {
        if (this == &m) return *this;
        if (m.refcount) __sync_fetch_and_add(m.refcount, 1);
        release();
        copy_fields(m);
        return *this;
    }

    void fill(float v) { fill<float>(v); }
    template <typename T>
    void fill(T v) {
        for (int q = 0; q < c; q++) {
            T* p = (T*)((unsigned char*)data + cstep * q * elemsize);
            const size_t n = (size_t)w * h * d;
            for (size_t i = 0; i < n; i++) p[i] = v;
        }
    }
    Mat clone(Allocator* = 0) const {
        if (empty()) return Mat();
        Mat m;
        if (dims == 1) m.create(w, elemsize, elempack);
        else if (dims == 2) m.create(w, h, elemsize, elempack);
        else m.create(w, h, c, elemsize, elempack);
        if (total() > 0) memcpy(m.data, data, total() * elemsize);
        return m;
    }

    void create(int _w, size_t _elemsize = 4u, Allocator* = 0) { create_nd(1, _w, 1, 1, _elemsize, 1); }
    void create(int _w, int _h, size_t _elemsize = 4u, Allocator* = 0) { create_nd(2, _w, _h, 1, _elemsize, 1); }
    void create(int _w, int _h, int _c, size_t _elemsize = 4u, Allocator* = 0) { create_nd(3, _w, _h, _c, _elemsize, 1); }
    void create(int _w, size_t _elemsize, int _elempack, Allocator* = 0) { create_nd(1, _w, 1, 1, _elemsize, _elempack); }
    void create(int _w, int _h, size_t _elemsize, int _elempack, Allocator* = 0) { create_nd(2, _w, _h, 1, _elemsize, _elempack); }
    void create(int _w, int _h, int _c, size_t _elemsize, int _elempack, Allocator* = 0) { create_nd(3, _w, _h, _c, _elemsize, _elempack); }

    void addref() { if (refcount) __sync_fetch_and_add(refcount, 1); }
    void release() {
        if (refcount && __sync_fetch_and_add(refcount, -1) == 1) free(alloc_base);
        reset();
    }
    bool empty() const { return data == 0 || total() == 0; }
    size_t total() const { return cstep * c; }

    Mat channel(int q) { return plane_view(q); }
    const Mat channel(int q) const { return plane_view(q); }
    float* row(int y) { return (float*)((unsigned char*)data + (size_t)w * y * elemsize); }
    const float* row(int y) const { return (const float*)((unsigned char*)data + (size_t)w * y * elemsize); }
    template <typename T> T* row(int y) { return (T*)((unsigned char*)data + (size_t)w * y * elemsize); }
    template <typename T> const T* row(int y) const { return (const T*)((unsigned char*)data + (size_t)w * y * elemsize); }
    template <typename T> operator T*() { return (T*)data; }
    template <typename T> operator const T*() const { return (const T*)data; }
    float& operator[](size_t i) { return ((float*)data)[i]; }
    const float& operator[](size_t i) const { return ((const float*)data)[i]; }

    // u8 interleaved -> fp32 planar, values 0..255 (ncnn Mat::from_pixels)
    static Mat from_pixels(const unsigned char* pixels, int type, int _w, int _h, Allocator* = 0) {
        Mat m(_w, _h, 3);
        const int from = type & PIXEL_FORMAT_MASK;
        const bool swap = (type & PIXEL_CONVERT_MASK) != 0 && ((type >> PIXEL_CONVERT_SHIFT) != from);
        float* p0 = m.channel(swap ? 2 : 0);
        float* p1 = m.channel(1);
        float* p2 = m.channel(swap ? 0 : 2);
        const size_t n = (size_t)_w * _h;
        for (size_t i = 0; i < n; i++) {
            p0[i] = (float)pixels[3 * i];
            p1[i] = (float)pixels[3 * i + 1];
            p2[i] = (float)pixels[3 * i + 2];
        }
        return m;
    }
    // fp32 planar -> u8 interleaved: (unsigned char) min(max((int)v, 0), 255) per element (ncnn SATURATE_CAST_UCHAR)
    void to_pixels(unsigned char* pixels, int type) const {
        const int from = type & PIXEL_FORMAT_MASK;
        const bool swap = (type & PIXEL_CONVERT_MASK) != 0 && ((type >> PIXEL_CONVERT_SHIFT) != from);
        const float* p0 = channel(swap ? 2 : 0);
        const float* p1 = channel(1);
        const float* p2 = channel(swap ? 0 : 2);
        const size_t n = (size_t)w * h;
        for (size_t i = 0; i < n; i++) {
            pixels[3 * i] = sat(p0[i]);
            pixels[3 * i + 1] = sat(p1[i]);
            pixels[3 * i + 2] = sat(p2[i]);
        }
    }

    void* data;
    int* refcount;
    size_t elemsize;
    int elempack;
    Allocator* allocator;
    int dims;
    int w, h, d, c;
    size_t cstep;

private:
    void* alloc_base;
    static unsigned char sat(float v) { return (unsigned char)std::min(std::max((int)v, 0), 255); }
    void reset() {
        data = 0; refcount = 0; elemsize = 0; elempack = 0; allocator = 0; dims = 0; w = h = d = c = 0; cstep = 0; alloc_base = 0;
    }
    void copy_fields(const Mat& m) {
        data = m.data; refcount = m.refcount; elemsize = m.elemsize; elempack = m.elempack; allocator = m.allocator;
        dims = m.dims; w = m.w; h = m.h; d = m.d; c = m.c; cstep = m.cstep; alloc_base = m.alloc_base;
    }
    void create_nd(int _dims, int _w, int _h, int _c, size_t _elemsize, int _elempack) {
        if (dims == _dims && w == _w && h == _h && c == _c && elemsize == _elemsize && elempack == _elempack && refcount && *refcount == 1) return;
        release();
        elemsize = _elemsize; elempack = _elempack; dims = _dims; w = _w; h = _h; d = 1; c = _c;
        const size_t plane = (size_t)w * h * elemsize;
        cstep = _dims == 3 ? ((plane + 15) & ~(size_t)15) / elemsize : (size_t)w * h;
        const size_t bytes = (total() * elemsize + 3) & ~(size_t)3;
        if (bytes == 0) return;
        // one allocation: [pad to 64][payload][refcount]
        alloc_base = malloc(bytes + 64 + sizeof(int));
        if (!alloc_base) { reset(); return; }
        data = (void*)(((uintptr_t)alloc_base + 63) & ~(uintptr_t)63);
        refcount = (int*)((unsigned char*)data + bytes);
        *refcount = 1;
    }
    Mat plane_view(int q) const {
        Mat m;
        m.data = (unsigned char*)data + cstep * q * elemsize;
        m.elemsize = elemsize; m.elempack = elempack; m.dims = dims - 1 > 0 ? dims - 1 : 1;
        m.w = w; m.h = h; m.d = 1; m.c = 1; m.cstep = (size_t)w * h;
        return m;
    }
};

// ------------------------------------------------------------------------------------------------ Vulkan side: declarations only
class VulkanDevice {
public:
    VkAllocator* acquire_blob_allocator() const { stub_vulkan_called("VulkanDevice::acquire_blob_allocator"); }
    VkAllocator* acquire_staging_allocator() const { stub_vulkan_called("VulkanDevice::acquire_staging_allocator"); }
    void reclaim_blob_allocator(VkAllocator*) const { stub_vulkan_called("VulkanDevice::reclaim_blob_allocator"); }
    void reclaim_staging_allocator(VkAllocator*) const { stub_vulkan_called("VulkanDevice::reclaim_staging_allocator"); }
};
inline VulkanDevice* get_gpu_device(int) { stub_vulkan_called("get_gpu_device"); }

class VkMat {
public:
    VkMat() : elemsize(0), elempack(0), dims(0), w(0), h(0), d(0), c(0), cstep(0) {}
    template <typename... A> void create(A&&...) { stub_vulkan_called("VkMat::create"); }
    void release() {}
    bool empty() const { return true; }
    size_t elemsize;
    int elempack;
    int dims;
    int w, h, d, c;
    size_t cstep;
};

union vk_specialization_type { int i; float f; uint32_t u32; };
union vk_constant_type { int i; float f; };

class Pipeline {
public:
    explicit Pipeline(const VulkanDevice*) {}
    virtual ~Pipeline() {}
    void set_optimal_local_size_xyz(int = 4, int = 4, int = 4) {}
    int create(const uint32_t*, size_t, const std::vector<vk_specialization_type>&) { stub_vulkan_called("Pipeline::create"); }
};

class VkCompute {
public:
    explicit VkCompute(const VulkanDevice*) {}
    template <typename... A> void record_clone(A&&...) { stub_vulkan_called("VkCompute::record_clone"); }
    template <typename... A> void record_pipeline(A&&...) { stub_vulkan_called("VkCompute::record_pipeline"); }
    int submit_and_wait() { stub_vulkan_called("VkCompute::submit_and_wait"); }
};

class Mutex {};
class MutexLockGuard {
public:
    explicit MutexLockGuard(Mutex&) {}
};

// ------------------------------------------------------------------------------------------------ Option / ParamDict / Layer
class Option {
public:
    Option()
        : lightmode(true), num_threads(1), blob_allocator(0), workspace_allocator(0), blob_vkallocator(0), workspace_vkallocator(0),
          staging_vkallocator(0), use_winograd_convolution(true), use_sgemm_convolution(true), use_int8_inference(true),
          use_vulkan_compute(false), use_bf16_storage(false), use_fp16_packed(true), use_fp16_storage(true), use_fp16_arithmetic(true),
          use_int8_packed(true), use_int8_storage(true), use_int8_arithmetic(false), use_packing_layout(true), use_shader_pack8(false) {}
    bool lightmode;
    int num_threads;
    Allocator* blob_allocator;
    Allocator* workspace_allocator;
    VkAllocator* blob_vkallocator;
    VkAllocator* workspace_vkallocator;
    VkAllocator* staging_vkallocator;
    bool use_winograd_convolution, use_sgemm_convolution, use_int8_inference, use_vulkan_compute, use_bf16_storage, use_fp16_packed,
        use_fp16_storage, use_fp16_arithmetic, use_int8_packed, use_int8_storage, use_int8_arithmetic, use_packing_layout, use_shader_pack8;
};

inline int compile_spirv_module(const char*, int, const Option&, std::vector<uint32_t>&) { stub_vulkan_called("compile_spirv_module"); }

class ParamDict {
public:
    ParamDict() { memset(kind, 0, sizeof(kind)); }
    void set(int id, int v) { kind[id] = 1; iv[id] = v; }
    void set(int id, float v) { kind[id] = 2; fv[id] = v; }
    void set(int id, const Mat& v) { kind[id] = 3; mv[id] = v; }
    int get(int id, int def) const { return kind[id] == 1 ? iv[id] : kind[id] == 2 ? (int)fv[id] : def; }
    float get(int id, float def) const { return kind[id] == 2 ? fv[id] : kind[id] == 1 ? (float)iv[id] : def; }
    Mat get(int id, const Mat& def) const { return kind[id] == 3 ? mv[id] : def; }

private:
    enum { NP = 32 };
    int kind[NP];
    int iv[NP];
    float fv[NP];
    Mat mv[NP];
};

class ModelBin;

class Layer {
public:
    Layer() : one_blob_only(false), support_inplace(false), support_vulkan(false), vkdev(0) {}
    virtual ~Layer() {}
    virtual int load_param(const ParamDict&) { return 0; }
    virtual int load_model(const ModelBin&) { return 0; }
    virtual int create_pipeline(const Option&) { return 0; }
    virtual int destroy_pipeline(const Option&) { return 0; }
    virtual int forward(const std::vector<Mat>& bottom_blobs, std::vector<Mat>& top_blobs, const Option& opt) const;
    virtual int forward(const Mat& bottom_blob, Mat& top_blob, const Option& opt) const;
    virtual int forward(const std::vector<VkMat>&, std::vector<VkMat>&, VkCompute&, const Option&) const { stub_vulkan_called("Layer::forward(VkMat)"); }
    virtual int forward(const VkMat&, VkMat&, VkCompute&, const Option&) const { stub_vulkan_called("Layer::forward(VkMat)"); }

    bool one_blob_only;
    bool support_inplace;
    bool support_vulkan;
    const VulkanDevice* vkdev;
    std::string type, name;
};

typedef Layer* (*layer_creator_func)(void*);
typedef void (*layer_destroyer_func)(Layer*, void*);
#define DEFINE_LAYER_CREATOR(name) \
    ::ncnn::Layer* name##_layer_creator(void* /*userdata*/) { return new name; }

// the three built-in layer types the reference instantiates by name (src/rife.cpp:297, 309, 321, 337): Interp, BinaryOp, Slice
Layer* create_layer(const char* type);

// ------------------------------------------------------------------------------------------------ Net / Extractor
class NetImpl;
class ExtractorImpl;

class Extractor {
public:
    ~Extractor();
    Extractor(const Extractor&);
    Extractor& operator=(const Extractor&);
    void set_light_mode(bool) {}
    void set_num_threads(int) {}
    void set_blob_vkallocator(VkAllocator*) {}
    void set_workspace_vkallocator(VkAllocator*) {}
    void set_staging_vkallocator(VkAllocator*) {}
    int input(const char* blob_name, const Mat& in);
    int extract(const char* blob_name, Mat& feat);
    int input(const char*, const VkMat&) { stub_vulkan_called("Extractor::input(VkMat)"); }
    int extract(const char*, VkMat&, VkCompute&) { stub_vulkan_called("Extractor::extract(VkMat)"); }

private:
    friend class Net;
    explicit Extractor(const NetImpl* net);
    std::shared_ptr<ExtractorImpl> d;
};

class Net {
public:
    Net();
    ~Net();
    Option opt;
    void set_vulkan_device(const VulkanDevice*) {}
    int register_custom_layer(const char* type, layer_creator_func creator, layer_destroyer_func destroyer = 0, void* userdata = 0);
    int load_param(const char* path);
    int load_model(const char* path);
    Extractor create_extractor() const;

private:
    Net(const Net&);
    Net& operator=(const Net&);
    NetImpl* d;
};

}  // namespace ncnn
