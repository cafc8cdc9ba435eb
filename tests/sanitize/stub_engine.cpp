// TEST INFRASTRUCTURE ONLY - never linked into a product library.
// A stand-in for librife_hip.so behind the same C-ABI (include/rife_hip.h) for the SANITIZER builds of the host side (make -C rife-ncnn-vulkan_amd/csrc sanitize):
// csrc/main.cpp (the 3-stage pipeline, bounded queues, frame cache, `-g` replicas: src/main.cpp:248-436, 819-904 is the threading model it mirrors),
// csrc/rife.cpp (the class shim), csrc/jpeg_codec.h and the band-parallel PNG writer run under ASan + UBSan and under TSan on a box WITHOUT a GPU, with the
// entry points the CLI uses answered by plain host code: "interpolation" = the rounded mean of the two frames, frames "resident" in heap copies.  It keeps the
// thread-safety contract of the real engine (process_frames is const and re-entrant, last_error is thread local) and nothing else: no arithmetic of the
// reference is restated here and no parity claim rests on it.
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rife_hip.h"

struct rife_hip { bool loaded = false; int gpuid = 0; };
struct rife_hip_frame { std::vector<uint8_t> px; int w = 0, h = 0; };

static thread_local std::string g_err;
static int fail(int code, const char* msg) { g_err = msg; return -code; }

extern "C" {

int rife_hip_device_count(void) { return 2; }      // two "devices": `-g 0,1` and `-g 0,0` both exercise the replica code

rife_hip_t* rife_hip_create(int gpuid, int, int, int, int, int, int) {
    if (gpuid < 0 || gpuid >= 2) { g_err = "stub: no such device"; return nullptr; }
    rife_hip* e = new rife_hip; e->gpuid = gpuid; return e;
}
void rife_hip_destroy(rife_hip_t* r) { delete r; }
int rife_hip_load(rife_hip_t* r, const char* modeldir) {
    if (!r || !modeldir) return fail(RIFE_HIP_EINVAL, "stub: null argument");
    r->loaded = true; return 0;
}
static void blend(const uint8_t* a, const uint8_t* b, size_t n, float t, uint8_t* out) {
    for (size_t i = 0; i < n; i++) out[i] = (uint8_t)((1.f - t) * a[i] + t * b[i] + 0.5f);
}
int rife_hip_process(const rife_hip_t* r, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep, uint8_t* out) {
    if (!r || !r->loaded || !in0 || !in1 || !out || w <= 0 || h <= 0) return fail(RIFE_HIP_EINVAL, "stub: bad argument");
    blend(in0, in1, (size_t)w * h * 3, timestep, out); return 0;
}
int rife_hip_frame_upload(const rife_hip_t* r, const uint8_t* rgb, int w, int h, rife_hip_frame_t** frame) {
    if (!r || !rgb || !frame || w <= 0 || h <= 0) return fail(RIFE_HIP_EINVAL, "stub: bad argument");
    rife_hip_frame* f = new rife_hip_frame; f->w = w; f->h = h; f->px.assign(rgb, rgb + (size_t)w * h * 3); *frame = f; return 0;
}
int rife_hip_process_frames(const rife_hip_t* r, const rife_hip_frame_t* f0, const rife_hip_frame_t* f1, float timestep, uint8_t* out) {
    if (!r || !r->loaded || !f0 || !f1 || !out || f0->w != f1->w || f0->h != f1->h) return fail(RIFE_HIP_EINVAL, "stub: bad argument");
    blend(f0->px.data(), f1->px.data(), f0->px.size(), timestep, out); return 0;
}
void rife_hip_frame_release(rife_hip_frame_t* frame) { delete frame; }
const char* rife_hip_last_error(void) { return g_err.c_str(); }

}  // extern "C"
