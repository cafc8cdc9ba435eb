// ORACLE BUILD RECIPE - TEST INFRASTRUCTURE ONLY.
// C entry points around the reference's OWN `class RIFE` (src/rife.h:11-52, compiled unmodified from /root/reference/src/rife.cpp by
// oracle/refbuild/Makefile) so that tests can drive it with ctypes: the role src/main.cpp:825-827, 332, 360 plays in the reference.
// gpuid is always -1: `vkdev == 0`, RIFE::process dispatches to process_cpu / process_v4_cpu (src/rife.cpp:383-393).
#include <stdint.h>
#include <string.h>

#include <string>

#include "rife.h"   // the reference's header, found through -I/root/reference/src

extern "C" {

void* ref_rife_create(int tta_mode, int tta_temporal_mode, int uhd_mode, int num_threads, int rife_v2, int rife_v4) {
    return new RIFE(-1, tta_mode != 0, tta_temporal_mode != 0, uhd_mode != 0, num_threads, rife_v2 != 0, rife_v4 != 0);
}

int ref_rife_load(void* r, const char* modeldir) { return ((RIFE*)r)->load(std::string(modeldir)); }

// in0 / in1 / out: tightly packed u8 RGB, w x h (the carriers of src/main.cpp:187 and :332)
int ref_rife_process(void* r, const uint8_t* in0, const uint8_t* in1, int w, int h, float timestep, uint8_t* out) {
    ncnn::Mat in0image(w, h, (void*)in0, (size_t)3, 3);
    ncnn::Mat in1image(w, h, (void*)in1, (size_t)3, 3);
    ncnn::Mat outimage(w, h, (void*)out, (size_t)3, 3);
    int ret = ((const RIFE*)r)->process(in0image, in1image, timestep, outimage);
    // timestep 0 / 1 rebinds outimage to an input instead of writing (src/rife.cpp:1216-1226, 3206-3216)
    if (ret == 0 && outimage.data != (void*)out) memcpy(out, outimage.data, (size_t)w * h * 3);
    return ret;
}

void ref_rife_destroy(void* r) { delete (RIFE*)r; }

}
