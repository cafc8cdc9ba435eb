/* librife_hip — C-ABI of the MI355X-native RIFE engine.
 *
 * This is the drop-in boundary for the reference's hot path `RIFE::load()` / `RIFE::process()`
 * (nihui/rife-ncnn-vulkan src/rife.h:11-52, src/rife.cpp:127-379, 381-1212, 2462-3202): every entry point
 * below replaces one member of that class.  Plain pointers and sizes only; no C++/torch types cross it.
 * The C++ class `RIFE` in rife-ncnn-vulkan_amd/csrc/rife.h (same name, same signatures as the reference)
 * is a thin shim over these calls, and INTEGRATION.md shows the binding a maintainer of the reference adds.
 *
 * Error convention (reference: `int` return, 0 = ok, src/rife.cpp:378,1211,3201): 0 on success, negative
 * on failure; rife_hip_last_error() returns a thread-local message.  Nothing throws across this boundary.
 * There is NO CPU fallback: every compute entry point fails (-RIFE_HIP_ENODEV) when no HIP device exists.
 */
#ifndef RIFE_HIP_H
#define RIFE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RIFE_HIP_EINVAL   1   /* bad argument */
#define RIFE_HIP_ENODEV   2   /* no such HIP device / no GPU */
#define RIFE_HIP_EIO      3   /* model file missing or malformed */
#define RIFE_HIP_EMODEL   4   /* graph is not a supported RIFE family */
#define RIFE_HIP_EHIP     5   /* HIP runtime error */
#define RIFE_HIP_ENOSYS   6   /* mode not implemented */

typedef struct rife_hip rife_hip_t;

/* ncnn::get_gpu_count() as used by the reference CLI (src/main.cpp:792-802). */
int rife_hip_device_count(void);

/* RIFE::RIFE(gpuid, tta_mode, tta_temporal_mode, uhd_mode, num_threads, rife_v2, rife_v4)  src/rife.cpp:27-47.
 * gpuid = HIP device ordinal (the reference's -1 = CPU device is not served by this library). */
rife_hip_t* rife_hip_create(int gpuid, int tta_mode, int tta_temporal_mode, int uhd_mode, int num_threads,
                            int rife_v2, int rife_v4);

/* RIFE::~RIFE()  src/rife.cpp:49-78 */
void rife_hip_destroy(rife_hip_t* r);

/* RIFE::load(modeldir)  src/rife.cpp:127-379 — reads <modeldir>/flownet.{param,bin} (+ contextnet, fusionnet
 * unless rife_v4), validates the topology against the compiled-in schedules, packs and uploads the weights. */
int rife_hip_load(rife_hip_t* r, const char* modeldir);

/* RIFE::process(in0image, in1image, timestep, outimage)  src/rife.cpp:381-1212 / 2462-3202.
 * Host buffers: tightly packed u8 HWC RGB, w*h*3 bytes each (the ncnn::Mat the CLI builds, src/main.cpp:187,332).
 * Re-entrant on a const object from several host threads (src/main.cpp:860-863). */
int rife_hip_process(const rife_hip_t* r, const uint8_t* in0_rgb, const uint8_t* in1_rgb, int w, int h,
                     float timestep, uint8_t* out_rgb);

/* Optional throughput entry point: n independent pairs from host memory in one call, spread over internal
 * streams so that the copies of one pair overlap the kernels of the others - what the reference gets from its proc threads
 * (src/main.cpp:849-866).  Same pixels as n rife_hip_process() calls; timestep 0 / 1 entries are copies.  A host frame that
 * appears in several pairs of the batch (in0[i + 1] == in1[i] in a sequence) is uploaded once. */
int rife_hip_process_batch(const rife_hip_t* r, int n, const uint8_t* const* in0_rgb, const uint8_t* const* in1_rgb, const float* timestep,
                           uint8_t* const* out_rgb, int w, int h);

/* Same, with all three frames already resident in device memory (what ncnn's VkMat path does internally between
 * record_clone and submit, src/rife.cpp:2522-2530,3176-3186).  Work is enqueued on `hip_stream` (a hipStream_t;
 * NULL = the engine's own stream) and the call returns without synchronising when a stream is given. */
int rife_hip_process_device(const rife_hip_t* r, const void* d_in0_rgb, const void* d_in1_rgb, int w, int h,
                            float timestep, void* d_out_rgb, void* hip_stream);

/* Streams that own a PART of the chip.  The reference keeps a GPU busy with several proc threads per device (-j, src/main.cpp:849-866), whose
 * command buffers share the whole device.  On MI355X (256 compute units in 8 XCDs, an L2 per XCD) the pairs in flight interfere less when each has
 * compute units of its own: rife_hip_stream_create returns a hipStream_t restricted to the compute units i with i % nparts == part
 * (hipExtStreamCreateWithCUMask), and rife_hip_process_device on such a stream sizes its persistent kernels for that part.  Frames are the same
 * bytes on every stream.  nparts = 1: an ordinary stream.  Measured (1920x1080, resident frames): 4 parts x 1 caller each 1,690 frames/s against
 * 1,450 - 1,590 from 3 ordinary streams; 3840x2160: 2 parts x 2 callers each 473 - 479 against 465 - 469.  Destroy with rife_hip_stream_destroy (after the work on it has finished), or
 * let rife_hip_destroy do it.
 * Synchronisation semantics: hipExtStreamCreateWithCUMask takes no flags, so a stream with nparts > 1 is a BLOCKING stream - it synchronises implicitly
 * with the legacy NULL stream (work enqueued on stream 0, plain hipMemcpy / hipMemset), while nparts = 1 returns a hipStreamNonBlocking stream.  Keep
 * NULL-stream work out of the process while partition streams carry pairs, or the "independent" parts serialise behind it; the engine itself never
 * enqueues on the NULL stream. */
int rife_hip_stream_create(const rife_hip_t* r, int part, int nparts, void** hip_stream);
int rife_hip_stream_destroy(const rife_hip_t* r, void* hip_stream);

/* n resident pairs in one call (SURVEY.md §8f-2 "batch >= 2 pairs per launch for the coarse blocks"): the pairs run two by two in lockstep on
 * internal streams, the trunk layers of the coarse IFBlocks as ONE launch per layer for both pairs of a group (models/rife-v4.6/flownet.param:14-42,
 * 66-94), forked from and joined into `hip_stream` with events: like rife_hip_process_device the call returns without synchronising when a
 * stream is given, and work enqueued on `hip_stream` afterwards sees all n results.  NULL synchronises before returning.  Same bytes as n
 * rife_hip_process_device calls; model families without the lockstep schedule (and -x / -z) run the pairs one after the other. */
int rife_hip_process_device_batch(const rife_hip_t* r, int n, const void* const* d_in0_rgb, const void* const* d_in1_rgb, const float* timestep,
                                  void* const* d_out_rgb, int w, int h, void* hip_stream);

/* Stream mode (absent in the reference, whose tasks upload both frames every time, src/main.cpp:315-334,
 * src/rife.cpp:2490-2530): in a frame sequence every frame is the second frame of one pair and the first frame of the next,
 * and with -n > 2N it serves several timesteps, so a caller can upload a frame ONCE and interpolate between resident frames.
 * A frame belongs to the device of the engine that uploaded it and may be used by any thread and by several calls at once
 * (it is read-only); release it after the last call that uses it has returned.  Pixels are identical to rife_hip_process(). */
typedef struct rife_hip_frame rife_hip_frame_t;
int rife_hip_frame_upload(const rife_hip_t* r, const uint8_t* rgb, int w, int h, rife_hip_frame_t** frame);
int rife_hip_process_frames(const rife_hip_t* r, const rife_hip_frame_t* frame0, const rife_hip_frame_t* frame1, float timestep,
                            uint8_t* out_rgb);
void rife_hip_frame_release(rife_hip_frame_t* frame);

const char* rife_hip_last_error(void);

/* ---- measurement hooks (bench.py / profiles) ---------------------------------------------------------------
 * When enabled, every launch of the engine's kernels is bracketed by HIP events on the launch stream; read-out
 * synchronises and returns per-kernel-class totals.  `names` receives '\n'-separated class names. */
int rife_hip_profile_enable(rife_hip_t* r, int on);
int rife_hip_profile_read(rife_hip_t* r, char* names, size_t names_cap, double* total_ms, long long* launches,
                          double* flops, int max_classes);   /* returns number of classes */

/* Dry run of the generic graph executor's loader on one ncnn .param file (no GPU needed): 0 if every layer of the graph has a
 * kernel, RIFE_HIP_EMODEL with the offending layer in rife_hip_last_error() otherwise.  The v1 family (models/rife, rife-HD,
 * rife-UHD, rife-anime) is executed from its .param layer by layer, like ncnn::Net does for every model (src/rife.cpp:112-121). */
int rife_hip_graph_check(const char* param_path_without_extension);
/* Structural hash of the sub-graph that produces blob `blob` of an ncnn .param file (layer types, parameters, topology; no names,
 * no weights) - what rife_hip_load() compares with the compiled-in constants of csrc/model_hashes.h to prove that a model directory
 * holds the graph a fused schedule was written for (the reference's models/rife-v4.6/flownet.param etc.).  CPU only. */
int rife_hip_param_hash(const char* param_path, const char* blob, uint64_t* hash_out);

/* ---- page-locked host frames (optional) -------------------------------------------------------------------------------
 * rife_hip_process() takes any host pointer, like RIFE::process() takes any ncnn::Mat (src/main.cpp:187, 332).  Copies from / to
 * pageable memory are staged by the HIP runtime and hold the calling thread for their whole duration; from page-locked memory they
 * are asynchronous DMA at PCIe rate, so that with two caller threads (the reference's default -j 1:2:2) one pair's copies run
 * under the other pair's kernels.  A caller without HIP headers can get such memory here: allocate frames with
 * rife_hip_host_alloc() (returns NULL on failure), or page-lock buffers it already owns for as long as it keeps them
 * (rife_hip_host_register / rife_hip_host_unregister; the range must stay mapped in between).  Pixels are identical either way. */
void* rife_hip_host_alloc(size_t bytes);
void rife_hip_host_free(void* p);
int rife_hip_host_register(void* p, size_t bytes);
int rife_hip_host_unregister(void* p);

#ifdef __cplusplus
}
#endif
#endif /* RIFE_HIP_H */
