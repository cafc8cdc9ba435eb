// RIFE — same class name, constructor and member signatures as the reference (src/rife.h:11-52), implemented on
// librife_hip.so (include/rife_hip.h) instead of ncnn + Vulkan.  `src/main.cpp` of the reference compiles against
// this header unchanged apart from the include path of the Mat type.
#ifndef RIFE_H
#define RIFE_H

#include <string>

#include "ncnn_mat.h"

struct rife_hip;
struct rife_hip_frame;

class RIFE
{
public:
    RIFE(int gpuid, bool tta_mode = false, bool tta_temporal_mode = false, bool uhd_mode = false, int num_threads = 1, bool rife_v2 = false, bool rife_v4 = false);
    ~RIFE();

    int load(const std::string& modeldir);

    int process(const ncnn::Mat& in0image, const ncnn::Mat& in1image, float timestep, ncnn::Mat& outimage) const;

    // The reference exposes its four back-ends publicly (src/rife.h:25-29).  Here the HIP engine is the only one:
    // process_v4 == process for rife_v4 objects; the *_cpu entry points report an error (no CPU path in this build).
    int process_cpu(const ncnn::Mat& in0image, const ncnn::Mat& in1image, float timestep, ncnn::Mat& outimage) const;
    int process_v4(const ncnn::Mat& in0image, const ncnn::Mat& in1image, float timestep, ncnn::Mat& outimage) const;
    int process_v4_cpu(const ncnn::Mat& in0image, const ncnn::Mat& in1image, float timestep, ncnn::Mat& outimage) const;

    // Extension, not in the reference (include/rife_hip.h "stream mode"): a frame uploaded once and used by several process()
    // calls - consecutive pairs share a frame, and every timestep of a pair shares both.  Same pixels as the host-buffer call.
    rife_hip_frame* upload(const ncnn::Mat& image) const;
    int process(const rife_hip_frame* frame0, const rife_hip_frame* frame1, float timestep, ncnn::Mat& outimage) const;
    static void release(rife_hip_frame* frame);

private:
    RIFE(const RIFE&);
    RIFE& operator=(const RIFE&);
    rife_hip* engine;
    int gpuid;
    bool rife_v4;
};

#endif // RIFE_H
