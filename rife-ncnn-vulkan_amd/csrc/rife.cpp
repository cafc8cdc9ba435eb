// RIFE class shim over the C-ABI (see rife.h).  Behaviour kept from the reference:
//   * constructor never fails; an unusable device surfaces at load()/process() (reference: vkdev lookup, rife.cpp:27-47)
//   * process() with timestep 0 / 1 rebinds outimage to an input Mat, sharing its buffer (rife.cpp:395-405, 2470-2480)
//   * return value 0 = ok (rife.cpp:1211, 3201); errors are printed to stderr like ncnn does and returned negative
#include "rife.h"

#include <cstdio>

#include "../../include/rife_hip.h"

RIFE::RIFE(int gpuid_, bool tta_mode, bool tta_temporal_mode, bool uhd_mode, int num_threads, bool rife_v2, bool rife_v4_)
    : engine(0), gpuid(gpuid_), rife_v4(rife_v4_)
{
    if (gpuid >= 0)
        engine = rife_hip_create(gpuid, tta_mode, tta_temporal_mode, uhd_mode, num_threads, rife_v2, rife_v4);
    if (!engine)
        fprintf(stderr, "RIFE: %s\n", gpuid < 0 ? "gpuid -1 (CPU device) is not served by the HIP engine" : rife_hip_last_error());
}

RIFE::~RIFE()
{
    if (engine) rife_hip_destroy(engine);
}

int RIFE::load(const std::string& modeldir)
{
    if (!engine) return -RIFE_HIP_ENODEV;
    int ret = rife_hip_load(engine, modeldir.c_str());
    if (ret) fprintf(stderr, "RIFE::load: %s\n", rife_hip_last_error());
    return ret;
}

int RIFE::process(const ncnn::Mat& in0image, const ncnn::Mat& in1image, float timestep, ncnn::Mat& outimage) const
{
    if (timestep == 0.f)
    {
        outimage = in0image;
        return 0;
    }

    if (timestep == 1.f)
    {
        outimage = in1image;
        return 0;
    }

    if (!engine) return -RIFE_HIP_ENODEV;
    if (in0image.w != in1image.w || in0image.h != in1image.h || outimage.w != in0image.w || outimage.h != in0image.h || !outimage.data)
    {
        fprintf(stderr, "RIFE::process: frame size mismatch\n");
        return -RIFE_HIP_EINVAL;
    }

    int ret = rife_hip_process(engine, (const unsigned char*)in0image.data, (const unsigned char*)in1image.data, in0image.w, in0image.h, timestep, (unsigned char*)outimage.data);
    if (ret) fprintf(stderr, "RIFE::process: %s\n", rife_hip_last_error());
    return ret;
}

rife_hip_frame* RIFE::upload(const ncnn::Mat& image) const
{
    if (!engine || !image.data) return 0;
    rife_hip_frame* f = 0;
    if (rife_hip_frame_upload(engine, (const unsigned char*)image.data, image.w, image.h, &f))
        fprintf(stderr, "RIFE::upload: %s\n", rife_hip_last_error());
    return f;
}

int RIFE::process(const rife_hip_frame* frame0, const rife_hip_frame* frame1, float timestep, ncnn::Mat& outimage) const
{
    if (!engine) return -RIFE_HIP_ENODEV;
    if (!outimage.data) return -RIFE_HIP_EINVAL;
    // the engine checks the two frames against each other; outimage must be the caller's w x h x 3 buffer as in main.cpp:332
    int ret = rife_hip_process_frames(engine, frame0, frame1, timestep, (unsigned char*)outimage.data);
    if (ret) fprintf(stderr, "RIFE::process: %s\n", rife_hip_last_error());
    return ret;
}

void RIFE::release(rife_hip_frame* frame)
{
    rife_hip_frame_release(frame);
}

int RIFE::process_v4(const ncnn::Mat& in0image, const ncnn::Mat& in1image, float timestep, ncnn::Mat& outimage) const
{
    return process(in0image, in1image, timestep, outimage);
}

int RIFE::process_cpu(const ncnn::Mat&, const ncnn::Mat&, float, ncnn::Mat&) const
{
    fprintf(stderr, "RIFE::process_cpu: this build has no CPU path\n");
    return -RIFE_HIP_ENOSYS;
}

int RIFE::process_v4_cpu(const ncnn::Mat&, const ncnn::Mat&, float, ncnn::Mat&) const
{
    fprintf(stderr, "RIFE::process_v4_cpu: this build has no CPU path\n");
    return -RIFE_HIP_ENOSYS;
}
