// ORACLE — TEST INFRASTRUCTURE ONLY (see ncnn_graph.h header).
// PINNED since round 4: tests/test_ref_build.py holds every function of this file bit for bit against the reference's OWN src/rife.cpp + src/warp.cpp,
// compiled unmodified against an ncnn look-alike (oracle/refbuild/ -> oracle/_ref/libref_rife.so), in all families and modes.  What stays
// "parity unpinned" is the arithmetic of the ncnn built-in layers underneath (ncnn_graph.cpp, conv_cpu.cpp), which both sides share.
//
// CPU restatement of the reference's `-g -1` path:
//   RIFE::load            reference src/rife.cpp:127-379  (only what the CPU path needs)
//   RIFE::process_v4_cpu  reference src/rife.cpp:3204-4401 (plain 4146-4389, temporal TTA, spatial TTA 3246-4145)
//   RIFE::process_cpu     reference src/rife.cpp:1214-2460 (rife_v2 plain branch 2139-2457 incl. UHD and temporal TTA)
// The nets are the reference's own ncnn graphs executed by the interpreter in ncnn_graph.cpp.
// Exposed through a small C interface so tests / bench.py can drive it with ctypes.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "ncnn_graph.h"

using oracle::Extractor;
using oracle::Mat;
using oracle::Net;

namespace {

struct RifeOracle {
    Net flownet, contextnet, fusionnet;
    bool tta_mode = false, tta_temporal_mode = false, uhd_mode = false, rife_v2 = false, rife_v4 = false;
    // 0: literal reference CPU behaviour (flat h*w copy that ignores the padded row pitch, rife.cpp:4375-4387,
    //    SURVEY App. F-1);  1: crop with the padded pitch like the GPU shader (rife_postproc.comp:42).
    //    Identical whenever w % 32 == 0.
    int gpu_crop = 0;
    int num_threads = 1;
    std::string err;
};

// Mat::from_pixels(PIXEL_RGB) followed by the x*(1/255.f) + zero-pad loops (rife.cpp:4152-4211)
Mat preproc_pad(const uint8_t* px, int w, int h, int wp, int hp) {
    Mat m(wp, hp, 3);
    for (int q = 0; q < 3; q++) {
        float* o = m.channel(q);
        for (int i = 0; i < hp; i++)
            for (int j = 0; j < wp; j++) {
                float v = 0.f;
                if (i < h && j < w) v = (float)px[((size_t)i * w + j) * 3 + q] * (1 / 255.f);
                o[(size_t)i * wp + j] = v;
            }
    }
    return m;
}

Mat filled(int w, int h, float v) {
    Mat m(w, h, 1);
    for (size_t i = 0; i < m.total(); i++) m.data[i] = v;
    return m;
}

// Mat::to_pixels(PIXEL_RGB): (unsigned char) min(max((int)v, 0), 255)  
inline uint8_t sat_u8(float v) {
    int i = (int)v;
    return (uint8_t)std::min(std::max(i, 0), 255);
}

// the 8 TTA orientations of a W x H plane (reference rife.cpp:3319-3413)
//   returns index into orientation ti's buffer for source pixel (i=row, j=col)
inline size_t tta_index(int ti, int i, int j, int W, int H) {
    switch (ti) {
        case 0: return (size_t)i * W + j;
        case 1: return (size_t)i * W + (W - 1 - j);
        case 2: return (size_t)(H - 1 - i) * W + (W - 1 - j);
        case 3: return (size_t)(H - 1 - i) * W + j;
        case 4: return (size_t)j * H + i;
        case 5: return (size_t)j * H + (H - 1 - i);
        case 6: return (size_t)(W - 1 - j) * H + (H - 1 - i);
        default: return (size_t)(W - 1 - j) * H + i;
    }
}

void make_orientations(const Mat& src, Mat out[8]) {
    const int W = src.w, H = src.h;
    out[0] = src;
    for (int ti = 1; ti < 8; ti++) out[ti].create(ti < 4 ? W : H, ti < 4 ? H : W, src.c);
    for (int q = 0; q < src.c; q++)
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++) {
                float v = src.channel(q)[(size_t)i * W + j];
                for (int ti = 1; ti < 8; ti++) out[ti].channel(q)[tta_index(ti, i, j, W, H)] = v;
            }
}

// v4 temporal merge of a flow / reversed-flow pair (rife.cpp:3477-3512, 4264-4297)
void v4_temporal_merge(Mat& f, Mat& r) {
    size_t n = (size_t)f.w * f.h;
    float *fx = f.channel(0), *fy = f.channel(1), *fz = f.channel(2), *fw = f.channel(3), *fm = f.channel(4);
    float *rx = r.channel(0), *ry = r.channel(1), *rz = r.channel(2), *rw = r.channel(3), *rm = r.channel(4);
    for (size_t i = 0; i < n; i++) {
        float x = (fx[i] + rz[i]) * 0.5f;
        float y = (fy[i] + rw[i]) * 0.5f;
        float z = (fz[i] + rx[i]) * 0.5f;
        float w = (fw[i] + ry[i]) * 0.5f;
        float m = (fm[i] - rm[i]) * 0.5f;
        fx[i] = x; fy[i] = y; fz[i] = z; fw[i] = w; fm[i] = m;
        rx[i] = z; ry[i] = w; rz[i] = x; rw[i] = y; rm[i] = -m;
    }
}

// 8-orientation flow(/mask) consensus, in place: v4 (5 channels, rife.cpp:3515-3665) and v2 (4 channels, no mask,
// rife.cpp:1543-1667); signs per SURVEY App. G
void v4_spatial_avg(Mat fl[8]) {
    const int W = fl[0].w, H = fl[0].h;
    const bool has_mask = fl[0].c > 4;
    if (fl[0].c == 2) {   // v1 family: one 2-channel flow (rife.cpp:1669-1716): the x / y rows of the same algebra
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++) {
                size_t id[8];
                for (int ti = 0; ti < 8; ti++) id[ti] = tta_index(ti, i, j, W, H);
                auto X = [&](int ti) -> float& { return fl[ti].channel(0)[id[ti]]; };
                auto Y = [&](int ti) -> float& { return fl[ti].channel(1)[id[ti]]; };
                float x = (X(0) + -X(1) + -X(2) + X(3) + Y(4) + Y(5) + -Y(6) + -Y(7)) * 0.125f;
                float y = (Y(0) + Y(1) + -Y(2) + -Y(3) + X(4) + -X(5) + -X(6) + X(7)) * 0.125f;
                X(0) = x; X(1) = -x; X(2) = -x; X(3) = x; X(4) = y; X(5) = -y; X(6) = -y; X(7) = y;
                Y(0) = y; Y(1) = y; Y(2) = -y; Y(3) = -y; Y(4) = x; Y(5) = x; Y(6) = -x; Y(7) = -x;
            }
        return;
    }
    for (int i = 0; i < H; i++)
        for (int j = 0; j < W; j++) {
            size_t id[8];
            for (int ti = 0; ti < 8; ti++) id[ti] = tta_index(ti, i, j, W, H);
            auto X = [&](int ti) -> float& { return fl[ti].channel(0)[id[ti]]; };
            auto Y = [&](int ti) -> float& { return fl[ti].channel(1)[id[ti]]; };
            auto Z = [&](int ti) -> float& { return fl[ti].channel(2)[id[ti]]; };
            auto Wc = [&](int ti) -> float& { return fl[ti].channel(3)[id[ti]]; };
            auto M = [&](int ti) -> float& { return fl[ti].channel(has_mask ? 4 : 0)[id[ti]]; };
            float x = (X(0) + -X(1) + -X(2) + X(3) + Y(4) + Y(5) + -Y(6) + -Y(7)) * 0.125f;
            float y = (Y(0) + Y(1) + -Y(2) + -Y(3) + X(4) + -X(5) + -X(6) + X(7)) * 0.125f;
            float z = (Z(0) + -Z(1) + -Z(2) + Z(3) + Wc(4) + Wc(5) + -Wc(6) + -Wc(7)) * 0.125f;
            float w = (Wc(0) + Wc(1) + -Wc(2) + -Wc(3) + Z(4) + -Z(5) + -Z(6) + Z(7)) * 0.125f;
            float m = has_mask ? (M(0) + M(1) + M(2) + M(3) + M(4) + M(5) + M(6) + M(7)) * 0.125f : 0.f;
            X(0) = x; X(1) = -x; X(2) = -x; X(3) = x; X(4) = y; X(5) = -y; X(6) = -y; X(7) = y;
            Y(0) = y; Y(1) = y; Y(2) = -y; Y(3) = -y; Y(4) = x; Y(5) = x; Y(6) = -x; Y(7) = -x;
            Z(0) = z; Z(1) = -z; Z(2) = -z; Z(3) = z; Z(4) = w; Z(5) = -w; Z(6) = -w; Z(7) = w;
            Wc(0) = w; Wc(1) = w; Wc(2) = -w; Wc(3) = -w; Wc(4) = z; Wc(5) = z; Wc(6) = -z; Wc(7) = -z;
            if (has_mask) for (int ti = 0; ti < 8; ti++) M(ti) = m;
        }
}

int v4_extract_flow(const Net& net, const Mat& a, const Mat& b, const Mat& t, Mat* flows, int fi, Mat& out) {
    Extractor ex(net);
    ex.input("in0", a); ex.input("in1", b); ex.input("in2", t);
    static const char* names[4] = {"flow0", "flow1", "flow2", "flow3"};
    for (int k = 0; k < fi; k++) ex.input(names[k], flows[k]);
    return ex.extract(names[fi], out);
}

int v4_extract_out(const Net& net, const Mat& a, const Mat& b, const Mat& t, Mat* flows, Mat& out) {
    Extractor ex(net);
    ex.input("in0", a); ex.input("in1", b); ex.input("in2", t);
    static const char* names[4] = {"flow0", "flow1", "flow2", "flow3"};
    if (flows) for (int k = 0; k < 4; k++) ex.input(names[k], flows[k]);
    return ex.extract("out0", out);
}

// reference rife.cpp:3204-4401
int process_v4_cpu(const RifeOracle& R, const uint8_t* p0, const uint8_t* p1, int w, int h, float timestep, uint8_t* outpx) {
    if (timestep == 0.f) { std::memcpy(outpx, p0, (size_t)w * h * 3); return 0; }   // rife.cpp:3206-3216 (shares the buffer)
    if (timestep == 1.f) { std::memcpy(outpx, p1, (size_t)w * h * 3); return 0; }
    const int wp = (w + 31) / 32 * 32, hp = (h + 31) / 32 * 32;
    Mat out(w, h, 3);
    int rc = 0;
    if (R.tta_mode) {
        Mat in0[8], in1[8], ts[2], tsr[2];
        make_orientations(preproc_pad(p0, w, h, wp, hp), in0);
        make_orientations(preproc_pad(p1, w, h, wp, hp), in1);
        ts[0] = filled(wp, hp, timestep); ts[1] = filled(hp, wp, timestep);
        tsr[0] = filled(wp, hp, 1.f - timestep); tsr[1] = filled(hp, wp, 1.f - timestep);
        Mat outp[8], outr[8];
        Mat flow[8][4], flowr[8][4];   // [ti][fi]
        for (int fi = 0; fi < 4; fi++) {
            Mat cur[8], curr[8];
            for (int ti = 0; ti < 8; ti++) {
                if ((rc = v4_extract_flow(R.flownet, in0[ti], in1[ti], ts[ti / 4], flow[ti], fi, cur[ti]))) return rc;
                cur[ti] = cur[ti].clone();
                if (R.tta_temporal_mode) {
                    if ((rc = v4_extract_flow(R.flownet, in1[ti], in0[ti], tsr[ti / 4], flowr[ti], fi, curr[ti]))) return rc;
                    curr[ti] = curr[ti].clone();
                    v4_temporal_merge(cur[ti], curr[ti]);
                }
            }
            v4_spatial_avg(cur);
            if (R.tta_temporal_mode) v4_spatial_avg(curr);
            for (int ti = 0; ti < 8; ti++) { flow[ti][fi] = cur[ti]; if (R.tta_temporal_mode) flowr[ti][fi] = curr[ti]; }
        }
        for (int ti = 0; ti < 8; ti++) {
            if ((rc = v4_extract_out(R.flownet, in0[ti], in1[ti], ts[ti / 4], flow[ti], outp[ti]))) return rc;
            if (R.tta_temporal_mode)
                if ((rc = v4_extract_out(R.flownet, in1[ti], in0[ti], tsr[ti / 4], flowr[ti], outr[ti]))) return rc;
        }
        // cut padding and postproc (rife.cpp:4056-4144): row()-based, i.e. pitch-correct
        for (int q = 0; q < 3; q++) {
            float* o = out.channel(q);
            for (int i = 0; i < h; i++)
                for (int j = 0; j < w; j++) {
                    float s[8], sr[8];
                    for (int ti = 0; ti < 8; ti++) {
                        size_t id = tta_index(ti, i, j, wp, hp);
                        s[ti] = outp[ti].channel(q)[id];
                        if (R.tta_temporal_mode) sr[ti] = outr[ti].channel(q)[id];
                    }
                    float v = (s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7]) / 8;
                    if (R.tta_temporal_mode) {
                        float vr = (sr[0] + sr[1] + sr[2] + sr[3] + sr[4] + sr[5] + sr[6] + sr[7]) / 8;
                        o[(size_t)i * w + j] = (v + vr) * 0.5f * 255.f + 0.5f;
                    } else {
                        o[(size_t)i * w + j] = v * 255.f + 0.5f;
                    }
                }
        }
    } else {
        Mat in0 = preproc_pad(p0, w, h, wp, hp), in1 = preproc_pad(p1, w, h, wp, hp);
        Mat ts = filled(wp, hp, timestep);
        Mat outp, outr;
        if (R.tta_temporal_mode) {
            Mat tsr = filled(wp, hp, 1.f - timestep);
            Mat flow[4], flowr[4];
            for (int fi = 0; fi < 4; fi++) {
                Mat a, b;
                if ((rc = v4_extract_flow(R.flownet, in0, in1, ts, flow, fi, a))) return rc;
                if ((rc = v4_extract_flow(R.flownet, in1, in0, tsr, flowr, fi, b))) return rc;
                a = a.clone(); b = b.clone();
                v4_temporal_merge(a, b);
                flow[fi] = a; flowr[fi] = b;
            }
            if ((rc = v4_extract_out(R.flownet, in0, in1, ts, flow, outp))) return rc;
            if ((rc = v4_extract_out(R.flownet, in1, in0, tsr, flowr, outr))) return rc;
        } else {
            if ((rc = v4_extract_out(R.flownet, in0, in1, ts, nullptr, outp))) return rc;
        }
        for (int q = 0; q < 3; q++) {
            float* o = out.channel(q);
            const float* p = outp.channel(q);
            const float* pr = R.tta_temporal_mode ? outr.channel(q) : nullptr;
            for (int i = 0; i < h; i++)
                for (int j = 0; j < w; j++) {
                    // literal: ptr++ over h*w contiguous elements (App. F-1); gpu_crop: padded pitch
                    size_t src = R.gpu_crop ? (size_t)i * wp + j : (size_t)i * w + j;
                    float v = pr ? (p[src] + pr[src]) * 0.5f * 255.f + 0.5f : p[src] * 255.f + 0.5f;
                    o[(size_t)i * w + j] = v;
                }
        }
    }
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            for (int q = 0; q < 3; q++) outpx[((size_t)i * w + j) * 3 + q] = sat_u8(out.channel(q)[(size_t)i * w + j]);
    return 0;
}

// v2 flow / reversed-flow merge (rife.cpp:2277-2303)
void v2_temporal_merge(Mat& f, Mat& r) {
    size_t n = (size_t)f.w * f.h;
    if (f.c == 2) {   // v1 family (rife.cpp:2304-2316, 1525-1538): the reversed flow is the negated flow
        float *fx = f.channel(0), *fy = f.channel(1), *rx = r.channel(0), *ry = r.channel(1);
        for (size_t i = 0; i < n; i++) {
            float x = (fx[i] - rx[i]) * 0.5f, y = (fy[i] - ry[i]) * 0.5f;
            fx[i] = x; fy[i] = y; rx[i] = -x; ry[i] = -y;
        }
        return;
    }
    float *fx = f.channel(0), *fy = f.channel(1), *fz = f.channel(2), *fw = f.channel(3);
    float *rx = r.channel(0), *ry = r.channel(1), *rz = r.channel(2), *rw = r.channel(3);
    for (size_t i = 0; i < n; i++) {
        float x = (fx[i] + rz[i]) * 0.5f, y = (fy[i] + rw[i]) * 0.5f, z = (fz[i] + rx[i]) * 0.5f, w = (fw[i] + ry[i]) * 0.5f;
        fx[i] = x; fy[i] = y; fz[i] = z; fw[i] = w;
        rx[i] = z; ry[i] = w; rz[i] = x; rw[i] = y;
    }
}

int v2_flow(const RifeOracle& R, const Mat& a, const Mat& b, Mat& flow) {
    Extractor ex(R.flownet);
    int rc;
    if (R.uhd_mode) {   // rife.cpp:2212-2229: half-res flow estimate, bilinear x2, doubled
        Mat ad, bd, fd, fh;
        oracle::interp_bilinear(a, ad, 0.5f, 0.5f);
        oracle::interp_bilinear(b, bd, 0.5f, 0.5f);
        ex.input("input0", ad); ex.input("input1", bd);
        if ((rc = ex.extract("flow", fd))) return rc;
        oracle::interp_bilinear(fd, fh, 2.f, 2.f);
        flow.create(fh.w, fh.h, fh.c);
        for (size_t i = 0; i < fh.total(); i++) flow.data[i] = fh.data[i] * 2.f;
        return 0;
    }
    ex.input("input0", a); ex.input("input1", b);
    if ((rc = ex.extract("flow", flow))) return rc;
    flow = flow.clone();
    return 0;
}

// reference rife.cpp:1214-2460, rife_v2 family, non-spatial-TTA branch 2139-2457
int process_cpu_v2(const RifeOracle& R, const uint8_t* p0, const uint8_t* p1, int w, int h, float timestep, uint8_t* outpx) {
    if (timestep == 0.f) { std::memcpy(outpx, p0, (size_t)w * h * 3); return 0; }   // rife.cpp:1216-1226
    if (timestep == 1.f) { std::memcpy(outpx, p1, (size_t)w * h * 3); return 0; }
    const int wp = (w + 31) / 32 * 32, hp = (h + 31) / 32 * 32;
    Mat in0 = preproc_pad(p0, w, h, wp, hp), in1 = preproc_pad(p1, w, h, wp, hp);
    Mat flow, flowr;
    int rc;
    auto synth = [&](const Mat& a, const Mat& b, const Mat& fl, Mat& o) -> int {   // slice -> contextnet x2 -> fusionnet
        int r;
        Mat c0[4], c1[4];
        static const char* fnames[4] = {"f1", "f2", "f3", "f4"};
        if (R.rife_v2) {   // Slice {-233,-233} (rife.cpp:2322-2330), each half bound to "flow.0"
            Mat f0 = fl.channel_range(0, 2).clone(), f1 = fl.channel_range(2, 2).clone();
            { Extractor ex(R.contextnet); ex.light = false; ex.input("input.1", a); ex.input("flow.0", f0);
              for (int k = 0; k < 4; k++) if ((r = ex.extract(fnames[k], c0[k]))) return r; }
            { Extractor ex(R.contextnet); ex.light = false; ex.input("input.1", b); ex.input("flow.0", f1);
              for (int k = 0; k < 4; k++) if ((r = ex.extract(fnames[k], c1[k]))) return r; }
        } else {           // v1 family: one 2-channel flow; img0 binds it to "flow.0", img1 to "flow.1" (= -flow.0 inside the graph), rife.cpp:2339-2362
            { Extractor ex(R.contextnet); ex.light = false; ex.input("input.1", a); ex.input("flow.0", fl);
              for (int k = 0; k < 4; k++) if ((r = ex.extract(fnames[k], c0[k]))) return r; }
            { Extractor ex(R.contextnet); ex.light = false; ex.input("input.1", b); ex.input("flow.1", fl);
              for (int k = 0; k < 4; k++) if ((r = ex.extract(fnames[k], c1[k]))) return r; }
        }
        Extractor ex(R.fusionnet);
        ex.input("img0", a); ex.input("img1", b); ex.input("flow", fl);
        static const char* n0[4] = {"3", "4", "5", "6"};
        static const char* n1[4] = {"7", "8", "9", "10"};
        for (int k = 0; k < 4; k++) { ex.input(n0[k], c0[k]); ex.input(n1[k], c1[k]); }
        if ((r = ex.extract("output", o))) return r;
        o = o.clone();
        return 0;
    };
    if (R.tta_mode) {
        // rife.cpp:1256-2138 (rife_v2 branches).  The reversed FusionNet pass of the reference re-uses the forward
        // contexts swapped (rife.cpp:2026-2047); after the final merge flow_reversed = (z, w, x, y) of flow, so
        // ContextNet(in1, flow_reversed[0:2]) is the same computation and synth() may simply recompute it.
        Mat a[8], b[8], fl[8], flr[8], o[8], orv[8];
        make_orientations(in0, a); make_orientations(in1, b);
        for (int ti = 0; ti < 8; ti++) if ((rc = v2_flow(R, a[ti], b[ti], fl[ti]))) return rc;             // 1420-1450
        if (R.tta_temporal_mode)
            for (int ti = 0; ti < 8; ti++) {                                                                // 1452-1540
                if ((rc = v2_flow(R, b[ti], a[ti], flr[ti]))) return rc;
                v2_temporal_merge(fl[ti], flr[ti]);
            }
        v4_spatial_avg(fl);                                                                                 // 1543-1667
        if (R.tta_temporal_mode) {
            v4_spatial_avg(flr);                                                                            // 1721-1896
            for (int ti = 0; ti < 8; ti++) v2_temporal_merge(fl[ti], flr[ti]);                              // 1898-1948
        }
        for (int ti = 0; ti < 8; ti++) {                                                                    // 1964-2048
            if ((rc = synth(a[ti], b[ti], fl[ti], o[ti]))) return rc;
            if (R.tta_temporal_mode && (rc = synth(b[ti], a[ti], flr[ti], orv[ti]))) return rc;
        }
        for (int i = 0; i < h; i++)                                                                         // 2050-2137
            for (int j = 0; j < w; j++)
                for (int q = 0; q < 3; q++) {
                    float sv[8], sr[8];
                    for (int ti = 0; ti < 8; ti++) {
                        size_t id = tta_index(ti, i, j, wp, hp);
                        sv[ti] = o[ti].channel(q)[id];
                        if (R.tta_temporal_mode) sr[ti] = orv[ti].channel(q)[id];
                    }
                    float v = (sv[0] + sv[1] + sv[2] + sv[3] + sv[4] + sv[5] + sv[6] + sv[7]) / 8;
                    float res;
                    if (R.tta_temporal_mode) {
                        float vr = (sr[0] + sr[1] + sr[2] + sr[3] + sr[4] + sr[5] + sr[6] + sr[7]) / 8;
                        res = (v + vr) * 0.5f * 255.f + 0.5f;
                    } else res = v * 255.f + 0.5f;
                    outpx[((size_t)i * w + j) * 3 + q] = sat_u8(res);
                }
        return 0;
    }
    if ((rc = v2_flow(R, in0, in1, flow))) return rc;
    if (R.tta_temporal_mode) {
        if ((rc = v2_flow(R, in1, in0, flowr))) return rc;
        v2_temporal_merge(flow, flowr);
    }
    // slice -> contextnet x2 -> fusionnet (rife.cpp:2322-2388); the reversed pass of -z re-uses the contexts swapped
    // (rife.cpp:2391-2413), which synth() recomputes to identical values
    Mat outp, outr;
    if ((rc = synth(in0, in1, flow, outp))) return rc;
    if (R.tta_temporal_mode) if ((rc = synth(in1, in0, flowr, outr))) return rc;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            for (int q = 0; q < 3; q++) {
                size_t src = R.gpu_crop ? (size_t)i * wp + j : (size_t)i * w + j;
                float v = R.tta_temporal_mode ? (outp.channel(q)[src] + outr.channel(q)[src]) * 0.5f * 255.f + 0.5f
                                              : outp.channel(q)[src] * 255.f + 0.5f;
                outpx[((size_t)i * w + j) * 3 + q] = sat_u8(v);
            }
    return 0;
}

}  // namespace

extern "C" {

void* oracle_create(int tta_mode, int tta_temporal_mode, int uhd_mode, int num_threads, int rife_v2, int rife_v4) {
    RifeOracle* R = new RifeOracle;
    R->tta_mode = tta_mode; R->tta_temporal_mode = tta_temporal_mode; R->uhd_mode = uhd_mode;
    R->num_threads = num_threads; R->rife_v2 = rife_v2; R->rife_v4 = rife_v4;
    return R;
}

void oracle_destroy(void* h) { delete (RifeOracle*)h; }

void oracle_set_gpu_crop(void* h, int v) { ((RifeOracle*)h)->gpu_crop = v; }

// reference rife.cpp:127-163 (load_param_model ×1 for v4, ×3 otherwise)
int oracle_load(void* h, const char* modeldir) {
    RifeOracle* R = (RifeOracle*)h;
    auto load = [&](Net& n, const char* name) {
        n.num_threads = R->num_threads;
        std::string base = std::string(modeldir) + "/" + name;
        int r = n.load_param(base + ".param");
        if (r) return r * 10 - 1;
        r = n.load_model(base + ".bin");
        if (r) return r * 10 - 2;
        return 0;
    };
    int r = load(R->flownet, "flownet");
    if (r) return r;
    if (!R->rife_v4) {
        if ((r = load(R->contextnet, "contextnet"))) return r - 1000;
        if ((r = load(R->fusionnet, "fusionnet"))) return r - 2000;
    }
    return 0;
}

// reference rife.cpp:381-393 dispatcher, CPU side
int oracle_process(void* h, const uint8_t* in0, const uint8_t* in1, int w, int hgt, float timestep, uint8_t* out) {
    RifeOracle* R = (RifeOracle*)h;
    if (R->rife_v4) return process_v4_cpu(*R, in0, in1, w, hgt, timestep, out);
    return process_cpu_v2(*R, in0, in1, w, hgt, timestep, out);   // v2 / v3 (rife_v2) and the v1 family share RIFE::process_cpu
}

// Debug tap for stage-wise parity: run the plain v4 graph and return any named blob (planar CHW fp32).
// `flows_in` optionally injects flow0..flow{n_inject-1} (each 6 x H/s x W/s).  Returns channel count or <0.
int oracle_v4_extract(void* h, const uint8_t* in0, const uint8_t* in1, int w, int hgt, float timestep, const char* blob,
                      const float* const* flows_in, int n_inject, float* out, int out_capacity, int* out_w, int* out_h) {
    RifeOracle* R = (RifeOracle*)h;
    const int wp = (w + 31) / 32 * 32, hp = (hgt + 31) / 32 * 32;
    Mat a = preproc_pad(in0, w, hgt, wp, hp), b = preproc_pad(in1, w, hgt, wp, hp), t = filled(wp, hp, timestep);
    Extractor ex(R->flownet);
    ex.input("in0", a); ex.input("in1", b); ex.input("in2", t);
    static const char* names[4] = {"flow0", "flow1", "flow2", "flow3"};
    static const int scale[4] = {8, 4, 2, 1};
    // blob flow{k}: PixelShuffle output (6 ch at 1/s, rife-v4.6) or Deconvolution output (5 ch at 1/2s, rife-v4)
    const int fb = R->flownet.find_blob("flow0");
    const bool deconv_flow = fb >= 0 && R->flownet.layers[R->flownet.producer[fb]].type == "Deconvolution";
    for (int k = 0; k < n_inject; k++) {
        const int div = deconv_flow ? 2 * scale[k] : scale[k];
        Mat f(wp / div, hp / div, deconv_flow ? 5 : 6);
        std::memcpy(f.data, flows_in[k], f.total() * sizeof(float));
        ex.input(names[k], f);
    }
    Mat o;
    int rc = ex.extract(blob, o);
    if (rc) return rc;
    if ((int)o.total() > out_capacity) return -40;
    std::memcpy(out, o.data, o.total() * sizeof(float));
    *out_w = o.w; *out_h = o.h;
    return o.c;
}

// generic single-net debug tap (used for the v2.3 nets): bind n_in named inputs, extract one blob
int oracle_net_extract(void* h, int which, int n_in, const char* const* names, const float* const* datas, const int* dims /*w,h,c each*/,
                       const char* blob, float* out, int out_capacity, int* out_w, int* out_h) {
    RifeOracle* R = (RifeOracle*)h;
    const Net& net = which == 0 ? R->flownet : which == 1 ? R->contextnet : R->fusionnet;
    Extractor ex(net);
    for (int i = 0; i < n_in; i++) {
        Mat m(dims[i * 3], dims[i * 3 + 1], dims[i * 3 + 2]);
        std::memcpy(m.data, datas[i], m.total() * sizeof(float));
        if (ex.input(names[i], m)) return -41;
    }
    Mat o;
    int rc = ex.extract(blob, o);
    if (rc) return rc;
    if ((int)o.total() > out_capacity) return -40;
    std::memcpy(out, o.data, o.total() * sizeof(float));
    *out_w = o.w; *out_h = o.h;
    return o.c;
}

size_t oracle_bin_bytes(void* h, int which, int total) {
    RifeOracle* R = (RifeOracle*)h;
    const Net& net = which == 0 ? R->flownet : which == 1 ? R->contextnet : R->fusionnet;
    return total ? net.bin_bytes_total : net.bin_bytes_consumed;
}

// ---- single-op entry points for per-kernel parity tests (planar CHW fp32 in/out) ----
void oracle_conv2d(const float* in, int w, int h, int c, const float* weight, const float* bias, int outc, int k, int stride, int pad,
                   int act_type, float act_p0, float* out, int num_threads) {
    Mat m(w, h, c); std::memcpy(m.data, in, m.total() * 4);
    Mat o; float ap[2] = {act_p0, 0.f};
    oracle::conv2d(m, o, weight, bias, outc, k, stride, pad, act_type, ap, num_threads);
    std::memcpy(out, o.data, o.total() * 4);
}

void oracle_deconv2d(const float* in, int w, int h, int c, const float* weight, const float* bias, int outc, int k, int stride, int pad,
                     int act_type, float act_p0, float* out, int num_threads) {
    Mat m(w, h, c); std::memcpy(m.data, in, m.total() * 4);
    Mat o; float ap[2] = {act_p0, 0.f};
    oracle::deconv2d(m, o, weight, bias, outc, k, stride, pad, act_type, ap, num_threads);
    std::memcpy(out, o.data, o.total() * 4);
}

void oracle_warp(const float* image, const float* flow, int w, int h, int c, float* out, int num_threads) {
    Mat m(w, h, c), f(w, h, 2); std::memcpy(m.data, image, m.total() * 4); std::memcpy(f.data, flow, f.total() * 4);
    Mat o; oracle::warp(m, f, o, num_threads);
    std::memcpy(out, o.data, o.total() * 4);
}

void oracle_interp(const float* in, int w, int h, int c, float hscale, float wscale, float* out) {
    Mat m(w, h, c); std::memcpy(m.data, in, m.total() * 4);
    Mat o; oracle::interp_bilinear(m, o, hscale, wscale);
    std::memcpy(out, o.data, o.total() * 4);
}

void oracle_pixelshuffle(const float* in, int w, int h, int c, int r, float* out) {
    Mat m(w, h, c); std::memcpy(m.data, in, m.total() * 4);
    Mat o; oracle::pixelshuffle(m, o, r);
    std::memcpy(out, o.data, o.total() * 4);
}

}  // extern "C"
