// engine_ctx.h: profiler, the per-pair workspace (Ctx), the engine object behind rife_hip_t (weights of every family, workspace pool)
// One translation unit (engine.hip includes the engine_*.h sections in dependency order; every function here is file-local).
// No include guard on purpose: a section is included exactly once, by engine.hip.

#include "graph_exec.h"      // the generic layer-wise graph executor (v1 family): needs launch_conv (engine_dispatch.h), knows nothing of the workspaces below
namespace rife {

// ------------------------------------------------------------------------------------------------
// profiler (rife_hip_profile_*): HIP events on the launch stream around every kernel
// ------------------------------------------------------------------------------------------------

struct Profiler {
    bool on = false;
    std::mutex mu;
    struct Rec { int cls; hipEvent_t e0, e1; double flops; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    std::vector<std::string> names;
    std::map<std::string, int> ids;
    std::vector<double> ms, flops;
    std::vector<long long> launches;
    int cls_id(const std::string& n) {
        auto it = ids.find(n);
        if (it != ids.end()) return it->second;
        int id = (int)names.size();
        names.push_back(n); ids[n] = id; ms.push_back(0); flops.push_back(0); launches.push_back(0);
        return id;
    }
    void begin(const std::string& cls, double fl, hipStream_t st, size_t& token) {
        token = (size_t)-1;
        if (!on) return;
        std::lock_guard<std::mutex> g(mu);
        Rec r; r.cls = cls_id(cls); r.flops = fl;
        if (pool.size() >= 2) { r.e0 = pool.back(); pool.pop_back(); r.e1 = pool.back(); pool.pop_back(); }   // events are recycled
        else if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
        (void)hipEventRecord(r.e0, st);
        recs.push_back(r); token = recs.size() - 1;
    }
    void end(size_t token, hipStream_t st) {
        if (token == (size_t)-1) return;
        std::lock_guard<std::mutex> g(mu);
        (void)hipEventRecord(recs[token].e1, st);
    }
    void collect() {
        std::lock_guard<std::mutex> g(mu);
        for (Rec& r : recs) {
            (void)hipEventSynchronize(r.e1);
            float t = 0.f;
            if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) { ms[r.cls] += t; flops[r.cls] += r.flops; launches[r.cls]++; }
            pool.push_back(r.e0); pool.push_back(r.e1);
        }
        recs.clear();
    }
};

// ------------------------------------------------------------------------------------------------
// per-pair workspace ("context"): everything one in-flight frame pair needs, sized for one padded resolution
// ------------------------------------------------------------------------------------------------
struct Ctx {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int cu_budget = 0;                                                // > 0: `stream` owns this many compute units only (pool partition, lease_ctx)
    int pool_parts = 1, pool_part = 0;                                // pool workspaces: the partition of the chip its stream owns (1 = the whole chip)
    std::mutex use;                                                   // rife_hip_process_device: one caller at a time per stream workspace
    int w = 0, h = 0, wp = 0, hp = 0;
    uint8_t *d_in0 = nullptr, *d_in1 = nullptr, *d_out = nullptr;   // staging for the host-buffer entry point
    uint32_t *img0 = nullptr, *img1 = nullptr;                       // padded RGBX u8
    float *X = nullptr, *S1 = nullptr, *T0 = nullptr, *T1 = nullptr; // block input, stem-1 output, trunk ping/pong
    float *T2 = nullptr;                                             // rife-v4 (4.0): stem-1 output kept for the block's residual add
    unsigned char* P[4][2] = {};                                     // per block: trunk ping / pong as S16 tensors (conv_t64.h), zero borders
    float* flow[4] = {nullptr, nullptr, nullptr, nullptr};           // [hp/s][wp/s][8]
    float4* F = nullptr; float* M = nullptr;                         // full-resolution flow (4ch) and mask logit
    float4* F2 = nullptr; float* M2 = nullptr;                       // the other pair of buffers for a flow update fused into the next stem (stem_fused.h UPD); F, M are swapped with them
    float4* outf = nullptr;                                          // TTA only: out0 as float, padded
    // hipGraph replay of the plain v4 schedule for launch-bound frame sizes: fixed staging buffers (d_in0 / d_in1 / d_out), the
    // timestep in device memory, one warm-up pass (lazy allocations, kernel attributes), then capture once and replay
    float* d_ts = nullptr;
    hipEvent_t ev_group = nullptr;                                    // rife_hip_process_batch: cross-stream hand-off around a batched coarse trunk
    hipGraphExec_t gexec = nullptr;
    bool g_warm = false;
    // rife-v2.x only
    bool v2 = false;
    float4 *acc = nullptr, *D = nullptr, *head = nullptr;           // running half-res flow, deconv output, fusion head
    float4 *h0 = nullptr, *h1 = nullptr, *acc_s = nullptr;          // UHD: half-resolution fp32 frames and their (quarter-res) flow
    float *I8 = nullptr, *ca = nullptr, *cb = nullptr, *cc = nullptr, *feat[4] = {nullptr, nullptr, nullptr, nullptr}, *ctmp[3] = {nullptr, nullptr, nullptr};
    float2* fl[4] = {nullptr, nullptr, nullptr, nullptr};           // ContextNet flow pyramid
    // the second ContextNet pass (img1, flow10): its own activations, so that both passes ride one launch per layer (gridDim.y = 2)
    float *ca2 = nullptr, *cb2 = nullptr, *cc2 = nullptr, *feat2[4] = {nullptr, nullptr, nullptr, nullptr}, *ctmp2[3] = {nullptr, nullptr, nullptr};
    float2* fl2[4] = {nullptr, nullptr, nullptr, nullptr};
    float *e0a = nullptr, *e0b = nullptr, *e0c = nullptr, *B1 = nullptr, *e1a = nullptr, *B2 = nullptr, *e2a = nullptr, *B3 = nullptr, *e3a = nullptr, *B4 = nullptr;
    float *U0 = nullptr, *U1 = nullptr, *U2 = nullptr, *U3 = nullptr;
    // rife-v2.x TTA: per orientation RGBX frames, half-res flows [direction][orientation], float outputs [direction][orientation]
    uint32_t *timg0[8] = {}, *timg1[8] = {};
    float4 *tflow[2][8] = {}, *toutf[2][8] = {};
    // v1 family (generic graph executor): blob storage per net instance, one set per frame orientation (w x h / h x w for TTA);
    // [.][0] flownet, [1] / [2] contextnet of frame 0 / 1, [3] fusionnet, [4] tensors outside the nets (frames, UHD resizes, TTA flows)
    std::unique_ptr<GraphInst> ginst[2][5];
    std::vector<void*> allocs;
    ~Ctx() {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (ev_group) (void)hipEventDestroy(ev_group);
        for (void* p : allocs) (void)hipFree(p);
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
};

template <typename T>
static int dalloc(Ctx& c, T*& p, size_t n) {
    void* v = nullptr;
    HIPCHK(hipMalloc(&v, n * sizeof(T)));
    c.allocs.push_back(v);
    p = (T*)v;
    return 0;
}

}  // namespace rife

using namespace rife;

// ------------------------------------------------------------------------------------------------
// the engine object behind rife_hip_t
// ------------------------------------------------------------------------------------------------
// Device buffers of released resident frames (rife_hip_frame_*), reused by the next upload of the same size: hipFree waits for
// the whole device, which would stall the pairs in flight every time a frame of a sequence retires.
struct FramePool {
    int gpuid = 0;
    std::mutex mu;
    std::vector<std::pair<size_t, uint8_t*>> idle;
    uint8_t* take(size_t nbytes) {
        {
            std::lock_guard<std::mutex> g(mu);
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].first == nbytes) { uint8_t* p = idle[i].second; idle.erase(idle.begin() + i); return p; }
        }
        uint8_t* p = nullptr;
        return hipMalloc((void**)&p, nbytes) == hipSuccess ? p : nullptr;
    }
    void give(uint8_t* p, size_t nbytes) {
        uint8_t* evict = nullptr;
        {
            std::lock_guard<std::mutex> g(mu);
            if (idle.size() >= 16) { evict = idle.front().second; idle.erase(idle.begin()); }      // oldest out: sizes may change over time
            idle.emplace_back(nbytes, p);
        }
        if (evict && hipSetDevice(gpuid) == hipSuccess) (void)hipFree(evict);
    }
    ~FramePool() {
        if (!idle.empty() && hipSetDevice(gpuid) == hipSuccess) for (auto& e : idle) (void)hipFree(e.second);
    }
};

struct rife_hip {
    int gpuid = 0;
    bool tta = false, tta_temporal = false, uhd = false, v2 = false, v4 = false;
    int num_threads = 1;
    bool loaded = false;
    // v4.x schedule: per block {stem0, stem1, res x8, head}
    struct Block { ConvLayer stem0, stem1, res[8], head; int c = 0, scale = 1; } blk[4];
    // rife-v4 (4.0) variant of the schedule: PReLU, plain trunk + one residual add, 5-channel deconv head at half the block
    // resolution (flow{b} is [hp/2s][wp/2s][8] instead of [hp/s][wp/s][8])
    bool v40 = false;
    // finest-block trunk on S16 tensors + the persistent conv_t64 kernel (RIFE_HIP_T64=0 at create time keeps conv_h2b: A/B and
    // the bit-equality test of the two trunk implementations)
    bool t64 = true;
    // block-3 trunk on the row-streaming kernel (conv_rs.h) instead of conv_t64 (RIFE_HIP_RS=0 at create time: A/B, bit-equality test)
    bool rs = true;
    // ... two layers per launch, layer A's rows LDS-resident (conv_rs2.h; RIFE_HIP_RS2=0 at create time: A/B, bit-equality test)
    bool rs2 = true;
    // coarse-block trunks on the weight-stationary K-split kernel (conv_ks.h): bit mask by channel count, see launch_ks (RIFE_HIP_KS at create time)
    int ks_mask = 0;
    // block 3: block-input assembly + both stem convolutions in one row-streaming kernel (stem_rs.h) instead of stem0_fused_kernel + conv_h2s2_kernel
    // (RIFE_HIP_STEM_RS=0 at create time: A/B, the comparison test)
    bool stem_rs = true;
    // block 3's head + the tail of the graph + postproc in one row-streaming kernel (tail_rs.h) instead of head_h2_kernel<EPI_FINAL, true>
    // (RIFE_HIP_TAIL_RS=0 at create time: A/B, the comparison test)
    bool tail_rs = true;
    bool tail_rs_always = false;      // RIFE_HIP_TAIL_RS=2: at every frame size (tests)
    // -x -z: temporal + spatial flow consensus of a block in one kernel (k_v4_consensus); RIFE_HIP_TTA_CONSENSUS=0 at create time: the two steps as
    // separate kernels (8 + 2 launches per block; A/B, the bit-identity test)
    bool tta_consensus = true;
    // RIFE_HIP_FUSE_FLOW=1 (A/B, parity taps): the flow updates after blocks 1 and 2 inside the fused stems of blocks 2 and 3 (stem_fused.h UPD)
    // instead of two k_flow_update launches.  Bit-identical, and measured SLOWER at 4K (432 vs 442 frames/s, same call): the update kernels
    // run at 6 - 7 TB/s, the stems are bound by gather latency and VALU issue and every load added to them costs more than the pass it removes
    // (stem0_b3 0.210 -> 0.285, stem0_b2 0.161 -> 0.272, flow_update 0.191 -> 0.034 ms per pair).  Off in the product.
    bool fuse_flow = false;
    int flow_div(int b) const { return v40 ? 2 * blk[b].scale : blk[b].scale; }
    // rife-v2.x schedule (IFNet + ContextNet + FusionNet)
    struct V2Block { ConvLayer stem0, stem1, conv[6], head; int c = 0, scale = 1; } fblk[4];
    // rife-v3.x: same ContextNet / FusionNet, IFNet of 3 blocks (scales 4, 2, 1; 160 channels; trunk = 3 x [conv, conv, + skip])
    bool v3 = false;
    bool prof_fine = false;                                              // RIFE_HIP_PROFILE_FINE=1: per-layer profile classes (load_v2)
    int n_fblk = 4;
    // v1 family (rife, rife-HD, rife-UHD, rife-anime): executed layer by layer from the .param (graph_exec.h)
    bool v1 = false;
    std::unique_ptr<GraphNet> gflow, gctx, gfus;
    ConvLayer ctxc[10];          // ContextNet convs in graph order
    ConvLayer fus[15];           // FusionNet: 10 down convs, 4 up deconvs, sigmoid head
    mutable Profiler prof;
    mutable std::mutex mu;
    mutable std::vector<std::unique_ptr<Ctx>> free_ctx;                  // pool for the host-buffer entry points (lease_ctx / release_ctx)
    mutable int leased = 0;                                              // pool workspaces in use = callers in flight
    mutable int lease_hist[32] = {};                                     // callers in flight at each of the last 32 leases: the pool is trimmed to their maximum
    mutable unsigned lease_n = 0;
    mutable std::mutex h2d_mu;                                            // upload token of the host-frame entry points (enqueue_host_pair)
    mutable int pool_parts_now = 1;                                      // the layout of the latest lease (pool_layout)
    mutable int part_live[5][4] = {};                                    // [parts][part]: leased workspaces per partition of the chip
    mutable std::vector<hipEvent_t> batch_fork;                          // rife_hip_process_device_batch: recycled fork events
    mutable std::map<void*, int> part_streams;                           // rife_hip_stream_create: CU-masked streams of this engine -> compute units they own
    mutable std::map<void*, std::unique_ptr<Ctx>> stream_ctx;            // one workspace per caller stream
    mutable std::mutex tta_mu;                                           // TTA passes share one set of workspaces
    std::shared_ptr<FramePool> frame_pool;                               // shared with the frames: they may outlive the engine
    mutable std::vector<hipStream_t> upload_streams;                     // rife_hip_frame_upload: one copy stream per concurrent uploader
    mutable std::unique_ptr<Ctx> tta_ctx[2][8];                          // [direction][orientation]
    static constexpr int NLANE = 4;                                      // spatial TTA: orientations run on 4 worker streams
    mutable hipStream_t tta_lane[NLANE] = {nullptr, nullptr, nullptr, nullptr};
    mutable int tta_lane_cus = 0;                                        // > 0: the lanes are CU-masked streams that own this many compute units each (run_v4_tta)
    mutable hipEvent_t tta_fork[6] = {}, tta_join[6][NLANE] = {};

    ~rife_hip() {
        (void)hipSetDevice(gpuid);
        free_ctx.clear(); stream_ctx.clear();
        for (auto& d : tta_ctx) for (auto& c : d) c.reset();
        for (auto& l : tta_lane) if (l) (void)hipStreamDestroy(l);
        for (auto& u : upload_streams) (void)hipStreamDestroy(u);
        for (auto& e : tta_fork) if (e) (void)hipEventDestroy(e);
        for (auto& e : batch_fork) if (e) (void)hipEventDestroy(e);
        for (auto& kv : part_streams) (void)hipStreamDestroy((hipStream_t)kv.first);
        for (auto& r : tta_join) for (auto& e : r) if (e) (void)hipEventDestroy(e);
        for (auto& b : blk) { free_layer(b.stem0); free_layer(b.stem1); for (auto& r : b.res) free_layer(r); free_layer(b.head); }
        for (auto& b : fblk) { free_layer(b.stem0); free_layer(b.stem1); for (auto& r : b.conv) free_layer(r); free_layer(b.head); }
        for (auto& l : ctxc) free_layer(l);
        for (auto& l : fus) free_layer(l);
    }
};
