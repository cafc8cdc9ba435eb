// engine_v4.h: rife-v4.x schedule: RIFE::process_v4 (rife.cpp:2462-3202) = run_v4, run_v4_tta, run_v4_group and their helpers
// One translation unit (engine.hip includes the engine_*.h sections in dependency order; every function here is file-local).
// No include guard on purpose: a section is included exactly once, by engine.hip.

namespace rife {


// structural hashes of the graphs the schedules below were written for (= the reference's
// models/rife-v4.6/flownet.param; tests/test_models.py proves the equivalence whenever /root/reference exists)
static const uint64_t V46_HASH_OUT0 = RIFE_V46_HASH_OUT0;

static std::atomic<bool> g_fuse_flow_buffers{false};                 // some engine of the process asked for RIFE_HIP_FUSE_FLOW=1: workspaces carry F2, M2

// (Re)allocate a workspace for frames of w x h (padded wp x hp).  `scratch` != null: borrow the big per-layer
// scratch tensors (block input, stem output, trunk ping/pong) from another context of the same pixel count —
// the TTA passes run one after another on one stream, only flows / F / M / images must persist per pass.
static int ensure_ctx_dims_impl(Ctx& c, int w, int h, int wp, int hp, const Ctx* scratch, bool own_images, bool want_outf) {
    if (!c.v2 && c.wp == wp && c.hp == hp && c.w == w && c.h == h && (!want_outf || c.outf)) return 0;
    c.v2 = false;
    if (c.gexec) { (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; }
    c.g_warm = false; c.d_ts = nullptr;
    for (void* p : c.allocs) (void)hipFree(p);
    c.allocs.clear();
    c.outf = nullptr; c.F2 = nullptr; c.M2 = nullptr;
    for (auto& pb : c.P) pb[0] = pb[1] = nullptr;                       // S16 trunk tensors: allocated by the first block that runs on them (ensure_s16)
    c.w = w; c.h = h; c.wp = wp; c.hp = hp;
    const size_t P = (size_t)wp * hp;
    int rc;
    if (own_images) {
        if ((rc = dalloc(c, c.img0, P))) return rc;
        if ((rc = dalloc(c, c.img1, P))) return rc;
    }
    if (scratch) { c.X = scratch->X; c.S1 = scratch->S1; c.T0 = scratch->T0; c.T1 = scratch->T1; c.T2 = scratch->T2; }
    else {
        if ((rc = dalloc(c, c.d_in0, (size_t)w * h * 3))) return rc;
        if ((rc = dalloc(c, c.d_in1, (size_t)w * h * 3))) return rc;
        if ((rc = dalloc(c, c.d_out, (size_t)w * h * 3))) return rc;
        if ((rc = dalloc(c, c.X, P * 16))) return rc;                  // block 3: full res x 16 ch
        if ((rc = dalloc(c, c.S1, P / 4 * 32))) return rc;             // block 3 stem-0 output: (hp/2 x wp/2) x 32
        if ((rc = dalloc(c, c.T0, P / 16 * 64))) return rc;            // block 3 trunk: (hp/4 x wp/4) x 64 (the largest trunk)
        if ((rc = dalloc(c, c.T1, P / 16 * 64))) return rc;
        if ((rc = dalloc(c, c.T2, P / 16 * 64))) return rc;
    }
    static const int sc[4] = {8, 4, 2, 1};
    for (int b = 0; b < 4; b++) {
        const size_t n = P / (sc[b] * sc[b]) * 8;
        if ((rc = dalloc(c, c.flow[b], n))) return rc;
        // on the workspace's own stream, not the legacy stream: a synchronous hipMemset from one caller thread while others create
        // streams / launch on theirs makes the runtime fail intermittently ("legacy stream depend on a capturing blocking stream", then
        // every later call of the process reports a capture error) - tools/reentrancy_stress.py, ~1 in 100 concurrent calls
        if (c.stream) HIPCHK(hipMemsetAsync(c.flow[b], 0, n * 4, c.stream));
        else HIPCHK(hipMemset(c.flow[b], 0, n * 4));
    }
    if ((rc = dalloc(c, c.F, P))) return rc;
    if ((rc = dalloc(c, c.M, P))) return rc;
    if (!want_outf && !scratch && g_fuse_flow_buffers) {               // the plain pass of an engine created with RIFE_HIP_FUSE_FLOW=1 (not the TTA workspaces, whose updates go through the consensus kernels)
        if ((rc = dalloc(c, c.F2, P))) return rc;
        if ((rc = dalloc(c, c.M2, P))) return rc;
    }
    if (!scratch && (rc = dalloc(c, c.d_ts, 4))) return rc;
    if (want_outf && (rc = dalloc(c, c.outf, P))) return rc;
    return 0;
}

// A workspace whose (re)allocation failed half way is emptied, so that the next call reports the error again instead of taking the
// "already sized" early return and running on freed memory.
static void reset_ctx(Ctx& c) {
    if (c.gexec) { (void)hipGraphExecDestroy(c.gexec); c.gexec = nullptr; }
    for (void* p : c.allocs) (void)hipFree(p);
    c.allocs.clear();
    c.w = c.h = c.wp = c.hp = 0; c.v2 = false; c.outf = nullptr; c.F2 = nullptr; c.M2 = nullptr; c.d_ts = nullptr; c.g_warm = false; for (auto& pb : c.P) pb[0] = pb[1] = nullptr;
}
static int ensure_ctx_dims(Ctx& c, int w, int h, int wp, int hp, const Ctx* scratch = nullptr, bool own_images = true, bool want_outf = false) {
    const int rc = ensure_ctx_dims_impl(c, w, h, wp, hp, scratch, own_images, want_outf);
    if (rc) reset_ctx(c);
    return rc;
}

static int ensure_ctx(Ctx& c, int w, int h) {
    return ensure_ctx_dims(c, w, h, (w + 31) / 32 * 32, (h + 31) / 32 * 32);   // pad to 32n, rife.cpp:2499-2500
}

struct Timed {
    Profiler& p; hipStream_t st; size_t tok;
    Timed(Profiler& p_, const std::string& cls, double fl, hipStream_t s) : p(p_), st(s) { p.begin(cls, fl, st, tok); }
    ~Timed() { p.end(tok, st); }
};

static inline dim3 grid2d(int w, int h) { return dim3((w + 255) / 256, h); }
// tiles of the kernels that touch all eight TTA orientations of a plane: a wave = 8 columns x 8 rows (32-byte and 16-byte elements: 256- / 128-byte
// runs in the straight AND in the transposed buffers) or 16 x 4 rows of a 16 x 16 block (4-byte elements: 64-byte runs both ways)
static inline dim3 tta_block(int elem_bytes) { return elem_bytes >= 16 ? dim3(8, 32) : dim3(16, 16); }
static inline dim3 tta_grid(int w, int h, int elem_bytes) { const dim3 b = tta_block(elem_bytes); return dim3((w + b.x - 1) / b.x, (h + b.y - 1) / b.y); }
// rife_preproc.comp: u8 HWC RGB -> zero-padded RGBX; four pixels per lane when the frame allows 4-byte loads
static inline void launch_preproc(hipStream_t st, const uint8_t* rgb, int w, int h, uint32_t* out, int wp, int hp) {
    if ((w & 3) == 0 && (reinterpret_cast<uintptr_t>(rgb) & 3) == 0) hipLaunchKernelGGL(k_preproc4, dim3((wp / 4 + 255) / 256, hp), dim3(256), 0, st, rgb, w, h, out, wp, hp);
    else hipLaunchKernelGGL(k_preproc, grid2d(wp, hp), dim3(256), 0, st, rgb, w, h, out, wp, hp);
}

}  // namespace rife
#include "graph_run.h"
namespace rife {

static int run_assemble(const rife_hip& E, Ctx& c, int b, float timestep, const float* tsp = nullptr) {
    hipStream_t st = c.stream;
    Timed t(E.prof, "assemble", 0, st);
    const int s = E.blk[b].scale;
    dim3 g = grid2d(c.wp / s, c.hp / s);
    if (b == 0) hipLaunchKernelGGL(k_assemble0, g, dim3(256), 0, st, c.img0, c.img1, timestep, tsp, c.X, c.wp, c.hp);
    else if (s == 4) hipLaunchKernelGGL(k_assemble<4>, g, dim3(256), 0, st, c.img0, c.img1, timestep, tsp, c.F, c.M, c.X, c.wp, c.hp);
    else if (s == 2) hipLaunchKernelGGL(k_assemble<2>, g, dim3(256), 0, st, c.img0, c.img1, timestep, tsp, c.F, c.M, c.X, c.wp, c.hp);
    else hipLaunchKernelGGL(k_assemble<1>, g, dim3(256), 0, st, c.img0, c.img1, timestep, tsp, c.F, c.M, c.X, c.wp, c.hp);
    HIPCHK(hipGetLastError());
    return 0;
}

// block 3 of rife-v4.6: frames + F, M -> the first S16 trunk tensor (stem_rs.h); two workgroups per CU, all resident
static int launch_stem_rs(const rife_hip& E, Ctx& c, const rife_hip::Block& B, unsigned char* out, int Hq, int Wq, float timestep, const float* tsp) {
    {
        int dev = 0; (void)hipGetDevice(&dev);
        static std::mutex mu; static std::map<int, bool> done;
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem_rs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, SRS_LDS));
            done[dev] = true;
        }
    }
    const S16Geom G(Hq, Wq);
    StemRsArgs a;
    a.img0 = c.img0; a.img1 = c.img1; a.F = c.F; a.M = c.M;
    a.w0 = B.stem0.d_wh; a.bias0 = B.stem0.d_bias; a.slope0 = B.stem0.d_slope;
    a.w1 = B.stem1.d_whp; a.bias1 = B.stem1.d_bias; a.slope1 = B.stem1.d_slope;
    a.out = out; a.timestep = timestep; a.tsp = tsp; a.wp = c.wp; a.hp = c.hp; a.Hq = Hq; a.Wq = Wq; a.pitch = G.pitch; a.plane = G.plane();
    a.nunits = ((Wq + SRS_SW - 1) / SRS_SW) * Hq;
    const int nwg = std::min(2 * device_cus(), a.nunits);
    hipLaunchKernelGGL((stem_rs_kernel<0>), dim3(nwg), dim3(SRS_NTHR), SRS_LDS, c.stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("stem_rs launch: ") + hipGetErrorString(e));
    return 0;
}
// block 3 of rife-v4.6: last S16 trunk tensor + F, M + frames -> u8 frame (tail_rs.h); two workgroups per CU, all resident
static int launch_tail_rs(const rife_hip::Block& B, const unsigned char* in, int Hq, int Wq, const FinalArgs& fin, hipStream_t st) {
    {
        int dev = 0; (void)hipGetDevice(&dev);
        static std::mutex mu; static std::map<int, bool> done;
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(tail_rs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, TRS_LDS));
            done[dev] = true;
        }
    }
    const S16Geom G(Hq, Wq);
    TailRsArgs a;
    a.in = in; a.w = B.head.d_wh; a.bias = B.head.d_bias; a.img0 = fin.img0; a.img1 = fin.img1; a.F = fin.F; a.M = fin.M; a.out = fin.out;
    a.w_ = fin.w; a.h_ = fin.h; a.wp = fin.wp; a.hp = fin.hp; a.Hq = Hq; a.Wq = Wq; a.pitch = G.pitch; a.plane = G.plane();
    a.nunits = ((Wq + 31) / 32) * Hq;
    const int nwg = std::min(2 * device_cus(), a.nunits);
    hipLaunchKernelGGL((tail_rs_kernel<0>), dim3(nwg), dim3(TRS_NTHR), TRS_LDS, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("tail_rs launch: ") + hipGetErrorString(e));
    return 0;
}

// can block b's two stems run as one stem_rs launch?  (64-channel block 3 at scale 1 on the S16 trunk, 12 -> 32 -> 64 channels, uniform shapes)
static bool block_on_stem_rs(const rife_hip& E, const Ctx& c, int b) {
    const rife_hip::Block& B = E.blk[b];
    return E.stem_rs && b == 3 && B.scale == 1 && B.c == 64 && B.stem0.d_wh && B.stem0.cout == 32 && B.stem1.d_whp && B.stem1.cout == 64 &&
           trunk_h2() && g_fuse_stem && (c.hp % 4) == 0 && (c.wp % 4) == 0 &&
           (long long)c.wp * c.hp <= (1ll << 27);                       // the kernel addresses F (16 B per pixel) with 32-bit byte offsets; larger frames take the tile stems
}

// One IFBlock: stems, 8 residual convs, head -> flow[b]   (flownet.param:11-46, 63-98, 116-151, 166-201)
// which trunk kernel serves block b at this frame size: 0 = conv_t64 / conv_rs (fine blocks), 1 = conv_row (coarse blocks, small grids)
static bool block_on_row_kernel(const rife_hip& E, const Ctx& c, int b) {
    const rife_hip::Block& B = E.blk[b];
    const int s = B.scale, Ht = c.hp / s / 4, Wt = c.wp / s / 4;
    const int ptiles = ((Ht + 7) / 8) * ((Wt + 31) / 32), cus = device_cus(true);      // kernel selection never depends on a CU partition: same bytes on every stream
    const bool row_small = b == 2 && B.c == 96 && (ptiles <= cus || (E.ks_mask & 4));          // fewer 8 x 32 tiles than the chip has CUs (or conv_ks at every size)
    return (b == 1 && B.c == 128) || (b == 0 && B.c == 192 && ((Wt + 31) / 32) * Ht <= cus * 5 / 8) || row_small;      // MI355X: 160 of 256
}

// Does block b run on S16 trunk tensors (conv_rs / conv_t64 / conv_row) at this frame size?  Blocks 3 / 2 on the persistent kernels, the coarse
// blocks on the row kernel where block_on_row_kernel says so; never for rife-v4 (4.0), RIFE_HIP_T64=0, or a tensor of 4 GB and more (the
// kernels address S16 tensors with 32-bit byte offsets).
static bool block_on_s16(const rife_hip& E, const Ctx& c, int b) {
    const rife_hip::Block& B = E.blk[b];
    const int s = B.scale, Ht = c.hp / s / 4, Wt = c.wp / s / 4;
    const bool rowk = block_on_row_kernel(E, c, b);
    bool s16 = E.t64 && !E.v40 && trunk_h2() && B.stem1.d_whp && B.head.d_wh && B.head.epi == EPI_DECONV_PS &&
               ((b == 3 && B.c == 64) || (b == 2 && B.c == 96) || rowk);
    for (int i = 0; i < 8 && s16; i++) s16 = B.res[i].d_t64 != nullptr && (!(rowk && B.c == 96) || B.res[i].d_row != nullptr);
    return s16 && (unsigned long long)S16Geom(Ht, Wt).bytes(B.c) < (1ull << 32);
}
// the block's two S16 tensors (trunk ping / pong), allocated on first use with their zero borders: a workspace only carries the tensors of
// the blocks that really run on the S16 kernels (TTA: 16 workspaces)
static int ensure_s16(Ctx& c, int b, int Ht, int Wt, int C) {
    if (c.P[b][0] && c.P[b][1]) return 0;
    const size_t nb = S16Geom(Ht, Wt).bytes(C);
    int rc;
    if (c.stream) {      // a lazy hipMalloc inside a hipGraph capture would be illegal: the warm-up pass before a capture allocates everything (run_v4_replay)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c.stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(RIFE_HIP_EHIP, "S16 trunk tensors requested while the stream is capturing");
    }
    for (int k = 0; k < 2; k++) {
        if ((rc = dalloc(c, c.P[b][k], nb))) {
            if (k == 1) {                                                // the first buffer goes back: it is the newest entry of the workspace's allocation list
                if (!c.allocs.empty() && c.allocs.back() == (void*)c.P[b][0]) c.allocs.pop_back();
                (void)hipFree(c.P[b][0]);
            }
            c.P[b][0] = c.P[b][1] = nullptr;
            return rc;
        }
        if (c.stream) HIPCHK(hipMemsetAsync(c.P[b][k], 0, nb, c.stream));
        else HIPCHK(hipMemset(c.P[b][k], 0, nb));
    }
    return 0;
}

enum { PH_STEMS = 1, PH_TRUNK = 2, PH_HEAD = 4, PH_ALL = 7 };
// phases != PH_ALL (rife_hip_process_batch): the S16 path only; PH_TRUNK is then the caller's batched launch
// Can the flow update after block b - 1 be left to block b's fused stem (stem_fused.h UPD)?  Blocks 2 and 3 of rife-v4.6 only: their stems
// visit every full-resolution pixel.
static bool flow_update_fused_into(const rife_hip& E, const Ctx& c, int b) {
    return E.fuse_flow && !E.v40 && (b == 2 || b == 3) && c.F2 && E.blk[b].stem0.d_wh && trunk_h2() && g_fuse_stem;
}

// upd_flow != null: the flow of block b - 1, whose update of F, M this block's stem applies itself (flow_update_fused_into); F, M swap with F2, M2
static int run_block_convs(const rife_hip& E, Ctx& c, int b, float timestep, const FinalArgs* fin = nullptr, const float* tsp = nullptr, int phases = PH_ALL,
                           const float* upd_flow = nullptr, const float* first_flow = nullptr) {
    const rife_hip::Block& B = E.blk[b];
    hipStream_t st = c.stream;
    const int s = B.scale, Hb = c.hp / s, Wb = c.wp / s;
    const int xin_ld = b == 0 ? 8 : 16;
    int rc;
    // block 3: one row-streaming kernel for the assembly and both stems (stem_rs.h), launched where stem 1 used to be
    const bool srs = !upd_flow && block_on_stem_rs(E, c, b) && block_on_s16(E, c, b);
    if (!(phases & PH_STEMS) || srs) goto after_stem0;
    if (b == 0 && (rc = run_assemble(E, c, 0, timestep, tsp))) return rc;
    if (b > 0 && B.stem0.d_wh && trunk_h2() && g_fuse_stem) {
        // assemble + stem-0 in one kernel (stem_fused.h): the block input never goes to HBM
        Timed t(E.prof, B.stem0.cls, B.stem0.flops_per_pixel * (Hb / 2) * (Wb / 2), st);
        StemFusedArgs fa;
        fa.img0 = c.img0; fa.img1 = c.img1; fa.F = c.F; fa.M = c.M; fa.wpk = B.stem0.d_wh; fa.bias = B.stem0.d_bias; fa.slope = B.stem0.d_slope;
        fa.out = c.S1; fa.timestep = timestep; fa.tsp = tsp; fa.wp = c.wp; fa.hp = c.hp; fa.Ho = Hb / 2; fa.Wo = Wb / 2; fa.out_ld = B.c / 2; fa.Cout = B.c / 2;
        fa.tiles_x = (fa.Wo + 31) / 32;
        const int nb = fa.tiles_x * ((fa.Ho + 3) / 4);
        {
            static std::mutex fmu; static std::map<int, bool> fdone;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(fmu);
            if (!fdone[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<2, 2, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<4, 2, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>()));
                fdone[dev] = true;
            }
        }
        if (upd_flow) {
            if (s > 2 || !c.F2) return fail(RIFE_HIP_EINVAL, "no fused flow update for this block");
            fa.pend.flow = upd_flow; fa.pend.Fw = c.F2; fa.pend.Mw = c.M2;
            if (s == 2) hipLaunchKernelGGL((stem0_fused_kernel<2, 2, 0, true>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
            else hipLaunchKernelGGL((stem0_fused_kernel<1, 1, 256, true>), dim3(nb), dim3(512), (stemf_lds_bytes<1, 256>()), st, fa);
            std::swap(c.F, c.F2); std::swap(c.M, c.M2);
        } else if (first_flow) {      // block 1 right after block 0: F, M are not materialised yet, the stem samples the first update itself (first_flow_merged)
            if (s != 4) return fail(RIFE_HIP_EINVAL, "the first flow update is sampled by the scale-4 stem only");
            fa.pend.flow = first_flow;
            hipLaunchKernelGGL((stem0_fused_kernel<4, 2, 0, 2>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
        } else if (s == 4) hipLaunchKernelGGL((stem0_fused_kernel<4, 2>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
        else if (s == 2) hipLaunchKernelGGL((stem0_fused_kernel<2, 2>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), st, fa);
        else hipLaunchKernelGGL((stem0_fused_kernel<1, 1, 256>), dim3(nb), dim3(512), (stemf_lds_bytes<1, 256>()), st, fa);      // 64-byte swizzled records, three workgroups per CU
        HIPCHK(hipGetLastError());
    } else {
        if (upd_flow || first_flow) return fail(RIFE_HIP_EINVAL, "fused flow update without the fused stem");
        if (b > 0 && (rc = run_assemble(E, c, b, timestep, tsp))) return rc;
        Timed t(E.prof, B.stem0.cls, B.stem0.flops_per_pixel * (Hb / 2) * (Wb / 2), st);
        if ((rc = launch_conv(B.stem0, {c.X, xin_ld, 0}, Hb, Wb, {c.S1, B.c / 2, 0}, nullptr, st))) return rc;
    }
after_stem0:
    FinalArgs fin_now;                                                   // the fused tail reads the F, M that are current AFTER this block's stem
    if (fin) { fin_now = *fin; fin_now.F = c.F; fin_now.M = c.M; fin = &fin_now; }
    const int Ht = Hb / 4, Wt = Wb / 4;
    // S16 trunk tensors: blocks 3 / 2 on the persistent LDS-DMA kernel (conv_t64.h; block 2 only when its grid fills a good part of the
    // chip), the coarse blocks 1 / 0 on the one-pass row kernel (conv_row.h).  (Blocks 1 / 0 as N-tiles of 64 output channels on the
    // persistent kernel were measured too: 4K trunk_b1 0.300 vs 0.285 ms per pair, trunk_b0 0.228 vs 0.206 - a chain of 8 - 12 dependent
    // steps whose fixed cost exceeds a step's matrix work at these sizes.)
    // block 0 on the row kernel only while its grid is small: at 4K all 272 workgroups stream the same 663 KB of weights through the L2 at
    // once (0.239 vs 0.208 ms per pair for the per-tile kernel), at 1080p (68 workgroups) it wins (0.133 vs 0.152); block 1 wins at both
    // block 2 on small grids (<= 256 tiles of 8 x 32: fewer tiles than CUs): the persistent kernel (one workgroup per CU for 96 channels) has at most one
    // tile per workgroup there and fills only part of the chip: 1080p (136 tiles) trunk_b2 0.229 -> 0.179 ms per pair on the row kernel, 4K (510 tiles)
    // 0.387 -> 0.401; block 3 (64 channels, two workgroups per CU) stays on the persistent kernel at every size (1080p 0.225 vs 0.233)
    const bool rowk = block_on_row_kernel(E, c, b);
    if (block_on_s16(E, c, b)) {
        if ((rc = ensure_s16(c, b, Ht, Wt, B.c))) return rc;
        unsigned char* const PA = c.P[b][0];
        unsigned char* const PB = c.P[b][1];
        // stem-1 writes the first S16 tensor, eight persistent trunk launches ping-pong between the two, the head reads the last one
        const S16Geom G(Ht, Wt);
        if ((phases & PH_STEMS) && srs) {
            Timed t(E.prof, "stems_b3", B.stem0.flops_per_pixel * (Hb / 2) * (Wb / 2) + B.stem1.flops_per_pixel * Ht * Wt, st);
            if ((rc = launch_stem_rs(E, c, B, PA, Ht, Wt, timestep, tsp))) return rc;
        } else if (phases & PH_STEMS) {
            Timed t(E.prof, B.stem1.cls, B.stem1.flops_per_pixel * Ht * Wt, st);
            if ((rc = launch_conv(B.stem1, {c.S1, B.c / 2, 0}, Hb / 2, Wb / 2, {reinterpret_cast<float*>(PA), B.c, 0}, nullptr, st, nullptr, G.pitch, G.plane()))) return rc;
        }
        unsigned char *pc = PA, *pn = PB;
        if (phases & PH_TRUNK) for (int i = 0; i < 8; i++) {
            if (!rowk && E.rs && E.rs2 && B.c == 64 && !(i & 1) && rs2_applies(Ht, Wt)) {      // layers i, i + 1 in one launch (conv_rs2.h)
                Timed t(E.prof, B.res[i].cls, (B.res[i].flops_per_pixel + B.res[i + 1].flops_per_pixel) * Ht * Wt, st);
                if ((rc = launch_rs2(B.res[i], B.res[i + 1], pc, pn, Ht, Wt, st, (i & 2) != 0))) return rc;
                std::swap(pc, pn); i++;
                continue;
            }
            Timed t(E.prof, B.res[i].cls, B.res[i].flops_per_pixel * Ht * Wt, st);
            if (rowk && ks_serves(E.ks_mask, B.c)) rc = launch_ks(B.res[i], pc, pn, Ht, Wt, st);
            else if (rowk) rc = launch_row(B.res[i], pc, pn, Ht, Wt, st);
            else if (E.rs && B.c == 64 && (Ht + 1) / 2 >= RS_MIN_PAIRS) rc = launch_rs(B.res[i], pc, pn, Ht, Wt, st, (i & 1) != 0);      // tiny tensors: conv_t64
            else rc = launch_t64(B.res[i], pc, pn, Ht, Wt, st, (i & 1) == 0);
            if (rc) return rc;
            std::swap(pc, pn);
        }
        if (!(phases & PH_HEAD)) return 0;
        Timed t(E.prof, B.head.cls, B.head.flops_per_pixel * Ht * Wt, st);      // eight layers: the trunk output is back in PA
        // the row-streaming tail where every workgroup has at least 16 steps to amortise its prologue over (4K: 32; 1080p: 8 - there the tile kernel
        // is as fast or faster: head_b3 0.037 vs 0.039 ms per pair, same call)
        if (fin && E.tail_rs && b == 3 && B.c == 64 && B.head.cout == 24 && B.head.d_wh && Ht * 4 == c.hp && Wt * 4 == c.wp &&
            (E.tail_rs_always || ((Wt + 31) / 32) * Ht >= 32 * device_cus(true)))
            return launch_tail_rs(B, PA, Ht, Wt, *fin, st);
        return launch_conv(B.head, {reinterpret_cast<float*>(PA), B.c, 0}, Ht, Wt, {c.flow[b], 8, 0}, nullptr, st, fin, G.pitch, G.plane());
    }
    if (phases != PH_ALL) return fail(RIFE_HIP_EINVAL, "phased block execution needs the S16 trunk path");
    float* const stem_out = E.v40 ? c.T2 : c.T0;
    {
        Timed t(E.prof, B.stem1.cls, B.stem1.flops_per_pixel * (Hb / 4) * (Wb / 4), st);
        if ((rc = launch_conv(B.stem1, {c.S1, B.c / 2, 0}, Hb / 2, Wb / 2, {stem_out, B.c, 0}, nullptr, st))) return rc;
    }
    float* cur = stem_out; float* nxt = E.v40 ? c.T0 : c.T1;
    for (int i = 0; i < 8; i++) {
        Timed t(E.prof, B.res[i].cls, B.res[i].flops_per_pixel * Ht * Wt, st);
        if ((rc = launch_conv(B.res[i], {cur, B.c, 0}, Ht, Wt, {nxt, B.c, 0}, nullptr, st))) return rc;   // v4.6: skip folded into the weights
        if (E.v40 && i == 0) { cur = c.T0; nxt = c.T1; }
        else std::swap(cur, nxt);
    }
    if (E.v40) {   // add_0 / add_3 / add_8 / add_12 (models/rife-v4/flownet.param): trunk output + stem output, no activation
        Timed t(E.prof, "v40_block_add", 0, st);
        const size_t n4 = (size_t)Ht * Wt * B.c / 4;
        hipLaunchKernelGGL(k_add_inplace, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<float4*>(cur), reinterpret_cast<const float4*>(c.T2), n4);
        HIPCHK(hipGetLastError());
    }
    {
        Timed t(E.prof, B.head.cls, B.head.flops_per_pixel * Ht * Wt, st);
        if ((rc = launch_conv(B.head, {cur, B.c, 0}, Ht, Wt, {c.flow[b], 8, 0}, nullptr, st, fin))) return rc;
    }
    return 0;
}

static int run_flow_update(const rife_hip& E, Ctx& c, int b) {
    hipStream_t st = c.stream;
    Timed t(E.prof, "flow_update", 0, st);
    dim3 g = grid2d(c.wp, c.hp);
    if (E.v40) {   // Interp x(2 x scale) of the 5-channel head output, then F (+)= u[0:4] * (2 x scale), M (+)= u[4]
        if (b == 0) hipLaunchKernelGGL((k_flow_update<16, true>), g, dim3(256), 0, st, c.flow[0], c.F, c.M, c.wp, c.hp);
        else if (b == 1) hipLaunchKernelGGL((k_flow_update<8, false>), g, dim3(256), 0, st, c.flow[1], c.F, c.M, c.wp, c.hp);
        else if (b == 2) hipLaunchKernelGGL((k_flow_update<4, false>), g, dim3(256), 0, st, c.flow[2], c.F, c.M, c.wp, c.hp);
        else hipLaunchKernelGGL((k_flow_update<2, false>), g, dim3(256), 0, st, c.flow[3], c.F, c.M, c.wp, c.hp);
        HIPCHK(hipGetLastError());
        return 0;
    }
    if (b == 0) hipLaunchKernelGGL((k_flow_update<8, true>), g, dim3(256), 0, st, c.flow[0], c.F, c.M, c.wp, c.hp);
    else if (b == 1) hipLaunchKernelGGL((k_flow_update<4, false>), g, dim3(256), 0, st, c.flow[1], c.F, c.M, c.wp, c.hp);
    else hipLaunchKernelGGL((k_flow_update<2, false>), g, dim3(256), 0, st, c.flow[2], c.F, c.M, c.wp, c.hp);
    HIPCHK(hipGetLastError());
    return 0;
}

// RIFE::process_v4, non-TTA branch (rife.cpp:2931-3173) on device-resident frames.
static int run_v4(const rife_hip& E, Ctx& c, const uint8_t* d_in0, const uint8_t* d_in1, float timestep, uint8_t* d_out, const float* tsp = nullptr) {
    hipStream_t st = c.stream;
    int rc;
    {
        Timed t(E.prof, "preproc", 0, st);
        launch_preproc(st, d_in0, c.w, c.h, c.img0, c.wp, c.hp);
        launch_preproc(st, d_in1, c.w, c.h, c.img1, c.wp, c.hp);
        HIPCHK(hipGetLastError());
    }
    const bool fuse_tail = !E.v40 && trunk_h2() && g_head_h2 && g_fuse_tail && E.blk[3].head.d_wh != nullptr;
    FinalArgs fin{c.img0, c.img1, c.F, c.M, d_out, c.w, c.h, c.wp, c.hp};
    const float* pending = nullptr;                                      // flow whose update of F, M the next block's stem applies
    // The update after block 0 never reaches HBM on its own (round 5): block 1's scale-4 stem samples it from flow0 (assemble_pixel UPD = 2) and ONE pass after
    // block 1 writes F, M with both updates applied (k_flow_update2) - bit for bit the tensors of the two-kernel sequence, one launch and 20 B / pixel of writes +
    // 20 B / pixel of reads less.  RIFE_HIP_MERGE_FLOW0=0 (A/B, test build): the three separate updates.
    const bool merge_env = read_switches().merge_flow0;      // per call
    const bool merge0 = merge_env && !E.v40 && trunk_h2() && g_fuse_stem && E.blk[1].stem0.d_wh != nullptr && E.blk[1].scale == 4 && !flow_update_fused_into(E, c, 1) &&
                        !flow_update_fused_into(E, c, 2);
    for (int b = 0; b < 4; b++) {
        if ((rc = run_block_convs(E, c, b, timestep, (b == 3 && fuse_tail) ? &fin : nullptr, tsp, PH_ALL, pending, (merge0 && b == 1) ? c.flow[0] : nullptr))) return rc;
        pending = nullptr;
        if (merge0 && b == 0) continue;
        if (merge0 && b == 1) {
            Timed t(E.prof, "flow_update", 0, st);
            hipLaunchKernelGGL((k_flow_update2<8, 4>), grid2d(c.wp, c.hp), dim3(256), 0, st, c.flow[0], c.flow[1], c.F, c.M, c.wp, c.hp);
            HIPCHK(hipGetLastError());
            continue;
        }
        if (b < 3 && flow_update_fused_into(E, c, b + 1)) pending = c.flow[b];
        else if ((b < 3 || E.v40) && (rc = run_flow_update(E, c, b))) return rc;
    }
    if (E.v40) {
        Timed t(E.prof, "final", 0, st);
        hipLaunchKernelGGL(k_blend_final, grid2d(c.w, c.h), dim3(256), 0, st, c.img0, c.img1, c.F, c.M, d_out, c.w, c.h, c.wp, c.hp);
        HIPCHK(hipGetLastError());
    } else if (!fuse_tail) {
        Timed t(E.prof, "final", 0, st);
        hipLaunchKernelGGL(k_final, grid2d(c.w, c.h), dim3(256), 0, st, c.img0, c.img1, c.F, c.M, c.flow[3], d_out, c.w, c.h, c.wp, c.hp);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// RIFE::process_v4 for G (2..4) pairs in LOCKSTEP (rife_hip_process_batch, SURVEY 8f-2 "batch >= 2 pairs per launch for the coarse blocks"):
// every pair keeps its own workspace and stream, so the fine blocks of different pairs overlap as before; the eight trunk layers of a block that
// runs on conv_row_kernel (the coarse blocks 1 / 0, block 2 on small grids; flownet.param:14-42, 66-94) are ONE launch per layer for all pairs
// (gridDim.y = G) on the first pair's stream, between two event hand-offs.  The workgroups of all pairs stream the layer's weights from the L2
// together, and the coarse grids - 68 / 255 workgroups per pair at 1080p - fill the chip in one round instead of G.
// Same kernels, same arguments per tensor: the frames are bit-identical to G single calls.
static int run_v4_group(const rife_hip& E, Ctx* const* cs, int G, const uint8_t* const* d_in0, const uint8_t* const* d_in1, const float* ts, uint8_t* const* d_out) {
    int rc;
    for (int g = 0; g < G; g++) {
        Ctx& c = *cs[g];
        if (!c.ev_group) HIPCHK(hipEventCreateWithFlags(&c.ev_group, hipEventDisableTiming));
        Timed t(E.prof, "preproc", 0, c.stream);
        launch_preproc(c.stream, d_in0[g], c.w, c.h, c.img0, c.wp, c.hp);
        launch_preproc(c.stream, d_in1[g], c.w, c.h, c.img1, c.wp, c.hp);
        HIPCHK(hipGetLastError());
    }
    const bool fuse_tail = trunk_h2() && g_head_h2 && g_fuse_tail && E.blk[3].head.d_wh != nullptr;
    const float* pend[4] = {nullptr, nullptr, nullptr, nullptr};         // per pair: flow whose update the next block's stem applies (run_v4)
    auto after_block = [&](Ctx& c, int g, int b) -> int {
        if (b < 3 && flow_update_fused_into(E, c, b + 1)) { pend[g] = c.flow[b]; return 0; }
        return b < 3 ? run_flow_update(E, c, b) : 0;
    };
    for (int b = 0; b < 4; b++) {
        const rife_hip::Block& B = E.blk[b];
        const bool batched = G >= 2 && block_on_row_kernel(E, *cs[0], b) && block_on_s16(E, *cs[0], b);
        for (int g = 0; g < G; g++) {
            Ctx& c = *cs[g];
            FinalArgs fin{c.img0, c.img1, c.F, c.M, d_out[g], c.w, c.h, c.wp, c.hp};
            if (!batched) {
                if ((rc = run_block_convs(E, c, b, ts[g], (b == 3 && fuse_tail) ? &fin : nullptr, nullptr, PH_ALL, pend[g]))) return rc;
                pend[g] = nullptr;
                if ((rc = after_block(c, g, b))) return rc;
            } else {
                if ((rc = run_block_convs(E, c, b, ts[g], nullptr, nullptr, PH_STEMS, pend[g]))) return rc;
                pend[g] = nullptr;
                if (g > 0) HIPCHK(hipEventRecord(c.ev_group, c.stream));
            }
        }
        if (!batched) continue;
        hipStream_t lead = cs[0]->stream;
        for (int g = 1; g < G; g++) HIPCHK(hipStreamWaitEvent(lead, cs[g]->ev_group, 0));
        {
            const int s = B.scale, Ht = cs[0]->hp / s / 4, Wt = cs[0]->wp / s / 4;
            const unsigned char* pin[4]; unsigned char* pout[4];
            for (int i = 0; i < 8; i++) {
                for (int g = 0; g < G; g++) { pin[g] = cs[g]->P[b][i & 1]; pout[g] = cs[g]->P[b][(i & 1) ^ 1]; }
                Timed t(E.prof, B.res[i].cls, B.res[i].flops_per_pixel * Ht * Wt * G, lead);
                if (ks_serves(E.ks_mask, B.c)) rc = launch_ks(B.res[i], nullptr, nullptr, Ht, Wt, lead, G, pin, pout);
                else rc = launch_row(B.res[i], nullptr, nullptr, Ht, Wt, lead, G, pin, pout);
                if (rc) return rc;
            }
        }
        HIPCHK(hipEventRecord(cs[0]->ev_group, lead));
        for (int g = 0; g < G; g++) {
            Ctx& c = *cs[g];
            if (g > 0) HIPCHK(hipStreamWaitEvent(c.stream, cs[0]->ev_group, 0));
            if ((rc = run_block_convs(E, c, b, ts[g], nullptr, nullptr, PH_HEAD))) return rc;
            if ((rc = after_block(c, g, b))) return rc;
        }
    }
    for (int g = 0; g < G; g++)
        if (!fuse_tail) {
            Ctx& c = *cs[g];
            hipLaunchKernelGGL(k_final, grid2d(c.w, c.h), dim3(256), 0, c.stream, c.img0, c.img1, c.F, c.M, c.flow[3], d_out[g], c.w, c.h, c.wp, c.hp);
            HIPCHK(hipGetLastError());
        }
    return 0;
}

// Plain v4 pass replayed from a hipGraph for small frames (<= 1920 x 1088 padded), opt-in with RIFE_HIP_GRAPH=1: one graph launch
// instead of ~50 kernel launches (+ two small device copies into the fixed staging buffers).  Measured on MI355X
// (tools/graph_bench.py, profiler off): 1080p 1.116 vs 1.119 ms per pair, 720p 0.782 vs 0.782, 360p 0.673 vs 0.674 - no gain: the
// chain of ~50 dependent kernels (fill / drain of each launch), not host launch overhead, sets the floor, and a replayed graph
// executes the same chain.  Kept off by default; the profiler (events around every launch) bypasses it.
static inline bool use_graph() { return process_switches().use_graph; }

static int run_v4_replay(const rife_hip& E, Ctx& c, const uint8_t* d_in0, const uint8_t* d_in1, float timestep, uint8_t* d_out) {
    const bool eligible = use_graph() && !E.prof.on && c.d_ts && (size_t)c.wp * c.hp <= (size_t)1920 * 1088;
    if (!eligible) return run_v4(E, c, d_in0, d_in1, timestep, d_out);
    hipStream_t st = c.stream;
    const size_t nbytes = (size_t)c.w * c.h * 3;
    if (d_in0 != c.d_in0) HIPCHK(hipMemcpyAsync(c.d_in0, d_in0, nbytes, hipMemcpyDeviceToDevice, st));
    if (d_in1 != c.d_in1) HIPCHK(hipMemcpyAsync(c.d_in1, d_in1, nbytes, hipMemcpyDeviceToDevice, st));
    uint32_t bits; std::memcpy(&bits, &timestep, 4);
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)c.d_ts, (int)bits, 1, st));
    int rc = 0;
    if (c.gexec) HIPCHK(hipGraphLaunch(c.gexec, st));
    else if (!c.g_warm) {
        if ((rc = run_v4(E, c, c.d_in0, c.d_in1, timestep, c.d_out, c.d_ts))) return rc;
        c.g_warm = true;
    } else {
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        rc = run_v4(E, c, c.d_in0, c.d_in1, timestep, c.d_out, c.d_ts);
        const hipError_t e = hipStreamEndCapture(st, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e != hipSuccess || !graph) return fail(RIFE_HIP_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        const hipError_t ei = hipGraphInstantiate(&c.gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) { c.gexec = nullptr; return fail(RIFE_HIP_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); }
        HIPCHK(hipGraphLaunch(c.gexec, st));
    }
    if (d_out != c.d_out) HIPCHK(hipMemcpyAsync(d_out, c.d_out, nbytes, hipMemcpyDeviceToDevice, st));
    return 0;
}

// RIFE::process_v4 with -x and/or -z (rife.cpp:2534-2930 spatial TTA, 3036-3135 temporal only; CPU twin 3246-4145):
// nori = 8 orientations or 1, ntemp = 2 directions (in0,in1,t) / (in1,in0,1-t) or 1.  Per IFBlock stage the flows of
// all passes are merged (temporal first, then spatial, like the reference) before any pass goes on.
static int run_v4_tta(const rife_hip& E, hipStream_t st, const uint8_t* d_in0, const uint8_t* d_in1, int w, int h, float timestep, uint8_t* d_out) {
    const int nori = E.tta ? 8 : 1, ntemp = E.tta_temporal ? 2 : 1;
    const int wp = (w + 31) / 32 * 32, hp = (h + 31) / 32 * 32;
    constexpr int NL = rife_hip::NLANE;
    const bool lanes = nori == 8;          // the 8 orientations are independent between consensus points: 4 worker streams
    int rc;
    // The four lanes are ordinary streams.  Round 6 measured them as CU-masked streams (RIFE_HIP_TTA_LANE_PARTS=2 / 4, test build: two per half / one per quarter
    // of the compute units - the layout that gives plain 4K pairs + 5 %, profiles/r6/layout_sweep.txt): SLOWER here, 22.4 - 22.7 / 22.0 - 22.4 against 25.3 - 25.4
    // frames/s at 4K and 59 - 60 against 80 at 1080p, identical bytes (profiles/r6/ab_tta_lane_parts.txt): the lanes meet at a consensus after every block, and a
    // lane that is done early leaves its part of the chip idle where an ordinary stream's neighbours would take it over.
    const int lane_parts = lanes ? process_switches().tta_lane_parts : 0;
    if (lanes && !E.tta_lane[0]) {
        const int ncu = device_cus(true);
        for (int l = 0; l < NL; l++) {
            if (lane_parts > 1) {
                std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
                int mine = 0;
                for (int cu = 0; cu < ncu; cu++) if (cu % lane_parts == l % lane_parts) { mask[cu / 32] |= 1u << (cu % 32); mine++; }
                HIPCHK(hipExtStreamCreateWithCUMask(&E.tta_lane[l], (uint32_t)mask.size(), mask.data()));
                E.tta_lane_cus = mine;
            } else HIPCHK(hipStreamCreateWithFlags(&E.tta_lane[l], hipStreamNonBlocking));
        }
        for (int i = 0; i < 6; i++) {
            HIPCHK(hipEventCreateWithFlags(&E.tta_fork[i], hipEventDisableTiming));
            for (int l = 0; l < NL; l++) HIPCHK(hipEventCreateWithFlags(&E.tta_join[i][l], hipEventDisableTiming));
        }
    }
    auto lane_of = [&](int ti) { return lanes ? E.tta_lane[ti % NL] : st; };
    const int caller_budget = tl_cu_budget, lane_budget = lanes && lane_parts > 1 ? E.tta_lane_cus : tl_cu_budget;      // persistent kernels size their grids for the stream they are enqueued on
    struct BudgetGuard { int keep; ~BudgetGuard() { tl_cu_budget = keep; } } budget_guard{caller_budget};
    for (int dir = 0; dir < ntemp; dir++)
        for (int ti = 0; ti < nori; ti++) {
            auto& up = E.tta_ctx[dir][ti];
            if (!up) up.reset(new Ctx);
            Ctx& c = *up;
            c.stream = lane_of(ti);
            const bool swap = ti >= 4;
            // per-layer scratch is shared by the passes of one lane (they run back to back on that lane's stream)
            const int owner = lanes ? ti % NL : 0;
            const Ctx* scratch = (dir == 0 && ti == owner) ? nullptr : E.tta_ctx[0][owner].get();
            if ((rc = ensure_ctx_dims(c, swap ? h : w, swap ? w : h, swap ? hp : wp, swap ? wp : hp, scratch, dir == 0, true))) return rc;
            if (dir == 1) { c.img0 = E.tta_ctx[0][ti]->img1; c.img1 = E.tta_ctx[0][ti]->img0; }   // reversed pass sees the frames swapped
        }
    int sync_id = 0;
    auto fork = [&]() -> int {             // lanes wait for everything enqueued on the caller's stream so far
        if (!lanes) return 0;
        HIPCHK(hipEventRecord(E.tta_fork[sync_id], st));
        for (int l = 0; l < NL; l++) HIPCHK(hipStreamWaitEvent(E.tta_lane[l], E.tta_fork[sync_id], 0));
        return 0;
    };
    auto join = [&]() -> int {             // the caller's stream waits for all lanes
        if (!lanes) return 0;
        for (int l = 0; l < NL; l++) {
            HIPCHK(hipEventRecord(E.tta_join[sync_id][l], E.tta_lane[l]));
            HIPCHK(hipStreamWaitEvent(st, E.tta_join[sync_id][l], 0));
        }
        sync_id++;
        return 0;
    };
    {
        Timed t(E.prof, "preproc", 0, st);
        if (nori == 8) {
            Ptr8 a, b;
            for (int ti = 0; ti < 8; ti++) { a.p[ti] = E.tta_ctx[0][ti]->img0; b.p[ti] = E.tta_ctx[0][ti]->img1; }
            hipLaunchKernelGGL(k_preproc_tta, tta_grid(wp, hp, 4), tta_block(4), 0, st, d_in0, w, h, a, wp, hp);
            hipLaunchKernelGGL(k_preproc_tta, tta_grid(wp, hp, 4), tta_block(4), 0, st, d_in1, w, h, b, wp, hp);
        } else {
            launch_preproc(st, d_in0, w, h, E.tta_ctx[0][0]->img0, wp, hp);
            launch_preproc(st, d_in1, w, h, E.tta_ctx[0][0]->img1, wp, hp);
        }
        HIPCHK(hipGetLastError());
    }
    const bool fused_consensus = ntemp == 2 && nori == 8 && E.tta_consensus;
    for (int fi = 0; fi < 4; fi++) {
        const int Wf = wp / E.flow_div(fi), Hf = hp / E.flow_div(fi);
        if ((rc = fork())) return rc;
        tl_cu_budget = lane_budget;
        for (int ti = 0; ti < nori; ti++) {
            hipStream_t ls = lane_of(ti);
            for (int dir = 0; dir < ntemp; dir++) {
                Ctx& c = *E.tta_ctx[dir][ti];
                if ((rc = run_block_convs(E, c, fi, dir ? 1.f - timestep : timestep))) return rc;
            }
            if (ntemp == 2 && !fused_consensus) {
                Timed t(E.prof, "tta_merge", 0, ls);
                const size_t npix = (size_t)Wf * Hf;
                hipLaunchKernelGGL(k_v4_temporal_merge, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ls,
                                   E.tta_ctx[0][ti]->flow[fi], E.tta_ctx[1][ti]->flow[fi], npix);
                HIPCHK(hipGetLastError());
            }
        }
        if ((rc = join())) return rc;
        tl_cu_budget = caller_budget;
        if (fused_consensus) {       // -x -z: temporal and spatial consensus of the sixteen flow tensors in one pass (k_v4_consensus)
            Timed t(E.prof, "tta_merge", 0, st);
            Ptr8x2 f;
            for (int ti = 0; ti < 8; ti++) { f.f[ti] = E.tta_ctx[0][ti]->flow[fi]; f.r[ti] = E.tta_ctx[1][ti]->flow[fi]; }
            hipLaunchKernelGGL(k_v4_consensus, tta_grid(Wf, Hf, 32), tta_block(32), 0, st, f, Wf, Hf);
            HIPCHK(hipGetLastError());
        } else if (nori == 8) {
            Timed t(E.prof, "tta_merge", 0, st);
            for (int dir = 0; dir < ntemp; dir++) {
                Ptr8 f;
                for (int ti = 0; ti < 8; ti++) f.p[ti] = E.tta_ctx[dir][ti]->flow[fi];
                hipLaunchKernelGGL(k_v4_spatial_avg, tta_grid(Wf, Hf, 32), tta_block(32), 0, st, f, Wf, Hf);
            }
            HIPCHK(hipGetLastError());
        }
        if (fi < 3 || E.v40) {
            if (lanes) {   // flow updates run on the lanes; they must see the consensus written on the caller's stream
                HIPCHK(hipEventRecord(E.tta_fork[5], st));
                for (int l = 0; l < NL; l++) HIPCHK(hipStreamWaitEvent(E.tta_lane[l], E.tta_fork[5], 0));
            }
            tl_cu_budget = lane_budget;
            for (int ti = 0; ti < nori; ti++)
                for (int dir = 0; dir < ntemp; dir++)
                    if ((rc = run_flow_update(E, *E.tta_ctx[dir][ti], fi))) return rc;
            tl_cu_budget = caller_budget;
        }
    }
    Ptr16 outs;
    for (int i = 0; i < 16; i++) outs.p[i] = nullptr;
    {
        if ((rc = fork())) return rc;
        for (int ti = 0; ti < nori; ti++)
            for (int dir = 0; dir < ntemp; dir++) {
                Ctx& c = *E.tta_ctx[dir][ti];
                Timed t(E.prof, "final", 0, c.stream);
                if (E.v40) hipLaunchKernelGGL(k_blend_final_float, grid2d(c.wp, c.hp), dim3(256), 0, c.stream, c.img0, c.img1, c.F, c.M, c.outf, c.wp, c.hp);
                else hipLaunchKernelGGL(k_final_float, grid2d(c.wp, c.hp), dim3(256), 0, c.stream, c.img0, c.img1, c.F, c.M, c.flow[3], c.outf, c.wp, c.hp);
                outs.p[dir * 8 + ti] = c.outf;
            }
        if ((rc = join())) return rc;
        Timed t(E.prof, "final", 0, st);
        hipLaunchKernelGGL(k_postproc_tta, tta_grid(w, h, 16), tta_block(16), 0, st, outs, nori, ntemp, d_out, w, h, wp, hp);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

}  // namespace rife
