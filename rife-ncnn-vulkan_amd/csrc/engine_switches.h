// engine_switches.h: errors, the switch table (every environment variable the library reads), HIPCHK
// One translation unit (engine.hip includes the engine_*.h sections in dependency order; every function here is file-local).
// No include guard on purpose: a section is included exactly once, by engine.hip.

namespace rife {


// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return -code; }

// ================================================================================================================================================
// THE SWITCH TABLE: every environment variable this library reads, parsed in ONE function (read_switches); nothing else in csrc/ may call getenv
// (tests/test_host_and_sharding.py::test_env_switches_live_in_one_table greps the sources for it and for `static ... = env_*()` initialisers).
// Round 5 found why it has to be this way: hipcc numbers the closure types of namespace-scope lambdas per `namespace rife { }` block, the engine re-opened
// the namespace four times and initialised its switches with immediately-invoked lambdas, so the initialiser of use_graph() carried the code of
// trunk_h2()'s - hipGraph replay was silently ON for four rounds (profiles/r5/static_init_bug_debug.txt).  No switch is a namespace-scope object any more:
// process-scope values sit in a function-local static (thread-safe, initialised at the first use, no order to get wrong).
// The PRODUCT reads four variables (RIFE_HIP_TRUNK=f32: fp32 matrix path; RIFE_HIP_GRAPH=1: hipGraph replay; RIFE_HIP_BATCH_WORKERS: process_batch worker
// threads; RIFE_HIP_PROFILE_FINE=1: per-layer profile classes); the kernel-selection and A/B switches exist only in the TEST build (librife_hip_test.so, the same
// sources with -DRIFE_HIP_TEST_BUILD) and the bench build - in the product ab() below is a null constant and every one of them keeps its default.
// Scopes: process = read once, at the first use; engine = read by rife_hip_create; call = read by every call of the entry point that uses it.
// ================================================================================================================================================
struct Switches {
    // ---- process scope
    bool trunk_h2;             // RIFE_HIP_TRUNK != "f32": split-f16 matrix path (product)
    bool use_graph;            // RIFE_HIP_GRAPH=1: hipGraph replay of the plain v4 pass (product; off)
    int batch_workers;         // RIFE_HIP_BATCH_WORKERS=1..16: worker threads of rife_hip_process_batch (product; 0 = the default of four)
    bool v2_stem16;            // RIFE_HIP_V2_STEM16=0: the 10-channel v2 stems back on the fp32 matrix path
    int ns3_rows4, rows4_max;  // RIFE_HIP_NS3_ROWS4 / RIFE_HIP_ROWS4_MAX: tile-height limits of conv_h2b (round 5 A/B)
    bool t64_loader_waves;     // RIFE_HIP_T64_LW=1: conv_t64 with loader waves
    bool rs_split;             // RIFE_HIP_RS_SPLIT=1: conv_rs epilogue shared by all four io waves
    int ks_div;                // RIFE_HIP_KS_DIV=1..8: conv_ks on a part of the chip
    bool ctx0_img;             // RIFE_HIP_CTX0_IMG=0: ContextNet conv0 from the fp32 NHWC8 copy instead of the RGBX frame
    bool v2_ctx_batch;         // RIFE_HIP_V2_CTX_BATCH=0: the two ContextNet passes one after the other
    bool v2_fused_stem;        // RIFE_HIP_V2_FUSED_STEM=0: k2_assemble + conv_h2s2 instead of stem2_fused_kernel
    bool v2_stem_r64;          // RIFE_HIP_V2_STEM_R64=0: scale-1 fused stem with two workgroups per CU
    bool v2_skip_copy;         // RIFE_HIP_V2_SKIP_COPY=1: U-Net skips copied (k2_copy_view) instead of stored twice
    int tta_lane_parts;        // RIFE_HIP_TTA_LANE_PARTS=2 / 4: the four orientation lanes of -x on CU-masked streams, two per half / one per quarter (default 0: ordinary streams - measured faster)
    bool h2d_token;            // RIFE_HIP_H2D_TOKEN=0: host-frame callers upload concurrently (A/B; default: one upload at a time, enqueue_host_pair)
    int pool_parts;            // RIFE_HIP_POOL_PARTS: 2 / 4 = pool streams own 1 / 2, 1 / 4 of the compute units; 3 = that layout from four callers in flight on; else whole-chip streams (default: measured faster, pool_layout)
    // ---- engine scope
    bool t64, rs, rs2, stem_rs, tta_consensus, tail_rs, tail_rs_always, fuse_flow;      // RIFE_HIP_T64 / RS / RS2 / STEM_RS / TTA_CONSENSUS / TAIL_RS (0, 2) / FUSE_FLOW=1
    int ks_mask;               // RIFE_HIP_KS=<bit mask of blocks on conv_ks>; -1 = not set
    // ---- call scope
    bool merge_flow0;          // RIFE_HIP_MERGE_FLOW0=0: three separate flow updates
    int batch_groups;          // RIFE_HIP_BATCH_GROUPS=1 / 0: force / forbid the lockstep groups of process_batch; -1 = by grid size
    bool profile_fine;         // RIFE_HIP_PROFILE_FINE=1 (product): per-layer profile classes
    int probe_lds; const char* probe_scrub; bool probe_quiet;      // RIFE_HIP_PROBE_LDS / SCRUB / QUIET: parameters of rife_hip_bench_stem_probe (bench build, bench_hooks.h)
};
static Switches read_switches() {
    auto on = [](const char* e) { return e && e[0] == '1'; };                   // default off, "=1" switches on
    auto not_off = [](const char* e) { return !(e && e[0] == '0'); };           // default on, "=0" switches off
    auto num = [](const char* e, int dflt, int lo, int hi) { if (!e) return dflt; const int v = atoi(e); return v >= lo && v <= hi ? v : dflt; };
    auto ab = [](const char* name) -> const char* {                             // test / bench builds only
#ifdef RIFE_HIP_TEST_BUILD
        return getenv(name);
#else
        (void)name; return nullptr;
#endif
    };
    Switches s;
    { const char* e = getenv("RIFE_HIP_TRUNK"); s.trunk_h2 = !(e && std::strcmp(e, "f32") == 0); }
    s.use_graph = on(getenv("RIFE_HIP_GRAPH"));
    s.batch_workers = num(getenv("RIFE_HIP_BATCH_WORKERS"), 0, 1, 16);
    s.profile_fine = on(getenv("RIFE_HIP_PROFILE_FINE"));
    s.v2_stem16 = not_off(ab("RIFE_HIP_V2_STEM16"));
    s.ns3_rows4 = num(ab("RIFE_HIP_NS3_ROWS4"), 0, INT_MIN, INT_MAX);
    s.rows4_max = num(ab("RIFE_HIP_ROWS4_MAX"), 400, INT_MIN, INT_MAX);
    s.t64_loader_waves = on(ab("RIFE_HIP_T64_LW"));
    s.rs_split = on(ab("RIFE_HIP_RS_SPLIT"));
    s.ks_div = num(ab("RIFE_HIP_KS_DIV"), 1, 1, 8);
    s.ctx0_img = not_off(ab("RIFE_HIP_CTX0_IMG"));
    s.v2_ctx_batch = not_off(ab("RIFE_HIP_V2_CTX_BATCH"));
    s.v2_fused_stem = not_off(ab("RIFE_HIP_V2_FUSED_STEM"));
    s.v2_stem_r64 = not_off(ab("RIFE_HIP_V2_STEM_R64"));
    s.v2_skip_copy = on(ab("RIFE_HIP_V2_SKIP_COPY"));
    s.pool_parts = num(ab("RIFE_HIP_POOL_PARTS"), -1, 0, 4);
    s.h2d_token = not_off(ab("RIFE_HIP_H2D_TOKEN"));
    s.tta_lane_parts = num(ab("RIFE_HIP_TTA_LANE_PARTS"), 0, 0, 4);
    s.t64 = not_off(ab("RIFE_HIP_T64"));
    s.rs = not_off(ab("RIFE_HIP_RS"));
    s.rs2 = not_off(ab("RIFE_HIP_RS2"));
    { const char* e = ab("RIFE_HIP_KS"); s.ks_mask = e && e[0] >= '0' && e[0] <= '9' ? atoi(e) : -1; }
    s.stem_rs = not_off(ab("RIFE_HIP_STEM_RS"));
    s.tta_consensus = not_off(ab("RIFE_HIP_TTA_CONSENSUS"));
    { const char* e = ab("RIFE_HIP_TAIL_RS"); s.tail_rs = not_off(e); s.tail_rs_always = e && e[0] == '2'; }
    s.fuse_flow = on(ab("RIFE_HIP_FUSE_FLOW"));
    s.merge_flow0 = not_off(ab("RIFE_HIP_MERGE_FLOW0"));
    { const char* e = ab("RIFE_HIP_BATCH_GROUPS"); s.batch_groups = e ? (e[0] != '0' ? 1 : 0) : -1; }
    { const char* e = ab("RIFE_HIP_PROBE_LDS"); s.probe_lds = e ? atoi(e) : -1; }
    s.probe_scrub = ab("RIFE_HIP_PROBE_SCRUB");
    s.probe_quiet = ab("RIFE_HIP_PROBE_QUIET") != nullptr;
    return s;
}
static const Switches& process_switches() { static const Switches s = read_switches(); return s; }
// ================================================================================================================================================ (end of the switch table)

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(RIFE_HIP_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace rife
