// engine_dispatch.h: kernel dispatch - launch_conv and the launchers of the specialised kernels (conv_t64 / conv_rs / conv_rs2 / conv_row / conv_ks / stems / tails)
// One translation unit (engine.hip includes the engine_*.h sections in dependency order; every function here is file-local).
// No include guard on purpose: a section is included exactly once, by engine.hip.

namespace rife {

// ------------------------------------------------------------------------------------------------
// kernel dispatch
// ------------------------------------------------------------------------------------------------
template <int STRIDE, int MS, int NS, int CC, int EPI, int TAG>
static hipError_t launch_cfg(const ConvArgs& a, int nblocks, hipStream_t st) {
    auto kfn = conv_mfma_kernel<STRIDE, MS, NS, CC, EPI, TAG>;
    constexpr int lds = conv_lds_bytes<STRIDE, MS, NS, CC, EPI, conv_ks<TAG>()>();
    static_assert(lds <= 64 * 1024, "tile does not fit the default dynamic LDS limit");
    hipLaunchKernelGGL(kfn, dim3(nblocks), dim3(256), lds, st, a);
    return hipGetLastError();
}

struct TensorView { float* p; int ld, coff; };

// split-K partial-sum workspace: one per (device, stream), grown on demand (used only by small layers)
static float* splitk_workspace(hipStream_t st, size_t floats) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, std::pair<float*, size_t>> ws;
    int dev = 0; (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    auto& e = ws[{dev, st}];
    if (e.second < floats) {
        if (e.first) { (void)hipStreamSynchronize(st); (void)hipFree(e.first); }
        if (hipMalloc(&e.first, floats * 4) != hipSuccess) { e.first = nullptr; e.second = 0; return nullptr; }
        e.second = floats;
    }
    return e.first;
}

// RIFE_HIP_TRUNK=f32 keeps the trunk convolutions on the fp32 matrix path (default: split-f16, see conv_h2_kernel): the documented
// numerics fallback, and bench.py's fp32 reference mode.  The round-1 A/B switches with a settled winner (fused stem, split-f16
// heads and stride-2 stems, split-K for tiny grids, fused tail, two-workgroup trunk kernel, 8-wave fp32 kernel, 96-wide N tiles,
// 4-row tiles below 400 workgroups) are constants now; the measurements behind them are in DESIGN.md and profiles/r1.
static inline bool trunk_h2() { return process_switches().trunk_h2; }      // RIFE_HIP_TRUNK=f32: the fp32 matrix path (switch table)
static constexpr bool g_fuse_stem = true, g_head_h2 = true, g_s2_h2 = true, g_splitk = true, g_fuse_tail = true, g_h2b = true, g_use_conv8 = true;

// x: NHWC input (H x W), y: output; for deconv layers y has 2H x 2W pixels (or the 4H x 4W flow tensor with EPI_DECONV_PS).
// s16_pitch > 0: the stride-2 stem writes / the head reads an S16 tensor (conv_t64.h) of that row pitch instead of NHWC fp32
static int launch_conv(const ConvLayer& L, TensorView x, int H, int W, TensorView y, const TensorView* res, hipStream_t st, const FinalArgs* fin = nullptr,
                       int s16_pitch = 0, unsigned s16_plane = 0, const float* in1 = nullptr, float* out1 = nullptr, const TensorView* y2 = nullptr) {
    // in1 / out1: a second tensor pair of the same geometry through the same launch (gridDim.y = 2; the stride-2 and stride-1 split-f16 kernels)
    ConvArgs a;
    a.in1 = in1; a.out1 = out1;
    const unsigned gy = in1 ? 2 : 1;
    // y2: the output goes to a second view as well (conv_h2b_kernel only: stride-1 split-f16 layers without split-K)
    if (y2) { a.out2 = y2->p; a.out2_ld = y2->ld; a.out2_coff = y2->coff; }
    if (y2 && !(L.nchunksh > 0 && L.stride == 1 && !L.deconv && trunk_h2() && res == nullptr && g_h2b && L.NS <= 2))
        return fail(RIFE_HIP_EINVAL, "no two-destination form of this convolution kernel");
    a.s16_pitch = s16_pitch; a.s16_plane = s16_plane;
    a.in = x.p; a.in_ld = x.ld; a.in_coff = x.coff; a.H = H; a.W = W;
    a.out = y.p; a.out_ld = y.ld; a.out_coff = y.coff;
    a.wpk = L.d_w; a.bias = L.d_bias; a.slope = L.d_slope;
    a.res = res ? res->p : nullptr; a.res_ld = res ? res->ld : 0; a.res_coff = res ? res->coff : 0;
    a.Ho = L.deconv ? H : (H + 2 - 3) / L.stride + 1;
    a.Wo = L.deconv ? W : (W + 2 - 3) / L.stride + 1;
    a.Cout = L.cout; a.nchunks = L.nchunks; a.nz = L.ntiles * L.npar;
    if (x.ld % 4 || x.coff % 4 || x.ld - x.coff < L.cin_p) return fail(RIFE_HIP_EINVAL, "conv input view is not padded to the channel chunk");
    if (L.epi != EPI_DECONV_PS && (y.ld % 4 || y.coff % 4 || L.cout % 4 || (res && (res->ld % 4 || res->coff % 4))))
        return fail(RIFE_HIP_EINVAL, "conv output / residual views must be 16-byte aligned per pixel (channel counts multiples of 4)");
    a.tiles_x = (a.Wo + 31) / 32;
    // rows per wave: 2 when that still gives every CU >= 1.5 workgroups, else 1 (more, smaller workgroups for the coarse blocks)
    int MS = L.MS;
    if (L.stride == 1) {
        const long wg2 = (long)a.tiles_x * ((a.Ho + 7) / 8) * a.nz;
        MS = wg2 >= 384 ? 2 : 1;
    }
    if (L.nchunksh > 0 && !L.deconv && L.stride == 2 && (L.cin >= 16 || L.cin == 10) && trunk_h2() && g_s2_h2 && res == nullptr) {
        if (x.ld - x.coff < 16 * L.nchunksh) return fail(RIFE_HIP_EINVAL, "conv input view is not padded to whole 16-channel chunks");
        a.ntiles_xy = a.tiles_x * ((a.Ho + 3) / 4);
        a.nchunks = L.nchunksh;
        a.wpk = reinterpret_cast<const float*>(L.d_wh);
        const int nb = a.ntiles_xy * a.nz;
        constexpr int ls1 = convh2s2_lds_bytes<1>(), ls2 = convh2s2_lds_bytes<2>(), ls3 = convh2s2_lds_bytes<3>();
        {
            static std::mutex smu; static std::map<int, bool> sdone;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(smu);
            if (!sdone[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2s2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, ls2));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2s2_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ls2));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2s2_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, ls3));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2s2_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, ls3));
                sdone[dev] = true;
            }
        }
        if (s16_pitch > 0) {
            if ((L.NS != 2 && L.NS != 3) || !L.d_whp) return fail(RIFE_HIP_EINVAL, "no S16 variant of this stride-2 layer");
            a.wpk = reinterpret_cast<const float*>(L.d_whp);
            if (L.NS == 2) hipLaunchKernelGGL((conv_h2s2_kernel<2, true>), dim3(nb, gy), dim3(256), ls2, st, a);
            else hipLaunchKernelGGL((conv_h2s2_kernel<3, true>), dim3(nb, gy), dim3(256), ls3, st, a);
        }
        else if (L.NS == 1) hipLaunchKernelGGL(conv_h2s2_kernel<1>, dim3(nb, gy), dim3(256), ls1, st, a);
        else if (L.NS == 2) hipLaunchKernelGGL(conv_h2s2_kernel<2>, dim3(nb, gy), dim3(256), ls2, st, a);
        else hipLaunchKernelGGL(conv_h2s2_kernel<3>, dim3(nb, gy), dim3(256), ls3, st, a);
        hipError_t eh = hipGetLastError();
        if (eh != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_h2s2 launch: ") + hipGetErrorString(eh));
        return 0;
    }
    if (fin && !(L.nchunksh > 0 && L.deconv && trunk_h2() && g_head_h2)) return fail(RIFE_HIP_EINVAL, "fused tail needs the split-f16 head kernel");
    if (L.nchunksh > 0 && L.deconv && trunk_h2() && g_head_h2) {
        if (in1) return fail(RIFE_HIP_EINVAL, "no two-tensor form of the head kernel");
        a.ntiles_xy = a.tiles_x * ((a.Ho + 7) / 8);
        a.nchunks = L.nchunksh;
        a.nz = (L.cout + 31) / 32;
        a.wpk = reinterpret_cast<const float*>(L.d_wh);
        {
            static std::mutex hmu; static std::map<int, bool> hdone;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(hmu);
            if (!hdone[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_DECONV_PS>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_DECONV>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_DECONV_SIG>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_FINAL>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_FINAL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(head_h2_kernel<EPI_DECONV_PS, true>), hipFuncAttributeMaxDynamicSharedMemorySize, headh2_lds_bytes()));
                hdone[dev] = true;
            }
        }
        const int nb = a.ntiles_xy * a.nz;
        if (L.epi == EPI_DECONV_PS && (L.cout != 24 || y.ld != 8 || y.coff != 0))
            return fail(RIFE_HIP_EINVAL, "the PixelShuffle head kernel writes the 6-channel flow tensor [4H][4W][8] only");
        if (s16_pitch > 0 && L.epi != EPI_DECONV_PS) return fail(RIFE_HIP_EINVAL, "no S16 variant of this head");
        if (s16_pitch > 0 && fin) hipLaunchKernelGGL((head_h2_kernel<EPI_FINAL, true>), dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, *fin);
        else if (s16_pitch > 0) hipLaunchKernelGGL((head_h2_kernel<EPI_DECONV_PS, true>), dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, FinalArgs{});
        else if (fin && L.epi == EPI_DECONV_PS) hipLaunchKernelGGL(head_h2_kernel<EPI_FINAL>, dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, *fin);
        else if (L.epi == EPI_DECONV_PS) hipLaunchKernelGGL(head_h2_kernel<EPI_DECONV_PS>, dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, FinalArgs{});
        else if (L.epi == EPI_DECONV) hipLaunchKernelGGL(head_h2_kernel<EPI_DECONV>, dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, FinalArgs{});
        else hipLaunchKernelGGL(head_h2_kernel<EPI_DECONV_SIG>, dim3(nb, gy), dim3(512), headh2_lds_bytes(), st, a, FinalArgs{});
        hipError_t eh = hipGetLastError();
        if (eh != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("head_h2 launch: ") + hipGetErrorString(eh));
        return 0;
    }
    // trunk layers: split-f16 matrix path (fp32-grade accuracy at 8x the fp32 MFMA rate) unless RIFE_HIP_TRUNK=f32
    if (L.nchunksh > 0 && L.stride == 1 && trunk_h2() && res == nullptr) {
        a.ntiles_xy = a.tiles_x * ((a.Ho + 7) / 8);
        a.nchunks = L.nchunksh;
        a.wpk = reinterpret_cast<const float*>(L.d_wh);
        const int nb = a.ntiles_xy * a.nz;
        constexpr int l29 = convh2_lds_bytes<2, 9>(), l210 = convh2_lds_bytes<2, 10>(), l39 = convh2_lds_bytes<3, 9>(), l310 = convh2_lds_bytes<3, 10>();
        {
            static std::mutex amu; static std::map<int, bool> done;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(amu);
            if (!done[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<2, 9, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, l29));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<2, 10, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, l210));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<2, 10, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, l210));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<3, 9, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, l39));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2_kernel<3, 10, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, l310));
                done[dev] = true;
            }
        }
        constexpr int lb9 = convh2b_lds_bytes<2, 9>(), lb10 = convh2b_lds_bytes<2, 10>();
        {
            static std::mutex bmu; static std::map<int, bool> bdone;
            int dev = 0; (void)hipGetDevice(&dev);
            std::lock_guard<std::mutex> g(bmu);
            if (!bdone[dev]) {
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 9, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lb9));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<1, 9, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, convh2b_lds_bytes<1, 9>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<1, 10, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, convh2b_lds_bytes<1, 10>()));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 10, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lb10));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 10, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lb10));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 10, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (convh2b_lds_bytes<2, 10, 4>())));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 9, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (convh2b_lds_bytes<2, 9, 4>())));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<3, 10, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (convh2b_lds_bytes<3, 10, 4>())));
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<3, 9, 0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (convh2b_lds_bytes<3, 9, 4>())));
                bdone[dev] = true;
            }
        }
        constexpr int lb19 = convh2b_lds_bytes<1, 9>();
        // split-K only for layers with a handful of workgroups (<= 64, i.e. under a quarter of the CUs): measured +35 % on the
        // 1080p block-0 trunk (30 workgroups); above that the partial-sum traffic and the extra launch eat the gain
        int nsplit = 1;
        if (g_splitk && g_h2b && L.NS == 2 && nb <= 64 && a.nchunks >= 4 && !in1 && !y2) nsplit = std::min(4, a.nchunks / 2);      // round-5 A/B of 2 / 8 slices and of the 64-workgroup limit: no change
        int nbl = nb;
        if (nsplit > 1) {
            a.nsplit = nsplit; a.cpad = L.ntiles * L.NS * 32;
            a.partial = splitk_workspace(st, (size_t)nsplit * a.Ho * a.Wo * a.cpad);
            if (!a.partial) return fail(RIFE_HIP_EHIP, "split-K workspace allocation failed");
            nbl = nb * nsplit;
        }
        // 4-row tiles (4 waves, three workgroups per CU) for layers whose 8-row tiles would occupy only part of the chip: twice the
        // workgroups, half the latency of each (below 400 8-row workgroups: round-1 A/B)
        constexpr int rows4_max = 400;
        const int ns3_rows4 = process_switches().ns3_rows4, rows4_lim = process_switches().rows4_max;      // A/B (round 5); the default limit is rows4_max
        static_assert(rows4_max == 400, "read_switches() carries the default of RIFE_HIP_ROWS4_MAX");
        const bool rows4 = g_h2b && (L.NS == 2 || L.NS == 3) && nsplit == 1 && (nb < rows4_lim || (L.NS == 3 && ns3_rows4));
        if (rows4) {
            constexpr int l4_9 = convh2b_lds_bytes<2, 9, 4>(), l4_10 = convh2b_lds_bytes<2, 10, 4>();
            constexpr int l43_9 = convh2b_lds_bytes<3, 9, 4>(), l43_10 = convh2b_lds_bytes<3, 10, 4>();      // 96-wide N-tiles: 63 KB, two workgroups per CU
            a.ntiles_xy = a.tiles_x * ((a.Ho + 3) / 4);
            const int nb4 = a.ntiles_xy * a.nz;
            if (L.NS == 3 && L.skip) hipLaunchKernelGGL((conv_h2b_kernel<3, 10, 0, 4>), dim3(nb4, gy), dim3(256), l43_10, st, a);
            else if (L.NS == 3) hipLaunchKernelGGL((conv_h2b_kernel<3, 9, 0, 4>), dim3(nb4, gy), dim3(256), l43_9, st, a);
            else if (L.skip) hipLaunchKernelGGL((conv_h2b_kernel<2, 10, 0, 4>), dim3(nb4, gy), dim3(256), l4_10, st, a);
            else hipLaunchKernelGGL((conv_h2b_kernel<2, 9, 0, 4>), dim3(nb4, gy), dim3(256), l4_9, st, a);
            hipError_t e4 = hipGetLastError();
            if (e4 != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_h2b (4-row) launch: ") + hipGetErrorString(e4));
            return 0;
        }
        const int nb_saved = nb; (void)nb_saved;
#define nb nbl
        constexpr int lb110 = convh2b_lds_bytes<1, 10>();
        if (L.NS == 1 && L.skip) hipLaunchKernelGGL((conv_h2b_kernel<1, 10, 0>), dim3(nb, gy), dim3(512), lb110, st, a);
        else if (L.NS == 1) hipLaunchKernelGGL((conv_h2b_kernel<1, 9, 0>), dim3(nb, gy), dim3(512), lb19, st, a);
        else if (g_h2b && L.NS == 2 && L.skip && L.tag == 3) hipLaunchKernelGGL((conv_h2b_kernel<2, 10, 3>), dim3(nb, gy), dim3(512), lb10, st, a);
        else if (g_h2b && L.NS == 2 && L.skip) hipLaunchKernelGGL((conv_h2b_kernel<2, 10, 0>), dim3(nb, gy), dim3(512), lb10, st, a);
        else if (g_h2b && L.NS == 2) hipLaunchKernelGGL((conv_h2b_kernel<2, 9, 0>), dim3(nb, gy), dim3(512), lb9, st, a);
        else if (L.NS == 2 && L.skip && L.tag == 3) hipLaunchKernelGGL((conv_h2_kernel<2, 10, 3>), dim3(nb, gy), dim3(512), l210, st, a);
        else if (L.NS == 2 && L.skip) hipLaunchKernelGGL((conv_h2_kernel<2, 10, 0>), dim3(nb, gy), dim3(512), l210, st, a);
        else if (L.NS == 2) hipLaunchKernelGGL((conv_h2_kernel<2, 9, 0>), dim3(nb, gy), dim3(512), l29, st, a);
        else if (L.skip) hipLaunchKernelGGL((conv_h2_kernel<3, 10, 0>), dim3(nb, gy), dim3(512), l310, st, a);
        else hipLaunchKernelGGL((conv_h2_kernel<3, 9, 0>), dim3(nb, gy), dim3(512), l39, st, a);
#undef nb
        hipError_t eh = hipGetLastError();
        if (eh != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_h2 launch: ") + hipGetErrorString(eh));
        if (nsplit > 1) {
            const size_t npix = (size_t)a.Ho * a.Wo, n = npix * (L.cout / 4);
            hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a.partial, nsplit, npix, a.cpad, L.cout, L.d_bias, L.d_slope,
                               y.p, y.ld, y.coff);
            eh = hipGetLastError();
            if (eh != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("splitk reduce launch: ") + hipGetErrorString(eh));
        }
        return 0;
    }
    if (in1) return fail(RIFE_HIP_EINVAL, "no two-tensor form of this convolution kernel");
    // layers with >= 2 full waves of 8-row tiles take the double-buffered 8-wave kernel
    if (L.nchunks8 > 0 && g_use_conv8) {
        const long wg8 = (long)a.tiles_x * ((a.Ho + 7) / 8) * a.nz;
        if (wg8 >= 448) {
            a.ntiles_xy = a.tiles_x * ((a.Ho + 7) / 8);
            a.nchunks = L.nchunks8;
            if (L.d_w8) a.wpk = L.d_w8;
            const int nb = a.ntiles_xy * a.nz;
            constexpr int lds28 = conv8_lds_bytes<2, 8>(), lds38 = conv8_lds_bytes<3, 8>();
            static_assert(lds28 <= 80 * 1024 && lds38 <= 160 * 1024, "LDS budget");
            {   // > 64 KB of dynamic LDS needs an opt-in per kernel and device
                static std::mutex amu; static std::map<int, bool> done;
                int dev = 0; (void)hipGetDevice(&dev);
                std::lock_guard<std::mutex> g(amu);
                if (!done[dev]) {
                    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma8_kernel<2, 8, 4, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds28));
                    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma8_kernel<2, 8, 4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds28));
                    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_mfma8_kernel<3, 8, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds38));
                    done[dev] = true;
                }
            }
            if (L.NS == 2 && L.tag == 3) hipLaunchKernelGGL((conv_mfma8_kernel<2, 8, 4, 3>), dim3(nb, gy), dim3(512), lds28, st, a);
            else if (L.NS == 2) hipLaunchKernelGGL((conv_mfma8_kernel<2, 8, 4, 0>), dim3(nb, gy), dim3(512), lds28, st, a);
            else hipLaunchKernelGGL((conv_mfma8_kernel<3, 8, 2, 0>), dim3(nb, gy), dim3(512), lds38, st, a);
            hipError_t e8 = hipGetLastError();
            if (e8 != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv8 launch: ") + hipGetErrorString(e8));
            return 0;
        }
    }
    a.ntiles_xy = a.tiles_x * ((a.Ho + 4 * MS - 1) / (4 * MS));
    const int nblocks = a.ntiles_xy * a.nz;
    hipError_t e = hipErrorInvalidValue;
#define RIFE_CFG(S_, MS_, NS_, CC_, E_, T_) \
    if (L.stride == S_ && MS == MS_ && L.NS == NS_ && L.CC == CC_ && L.epi == E_ && L.tag == T_) e = launch_cfg<S_, MS_, NS_, CC_, E_, T_>(a, nblocks, st); else
    RIFE_CFG(2, 1, 1, 8, EPI_STORE, 5)
    RIFE_CFG(1, 2, 1, 8, EPI_STORE, 5)
    RIFE_CFG(1, 1, 1, 8, EPI_STORE, 5)
    RIFE_CFG(2, 1, 1, 8, EPI_STORE, 0)
    RIFE_CFG(2, 1, 2, 8, EPI_STORE, 0)
    RIFE_CFG(2, 1, 3, 8, EPI_STORE, 0)
    RIFE_CFG(1, 2, 2, 16, EPI_STORE, 3)
    RIFE_CFG(1, 1, 2, 16, EPI_STORE, 3)
    RIFE_CFG(1, 2, 1, 16, EPI_STORE, 0)
    RIFE_CFG(1, 2, 2, 16, EPI_STORE, 0)
    RIFE_CFG(1, 2, 3, 8, EPI_STORE, 0)
    RIFE_CFG(1, 1, 1, 16, EPI_STORE, 0)
    RIFE_CFG(1, 1, 2, 16, EPI_STORE, 0)
    RIFE_CFG(1, 1, 3, 8, EPI_STORE, 0)
    RIFE_CFG(1, 2, 1, 16, EPI_DECONV_PS, 0)
    RIFE_CFG(1, 1, 1, 16, EPI_DECONV_PS, 0)
    RIFE_CFG(1, 2, 1, 16, EPI_DECONV, 0)
    RIFE_CFG(1, 2, 2, 16, EPI_DECONV, 0)
    RIFE_CFG(1, 2, 3, 8, EPI_DECONV, 0)
    RIFE_CFG(1, 1, 1, 16, EPI_DECONV, 0)
    RIFE_CFG(1, 1, 2, 16, EPI_DECONV, 0)
    RIFE_CFG(1, 1, 3, 8, EPI_DECONV, 0)
    RIFE_CFG(1, 2, 1, 16, EPI_DECONV_SIG, 0)
    RIFE_CFG(1, 1, 1, 16, EPI_DECONV_SIG, 0)
    { return fail(RIFE_HIP_ENOSYS, "no conv kernel instantiation for this layer shape"); }
#undef RIFE_CFG
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv launch: ") + hipGetErrorString(e));
    return 0;
}

// S16 tensor geometry for an H x W pixel grid (conv_t64.h): T64_TH x 32 tiles, one pixel of zero border on every side of every plane
struct S16Geom {
    int tiles_x, tiles_y, pitch, rows;
    S16Geom(int H, int W) : tiles_x((W + 31) / 32), tiles_y((H + T64_TH - 1) / T64_TH), pitch(tiles_x * 32 + 2), rows(tiles_y * T64_TH + 2) {}
    unsigned plane() const { return (unsigned)rows * pitch * 32u; }             // one [chunk][hi | lo] plane
    size_t bytes(int C) const { return (size_t)plane() * (C / 8); }
};

// compute units of the current device (cached): grid sizes of the persistent kernels and the kernel-selection thresholds below.
// tl_cu_budget > 0: the calling thread is enqueueing on a stream that owns only a PART of the chip (CU-masked stream, rife_hip_stream_create):
// persistent grids are sized for that part.
static thread_local int tl_cu_budget = 0;
static int device_cus(bool physical = false) {
    if (tl_cu_budget > 0 && !physical) return tl_cu_budget;
    int dev = 0; (void)hipGetDevice(&dev);
    static std::mutex mu; static std::map<int, int> ncu;
    std::lock_guard<std::mutex> g(mu);
    auto it = ncu.find(dev);
    if (it == ncu.end()) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        it = ncu.emplace(dev, n).first;
    }
    return it->second;
}

// one C -> C (C = 64, 96) residual trunk convolution, S16 in / S16 out, persistent workgroups (two / one per CU)
// reverse: walk the tiles last to first.  Consecutive trunk layers alternate, so that a layer starts on what its predecessor wrote
// last - still in the L2 / Infinity Cache (134 MB in + 134 MB out per 4K layer against 256 MB of cache: in one direction only the
// first rows of a layer's input were written more than a cache-full of traffic ago by the time they are read).
// RIFE_HIP_T64_LW=1: the 96-channel trunk with two loader waves (conv_t64.h, template parameter LW).  Off by default: measured equal or 2 % slower
// (4K, same call: trunk_b2 0.401 vs 0.392 - 0.396 ms per pair) - unlike in conv_rs_kernel, whose consumers also lose the weight stream and the stores
static inline bool t64_loader_waves() { return process_switches().t64_loader_waves; }
static int launch_t64(const ConvLayer& L, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, bool reverse = false) {
    if (!L.d_t64) return fail(RIFE_HIP_EINVAL, "layer has no conv_t64 image");
    const int NS = t64_ns(L.cout);
    int dev = 0; (void)hipGetDevice(&dev);
    static std::mutex mu; static std::map<int, int> ncu;
    int cus;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = ncu.find(dev);
        if (it == ncu.end()) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_t64_kernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, t64_lds(2)));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_t64_kernel<2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, t64_lds(3)));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_t64_kernel<2, 3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, t64_lds(3)));
            int n = 0;
            HIPCHK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
            it = ncu.emplace(dev, std::max(8, n / 8 * 8)).first;
        }
        cus = it->second;
    }
    if (tl_cu_budget > 0) cus = std::max(8, std::min(cus, tl_cu_budget) / 8 * 8);
    const S16Geom G(H, W);
    T64Args a;
    a.in = in; a.out = out; a.img = L.d_t64; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane(); a.tiles_x = G.tiles_x; a.ntiles = G.tiles_x * G.tiles_y; a.reverse = reverse ? 1 : 0;
    a.nchunks = L.cout / 16; a.nnt = L.cout / (32 * NS);
    const int nwg = std::min(t64_wg_per_cu(NS) * cus, (a.ntiles * a.nnt + 7) / 8 * 8);      // all workgroups resident at once
    if (L.cout == 64) hipLaunchKernelGGL((conv_t64_kernel<3, 2>), dim3(nwg), dim3(T64_NTHR), t64_lds(2), st, a);       // TAG: the profile class (trunk_b3 .. trunk_b0)
    else if (L.cout == 96 && t64_loader_waves()) hipLaunchKernelGGL((conv_t64_kernel<2, 3, 2>), dim3(nwg), dim3(T64_NTHR + 128), t64_lds(3), st, a);      // two loader waves
    else if (L.cout == 96) hipLaunchKernelGGL((conv_t64_kernel<2, 3>), dim3(nwg), dim3(T64_NTHR), t64_lds(3), st, a);
    else return fail(RIFE_HIP_EINVAL, "conv_t64 serves 64 and 96 channels");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_t64 launch: ") + hipGetErrorString(e));
    return 0;
}

// the same 64 -> 64 layer on the row-streaming kernel (conv_rs.h): one workgroup per CU, specialised waves.  descend: walk every
// workgroup's range bottom-up; consecutive layers alternate so that a layer starts on the rows its predecessor wrote last.
static inline bool rs_split() { return process_switches().rs_split; }      // A/B: epilogue shared by all four io waves (conv_rs.h, SPLIT)
static int launch_rs(const ConvLayer& L, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, bool descend = false) {
    if (!L.d_t64 || L.cout != 64) return fail(RIFE_HIP_EINVAL, "layer has no 64-channel conv_t64 image");
    if ((H + 1) / 2 < RS_MIN_PAIRS) return fail(RIFE_HIP_EINVAL, "conv_rs needs at least " + std::to_string(2 * RS_MIN_PAIRS - 1) + " rows");
    int dev = 0; (void)hipGetDevice(&dev);
    static std::mutex mu; static std::map<int, int> ncu;
    int cus;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = ncu.find(dev);
        if (it == ncu.end()) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS));
            int n = 0;
            HIPCHK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
            it = ncu.emplace(dev, std::max(1, n)).first;
        }
        cus = it->second;
    }
    if (tl_cu_budget > 0) cus = std::min(cus, tl_cu_budget);
    const S16Geom G(H, W);
    RsArgs a;
    a.in = in; a.out = out; a.img = L.d_t64; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane();
    a.npairs = (H + 1) / 2; a.nunits = G.tiles_x * a.npairs; a.descend = descend ? 1 : 0;
    const int nwg = std::min(cus, a.nunits);                             // one workgroup per CU (154 KB of LDS), all resident
    if (rs_split()) hipLaunchKernelGGL((conv_rs_kernel<0, 1>), dim3(nwg), dim3(RS_NTHR), RS_LDS, st, a);
    else hipLaunchKernelGGL((conv_rs_kernel<0>), dim3(nwg), dim3(RS_NTHR), RS_LDS, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_rs launch: ") + hipGetErrorString(e));
    return 0;
}

// TWO consecutive 64 -> 64 layers in one launch of the depth-fused row-streaming kernel (conv_rs2.h): layer A's rows stay in LDS.  Strips of 30 columns,
// every strip cut into kparts equal row ranges so that every CU of the (part of the) chip has one segment.  rs2_applies: false where the fused form
// does not apply - fewer than RS2_MIN_ROWS rows per segment, or a tensor of 2 GB and more (signed 32-bit DMA offsets) - and the caller runs two
// conv_rs launches instead: the bytes are the same either way.
static int rs2_plan(int H, int W, int cus, int& kparts, int& nstrips) {
    nstrips = (W + RS2_SW - 1) / RS2_SW;
    kparts = std::max(1, cus / nstrips);
    kparts = std::min(kparts, std::max(1, H / RS2_MIN_ROWS));
    return H / kparts;                                                  // rows of the shortest segment
}
static int rs2_cus() {
    int dev = 0; (void)hipGetDevice(&dev);
    static std::mutex mu; static std::map<int, int> ncu;
    int cus;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = ncu.find(dev);
        if (it == ncu.end()) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, RS2_LDS) != hipSuccess) return 0;
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
            it = ncu.emplace(dev, std::max(1, n)).first;
        }
        cus = it->second;
    }
    if (tl_cu_budget > 0) cus = std::min(cus, tl_cu_budget);
    return cus;
}
static bool rs2_applies(int H, int W) {
    int kparts, nstrips;
    const int cus = rs2_cus();
    return cus > 0 && S16Geom(H, W).bytes(64) < (1ull << 31) && rs2_plan(H, W, cus, kparts, nstrips) >= RS2_MIN_ROWS;
}
static int launch_rs2(const ConvLayer& LA, const ConvLayer& LB, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, bool descend = false) {
    if (!LA.d_t64 || LA.cout != 64 || !LB.d_t64 || LB.cout != 64) return fail(RIFE_HIP_EINVAL, "layer has no 64-channel conv_t64 image");
    const S16Geom G(H, W);
    const int cus = rs2_cus();
    int kparts, nstrips;
    if (cus <= 0 || G.bytes(64) >= (1ull << 31) || rs2_plan(H, W, cus, kparts, nstrips) < RS2_MIN_ROWS) return fail(RIFE_HIP_EINVAL, "conv_rs2 does not apply to this tensor");
    Rs2Args a;
    a.in = in; a.out = out; a.imgA = LA.d_t64; a.imgB = LB.d_t64; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane(); a.rowmax = G.pitch - 2;
    a.kparts = kparts; a.nseg = nstrips * kparts; a.descend = descend ? 1 : 0; a.limit = (int)(G.bytes(64) - 16);
    const int nwg = std::min(cus, a.nseg);                               // one workgroup per CU (all of its LDS), all resident
    hipLaunchKernelGGL((conv_rs2_kernel<0>), dim3(nwg), dim3(RS2_NTHR), RS2_LDS, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_rs2 launch: ") + hipGetErrorString(e));
    return 0;
}

// one C -> C (C = 128, 192) residual trunk convolution of a coarse block, S16 in / S16 out: one workgroup per ROWS x 32 pixels (conv_row.h)
// nb > 0: one launch for the tensors inb[k] -> outb[k] of nb pairs in flight (gridDim.y = nb)
static int launch_row(const ConvLayer& L, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, int nb = 0,
                      const unsigned char* const* inb = nullptr, unsigned char* const* outb = nullptr) {
    const unsigned char* const rimg = L.cout == 96 ? L.d_row : L.d_t64;
    if (!rimg) return fail(RIFE_HIP_EINVAL, "layer has no conv_row image");
    {
        int dev = 0; (void)hipGetDevice(&dev);
        static std::mutex mu; static std::map<int, bool> done;
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_row_kernel<192, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (convrow_lds_bytes<192, 1>())));
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_row_kernel<128, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (convrow_lds_bytes<128, 2>())));
            done[dev] = true;
        }
    }
    const S16Geom G(H, W);
    RowArgs a;
    a.in = in; a.out = out; a.img = rimg; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane(); a.tiles_x = G.tiles_x;
    if (nb > 4) return fail(RIFE_HIP_EINVAL, "conv_row batches at most four pairs");
    a.nb = nb;
    for (int k = 0; k < nb; k++) { a.inb[k] = inb[k]; a.outb[k] = outb[k]; }
    const unsigned gy = nb > 0 ? nb : 1;
    if (L.cout == 192) { a.ntiles = a.tiles_x * H; hipLaunchKernelGGL((conv_row_kernel<192, 1, 0>), dim3(a.ntiles, gy), dim3(384), (convrow_lds_bytes<192, 1>()), st, a); }
    else if (L.cout == 128) {
        a.ntiles = a.tiles_x * ((H + 1) / 2);
        hipLaunchKernelGGL((conv_row_kernel<128, 2, 1>), dim3(a.ntiles, gy), dim3(256), (convrow_lds_bytes<128, 2>()), st, a);
    }
    else if (L.cout == 96) { a.ntiles = a.tiles_x * ((H + 1) / 2); hipLaunchKernelGGL((conv_row_kernel<96, 2, 2>), dim3(a.ntiles, gy), dim3(192), (convrow_lds_bytes<96, 2, 2>()), st, a); }
    else return fail(RIFE_HIP_EINVAL, "conv_row serves 96, 128 and 192 channels");
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_row launch: ") + hipGetErrorString(e));
    return 0;
}

#ifdef RIFE_HIP_TEST_BUILD
// the same coarse-block layers on the weight-stationary K-split kernel (conv_ks.h; round 4): C = 128 (block 1) and C = 96 (block 2).
// RIFE_HIP_KS (create time) = bit mask: 1 = 128 channels, 2 = 96 channels where conv_row served them (small grids), 4 = 96 channels at every
// size (instead of conv_t64), 0 = conv_row / conv_t64 as in round 3 (A/B, tests/test_gpu_ks.py).
template <int C, int NB, int CPW>
static int launch_ks_cfg(const unsigned char* img, const KsArgs& a0, int tiles_x, int gy, hipStream_t st) {
    using K = KsCfg<C, NB, CPW>;
    {
        int dev = 0; (void)hipGetDevice(&dev);
        static std::mutex mu; static std::map<int, bool> done;
        std::lock_guard<std::mutex> g(mu);
        if (!done[dev]) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_ks_kernel<C, NB, CPW, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS));
            done[dev] = true;
        }
    }
    KsArgs a = a0;
    a.img = img;
    // ranges per N group: one workgroup per CU (150 KB of LDS), all resident at once also when gy pairs share the launch; a multiple of the
    // strip count where possible, so that no range crosses a strip (a crossing costs a pipeline drain and refill)
    const int div = process_switches().ks_div;      // A/B: part of the chip only
    int G = std::max(1, device_cus() / (K::NG * gy * div));
    G = std::min(G, a.nunits);
    if (G >= tiles_x) G = G / tiles_x * tiles_x;
    hipLaunchKernelGGL((conv_ks_kernel<C, NB, CPW, 0>), dim3(G * K::NG, gy), dim3(K::NTHR), K::LDS, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RIFE_HIP_EHIP, std::string("conv_ks launch: ") + hipGetErrorString(e));
    return 0;
}
static bool ks_serves(int ks_mask, int C) { return (C == 128 && (ks_mask & 1)) || (C == 96 && (ks_mask & 6)); }
static int launch_ks(const ConvLayer& L, const unsigned char* in, unsigned char* out, int H, int W, hipStream_t st, int nb = 0,
                     const unsigned char* const* inb = nullptr, unsigned char* const* outb = nullptr) {
    const unsigned char* const rimg = L.cout == 96 ? L.d_row : L.d_t64;
    if (!rimg) return fail(RIFE_HIP_EINVAL, "layer has no conv_row image");
    if (nb > 4) return fail(RIFE_HIP_EINVAL, "conv_ks batches at most four pairs");
    const S16Geom G(H, W);
    KsArgs a;
    a.in = in; a.out = out; a.img = rimg; a.H = H; a.W = W; a.pitch = G.pitch; a.plane = G.plane(); a.nunits = G.tiles_x * H; a.skip = 1;
    a.nb = nb;
    for (int k = 0; k < nb; k++) { a.inb[k] = inb[k]; a.outb[k] = outb[k]; }
    const int gy = nb > 0 ? nb : 1;
    if (L.cout == 128) return launch_ks_cfg<128, 2, 2>(rimg, a, G.tiles_x, gy, st);
    if (L.cout == 96) return launch_ks_cfg<96, 3, 2>(rimg, a, G.tiles_x, gy, st);
    return fail(RIFE_HIP_EINVAL, "conv_ks serves 96 and 128 channels");
}
#else      // product: no conv_ks
static inline bool ks_serves(int, int) { return false; }
static inline int launch_ks(const ConvLayer&, const unsigned char*, unsigned char*, int, int, hipStream_t, int = 0, const unsigned char* const* = nullptr, unsigned char* const* = nullptr) {
    return fail(RIFE_HIP_ENOSYS, "conv_ks is not part of the product build");
}
#endif

}  // namespace rife
