// Measurement and probe entry points of librife_hip.so - not part of the product path (nothing in include/rife_hip.h declares them;
// tools/*.py and tests/test_gpu_kernels.py bind them by name): kernel ablation benches, the matrix-pipe instruction-mix benchmark, the
// f16-subnormal and fp8 probes, the phase-stamp trace of the dominant kernel.  Included by engine.hip inside its extern "C" block.
// bench-only: time the 8-wave trunk kernel on a synthetic (h x w x c) -> c layer; variant bits: 256 no stores,
// 512 no global loads after chunk 1, 1024 no barriers (the last two compute garbage; timing ablations only)
int rife_hip_bench_conv8(int gpuid, int c, int h, int w, int variant, int iters, float* ms_out) {
    tl_cu_budget = 0;                                                    // bench hooks size their grids for the whole chip, whatever stream this thread used last
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    if (c != 64) return fail(RIFE_HIP_EINVAL, "bench supports c = 64");
    std::vector<float> wts((size_t)c * c * 9, 0.01f), bias(c, 0.f);
    ConvLayer L; L.cin = c; L.cout = c; L.stride = 1; L.epi = EPI_STORE;
    if ((rc = upload_layer(L, wts.data(), bias.data(), nullptr, 0.2f))) return rc;
    float *x = nullptr, *y = nullptr;
    HIPCHK(hipMalloc(&x, (size_t)h * w * c * 4)); HIPCHK(hipMalloc(&y, (size_t)h * w * c * 4));
    HIPCHK(hipMemset(x, 0, (size_t)h * w * c * 4));
    ConvArgs a;
    a.in = x; a.in_ld = c; a.in_coff = 0; a.H = h; a.W = w; a.out = y; a.out_ld = c; a.out_coff = 0;
    a.wpk = L.d_w8; a.bias = L.d_bias; a.slope = L.d_slope; a.res = nullptr; a.res_ld = 0; a.res_coff = 0;
    a.Ho = h; a.Wo = w; a.Cout = c; a.nchunks = L.nchunks8; a.nz = 1; a.tiles_x = (w + 31) / 32; a.ntiles_xy = a.tiles_x * ((h + 7) / 8);
    constexpr int lds = conv8_lds_bytes<2, 8>();
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto run = [&](auto kfn) -> int {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kfn, dim3(a.ntiles_xy), dim3(512), lds, 0, a);
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kfn, dim3(a.ntiles_xy), dim3(512), lds, 0, a);
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = t / iters;
        return 0;
    };
    switch (variant) {
        case 0: rc = run(conv_mfma8_kernel<2, 8, 4, 4096>); break;
        case 256: rc = run(conv_mfma8_kernel<2, 8, 4, 4096 + 256>); break;
        case 512: rc = run(conv_mfma8_kernel<2, 8, 4, 4096 + 512>); break;
        case 1024: rc = run(conv_mfma8_kernel<2, 8, 4, 4096 + 1024>); break;
        case 768: rc = run(conv_mfma8_kernel<2, 8, 4, 4096 + 768>); break;
        case 1792: rc = run(conv_mfma8_kernel<2, 8, 4, 4096 + 1792>); break;
        default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
    }
    (void)hipFree(x); (void)hipFree(y); free_layer(L);
    return rc;
}

// bench-only: ONE layer of the product schedule through launch_conv() on `nstreams` concurrent streams (own tensors per stream, shared weights),
// `iters` launches per stream back to back: ms_out[0] = wall time per launch at saturation (elapsed / (iters * nstreams)), ms_out[1] = one stream alone.
// kind 0: 3x3 stride 1 + PReLU, 1: 3x3 stride 2 + PReLU, 2: Deconvolution 4x4 s2 + PReLU.  in_ld = input pixel stride (>= cin padded to 16).
int rife_hip_bench_layer(int gpuid, int cin, int cout, int h, int w, int kind, int in_ld, int nstreams, int iters, float* ms_out) {
    tl_cu_budget = 0;
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    if (nstreams < 1 || nstreams > 16 || iters < 1) return fail(RIFE_HIP_EINVAL, "bad bench arguments");
    const int kk = kind == 2 ? 16 : 9;
    std::vector<float> wts((size_t)cin * cout * kk), bias(cout), slope(cout);
    uint32_t lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) / (float)(1 << 22); };   // [-1, 1)
    const float ws = std::sqrt(2.0f / (float)(cin * 9));
    for (auto& v : wts) v = (float)(_Float16)(rnd() * ws);
    for (auto& v : bias) v = rnd() * 0.01f;
    for (auto& v : slope) v = 0.25f + 0.25f * rnd();
    ConvLayer L; L.cin = cin; L.cout = cout; L.stride = kind == 1 ? 2 : 1; L.deconv = kind == 2; L.epi = kind == 2 ? EPI_DECONV : EPI_STORE; L.cls = "bench";
    if ((rc = upload_layer(L, wts.data(), bias.data(), slope.data(), 1.0f))) return rc;
    const int ho = kind == 1 ? (h - 1) / 2 + 1 : kind == 2 ? 2 * h : h, wo = kind == 1 ? (w - 1) / 2 + 1 : kind == 2 ? 2 * w : w;
    const size_t nin = (size_t)h * w * in_ld, nout = (size_t)ho * wo * cout;
    std::vector<float> hx(nin);
    for (auto& v : hx) v = rnd();
    std::vector<float*> xs(nstreams, nullptr), ys(nstreams, nullptr);
    std::vector<hipStream_t> st(nstreams, nullptr);
    for (int s = 0; s < nstreams; s++) {
        HIPCHK(hipMalloc(&xs[s], nin * 4)); HIPCHK(hipMalloc(&ys[s], nout * 4));
        HIPCHK(hipMemcpy(xs[s], hx.data(), nin * 4, hipMemcpyHostToDevice));
        HIPCHK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
    }
    auto region = [&](int ns, float& per_launch) -> int {
        for (int s = 0; s < ns; s++)
            for (int i = 0; i < 2; i++) if ((rc = launch_conv(L, {xs[s], in_ld, 0}, h, w, {ys[s], cout, 0}, nullptr, st[s]))) return rc;
        HIPCHK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; i++)
            for (int s = 0; s < ns; s++) if ((rc = launch_conv(L, {xs[s], in_ld, 0}, h, w, {ys[s], cout, 0}, nullptr, st[s]))) return rc;
        HIPCHK(hipDeviceSynchronize());
        per_launch = (float)(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / ((double)iters * ns));
        return 0;
    };
    rc = region(nstreams, ms_out[0]);
    if (!rc) rc = region(1, ms_out[1]);
    for (int s = 0; s < nstreams; s++) { (void)hipFree(xs[s]); (void)hipFree(ys[s]); (void)hipStreamDestroy(st[s]); }
    free_layer(L);
    return rc;
}

// hardware probe: does v_mfma_f32_32x32x16_f16 keep f16 subnormal inputs?  out[0] = sum over k of a_k*b_k with
// a_k = 2^-20 (f16 subnormal), b_k = 1  -> 16 * 2^-20 = 1.52587890625e-05 if preserved, 0 if flushed.
__global__ void k_probe_f16_denorm(float* out) {
    f16x8 a, b;
    for (int e = 0; e < 8; e++) { a[e] = (_Float16)9.5367431640625e-07f; b[e] = (_Float16)1.0f; }
    f32x16 c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

int rife_hip_probe_f16_denorm(int gpuid, float* out) {
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    float* d = nullptr;
    HIPCHK(hipMalloc(&d, 4));
    hipLaunchKernelGGL(k_probe_f16_denorm, dim3(1), dim3(64), 0, 0, d);
    HIPCHK(hipMemcpy(out, d, 4, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
}

// bench-only: what the matrix pipe alone sustains, on random operands, for the instruction mix of one 32 px x 64 ch trunk tile
// (64 input channels x 10 taps): MIX 0 = today's 80 + 80 v_mfma_f32_32x32x16_f16 (hi + lo), MIX 1 = 80 f16 (hi) + 20
// v_mfma_scale_f32_32x32x64_f8f6f4 (lo as scaled fp8), MIX 2 = the 80 hi instructions alone, MIX 3 = 80 f16 (hi) + 80 v_mfma_f32_32x32x16_fp8_fp8 (lo).  16 waves per CU like conv_h2b.
typedef int i32x8 __attribute__((ext_vector_type(8)));
extern "C++" {
template <int MIX>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_bench_mfma_mix(const int* __restrict__ src, float* out, int tiles) {
    const int tid = threadIdx.x;
    f16x8 wa[4], xb[4];
    i32x8 wq[2], xq[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int4 v = reinterpret_cast<const int4*>(src)[(i * 512 + tid) & 4095];
        wa[i] = *reinterpret_cast<f16x8*>(&v);
        v = reinterpret_cast<const int4*>(src)[(2048 + i * 512 + tid) & 4095];
        xb[i] = *reinterpret_cast<f16x8*>(&v);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            wq[i][j] = src[(i * 4096 + j * 512 + tid) & 16383] & 0x7f7f7f7f;      // positive fp8 bytes below NaN
            xq[i][j] = src[(8192 + i * 4096 + j * 512 + tid) & 16383] & 0x7f7f7f7f;
        }
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    if (MIX == 4) {      // round 6: FOUR independent accumulation chains per wave (160 instructions per tile like MIX 0): no wave ever waits for its own previous result
        f32x16 acc4[4];
#pragma unroll
        for (int n = 0; n < 4; n++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc4[n][r] = 0.f;
        for (int t = 0; t < tiles; t++) {
#pragma unroll
            for (int k = 0; k < 40; k++)
#pragma unroll
                for (int n = 0; n < 4; n++) acc4[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[(k + n) & 3], xb[(k + (n >> 1)) & 3], acc4[n], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r++) { acc[0][r] = acc4[0][r] + acc4[2][r]; acc[1][r] = acc4[1][r] + acc4[3][r]; }
    } else
    for (int t = 0; t < tiles; t++) {
#pragma unroll
        for (int k = 0; k < 40; k++) {
#pragma unroll
            for (int n = 0; n < 2; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[(k + n) & 3], xb[k & 3], acc[n], 0, 0, 0);
            if (MIX == 0) {
#pragma unroll
                for (int n = 0; n < 2; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[(k + n + 1) & 3], xb[(k + 2) & 3], acc[n], 0, 0, 0);
            }
            if (MIX == 3) {
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(((long)wq[n][(k & 3) * 2 + 1] << 32) | (unsigned)wq[n][(k & 3) * 2],
                                                                        ((long)xq[k & 1][((k >> 1) & 3) * 2 + 1] << 32) | (unsigned)xq[k & 1][((k >> 1) & 3) * 2], acc[n], 0, 0, 0);
            }
            if (MIX == 1 && (k & 3) == 3) {
#pragma unroll
                for (int n = 0; n < 2; n++)
                    acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wq[n], xq[(k >> 2) & 1], acc[n], 0, 0, 0, 127 - 9, 0, 127 - 13);
            }
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < 2; n++)
#pragma unroll
        for (int r = 0; r < 16; r++) sum += acc[n][r];
    if (sum == 123.456f) out[0] = sum;
}
}  // extern "C++"

// probe: fp8 conventions of gfx950 (OCP e4m3fn expected: 0x38 = 1.0, 0x7e = 448) for the conversion and both fp8 MFMA flavours
__global__ void k_probe_fp8(float* out) {
    const int lane = threadIdx.x;
    const long a1 = 0x3838383838383838L, b2 = 0x4040404040404040L;        // 1.0 x 2.0 in e4m3fn, K = 16 -> 32
    f32x16 c;
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b2, c, 0, 0, 0);
    const int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(1.0f, 448.0f, 0, false);
    const int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(1000.0f, -0.3f, 0, false);
    const int p2 = __builtin_amdgcn_cvt_pk_fp8_f32(0.001953125f, 0.0009765625f, 0, false);      // 2^-9 (min subnormal), 2^-10
    i32x8 a8, b8;
    for (int j = 0; j < 8; j++) { a8[j] = 0x38383838; b8[j] = 0x40404040; }
    f32x16 d;
    for (int r = 0; r < 16; r++) d[r] = 0.f;
    d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, d, 0, 0, 0, 127 - 3, 0, 127 + 1);        // 64 * 2 * 2^-3 * 2^1 = 32
    // lane-dependent operands: which k goes with which k (legacy fp8, K = 16): A = 1.0 everywhere, B byte j of lane half g = 2^(j + 8 g - 6)?  too wide:
    // use B byte j = 1.0 only for j == 3, half 1 -> result must equal A's byte (j = 3, half 1) value for every A pattern
    long aj = 0, bj = 0;
    for (int j = 0; j < 8; j++) aj |= (long)(0x30 + 8 * ((j + (lane >> 5) * 3) & 3)) << (8 * j);       // 0.5, 1, 2, 4 patterns
    if (lane >= 32) bj = 0x38L << 24;
    f32x16 e;
    for (int r = 0; r < 16; r++) e[r] = 0.f;
    e = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(aj, bj, e, 0, 0, 0);
    if (lane == 0) {
        out[0] = c[0]; out[1] = (float)(p0 & 0xffff); out[2] = (float)(p1 & 0xffff); out[3] = (float)(p2 & 0xffff); out[4] = d[0]; out[5] = e[0];
    }
}

int rife_hip_probe_fp8(int gpuid, float* out6) {
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    float* d = nullptr;
    HIPCHK(hipMalloc(&d, 24));
    hipLaunchKernelGGL(k_probe_fp8, dim3(1), dim3(64), 0, 0, d);
    HIPCHK(hipMemcpy(out6, d, 24, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return 0;
}

int rife_hip_bench_mfma_mix(int gpuid, int mix, int tiles, int iters, float* ms_out) {
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    std::vector<int> h(16384);
    uint32_t lcg = 777u;
    for (auto& v : h) {                                  // pairs of f16 in [-1, 1): exponent field 0x30..0x3b, random sign and mantissa
        uint32_t w = 0;
        for (int k = 0; k < 2; k++) {
            lcg = lcg * 1664525u + 1013904223u;
            const uint32_t e = 0x0c + ((lcg >> 28) % 3), m = (lcg >> 8) & 0x3ff, sg = (lcg >> 27) & 1;
            w |= ((sg << 15) | (e << 10) | m) << (16 * k);
        }
        v = (int)w;
    }
    const bool one_wave = (mix & 0x200) != 0;      // 0x200: 256 workgroups of 4 waves = ONE wave per SIMD (the matrix waves of conv_rs / conv_rs2 run like this)
    if (mix & 0x100) for (auto& v : h) v = 0;
    mix &= 0xff;      // 0x100: all-zero operands - the same instruction stream without the data's toggling (tools/mfma_power_peak.py)
    int* d = nullptr; float* o = nullptr;
    HIPCHK(hipMalloc(&d, h.size() * 4)); HIPCHK(hipMalloc(&o, 4));
    HIPCHK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (one_wave) {
            if (mix == 4) hipLaunchKernelGGL(k_bench_mfma_mix<4>, dim3(256), dim3(256), 0, 0, d, o, tiles);
            else hipLaunchKernelGGL(k_bench_mfma_mix<0>, dim3(256), dim3(256), 0, 0, d, o, tiles);
            return;
        }
        if (mix == 0) hipLaunchKernelGGL(k_bench_mfma_mix<0>, dim3(512), dim3(512), 0, 0, d, o, tiles);
        else if (mix == 1) hipLaunchKernelGGL(k_bench_mfma_mix<1>, dim3(512), dim3(512), 0, 0, d, o, tiles);
        else if (mix == 3) hipLaunchKernelGGL(k_bench_mfma_mix<3>, dim3(512), dim3(512), 0, 0, d, o, tiles);
        else if (mix == 4) hipLaunchKernelGGL(k_bench_mfma_mix<4>, dim3(512), dim3(512), 0, 0, d, o, tiles);
        else hipLaunchKernelGGL(k_bench_mfma_mix<2>, dim3(512), dim3(512), 0, 0, d, o, tiles);
    };
    for (int i = 0; i < 3; i++) launch();
    HIPCHK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) launch();
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / iters;
    (void)hipFree(d); (void)hipFree(o); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}

// bench-only: ablations of the split-f16 trunk kernel (variant bits: 256 no stores, 512 no prefetch loads, 1024 no barriers,
// 2048 no LDS staging writes after the first chunk; all but 0/256 compute garbage — timing only)
int rife_hip_bench_h2b(int gpuid, int h, int w, int variant, int iters, float* ms_out) {
    tl_cu_budget = 0;                                                    // bench hooks size their grids for the whole chip, whatever stream this thread used last
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    const int c = 64;
    // realistic operands (fp16-exact random weights, random activations): the matrix pipe clocks down under real data, an
    // all-zero tensor flatters the kernel by ~20 %
    std::vector<float> wts((size_t)c * c * 9), bias(c, 0.f);
    uint32_t lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) / (float)(1 << 22); };   // [-1, 1)
    for (auto& v : wts) v = (float)(_Float16)(rnd() * 0.03f);
    ConvLayer L; L.cin = c; L.cout = c; L.stride = 1; L.epi = EPI_STORE; L.skip = true;
    if ((rc = upload_layer(L, wts.data(), bias.data(), nullptr, 0.2f))) return rc;
    float *x = nullptr, *y = nullptr;
    HIPCHK(hipMalloc(&x, (size_t)h * w * c * 4)); HIPCHK(hipMalloc(&y, (size_t)h * w * c * 4));
    if (variant & 16384) { HIPCHK(hipMemset(x, 0, (size_t)h * w * c * 4)); variant &= ~16384; }
    else {
        std::vector<float> hx((size_t)h * w * c);
        for (auto& v : hx) v = rnd();
        HIPCHK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    }
    ConvArgs a;
    a.in = x; a.in_ld = c; a.in_coff = 0; a.H = h; a.W = w; a.out = y; a.out_ld = c; a.out_coff = 0;
    a.wpk = reinterpret_cast<const float*>(L.d_wh); a.bias = L.d_bias; a.slope = L.d_slope; a.res = nullptr; a.res_ld = 0; a.res_coff = 0;
    a.Ho = h; a.Wo = w; a.Cout = c; a.nchunks = L.nchunksh; a.nz = 1; a.tiles_x = (w + 31) / 32; a.ntiles_xy = a.tiles_x * ((h + 7) / 8);
    constexpr int lds = convh2b_lds_bytes<2, 10>();
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const bool pingpong = variant == 8192;
    auto run = [&](auto kfn) -> int {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kfn, dim3(a.ntiles_xy), dim3(512), lds, 0, a);
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) {
            if (pingpong) { a.in = (i & 1) ? y : x; a.out = (i & 1) ? x : y; }     // like consecutive trunk layers: read what the last launch wrote
            hipLaunchKernelGGL(kfn, dim3(a.ntiles_xy), dim3(512), lds, 0, a);
        }
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = t / iters;
        return 0;
    };
    if (variant == 32768) {          // phase stamps: one launch on an idle GPU, [workgroup][16] 64-bit slots copied to ms_out's neighbour buffer
        long long* stamps = nullptr;
        const size_t nst = (size_t)a.ntiles_xy * 16;
        HIPCHK(hipMalloc(&stamps, nst * 8));
        HIPCHK(hipMemset(stamps, 0, nst * 8));
        a.partial = reinterpret_cast<float*>(stamps);
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_h2b_kernel<2, 10, 4096 + 32768>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL((conv_h2b_kernel<2, 10, 4096 + 32768>), dim3(a.ntiles_xy), dim3(512), lds, 0, a);
        HIPCHK(hipDeviceSynchronize());
        std::vector<long long> hs(nst);
        HIPCHK(hipMemcpy(hs.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen("gpurun_out/h2b_stamps.bin", "wb")) { fwrite(hs.data(), 8, nst, f); fclose(f); }
        *ms_out = (float)a.ntiles_xy;
        (void)hipFree(stamps); (void)hipFree(x); (void)hipFree(y); free_layer(L);
        return 0;
    }
    switch (variant) {
        case 8192: rc = run(conv_h2b_kernel<2, 10, 4096>); break;
        case 0: rc = run(conv_h2b_kernel<2, 10, 4096>); break;
        case 256: rc = run(conv_h2b_kernel<2, 10, 4096 + 256>); break;
        case 512: rc = run(conv_h2b_kernel<2, 10, 4096 + 512>); break;
        case 1024: rc = run(conv_h2b_kernel<2, 10, 4096 + 1024>); break;
        case 2048: rc = run(conv_h2b_kernel<2, 10, 4096 + 2048>); break;
        case 2560: rc = run(conv_h2b_kernel<2, 10, 4096 + 2560>); break;
        case 2816: rc = run(conv_h2b_kernel<2, 10, 4096 + 2816>); break;
        case 3840: rc = run(conv_h2b_kernel<2, 10, 4096 + 3840>); break;
        default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
    }
    (void)hipFree(x); (void)hipFree(y); free_layer(L);
    return rc;
}

// bench-only: the persistent S16 trunk kernel (conv_t64.h) on an h x w tensor of random records, ping-pong between two tensors like
// consecutive trunk layers; variant = ablation bits of conv_t64.h (0 = the product kernel); T64_STAMPS writes gpurun_out/t64_stamps.bin
int rife_hip_bench_t64(int gpuid, int h, int w, int variant, int iters, float* ms_out) {
    tl_cu_budget = 0;                                                    // bench hooks size their grids for the whole chip, whatever stream this thread used last
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    std::vector<float> wts((size_t)64 * 64 * 9), bias(64, 0.f);
    uint32_t lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) / (float)(1 << 22); };   // [-1, 1)
    for (auto& v : wts) v = (float)(_Float16)(rnd() * 0.03f);
    std::vector<unsigned char> img = pack_t64_image(wts.data(), bias.data(), 0.2f);
    const S16Geom G(h, w);
    const size_t nb = G.bytes(64);
    unsigned char *x = nullptr, *y = nullptr, *dimg = nullptr;
    HIPCHK(hipMalloc(&x, nb)); HIPCHK(hipMalloc(&y, nb)); HIPCHK(hipMalloc(&dimg, img.size()));
    HIPCHK(hipMemcpy(dimg, img.data(), img.size(), hipMemcpyHostToDevice));
    {   // random {hi, lo} entries in the interior of every plane, zero border
        std::vector<_Float16> hx(nb / 2, (_Float16)0.f);
        const size_t pl = G.plane() / 2;                       // f16 elements per plane
        for (int yy = 0; yy < h; yy++)
            for (int xx = 0; xx < w; xx++)
                for (int c = 0; c < 4; c++)
                    for (int e = 0; e < 16; e++) {
                        const float v = rnd(); const _Float16 hh = (_Float16)v;
                        const size_t px = ((size_t)(yy + 1) * G.pitch + xx + 1) * 16 + e;
                        hx[(2 * c) * pl + px] = hh; hx[(2 * c + 1) * pl + px] = (_Float16)(v - (float)hh);
                    }
        HIPCHK(hipMemcpy(x, hx.data(), nb, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(y, 0, nb));
    }
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpuid));
    T64Args a;
    a.in = x; a.out = y; a.img = dimg; a.H = h; a.W = w; a.pitch = G.pitch; a.plane = G.plane(); a.tiles_x = G.tiles_x; a.ntiles = G.tiles_x * G.tiles_y; a.reverse = 0; a.nchunks = 4; a.nnt = 1;
    const bool alternate = (variant & 0x10000) != 0; variant &= ~0x10000;
    const int nwg = std::min(t64_wg_per_cu(2) * (cus / 8 * 8), (a.ntiles + 7) / 8 * 8);
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    long long* clk_base = nullptr;
    auto run = [&](auto kfn) -> int {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, T64_LDS));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kfn, dim3(nwg), dim3(T64_NTHR), T64_LDS, 0, a);
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) {
            a.in = (i & 1) ? y : x; a.out = (i & 1) ? x : y;
            a.reverse = alternate ? (i & 1) : 0;
            if (clk_base) a.stamps = clk_base + (size_t)std::min(i, 255) * nwg * 4;     // clock probe: launches 0 .. 254 keep their own records
            hipLaunchKernelGGL(kfn, dim3(nwg), dim3(T64_NTHR), T64_LDS, 0, a);
        }
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = t / iters;
        return 0;
    };
    if (variant & T64_CLK) {       // shader clock during the launch: per workgroup (cycles, 100 MHz ticks) -> stderr
        HIPCHK(hipMalloc(&clk_base, (size_t)nwg * 32 * 256));
        HIPCHK(hipMemset(clk_base, 0, (size_t)nwg * 32 * 256));
        a.stamps = clk_base;
        switch (variant & ~T64_CLK) {
            case 0: rc = run(conv_t64_kernel<T64_CLK>); break;
            case T64_NOSTORE: rc = run(conv_t64_kernel<T64_CLK | T64_NOSTORE>); break;
            case T64_NODMA: rc = run(conv_t64_kernel<T64_CLK | T64_NODMA>); break;
            case T64_NOMATH: rc = run(conv_t64_kernel<T64_CLK | T64_NOMATH>); break;
            case T64_NODMA | T64_NOSTORE: rc = run(conv_t64_kernel<T64_CLK | T64_NODMA | T64_NOSTORE>); break;
            case T64_NOMATH | T64_NOSTORE: rc = run(conv_t64_kernel<T64_CLK | T64_NOMATH | T64_NOSTORE>); break;
            case T64_NOMATH | T64_NODMA: rc = run(conv_t64_kernel<T64_CLK | T64_NOMATH | T64_NODMA>); break;
            default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
        }
        const int nl = std::min(iters, 255);
        std::vector<long long> hs((size_t)nwg * 4 * nl);
        HIPCHK(hipMemcpy(hs.data(), clk_base, hs.size() * 8, hipMemcpyDeviceToHost));
        long long prev_end = 0;
        fprintf(stderr, "t64 clk: variant 0x%x  launch: workgroup life median us | first start -> last end us | gap to the previous launch us | GHz\n", variant & ~T64_CLK);
        for (int l = 0; l < nl; l++) {
            const long long* p = hs.data() + (size_t)l * nwg * 4;
            std::vector<long long> rt;
            long long s0 = LLONG_MAX, e1 = 0, cyc = 0, tk = 0;
            for (int i = 0; i < nwg; i++) {
                rt.push_back(p[4 * i + 2] - p[4 * i + 1]); cyc += p[4 * i]; tk += p[4 * i + 2] - p[4 * i + 1];
                s0 = std::min(s0, p[4 * i + 1]); e1 = std::max(e1, p[4 * i + 2]);
            }
            std::sort(rt.begin(), rt.end());
            if (l < 4 || l % 20 == 0 || l == nl - 1)
                fprintf(stderr, "   %3d: %.2f | %.2f | %.2f | %.3f\n", l, rt[nwg / 2] * 0.01, (e1 - s0) * 0.01, l ? (s0 - prev_end) * 0.01 : 0.0, (double)cyc / (double)tk * 0.1);
            prev_end = e1;
        }
        char fn[64]; snprintf(fn, sizeof fn, "gpurun_out/t64_clk_%x.bin", variant & ~T64_CLK);
        if (FILE* f = fopen(fn, "wb")) { fwrite(hs.data() + (size_t)(nl - 1) * nwg * 4, 8, (size_t)nwg * 4, f); fclose(f); }
        (void)hipFree(clk_base);
    } else if (variant == T64_STAMPS) {
        const size_t nst = (size_t)nwg * T64_TH * 32 * 4;
        HIPCHK(hipMalloc(&a.stamps, nst * 8));
        HIPCHK(hipMemset(a.stamps, 0, nst * 8));
        rc = run(conv_t64_kernel<T64_STAMPS>);
        std::vector<long long> hs(nst);
        HIPCHK(hipMemcpy(hs.data(), a.stamps, nst * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen("gpurun_out/t64_stamps.bin", "wb")) { fwrite(hs.data(), 8, nst, f); fclose(f); }
        (void)hipFree(a.stamps);
    } else switch (variant) {
        case 0: rc = run(conv_t64_kernel<0>); break;
        case T64_NOSTORE: rc = run(conv_t64_kernel<T64_NOSTORE>); break;
        case T64_NODMA: rc = run(conv_t64_kernel<T64_NODMA>); break;
        case T64_NOMATH: rc = run(conv_t64_kernel<T64_NOMATH>); break;
        case T64_NOVMWAIT: rc = run(conv_t64_kernel<T64_NOVMWAIT>); break;
        case T64_NODMA | T64_NOSTORE: rc = run(conv_t64_kernel<T64_NODMA | T64_NOSTORE>); break;
        case T64_NOMATH | T64_NOSTORE: rc = run(conv_t64_kernel<T64_NOMATH | T64_NOSTORE>); break;
        case T64_NOMATH | T64_NODMA: rc = run(conv_t64_kernel<T64_NOMATH | T64_NODMA>); break;
        default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
    }
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(dimg); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

// bench-only: the row-streaming trunk kernel (conv_rs.h) on an h x w tensor of random records.  First its output (walking down, then walking
// up) is compared with conv_t64_kernel's on the same input: stats[0] / stats[1] = differing bytes (down / up), stats[2] / stats[3] = the largest
// difference of the stored values hi + lo x 1e9, stats[4], [5], [7] = chunk, padded row, padded column of it (down), stats[6] = bytes compared.  Then `iters` launches ping-pong between two
// tensors like consecutive trunk layers.  variant = ablation bits of conv_rs.h | 0x10000 (layers alternate direction) | 0x20000 (always up)
// | 0x1000000 * g (g > 0: launch g workgroups instead of one per CU).
// conv_ks_kernel (conv_ks.h) alone on a random C-channel S16 tensor of h x w pixels: ms per launch over `iters` back-to-back launches (ping-pong
// tensors), ablation variants (KS_* bits), and with KS_CLK the per-workgroup timeline on the 100 MHz counter:
// stamps_out[16 * nwg] of the LAST launch (see KS_STAMP in conv_ks.h), *nwg_out = workgroups.  div: ranges = CUs / (NG * div).
extern "C++" {
template <int C, int NB, int CPW>
static int bench_ks_cfg(int gpuid, int h, int w, int variant, int iters, int div, float* ms_out, long long* stamps_out, int* nwg_out) {
    using K = KsCfg<C, NB, CPW>;
    std::vector<float> wts((size_t)C * C * 9), bias(C);
    uint32_t lcg = 4321u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) / (float)(1 << 22); };
    for (auto& v : wts) v = (float)(_Float16)(rnd() * 0.02f);
    for (auto& v : bias) v = rnd() * 0.1f;
    std::vector<unsigned char> img = pack_t64_image(wts.data(), bias.data(), 0.2f, C, 1);
    const S16Geom G(h, w);
    const size_t nb = G.bytes(C);
    unsigned char *x = nullptr, *y = nullptr, *dimg = nullptr;
    HIPCHK(hipMalloc(&x, nb)); HIPCHK(hipMalloc(&y, nb)); HIPCHK(hipMalloc(&dimg, img.size()));
    HIPCHK(hipMemcpy(dimg, img.data(), img.size(), hipMemcpyHostToDevice));
    {
        std::vector<_Float16> hx(nb / 2, (_Float16)0.f);
        const size_t pl = G.plane() / 2;
        for (int yy = 0; yy < h; yy++)
            for (int xx = 0; xx < w; xx++)
                for (int c = 0; c < C / 16; c++)
                    for (int e = 0; e < 16; e++) {
                        const float v = rnd(); const _Float16 hh = (_Float16)v;
                        const size_t px = ((size_t)(yy + 1) * G.pitch + xx + 1) * 16 + e;
                        hx[(2 * c) * pl + px] = hh; hx[(2 * c + 1) * pl + px] = (_Float16)(v - (float)hh);
                    }
        HIPCHK(hipMemcpy(x, hx.data(), nb, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(y, 0, nb));
    }
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpuid));
    KsArgs a;
    a.in = x; a.out = y; a.img = dimg; a.H = h; a.W = w; a.pitch = G.pitch; a.plane = G.plane(); a.nunits = G.tiles_x * h; a.skip = 1;
    int Gr = std::max(1, cus / (K::NG * std::max(1, div)));
    Gr = std::min(Gr, a.nunits);
    if (Gr >= G.tiles_x) Gr = Gr / G.tiles_x * G.tiles_x;
    const int nwg = Gr * K::NG;
    if (nwg_out) *nwg_out = nwg;
    long long* dst = nullptr;
    HIPCHK(hipMalloc(&dst, (size_t)nwg * 16 * 8));
    HIPCHK(hipMemset(dst, 0, (size_t)nwg * 16 * 8));
    a.stamps = dst;
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto run = [&](auto kfn) -> int {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kfn, dim3(nwg), dim3(K::NTHR), K::LDS, 0, a);
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) {
            a.in = (i & 1) ? y : x; a.out = (i & 1) ? x : y;
            hipLaunchKernelGGL(kfn, dim3(nwg), dim3(K::NTHR), K::LDS, 0, a);
        }
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = t / iters;
        return 0;
    };
    int rc;
    switch (variant) {
        case 0: rc = run(conv_ks_kernel<C, NB, CPW, 0>); break;
        case KS_CLK: rc = run(conv_ks_kernel<C, NB, CPW, KS_CLK>); break;
        case KS_NOMATH: rc = run(conv_ks_kernel<C, NB, CPW, KS_NOMATH>); break;
        case KS_NODMA: rc = run(conv_ks_kernel<C, NB, CPW, KS_NODMA>); break;
        case KS_NOSTORE: rc = run(conv_ks_kernel<C, NB, CPW, KS_NOSTORE>); break;
        case KS_NOWEIGHTS: rc = run(conv_ks_kernel<C, NB, CPW, KS_NOWEIGHTS>); break;
        case KS_NOMATH | KS_NODMA | KS_NOSTORE | KS_NOWEIGHTS: rc = run(conv_ks_kernel<C, NB, CPW, KS_NOMATH | KS_NODMA | KS_NOSTORE | KS_NOWEIGHTS>); break;
        case KS_NODMA | KS_NOSTORE: rc = run(conv_ks_kernel<C, NB, CPW, KS_NODMA | KS_NOSTORE>); break;
        default: rc = fail(RIFE_HIP_EINVAL, "unknown conv_ks bench variant");
    }
    if (!rc && stamps_out) HIPCHK(hipMemcpy(stamps_out, dst, (size_t)nwg * 16 * 8, hipMemcpyDeviceToHost));
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(dimg); (void)hipFree(dst);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}
}  // extern "C++"
int rife_hip_bench_ks(int gpuid, int C, int h, int w, int variant, int iters, int div, float* ms_out, long long* stamps_out, int* nwg_out) {
    tl_cu_budget = 0;                                                    // bench hooks size their grids for the whole chip, whatever stream this thread used last
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    if (C == 128) return bench_ks_cfg<128, 2, 2>(gpuid, h, w, variant, iters, div, ms_out, stamps_out, nwg_out);
    if (C == 96) return bench_ks_cfg<96, 3, 2>(gpuid, h, w, variant, iters, div, ms_out, stamps_out, nwg_out);
    return fail(RIFE_HIP_EINVAL, "conv_ks bench: C = 96 or 128");
}

int rife_hip_bench_rs(int gpuid, int h, int w, int variant, int iters, float* ms_out, long long* stats) {
    tl_cu_budget = 0;                                                    // bench hooks size their grids for the whole chip, whatever stream this thread used last
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    if ((h + 1) / 2 < RS_MIN_PAIRS) return fail(RIFE_HIP_EINVAL, "conv_rs needs at least 7 rows");
    std::vector<float> wts((size_t)64 * 64 * 9), bias(64);
    uint32_t lcg = 12345u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) / (float)(1 << 22); };   // [-1, 1)
    for (auto& v : wts) v = (float)(_Float16)(rnd() * 0.03f);
    for (auto& v : bias) v = rnd() * 0.1f;
    std::vector<unsigned char> img = pack_t64_image(wts.data(), bias.data(), 0.2f);
    const S16Geom G(h, w);
    const size_t nb = G.bytes(64);
    unsigned char *x = nullptr, *y = nullptr, *yr = nullptr, *dimg = nullptr;
    HIPCHK(hipMalloc(&x, nb)); HIPCHK(hipMalloc(&y, nb)); HIPCHK(hipMalloc(&yr, nb)); HIPCHK(hipMalloc(&dimg, img.size()));
    HIPCHK(hipMemcpy(dimg, img.data(), img.size(), hipMemcpyHostToDevice));
    {   // random {hi, lo} entries in the interior of every plane, zero border
        std::vector<_Float16> hx(nb / 2, (_Float16)0.f);
        const size_t pl = G.plane() / 2;
        for (int yy = 0; yy < h; yy++)
            for (int xx = 0; xx < w; xx++)
                for (int c = 0; c < 4; c++)
                    for (int e = 0; e < 16; e++) {
                        const float v = rnd(); const _Float16 hh = (_Float16)v;
                        const size_t px = ((size_t)(yy + 1) * G.pitch + xx + 1) * 16 + e;
                        hx[(2 * c) * pl + px] = hh; hx[(2 * c + 1) * pl + px] = (_Float16)(v - (float)hh);
                    }
        HIPCHK(hipMemcpy(x, hx.data(), nb, hipMemcpyHostToDevice));
    }
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpuid));
    RsArgs a;
    a.in = x; a.out = y; a.img = dimg; a.H = h; a.W = w; a.pitch = G.pitch; a.plane = G.plane();
    a.npairs = (h + 1) / 2; a.nunits = G.tiles_x * a.npairs; a.descend = 0;
    const int gover = (variant >> 24) & 0xff;
    const int nwg = std::min(gover ? gover : cus, a.nunits);
    const bool alternate = (variant & 0x10000) != 0, up = (variant & 0x20000) != 0;
    variant &= 0xffff | RS_CLK;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS));
    if (stats) {
        for (int i = 0; i < 8; i++) stats[i] = -1;
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_t64_kernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, T64_LDS));
        T64Args t;
        t.in = x; t.out = yr; t.img = dimg; t.H = h; t.W = w; t.pitch = G.pitch; t.plane = G.plane(); t.tiles_x = G.tiles_x; t.ntiles = G.tiles_x * G.tiles_y; t.reverse = 0; t.nchunks = 4; t.nnt = 1;
        const int twg = std::min(t64_wg_per_cu(2) * (cus / 8 * 8), (t.ntiles + 7) / 8 * 8);
        HIPCHK(hipMemset(yr, 0, nb));
        hipLaunchKernelGGL((conv_t64_kernel<3, 2>), dim3(twg), dim3(T64_NTHR), T64_LDS, 0, t);
        HIPCHK(hipDeviceSynchronize());
        std::vector<unsigned char> ref(nb), got(nb);
        HIPCHK(hipMemcpy(ref.data(), yr, nb, hipMemcpyDeviceToHost));
        for (int dir = 0; dir < 2; dir++) {
            HIPCHK(hipMemset(y, 0, nb));
            a.descend = dir;
            hipLaunchKernelGGL((conv_rs_kernel<0>), dim3(nwg), dim3(RS_NTHR), RS_LDS, 0, a);
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemcpy(got.data(), y, nb, hipMemcpyDeviceToHost));
            // bytes that differ, and the largest difference of the values hi + lo the two kernels stored (another summation order only
            // moves the last bits: ~1e-6; a wrong tap, row or channel is O(0.1))
            long long bad = 0;
            double maxd = 0.0;
            const size_t ple = G.plane() / 2;                            // f16 elements per plane
            const _Float16* rh = reinterpret_cast<const _Float16*>(ref.data()); const _Float16* gh = reinterpret_cast<const _Float16*>(got.data());
            for (size_t i = 0; i < nb; i++) bad += ref[i] != got[i];
            for (int c = 0; c < 4; c++)
                for (size_t e = 0; e < ple; e++) {
                    const double vr = (double)(float)rh[(2 * c) * ple + e] + (double)(float)rh[(2 * c + 1) * ple + e];
                    const double vg = (double)(float)gh[(2 * c) * ple + e] + (double)(float)gh[(2 * c + 1) * ple + e];
                    const double d = vr > vg ? vr - vg : vg - vr;
                    if (!(d <= maxd)) {                                  // also catches NaN
                        maxd = d == d ? d : 1e9;
                        if (dir == 0) { stats[4] = (long long)c; stats[5] = (long long)(e / ((size_t)G.pitch * 16)); stats[7] = (long long)(e % ((size_t)G.pitch * 16) / 16); }
                    }
                }
            stats[dir] = bad;
            stats[2 + dir] = (long long)(maxd * 1e9);
        }
        stats[6] = (long long)nb;
        a.descend = 0;
    }
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    long long* clk_base = nullptr;
    auto run = [&](auto kfn) -> int {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kfn, dim3(nwg), dim3(RS_NTHR), RS_LDS, 0, a);
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) {
            a.in = (i & 1) ? y : x; a.out = (i & 1) ? x : y;
            a.descend = up ? 1 : (alternate ? (i & 1) : 0);
            if (clk_base) a.stamps = clk_base + (size_t)std::min(i, 255) * nwg * 4;
            hipLaunchKernelGGL(kfn, dim3(nwg), dim3(RS_NTHR), RS_LDS, 0, a);
        }
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = t / iters;
        return 0;
    };
    if (variant & RS_CLK) {
        HIPCHK(hipMalloc(&clk_base, (size_t)nwg * 32 * 256));
        HIPCHK(hipMemset(clk_base, 0, (size_t)nwg * 32 * 256));
        a.stamps = clk_base;
        switch (variant & ~RS_CLK) {
            case 0: rc = run(conv_rs_kernel<RS_CLK>); break;
            case RS_NOMATH: rc = run(conv_rs_kernel<RS_CLK | RS_NOMATH>); break;
            case RS_NODMA | RS_NOSTORE: rc = run(conv_rs_kernel<RS_CLK | RS_NODMA | RS_NOSTORE>); break;
            default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
        }
        const int nl = std::min(iters, 255);
        std::vector<long long> hs((size_t)nwg * 4 * nl);
        HIPCHK(hipMemcpy(hs.data(), clk_base, hs.size() * 8, hipMemcpyDeviceToHost));
        long long prev_end = 0;
        fprintf(stderr, "rs clk: variant 0x%x  launch: workgroup life median us | first start -> last end us | gap to the previous launch us | GHz\n", variant & ~RS_CLK);
        for (int l = 0; l < nl; l++) {
            const long long* p = hs.data() + (size_t)l * nwg * 4;
            std::vector<long long> rt;
            long long s0 = LLONG_MAX, e1c = 0, cyc = 0, tk = 0;
            for (int i = 0; i < nwg; i++) {
                rt.push_back(p[4 * i + 2] - p[4 * i + 1]); cyc += p[4 * i]; tk += p[4 * i + 2] - p[4 * i + 1];
                s0 = std::min(s0, p[4 * i + 1]); e1c = std::max(e1c, p[4 * i + 2]);
            }
            std::sort(rt.begin(), rt.end());
            if (l < 4 || l % 20 == 0 || l == nl - 1)
                fprintf(stderr, "   %3d: %.2f | %.2f | %.2f | %.3f\n", l, rt[nwg / 2] * 0.01, (e1c - s0) * 0.01, l ? (s0 - prev_end) * 0.01 : 0.0, (double)cyc / (double)std::max(1LL, tk) * 0.1);
            prev_end = e1c;
        }
        (void)hipFree(clk_base);
    } else switch (variant) {
        case 0: rc = run(conv_rs_kernel<0>); break;
        case RS_NOSTORE: rc = run(conv_rs_kernel<RS_NOSTORE>); break;
        case RS_NODMA: rc = run(conv_rs_kernel<RS_NODMA>); break;
        case RS_NOMATH: rc = run(conv_rs_kernel<RS_NOMATH>); break;
        case RS_PRIO: rc = run(conv_rs_kernel<RS_PRIO>); break;
        case RS_NTLOAD: rc = run(conv_rs_kernel<RS_NTLOAD>); break;
        case RS_NTSTORE: rc = run(conv_rs_kernel<RS_NTSTORE>); break;
        case RS_NTLOAD | RS_NTSTORE: rc = run(conv_rs_kernel<RS_NTLOAD | RS_NTSTORE>); break;
        case RS_NODMA | RS_NOSTORE: rc = run(conv_rs_kernel<RS_NODMA | RS_NOSTORE>); break;
        case RS_NOMATH | RS_NOSTORE: rc = run(conv_rs_kernel<RS_NOMATH | RS_NOSTORE>); break;
        case RS_NOMATH | RS_NODMA: rc = run(conv_rs_kernel<RS_NOMATH | RS_NODMA>); break;
        default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
    }
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(yr); (void)hipFree(dimg); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

// bench-only: the depth-fused trunk kernel (conv_rs2.h) on an h x w tensor of random records, two random layers A and B.  With `stats`: its output (walking
// down, then up) against conv_rs_kernel(A) followed by conv_rs_kernel(B) - the same bytes are expected: stats[0] / stats[1] = differing bytes (down / up),
// stats[2] = bytes compared, stats[3], [4], [5] = plane, padded row, padded column of the first difference (down), stats[6] = segments per strip, stats[7] = workgroups.
// Then `iters` launches ping-pong between two tensors.  variant = ablation bits of conv_rs.h | 0x10000 (launches alternate direction) | 0x20000 (always up)
// | 0x1000000 * g (g > 0: plan for g compute units instead of the chip's).
int rife_hip_bench_rs2(int gpuid, int h, int w, int variant, int iters, float* ms_out, long long* stats) {
    tl_cu_budget = 0;
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    std::vector<float> wts((size_t)64 * 64 * 9), bias(64);
    uint32_t lcg = 777u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) / (float)(1 << 22); };   // [-1, 1)
    std::vector<unsigned char> img[2];
    for (int l = 0; l < 2; l++) {
        for (auto& v : wts) v = (float)(_Float16)(rnd() * 0.03f);
        for (auto& v : bias) v = rnd() * 0.1f;
        img[l] = pack_t64_image(wts.data(), bias.data(), 0.2f);
    }
    const S16Geom G(h, w);
    const size_t nb = G.bytes(64);
    unsigned char *x = nullptr, *y = nullptr, *yr = nullptr, *tm = nullptr, *dimg[2] = {nullptr, nullptr};
    HIPCHK(hipMalloc(&x, nb)); HIPCHK(hipMalloc(&y, nb)); HIPCHK(hipMalloc(&yr, nb)); HIPCHK(hipMalloc(&tm, nb));
    for (int l = 0; l < 2; l++) { HIPCHK(hipMalloc(&dimg[l], img[l].size())); HIPCHK(hipMemcpy(dimg[l], img[l].data(), img[l].size(), hipMemcpyHostToDevice)); }
    {
        std::vector<_Float16> hx(nb / 2, (_Float16)0.f);
        const size_t pl = G.plane() / 2;
        for (int yy = 0; yy < h; yy++)
            for (int xx = 0; xx < w; xx++)
                for (int c = 0; c < 4; c++)
                    for (int e = 0; e < 16; e++) {
                        const float v = rnd(); const _Float16 hh = (_Float16)v;
                        const size_t px = ((size_t)(yy + 1) * G.pitch + xx + 1) * 16 + e;
                        hx[(2 * c) * pl + px] = hh; hx[(2 * c + 1) * pl + px] = (_Float16)(v - (float)hh);
                    }
        HIPCHK(hipMemcpy(x, hx.data(), nb, hipMemcpyHostToDevice));
    }
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpuid));
    const int gover = (variant >> 24) & 0xff;
    const int plan_cus = gover ? gover : cus;
    int kparts, nstrips;
    const int rmin = rs2_plan(h, w, plan_cus, kparts, nstrips);
    if (rmin < 1) return fail(RIFE_HIP_EINVAL, "conv_rs2 bench: empty segments");
    Rs2Args a;
    a.in = x; a.out = y; a.imgA = dimg[0]; a.imgB = dimg[1]; a.H = h; a.W = w; a.pitch = G.pitch; a.plane = G.plane(); a.rowmax = G.pitch - 2;
    a.kparts = kparts; a.nseg = nstrips * kparts; a.descend = 0; a.limit = (int)(nb - 16);
    const int nwg = std::min(plan_cus, a.nseg);
    const bool alternate = (variant & 0x10000) != 0, up = (variant & 0x20000) != 0;
    // 0x80000: "cold" mode - the input is one of four fixed random tensors, the output one of four others, in rotation (1.07 GB at 4K, four times the 256 MB
    // MALL): every launch reads from HBM like a launch inside a pass does, and the data stays what it is (the default mode feeds every launch its predecessor's
    // output, in and out 267 MB together: after a few thousand launches the tensor is whatever x <- layerB(layerA(x)) converges to, and half of it sits in the MALL)
    const bool cold = (variant & 0x80000) != 0;
    const bool fixed = (variant & 0x100000) != 0;      // 0x100000: every launch reads the SAME random tensor x and writes y (267 MB like the ping-pong mode, but the data never changes)
    variant &= 0xffff | RS_CLK;
    unsigned char *xs[4] = {x, nullptr, nullptr, nullptr}, *ys[4] = {y, nullptr, nullptr, nullptr};
    if (cold) for (int k = 1; k < 4; k++) {
        HIPCHK(hipMalloc(&xs[k], nb)); HIPCHK(hipMalloc(&ys[k], nb));
        HIPCHK(hipMemcpy(xs[k], x, nb, hipMemcpyDeviceToDevice)); HIPCHK(hipMemset(ys[k], 0, nb));
    }
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs2_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, RS2_LDS));
    if (stats) {
        for (int i = 0; i < 8; i++) stats[i] = -1;
        std::vector<unsigned char> ref(nb), got(nb);
        if ((h + 1) / 2 >= RS_MIN_PAIRS) {
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_rs_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS));
            RsArgs r;
            r.in = x; r.out = tm; r.img = dimg[0]; r.H = h; r.W = w; r.pitch = G.pitch; r.plane = G.plane(); r.npairs = (h + 1) / 2; r.nunits = G.tiles_x * r.npairs; r.descend = 0;
            const int rwg = std::min(cus, r.nunits);
            HIPCHK(hipMemset(tm, 0, nb)); HIPCHK(hipMemset(yr, 0, nb));
            hipLaunchKernelGGL((conv_rs_kernel<0>), dim3(rwg), dim3(RS_NTHR), RS_LDS, 0, r);
            r.in = tm; r.out = yr; r.img = dimg[1];
            hipLaunchKernelGGL((conv_rs_kernel<0>), dim3(rwg), dim3(RS_NTHR), RS_LDS, 0, r);
        } else {                                                         // tiny tensors: conv_t64 twice (conv_rs needs 7 rows) - another summation order, bytes differ
            return fail(RIFE_HIP_EINVAL, "conv_rs2 bench check needs at least 7 rows");
        }
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(ref.data(), yr, nb, hipMemcpyDeviceToHost));
        for (int dir = 0; dir < 2; dir++) {
            HIPCHK(hipMemset(y, 0, nb));
            a.descend = dir;
            hipLaunchKernelGGL((conv_rs2_kernel<0>), dim3(nwg), dim3(RS2_NTHR), RS2_LDS, 0, a);
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemcpy(got.data(), y, nb, hipMemcpyDeviceToHost));
            long long bad = 0;
            for (size_t i = 0; i < nb; i++)
                if (ref[i] != got[i]) {
                    if (!bad && dir == 0) { stats[3] = (long long)(i / G.plane()); stats[4] = (long long)(i % G.plane() / ((size_t)G.pitch * 32)); stats[5] = (long long)(i % ((size_t)G.pitch * 32) / 32); }
                    bad++;
                }
            stats[dir] = bad;
        }
        stats[2] = (long long)nb; stats[6] = kparts; stats[7] = nwg;
        a.descend = 0;
    }
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto run = [&](auto kfn) -> int {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, RS2_LDS));
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kfn, dim3(nwg), dim3(RS2_NTHR), RS2_LDS, 0, a);
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) {
            if (cold) { a.in = xs[i & 3]; a.out = ys[i & 3]; }
            else if (fixed) { a.in = x; a.out = y; }
            else { a.in = (i & 1) ? y : x; a.out = (i & 1) ? x : y; }
            a.descend = up ? 1 : (alternate ? (i & 1) : 0);
            hipLaunchKernelGGL(kfn, dim3(nwg), dim3(RS2_NTHR), RS2_LDS, 0, a);
        }
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = t / iters;
        return 0;
    };
    if (iters > 0 && (variant & RS_CLK)) {
        // per-workgroup life of the LAST launch: shader cycles and 100 MHz real-time ticks at start / end (conv_rs2.h, TAG & RS_CLK)
        long long* dst = nullptr;
        HIPCHK(hipMalloc(&dst, (size_t)nwg * 32)); HIPCHK(hipMemset(dst, 0, (size_t)nwg * 32));
        a.stamps = dst;
        rc = run(conv_rs2_kernel<RS_CLK>);
        std::vector<long long> hs((size_t)nwg * 4);
        HIPCHK(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(dst);
        long long s0 = LLONG_MAX, e1c = 0;
        std::vector<double> life(nwg), ghz(nwg), start(nwg);
        for (int i = 0; i < nwg; i++) { s0 = std::min(s0, hs[4 * i + 1]); e1c = std::max(e1c, hs[4 * i + 2]); }
        for (int i = 0; i < nwg; i++) {
            life[i] = (hs[4 * i + 2] - hs[4 * i + 1]) * 0.01; start[i] = (hs[4 * i + 1] - s0) * 0.01;
            ghz[i] = (double)hs[4 * i] / std::max(1.0, (double)(hs[4 * i + 2] - hs[4 * i + 1])) * 0.1;
        }
        auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
        fprintf(stderr, "rs2 clk (%s), last launch, %d workgroups: first start -> last end %.1f us | workgroup life us min %.1f p10 %.1f median %.1f p90 %.1f max %.1f | start spread us median %.1f max %.1f | GHz median %.3f min %.3f max %.3f\n",
                cold ? "cold" : fixed ? "fixed input" : "ping-pong", nwg, (e1c - s0) * 0.01, pct(life, 0), pct(life, 0.1), pct(life, 0.5), pct(life, 0.9), pct(life, 1), pct(start, 0.5), pct(start, 1), pct(ghz, 0.5), pct(ghz, 0), pct(ghz, 1));
        for (int x = 0; x < 8; x++) {                                    // workgroup b runs on XCD b % 8
            std::vector<double> v; for (int i = x; i < nwg; i += 8) v.push_back(life[i]);
            if (!v.empty()) fprintf(stderr, "   XCD %d: life median %.1f max %.1f us\n", x, pct(v, 0.5), pct(v, 1));
        }
    } else if (iters > 0) switch (variant) {
        case 0: rc = run(conv_rs2_kernel<0>); break;
        case RS_NOSTORE: rc = run(conv_rs2_kernel<RS_NOSTORE>); break;
        case RS_NODMA: rc = run(conv_rs2_kernel<RS_NODMA>); break;
        case RS_NOMATH: rc = run(conv_rs2_kernel<RS_NOMATH>); break;
        case RS_NODMA | RS_NOSTORE: rc = run(conv_rs2_kernel<RS_NODMA | RS_NOSTORE>); break;
        case RS_NOMATH | RS_NOSTORE: rc = run(conv_rs2_kernel<RS_NOMATH | RS_NOSTORE>); break;
        case RS_NOMATH | RS_NODMA: rc = run(conv_rs2_kernel<RS_NOMATH | RS_NODMA>); break;
        case RS_NODMA | RS_NOSTORE | RS2_NOFRAG: rc = run(conv_rs2_kernel<RS_NODMA | RS_NOSTORE | RS2_NOFRAG>); break;
        case RS_NODMA | RS_NOSTORE | RS2_NOLO: rc = run(conv_rs2_kernel<RS_NODMA | RS_NOSTORE | RS2_NOLO>); break;
        case RS2_STAMPS: case RS2_STAMPS | RS_NODMA | RS_NOSTORE: {
            // barrier trace of one workgroup (the middle one), last launch: per barrier, cycles from the previous release to each wave's arrival, and who came last
            long long* dst = nullptr;
            const size_t ns = (size_t)8 * RS2_NSTAMP * 2;
            HIPCHK(hipMalloc(&dst, ns * 8)); HIPCHK(hipMemset(dst, 0, ns * 8));
            a.stamps = dst; a.stamp_wg = nwg / 2;
            rc = variant == RS2_STAMPS ? run(conv_rs2_kernel<RS2_STAMPS>) : run(conv_rs2_kernel<RS2_STAMPS | RS_NODMA | RS_NOSTORE>);
            std::vector<long long> hs(ns);
            HIPCHK(hipMemcpy(hs.data(), dst, ns * 8, hipMemcpyDeviceToHost));
            (void)hipFree(dst);
            const int nbar = std::min(RS2_NSTAMP, h / kparts + 9);
            fprintf(stderr, "rs2 barrier trace, variant 0x%x, workgroup %d of %d, %d rows per segment: barrier | cycles since the previous release: arrival of CA0 CA1 CB0 CB1 L EA0 EA1 EB | release | last\n", variant, a.stamp_wg, nwg, h / kparts);
            static const char* const names[8] = {"CA0", "CA1", "CB0", "CB1", "L", "EA0", "EA1", "EB"};
            for (int b = 1; b < nbar; b++) {
                long long prev = 0, rel = 0;
                for (int wv = 0; wv < 8; wv++) { prev = std::max(prev, hs[((size_t)wv * RS2_NSTAMP + b - 1) * 2 + 1]); rel = std::max(rel, hs[((size_t)wv * RS2_NSTAMP + b) * 2 + 1]); }
                int last = 0; long long la = 0;
                fprintf(stderr, "  %3d |", b);
                for (int wv = 0; wv < 8; wv++) {
                    const long long arr = hs[((size_t)wv * RS2_NSTAMP + b) * 2];
                    if (arr > la) { la = arr; last = wv; }
                    fprintf(stderr, " %5lld", arr - prev);
                }
                fprintf(stderr, " | %5lld | %s\n", rel - prev, names[last]);
            }
            break;
        }
        default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
    }
    for (int k = 1; k < 4; k++) { if (xs[k]) (void)hipFree(xs[k]); if (ys[k]) (void)hipFree(ys[k]); }
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(yr); (void)hipFree(tm); (void)hipFree(dimg[0]); (void)hipFree(dimg[1]); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return rc;
}

// probe of the block-scaled fp8 matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4, both operands e4m3): raw per-lane operand dwords in, the
// wave's 16 accumulator registers per lane out; the scale dwords go through VGPRs (tools/probes/mx_probe.py pins the operand layout against numpy)
__global__ void k_probe_mx(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b, uint32_t sa, uint32_t sb, float* __restrict__ d) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
#pragma unroll
    for (int j = 0; j < 8; j++) { av[j] = (int)a[lane * 8 + j]; bv[j] = (int)b[lane * 8 + j]; }
    int va = (int)sa, vb = (int)sb;
    asm volatile("" : "+v"(va), "+v"(vb));
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; r++) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, va, 0, vb);
#pragma unroll
    for (int r = 0; r < 16; r++) d[lane * 16 + r] = c[r];
}
int rife_hip_probe_mx(int gpuid, const uint32_t* a, const uint32_t* b, uint32_t sa, uint32_t sb, float* d) {
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    uint32_t *da = nullptr, *db = nullptr; float* dd = nullptr;
    HIPCHK(hipMalloc(&da, 64 * 8 * 4)); HIPCHK(hipMalloc(&db, 64 * 8 * 4)); HIPCHK(hipMalloc(&dd, 64 * 16 * 4));
    HIPCHK(hipMemcpy(da, a, 64 * 8 * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(db, b, 64 * 8 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe_mx, dim3(1), dim3(64), 0, 0, da, db, sa, sb, dd);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(d, dd, 64 * 16 * 4, hipMemcpyDeviceToHost));
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd);
    return 0;
}

// fills the workgroup's whole LDS allocation with a pattern and leaves: launched between the probe's launches, it decides what the next
// kernel finds in LDS locations it does not write itself
__global__ void k_lds_scrub(uint32_t pattern, int ndw, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    uint32_t* l = reinterpret_cast<uint32_t*>(ldsb);
    for (int i = threadIdx.x; i < ndw; i += blockDim.x) l[i] = pattern;
    __syncthreads();
    if (pattern == 0x12345678u && l[(threadIdx.x * 7) % ndw] != pattern) sink[0] = 1;      // keeps the stores alive
}

// tools/probes/stem_bisect.py: path of a code object whose stem0_fused_kernel<4, 2, 0> / <2, 2, 0> rife_hip_probe_stem_det launches instead of the built-in ones
static std::string g_stem_hsaco;
static long long g_probe_extra[3] = {0, 0, 0};      // launch 0 of the external kernel vs the built-in one: differing floats, NaNs
int rife_hip_probe_set_stem_hsaco(const char* path) { g_stem_hsaco = path ? path : ""; return 0; }
int rife_hip_probe_last_extra(long long* out3) { out3[0] = g_probe_extra[0]; out3[1] = g_probe_extra[1]; out3[2] = g_probe_extra[2]; return 0; }      // [2]: input buffers modified by the launches

// probe: is the fused stem kernel deterministic in isolation?  Random frames, flows (some leaving the frame), mask and weights; `reps` launches
// into separate outputs, compared on the host: mismatch[r] = floats of launch r that differ from launch 0.  variant = S + 16 x ABL.
int rife_hip_probe_stem_det(int gpuid, int variant, int wp, int hp, int reps, long long* mismatch) {
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    const int S = variant & 15, ABL = variant >> 4;
    const size_t P = (size_t)wp * hp;
    const int Hb = hp / S, Wb = wp / S, Ho = Hb / 2, Wo = Wb / 2;
    const int cout = S == 1 ? 32 : (S == 2 ? 48 : 64), NSv = S == 1 ? 1 : 2;
    std::vector<uint32_t> hi0(P), hi1(P);
    std::vector<float> hF(P * 4), hM(P), hb(64), hs(64);
    std::vector<_Float16> hw((size_t)9 * 2 * NSv * 32 * 8);
    uint32_t lcg = 777u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) / (float)(1 << 22); };
    for (size_t i = 0; i < P; i++) { lcg = lcg * 1664525u + 1013904223u; hi0[i] = lcg & 0xffffffu; lcg = lcg * 1664525u + 1013904223u; hi1[i] = lcg & 0xffffffu; }
    for (auto& v : hF) v = rnd() * 9.f;
    for (auto& v : hM) v = rnd();
    for (auto& v : hb) v = rnd() * 0.1f;
    for (auto& v : hs) v = 0.2f;
    for (auto& v : hw) v = (_Float16)(rnd() * 0.2f);
    uint32_t *i0 = nullptr, *i1 = nullptr; float4* F = nullptr; float *M = nullptr, *bias = nullptr, *slope = nullptr; void* wh = nullptr;
    HIPCHK(hipMalloc(&i0, P * 4)); HIPCHK(hipMalloc(&i1, P * 4)); HIPCHK(hipMalloc(&F, P * 16)); HIPCHK(hipMalloc(&M, P * 4));
    HIPCHK(hipMalloc(&bias, 256)); HIPCHK(hipMalloc(&slope, 256)); HIPCHK(hipMalloc(&wh, hw.size() * 2));
    HIPCHK(hipMemcpy(i0, hi0.data(), P * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(i1, hi1.data(), P * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(F, hF.data(), P * 16, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(M, hM.data(), P * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(bias, hb.data(), 256, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(slope, hs.data(), 256, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(wh, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    const size_t nout = (size_t)Ho * Wo * cout;
    std::vector<float*> outs(reps, nullptr);
    for (int r = 0; r < reps; r++) { HIPCHK(hipMalloc(&outs[r], nout * 4)); HIPCHK(hipMemset(outs[r], 0, nout * 4)); }
    const int nb_dbg = ((Wo + 31) / 32) * ((Ho + 3) / 4);
    const size_t ndbg = (size_t)nb_dbg * 512 * 12;
    std::vector<float*> dbgs(reps, nullptr);
    if (ABL & 1024) for (int r = 0; r < reps; r++) { HIPCHK(hipMalloc(&dbgs[r], ndbg * 4)); HIPCHK(hipMemset(dbgs[r], 0, ndbg * 4)); }
    StemFusedArgs fa;
    fa.img0 = i0; fa.img1 = i1; fa.F = F; fa.M = M; fa.wpk = wh; fa.bias = bias; fa.slope = slope; fa.timestep = 0.5f; fa.tsp = nullptr;
    fa.wp = wp; fa.hp = hp; fa.Ho = Ho; fa.Wo = Wo; fa.out_ld = cout; fa.Cout = cout; fa.tiles_x = (Wo + 31) / 32;
    const int nb = fa.tiles_x * ((Ho + 3) / 4);
    auto run = [&](auto kfn, int lds) -> int {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int r = 0; r < reps; r++) { fa.out = outs[r]; fa.dbg = dbgs[r]; hipLaunchKernelGGL(kfn, dim3(nb), dim3(512), lds, 0, fa); }
        HIPCHK(hipDeviceSynchronize());
        return 0;
    };
    if (!g_stem_hsaco.empty() && (S == 4 || S == 2) && (ABL == 0 || ABL == 4096)) {
        // the kernel from an externally assembled code object (tools/probes/stem_bisect.py: the compiler's assembly with wait states inserted)
        hipModule_t mod = nullptr; hipFunction_t fn = nullptr;
        HIPCHK(hipModuleLoad(&mod, g_stem_hsaco.c_str()));
        const std::string fname = std::string("_ZN4rife18stem0_fused_kernelILi") + (S == 4 ? "4" : "2") + "ELi2ELi" + (ABL ? "4096" : "0") + "EEEvNS_13StemFusedArgsE";
        HIPCHK(hipModuleGetFunction(&fn, mod, fname.c_str()));
        const Switches psw = read_switches();
        const int ldsb_ext = psw.probe_lds >= 0 ? psw.probe_lds : stemf_lds_bytes<2>();      // > 80 KB: one workgroup per CU
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb_ext));
        const char* scrub = psw.probe_scrub;
        uint32_t* sink = nullptr;
        if (scrub) { HIPCHK(hipMalloc(&sink, 4)); HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_scrub), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
        for (int r = 0; r < reps; r++) {
            if (scrub) {      // every CU's LDS (one 160 KB workgroup per CU at a time, many rounds) <- pattern, alternating if "alt"
                const uint32_t pat = std::strcmp(scrub, "alt") == 0 ? ((r & 1) ? 0x7fc00000u : 0u) : (uint32_t)std::strtoul(scrub, nullptr, 16);
                hipLaunchKernelGGL(k_lds_scrub, dim3(2048), dim3(512), 160 * 1024, 0, pat, 160 * 1024 / 4, sink);
            }
            fa.out = outs[r]; fa.dbg = dbgs[r];
            size_t sz = sizeof(fa);
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &fa, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            HIPCHK(hipModuleLaunchKernel(fn, nb, 1, 1, 512, 1, 1, ldsb_ext, 0, nullptr, cfg));
        }
        HIPCHK(hipDeviceSynchronize());
        (void)hipModuleUnload(mod);
        if (sink) (void)hipFree(sink);
        {   // did the launches modify their INPUTS (an out-of-bounds store would explain launches that differ from launch 0)?
            std::vector<uint32_t> c0(P), c1(P); std::vector<float> cF(P * 4), cM(P);
            HIPCHK(hipMemcpy(c0.data(), i0, P * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(c1.data(), i1, P * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(cF.data(), F, P * 16, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(cM.data(), M, P * 4, hipMemcpyDeviceToHost));
            std::vector<_Float16> cw(hw.size()); HIPCHK(hipMemcpy(cw.data(), wh, hw.size() * 2, hipMemcpyDeviceToHost));
            g_probe_extra[2] = (long long)(std::memcmp(c0.data(), hi0.data(), P * 4) != 0) + (std::memcmp(c1.data(), hi1.data(), P * 4) != 0) + (std::memcmp(cF.data(), hF.data(), P * 16) != 0)
                               + (std::memcmp(cM.data(), hM.data(), P * 4) != 0) + (std::memcmp(cw.data(), hw.data(), hw.size() * 2) != 0);
        }
        {   // launch 0 of the external kernel against the built-in (library flags) kernel on the same inputs: differing floats, NaNs
            float* refo = nullptr;
            HIPCHK(hipMalloc(&refo, nout * 4)); HIPCHK(hipMemset(refo, 0, nout * 4));
            fa.out = refo; fa.dbg = nullptr;
            if (S == 4) { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<4, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>())); hipLaunchKernelGGL((stem0_fused_kernel<4, 2, 0>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), 0, fa); }
            else { HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(stem0_fused_kernel<2, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, stemf_lds_bytes<2>())); hipLaunchKernelGGL((stem0_fused_kernel<2, 2, 0>), dim3(nb), dim3(512), stemf_lds_bytes<2>(), 0, fa); }
            HIPCHK(hipDeviceSynchronize());
            std::vector<float> a0(nout), a1(nout);
            HIPCHK(hipMemcpy(a0.data(), outs[0], nout * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(a1.data(), refo, nout * 4, hipMemcpyDeviceToHost));
            g_probe_extra[0] = g_probe_extra[1] = 0;
            for (size_t k = 0; k < nout; k++) { g_probe_extra[0] += std::memcmp(&a0[k], &a1[k], 4) != 0; g_probe_extra[1] += a0[k] != a0[k]; }
            (void)hipFree(refo);
        }
    } else
    switch (variant) {
        case 4: rc = run(stem0_fused_kernel<4, 2, 0>, stemf_lds_bytes<2>()); break;
        case 2: rc = run(stem0_fused_kernel<2, 2, 0>, stemf_lds_bytes<2>()); break;
        case 1 + 16 * 256: rc = run(stem0_fused_kernel<1, 1, 256>, (stemf_lds_bytes<1, 256>())); break;
        case 1: rc = run(stem0_fused_kernel<1, 1, 0>, stemf_lds_bytes<1>()); break;
        case 4 + 16 * 1024: rc = run(stem0_fused_kernel<4, 2, 1024>, stemf_lds_bytes<2>()); break;
        case 4 + 16 * 4096: rc = run(stem0_fused_kernel<4, 2, 4096>, stemf_lds_bytes<2>()); break;
        case 2 + 16 * 4096: rc = run(stem0_fused_kernel<2, 2, 4096>, stemf_lds_bytes<2>()); break;
        default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
    }
    if (!rc) {
        std::vector<float> h0(nout), hr(nout);
        HIPCHK(hipMemcpy(h0.data(), outs[0], nout * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < reps; r++) {
            HIPCHK(hipMemcpy(hr.data(), outs[r], nout * 4, hipMemcpyDeviceToHost));
            long long n = 0;
            int shown = 0;
            for (size_t k = 0; k < nout; k++)
                if (std::memcmp(&h0[k], &hr[k], 4) != 0) {
                    n++;
                    const size_t px = k / cout; const int ch = (int)(k % cout);
                    if ((ch == 0 || ch == 63 % cout) && shown < (read_switches().probe_quiet ? 0 : 40)) { fprintf(stderr, "  launch %d: out (y %zu, x %zu) ch %d: %.9g vs %.9g\n", r, px / Wo, px % Wo, ch, h0[k], hr[k]); shown++; }
                }
            mismatch[r] = n;
        }
    }
    if (!rc && (ABL & 1024)) {      // which thread's gathered pixel differs, and in which of the 12 channels {warp0 rgb, warp1 rgb, t, M, F / S xyzw}
        std::vector<float> d0(ndbg), dr(ndbg);
        HIPCHK(hipMemcpy(d0.data(), dbgs[0], ndbg * 4, hipMemcpyDeviceToHost));
        for (int r = 1; r < reps; r++) {
            HIPCHK(hipMemcpy(dr.data(), dbgs[r], ndbg * 4, hipMemcpyDeviceToHost));
            int shown = 0;
            for (size_t k = 0; k < ndbg && shown < 60; k++)
                if (std::memcmp(&d0[k], &dr[k], 4) != 0) {
                    const size_t t = k / 12;
                    fprintf(stderr, "  dbg launch %d: workgroup %zu tid %zu (halo row %zu col %zu) channel %d: %.9g vs %.9g\n", r, t / 512, t % 512, (t % 512) / 65, (t % 512) % 65, (int)(k % 12), d0[k], dr[k]);
                    shown++;
                }
        }
    }
    for (float* o : dbgs) (void)hipFree(o);
    for (float* o : outs) (void)hipFree(o);
    (void)hipFree(i0); (void)hipFree(i1); (void)hipFree(F); (void)hipFree(M); (void)hipFree(bias); (void)hipFree(slope); (void)hipFree(wh);
    return rc;
}

// bench-only: tail_rs_kernel (tail_rs.h) on a wp x hp frame with random trunk / flows / weights, variant = TRS_* ablation bits
int rife_hip_bench_tail_rs(int gpuid, int wp, int hp, int variant, int iters, float* ms_out) {
    tl_cu_budget = 0;                                                    // bench hooks size their grids for the whole chip, whatever stream this thread used last
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    const size_t P = (size_t)wp * hp;
    const int Hq = hp / 4, Wq = wp / 4;
    const S16Geom G(Hq, Wq);
    uint32_t lcg = 4242u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) / (float)(1 << 22); };
    std::vector<uint32_t> himg(P); for (auto& v : himg) { lcg = lcg * 1664525u + 1013904223u; v = lcg & 0xffffffu; }
    std::vector<float> hF(P * 4), hM(P), hb(32);
    for (auto& v : hF) v = rnd() * 6.f;
    for (auto& v : hM) v = rnd();
    for (auto& v : hb) v = rnd() * 0.1f;
    std::vector<_Float16> hw((size_t)4 * 16 * 2 * 32 * 8), hx(G.bytes(64) / 2);
    for (auto& v : hw) v = (_Float16)(rnd() * 0.05f);
    for (auto& v : hx) v = (_Float16)(rnd() * 0.5f);
    uint32_t *i0 = nullptr, *i1 = nullptr; float4* F = nullptr; float *M = nullptr, *bias = nullptr; void* w = nullptr; unsigned char *x = nullptr, *out = nullptr;
    HIPCHK(hipMalloc(&i0, P * 4)); HIPCHK(hipMalloc(&i1, P * 4)); HIPCHK(hipMalloc(&F, P * 16)); HIPCHK(hipMalloc(&M, P * 4));
    HIPCHK(hipMalloc(&bias, 128)); HIPCHK(hipMalloc(&w, hw.size() * 2)); HIPCHK(hipMalloc(&x, G.bytes(64))); HIPCHK(hipMalloc(&out, P * 3));
    HIPCHK(hipMemcpy(i0, himg.data(), P * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(i1, himg.data(), P * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(F, hF.data(), P * 16, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(M, hM.data(), P * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(bias, hb.data(), 128, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(x, hx.data(), G.bytes(64), hipMemcpyHostToDevice));
    TailRsArgs a;
    a.in = x; a.w = w; a.bias = bias; a.img0 = i0; a.img1 = i1; a.F = F; a.M = M; a.out = reinterpret_cast<uint8_t*>(out);
    a.w_ = wp; a.h_ = hp; a.wp = wp; a.hp = hp; a.Hq = Hq; a.Wq = Wq; a.pitch = G.pitch; a.plane = G.plane();
    a.nunits = ((Wq + 31) / 32) * Hq;
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpuid));
    const int nwg = std::min(2 * cus, a.nunits);
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto run = [&](auto kfn) -> int {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, TRS_LDS));
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(kfn, dim3(nwg), dim3(TRS_NTHR), TRS_LDS, 0, a);
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kfn, dim3(nwg), dim3(TRS_NTHR), TRS_LDS, 0, a);
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = t / iters;
        return 0;
    };
    switch (variant) {
        case 0: rc = run(tail_rs_kernel<0>); break;
        case TRS_NOTAPS: rc = run(tail_rs_kernel<TRS_NOTAPS>); break;
        case TRS_NOFM: rc = run(tail_rs_kernel<TRS_NOFM>); break;
        case TRS_NOROW: rc = run(tail_rs_kernel<TRS_NOROW>); break;
        case TRS_NOTAPS | TRS_NOFM | TRS_NOROW: rc = run(tail_rs_kernel<TRS_NOTAPS | TRS_NOFM | TRS_NOROW>); break;
        case TRS_NOMATH: rc = run(tail_rs_kernel<TRS_NOMATH>); break;
        case TRS_NOSTORE: rc = run(tail_rs_kernel<TRS_NOSTORE>); break;
        case TRS_NOPIX: rc = run(tail_rs_kernel<TRS_NOPIX>); break;
        case TRS_NOTAPS | TRS_NOFM | TRS_NOROW | TRS_NOPIX: rc = run(tail_rs_kernel<TRS_NOTAPS | TRS_NOFM | TRS_NOROW | TRS_NOPIX>); break;
        case TRS_NOTAPS | TRS_NOFM | TRS_NOROW | TRS_NOMATH | TRS_NOSTORE: rc = run(tail_rs_kernel<TRS_NOTAPS | TRS_NOFM | TRS_NOROW | TRS_NOMATH | TRS_NOSTORE>); break;
        default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
    }
    (void)hipFree(i0); (void)hipFree(i1); (void)hipFree(F); (void)hipFree(M); (void)hipFree(bias); (void)hipFree(w); (void)hipFree(x); (void)hipFree(out);
    return rc;
}

// bench-only: stem_rs_kernel (stem_rs.h) on a wp x hp frame with random frames / flows / weights, variant = SRS_* ablation bits | 0x100 * g
// (g > 0: g workgroups per CU instead of two)
int rife_hip_bench_stem_rs(int gpuid, int wp, int hp, int variant, int iters, float* ms_out, long long* stamps_out) {
    tl_cu_budget = 0;                                                    // bench hooks size their grids for the whole chip, whatever stream this thread used last
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    const size_t P = (size_t)wp * hp;
    const int Hq = hp / 4, Wq = wp / 4;
    const S16Geom G(Hq, Wq);
    uint32_t lcg = 777u;
    auto rnd = [&]() { lcg = lcg * 1664525u + 1013904223u; return (float)((int)(lcg >> 9) - (1 << 22)) / (float)(1 << 22); };   // [-1, 1)
    std::vector<uint32_t> himg(P); for (auto& v : himg) { lcg = lcg * 1664525u + 1013904223u; v = lcg & 0xffffffu; }
    std::vector<float> hF(P * 4), hM(P), hb(64 + 64);
    for (auto& v : hF) v = rnd() * 6.f;
    for (auto& v : hM) v = rnd();
    for (auto& v : hb) v = rnd() * 0.1f;
    std::vector<_Float16> hw0(9 * 2 * 32 * 8), hw1(2 * 9 * 2 * 64 * 8);
    for (auto& v : hw0) v = (_Float16)(rnd() * 0.1f);
    for (auto& v : hw1) v = (_Float16)(rnd() * 0.05f);
    uint32_t *i0 = nullptr, *i1 = nullptr; float4* F = nullptr; float *M = nullptr, *bias = nullptr, *slope = nullptr; void *w0 = nullptr, *w1 = nullptr; unsigned char* out = nullptr;
    HIPCHK(hipMalloc(&i0, P * 4)); HIPCHK(hipMalloc(&i1, P * 4)); HIPCHK(hipMalloc(&F, P * 16)); HIPCHK(hipMalloc(&M, P * 4));
    HIPCHK(hipMalloc(&bias, 512)); HIPCHK(hipMalloc(&slope, 512)); HIPCHK(hipMalloc(&w0, hw0.size() * 2)); HIPCHK(hipMalloc(&w1, hw1.size() * 2)); HIPCHK(hipMalloc(&out, G.bytes(64)));
    HIPCHK(hipMemcpy(i0, himg.data(), P * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(i1, himg.data(), P * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(F, hF.data(), P * 16, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(M, hM.data(), P * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(bias, hb.data(), 512, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(slope, hb.data(), 512, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(w0, hw0.data(), hw0.size() * 2, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(w1, hw1.data(), hw1.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(out, 0, G.bytes(64)));
    StemRsArgs a;
    a.img0 = i0; a.img1 = i1; a.F = F; a.M = M; a.w0 = w0; a.bias0 = bias; a.slope0 = slope; a.w1 = w1; a.bias1 = bias + 32; a.slope1 = slope + 32;
    a.out = out; a.timestep = 0.5f; a.tsp = nullptr; a.wp = wp; a.hp = hp; a.Hq = Hq; a.Wq = Wq; a.pitch = G.pitch; a.plane = G.plane();
    a.nunits = ((Wq + SRS_SW - 1) / SRS_SW) * Hq;
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, gpuid));
    const int per = (variant >> 8) & 0xff;
    const int nwg = std::min((per ? per : 2) * cus, a.nunits);
    long long* dst = nullptr;
    HIPCHK(hipMalloc(&dst, (size_t)nwg * 64 * 8));
    HIPCHK(hipMemset(dst, 0, (size_t)nwg * 64 * 8));
    a.stamps = dst;
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto run = [&](auto kfn) -> int {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, SRS_LDS));
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(kfn, dim3(nwg), dim3(SRS_NTHR), SRS_LDS, 0, a);
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kfn, dim3(nwg), dim3(SRS_NTHR), SRS_LDS, 0, a);
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = t / iters;
        return 0;
    };
    switch (variant & 0xff) {
        case 0: rc = run(stem_rs_kernel<0>); break;
        case SRS_NOTAPS: rc = run(stem_rs_kernel<SRS_NOTAPS>); break;
        case SRS_NOFM: rc = run(stem_rs_kernel<SRS_NOFM>); break;
        case SRS_NOTAPS | SRS_NOFM: rc = run(stem_rs_kernel<SRS_NOTAPS | SRS_NOFM>); break;
        case SRS_NOMATH: rc = run(stem_rs_kernel<SRS_NOMATH>); break;
        case SRS_NOSTORE: rc = run(stem_rs_kernel<SRS_NOSTORE>); break;
        case SRS_NOFINISH: rc = run(stem_rs_kernel<SRS_NOFINISH>); break;
        case SRS_NOTAPS | SRS_NOFM | SRS_NOFINISH: rc = run(stem_rs_kernel<SRS_NOTAPS | SRS_NOFM | SRS_NOFINISH>); break;
        case SRS_NOTAPS | SRS_NOFM | SRS_NOFINISH | SRS_NOSTORE: rc = run(stem_rs_kernel<SRS_NOTAPS | SRS_NOFM | SRS_NOFINISH | SRS_NOSTORE>); break;
        case SRS_NOTAPS | SRS_NOFM | SRS_NOMATH | SRS_NOSTORE: rc = run(stem_rs_kernel<SRS_NOTAPS | SRS_NOFM | SRS_NOMATH | SRS_NOSTORE>); break;
        case SRS_CLK: rc = run(stem_rs_kernel<SRS_CLK>); break;
        case SRS_CLK | SRS_NOTAPS | SRS_NOFM | SRS_NOFINISH | SRS_NOSTORE: rc = run(stem_rs_kernel<SRS_CLK | SRS_NOTAPS | SRS_NOFM | SRS_NOFINISH | SRS_NOSTORE>); break;
        default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
    }
    if (rc == 0 && stamps_out && (variant & SRS_CLK)) {     // mean over the workgroups of the LAST launch, per wave and phase, in cycles per step
        std::vector<long long> hs((size_t)nwg * 64);
        HIPCHK(hipMemcpy(hs.data(), dst, hs.size() * 8, hipMemcpyDeviceToHost));
        const double steps = (double)a.nunits / nwg;
        for (int wvi = 0; wvi < 8; wvi++)
            for (int ph = 0; ph < 8; ph++) {
                double sum = 0;
                for (int g = 0; g < nwg; g++) sum += (double)hs[((size_t)g * 8 + wvi) * 8 + ph];
                stamps_out[wvi * 8 + ph] = (long long)(sum / nwg / steps);
            }
    }
    (void)hipFree(dst);
    (void)hipFree(i0); (void)hipFree(i1); (void)hipFree(F); (void)hipFree(M); (void)hipFree(bias); (void)hipFree(slope); (void)hipFree(w0); (void)hipFree(w1); (void)hipFree(out);
    return rc;
}

// bench-only: ablations of stem0_fused_kernel<1,1> on a wp x hp frame (variant = ABL bits, see stem_fused.h)
int rife_hip_bench_stemf(int gpuid, int wp, int hp, int variant, int iters, float* ms_out) {
    tl_cu_budget = 0;                                                    // bench hooks size their grids for the whole chip, whatever stream this thread used last
    int rc;
    if ((rc = check_device(gpuid))) return rc;
    const size_t P = (size_t)wp * hp;
    uint32_t *i0 = nullptr, *i1 = nullptr; float4* F = nullptr; float *M = nullptr, *out = nullptr, *bias = nullptr; void* wh = nullptr;
    HIPCHK(hipMalloc(&i0, P * 4)); HIPCHK(hipMalloc(&i1, P * 4)); HIPCHK(hipMalloc(&F, P * 16)); HIPCHK(hipMalloc(&M, P * 4));
    HIPCHK(hipMalloc(&out, P / 4 * 32 * 4)); HIPCHK(hipMalloc(&bias, 64 * 4)); HIPCHK(hipMalloc(&wh, 9 * 2 * 32 * 16));
    HIPCHK(hipMemset(i0, 0x40, P * 4)); HIPCHK(hipMemset(i1, 0x60, P * 4)); HIPCHK(hipMemset(F, 0, P * 16)); HIPCHK(hipMemset(M, 0, P * 4));
    HIPCHK(hipMemset(bias, 0, 256)); HIPCHK(hipMemset(wh, 0, 9 * 2 * 32 * 16));
    StemFusedArgs fa;
    fa.img0 = i0; fa.img1 = i1; fa.F = F; fa.M = M; fa.wpk = wh; fa.bias = bias; fa.slope = bias; fa.out = out; fa.timestep = 0.5f; fa.tsp = nullptr;
    fa.wp = wp; fa.hp = hp; fa.Ho = hp / 2; fa.Wo = wp / 2; fa.out_ld = 32; fa.Cout = 32; fa.tiles_x = (fa.Wo + 31) / 32;
    const int nb = fa.tiles_x * ((fa.Ho + 3) / 4);
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto run = [&](auto kfn) -> int {
        for (int i = 0; i < 2; i++) hipLaunchKernelGGL(kfn, dim3(nb), dim3(512), stemf_lds_bytes<1>(), 0, fa);
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kfn, dim3(nb), dim3(512), stemf_lds_bytes<1>(), 0, fa);
        HIPCHK(hipEventRecord(e1, 0));
        HIPCHK(hipEventSynchronize(e1));
        float t = 0; HIPCHK(hipEventElapsedTime(&t, e0, e1));
        *ms_out = t / iters;
        return 0;
    };
    switch (variant) {
        case 0: rc = run(stem0_fused_kernel<1, 1, 0>); break;
        case 1: rc = run(stem0_fused_kernel<1, 1, 1>); break;
        case 16: rc = run(stem0_fused_kernel<1, 1, 16>); break;
        case 32: rc = run(stem0_fused_kernel<1, 1, 32>); break;
        default: rc = fail(RIFE_HIP_EINVAL, "unknown variant");
    }
    (void)hipFree(i0); (void)hipFree(i1); (void)hipFree(F); (void)hipFree(M); (void)hipFree(out); (void)hipFree(bias); (void)hipFree(wh);
    return rc;
}

