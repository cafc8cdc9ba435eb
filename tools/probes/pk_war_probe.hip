// GPU box probe: is a VALU write to a VGPR safe right behind a packed-fp32 instruction that READS that VGPR (write-after-read)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/pk_war_probe.hip -o /tmp/pk_war_probe && /tmp/pk_war_probe
// The SLP-vectorized build of the fused stem kernels contained   v_pk_mul_f32 v[34:35], v[18:19], 0.5   directly followed by   v_mov_b32 v18, v23
// and was not run-to-run stable in quarter-wave groups of lanes.  This probe runs exactly that pair, many times, on a
// full chip (4 waves per SIMD, with and without memory traffic next to it) and counts lanes whose product used the NEW value of v18.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int GAP>
__global__ __launch_bounds__(512) void k_probe(const float* __restrict__ in, unsigned long long* bad, unsigned long long* badq, int iters, const float4* __restrict__ traffic, float* sink) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = gridDim.x * blockDim.x;
    float a = in[tid], b = in[tid + n], c = in[tid + 2 * n];
    unsigned long long nb = 0;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        float r0, r1;
        if (traffic) { const float4 t = traffic[(size_t)((tid * 7 + it * 131) & 0xfffff)]; acc += t.x; }
        if (GAP == 0)
            asm volatile("v_mov_b32 v18, %2\n v_mov_b32 v19, %3\n s_nop 4\n"
                         "v_pk_mul_f32 v[34:35], v[18:19], 0.5 op_sel_hi:[1,0]\n"
                         "v_mov_b32 v18, %4\n"
                         "s_nop 7\n v_mov_b32 %0, v34\n v_mov_b32 %1, v35\n"
                         : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "v"(c) : "v18", "v19", "v34", "v35");
        else
            asm volatile("v_mov_b32 v18, %2\n v_mov_b32 v19, %3\n s_nop 4\n"
                         "v_pk_mul_f32 v[34:35], v[18:19], 0.5 op_sel_hi:[1,0]\n"
                         "s_nop 1\n"
                         "v_mov_b32 v18, %4\n"
                         "s_nop 7\n v_mov_b32 %0, v34\n v_mov_b32 %1, v35\n"
                         : "=v"(r0), "=v"(r1) : "v"(a), "v"(b), "v"(c) : "v18", "v19", "v34", "v35");
        const bool w0 = r0 != a * 0.5f, w1 = r1 != b * 0.5f;
        nb += w0 + w1;
        if (w0 || w1) atomicAdd(badq + ((threadIdx.x & 63) >> 4), 1ull);
        a += 1.0f; c += 3.0f; b += 0.5f;
    }
    if (nb) atomicAdd(bad, nb);
    if (acc == 123.456f) sink[0] = acc;
}
int main() {
    const int nwg = 2048, nt = 512, n = nwg * nt;
    float* h = new float[(size_t)3 * n];
    for (int i = 0; i < 3 * n; i++) h[i] = (float)((i * 2654435761u) >> 20) * 0.25f + 1.0f;
    float* in; unsigned long long *bad, *badq; float4* traffic; float* sink;
    hipMalloc(&in, (size_t)3 * n * 4); hipMalloc(&bad, 8); hipMalloc(&badq, 32); hipMalloc(&traffic, (size_t)(1 << 20) * 16); hipMalloc(&sink, 4);
    hipMemcpy(in, h, (size_t)3 * n * 4, hipMemcpyHostToDevice); hipMemset(traffic, 0, (size_t)(1 << 20) * 16);
    for (int variant = 0; variant < 4; variant++) {
        hipMemset(bad, 0, 8); hipMemset(badq, 0, 32);
        const float4* tr = (variant & 1) ? traffic : nullptr;
        for (int rep = 0; rep < 5; rep++) {
            if (variant & 2) hipLaunchKernelGGL(k_probe<1>, dim3(nwg), dim3(nt), 0, 0, in, bad, badq, 2000, tr, sink);
            else hipLaunchKernelGGL(k_probe<0>, dim3(nwg), dim3(nt), 0, 0, in, bad, badq, 2000, tr, sink);
        }
        hipDeviceSynchronize();
        unsigned long long nb = 0, q[4];
        hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(q, badq, 32, hipMemcpyDeviceToHost);
        printf("%s, %s: %llu wrong products of %llu; by quarter wave (lanes 0-15, 16-31, 32-47, 48-63): %llu %llu %llu %llu\n",
               (variant & 2) ? "s_nop 1 between" : "back to back", (variant & 1) ? "with memory traffic" : "VALU only", nb, 5ull * n * 2000 * 2, q[0], q[1], q[2], q[3]);
    }
    return 0;
}
