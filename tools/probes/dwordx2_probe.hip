// GPU box probe: do 8-byte global loads at 4-byte (not 8-byte) aligned addresses always return the right two dwords?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dwordx2_probe.hip -o /tmp/dwordx2_probe && /tmp/dwordx2_probe
// Buffer B[i] = hash(i); every lane loads uint2 at a pseudo-random dword index near its pixel (like the paired warp taps of
// warp_rgbx) next to 16-byte loads of a second buffer, many rounds; counts the dwords that differ from hash(idx), hash(idx + 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__host__ __device__ inline uint32_t hsh(uint32_t i) { i *= 2654435761u; i ^= i >> 15; i *= 2246822519u; i ^= i >> 13; return i; }
__global__ void k_probe(const uint32_t* __restrict__ B, const float4* __restrict__ F, int w, int h, int rounds, unsigned long long* bad, float* sink) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    unsigned long long nb = 0;
    float acc = 0.f;
    for (int r = 0; r < rounds; r++) {
        const float4 f = F[(size_t)((y + r) % h) * w + x];
        acc += f.x + f.w;
        uint32_t s = hsh((uint32_t)(y * w + x) * 31u + (uint32_t)r);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            s = hsh(s + k);
            int xx = x + (int)(s & 15) - 8, yy = y + (int)((s >> 4) & 15) - 8;
            xx = min(max(xx, 0), w - 2); yy = min(max(yy, 0), h - 1);
            const int idx = yy * w + xx;
            uint2 v;
            __builtin_memcpy(&v, __builtin_assume_aligned(B + idx, 4), 8);
            nb += (v.x != hsh((uint32_t)idx)) + (v.y != hsh((uint32_t)idx + 1u));
        }
    }
    if (nb) atomicAdd(bad, nb);
    if (acc == 123.456f) sink[0] = acc;
}
int main() {
    const int w = 3840, h = 2176;
    uint32_t* hb = new uint32_t[(size_t)w * h];
    for (size_t i = 0; i < (size_t)w * h; i++) hb[i] = hsh((uint32_t)i);
    uint32_t* B; float4* F; unsigned long long* bad; float* sink;
    hipMalloc(&B, (size_t)w * h * 4); hipMalloc(&F, (size_t)w * h * 16); hipMalloc(&bad, 8); hipMalloc(&sink, 4);
    hipMemcpy(B, hb, (size_t)w * h * 4, hipMemcpyHostToDevice); hipMemset(F, 0, (size_t)w * h * 16); hipMemset(bad, 0, 8);
    for (int rep = 0; rep < 20; rep++) hipLaunchKernelGGL(k_probe, dim3((w + 255) / 256, h), dim3(256), 0, 0, B, F, w, h, 4, bad, sink);
    hipDeviceSynchronize();
    unsigned long long nb = 0; hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
    printf("dwordx2 at dword alignment: %llu wrong dwords of %llu loaded\n", nb, 20ull * w * h * 4 * 8 * 2);
    return nb != 0;
}
