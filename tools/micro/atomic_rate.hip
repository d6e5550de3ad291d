// Micro-benchmark: throughput of scattered device-scope integer atomics on MI355X (binning design study).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/atomic_rate.hip -o var/atomic_rate && var/atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k_noret(uint32_t* c, int n, int naddr) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)i * 2654435761u;
    atomicAdd(&c[(h >> 8) % naddr], 1u);
}
__global__ void k_ret(uint32_t* c, uint32_t* out, int n, int naddr) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)i * 2654435761u;
    out[i] = atomicAdd(&c[(h >> 8) % naddr], 1u);
}
// tiles of neighbouring threads correlate (a Gaussian's instances hit adjacent tiles)
__global__ void k_ret_local(uint32_t* c, uint32_t* out, int n, int naddr) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t g = (uint32_t)(i / 10) * 2654435761u;
    out[i] = atomicAdd(&c[((g >> 8) + (i % 10)) % naddr], 1u);
}
int main() {
    const int n = 4400000;
    uint32_t *c, *out;
    hipMalloc(&c, 1 << 20); hipMalloc(&out, (size_t)n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int naddr : {1, 64, 7500, 65536, 1000000 / 4}) {
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                hipMemset(c, 0, 1 << 20);
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_noret, dim3((n + 255) / 256), dim3(256), 0, 0, c, n, naddr);
                if (mode == 1) hipLaunchKernelGGL(k_ret, dim3((n + 255) / 256), dim3(256), 0, 0, c, out, n, naddr);
                if (mode == 2) hipLaunchKernelGGL(k_ret_local, dim3((n + 255) / 256), dim3(256), 0, 0, c, out, n, naddr);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("addresses %7d  %s  %.3f ms  -> %.1f atomics/us\n", naddr, mode == 0 ? "no-return " : mode == 1 ? "returning " : "ret, local", best, n / (best * 1e3));
        }
    }
    return 0;
}
