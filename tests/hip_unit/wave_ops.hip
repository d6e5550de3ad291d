// Standalone unit check of the wave64 primitives in g4s_device.h (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tests/hip_unit/wave_ops.hip -o /tmp/wave_ops && /tmp/wave_ops
#include <cstdio>
#include <vector>
#include "../../g4splat_amd/csrc/g4s_device.h"
using namespace g4s;

__global__ void k_sum4(const float* in, float* out) {
    const int l = threadIdx.x;
    out[l] = wave_sum4_to_rows(in[l], in[64 + l], in[128 + l], in[192 + l]);
}
__global__ void k_sum16(const float* in, float* out) {
    const int l = threadIdx.x;
    float v[16], r[4];
    for (int t = 0; t < 16; t++) v[t] = in[t * 64 + l];
    wave_sum16_to_rows(v, r);
    for (int i = 0; i < 4; i++) out[i * 64 + l] = r[i];
}
__global__ void k_sum16q(const float* in, float* out) {
    const int l = threadIdx.x;
    float v[16];
    for (int t = 0; t < 16; t++) v[t] = in[t * 64 + l];
    out[l] = wave_sum16_to_quads(v, (l & 8) != 0, (l & 4) != 0);
}
__global__ void k_sum63(const float* in, float* out) { out[threadIdx.x] = wave_sum_to_lane63(in[threadIdx.x]); }
__global__ void k_swap32(const float* in, float* out) { out[threadIdx.x] = swap32_sum(in[threadIdx.x], in[64 + threadIdx.x]); }
__global__ void k_swap16(const float* in, float* out) { out[threadIdx.x] = swap16_sum(in[threadIdx.x], in[64 + threadIdx.x]); }

int main() {
    std::vector<float> h(256);
    for (int v = 0; v < 4; v++) for (int l = 0; l < 64; l++) h[v * 64 + l] = (float)((v + 1) * 1000 + l);  // exact in f32
    float *din, *dout;
    hipMalloc(&din, 1024); hipMalloc(&dout, 256);
    hipMemcpy(din, h.data(), 1024, hipMemcpyHostToDevice);
    std::vector<float> o(64);
    int bad = 0;
    double want[4];
    for (int v = 0; v < 4; v++) { want[v] = 0; for (int l = 0; l < 64; l++) want[v] += h[v * 64 + l]; }
    hipLaunchKernelGGL(k_sum4, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o.data(), dout, 256, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) if (o[l] != (float)want[l >> 4]) { bad++; if (bad < 8) printf("sum4 lane %d got %f want %f\n", l, o[l], want[l >> 4]); }
    hipLaunchKernelGGL(k_sum63, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o.data(), dout, 256, hipMemcpyDeviceToHost);
    if (o[63] != (float)want[0]) { bad++; printf("sum63 got %f want %f\n", o[63], want[0]); }
    hipLaunchKernelGGL(k_swap32, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o.data(), dout, 256, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) { float w = l < 32 ? h[l] + h[l + 32] : h[64 + l - 32] + h[64 + l]; if (o[l] != w) { bad++; if (bad < 16) printf("swap32 lane %d got %f want %f\n", l, o[l], w); } }
    hipLaunchKernelGGL(k_swap16, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o.data(), dout, 256, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++) { int r = l >> 4, c = l & 15; float w = (r == 0) ? h[c] + h[16 + c] : (r == 1) ? h[64 + c] + h[64 + 16 + c] : (r == 2) ? h[32 + c] + h[48 + c] : h[64 + 32 + c] + h[64 + 48 + c]; if (o[l] != w) { bad++; if (bad < 24) printf("swap16 lane %d got %f want %f\n", l, o[l], w); } }
    {   // sixteen terms: out[i] of a lane in row k must be the sum of term 4k+i
        std::vector<float> h16(1024), o16(256);
        double w16[16];
        for (int t = 0; t < 16; t++) { w16[t] = 0; for (int l = 0; l < 64; l++) { h16[t * 64 + l] = (float)((t + 1) * 512 + ((l * 7 + t) % 64)); w16[t] += h16[t * 64 + l]; } }
        float *d16, *o16d;
        hipMalloc(&d16, 4096); hipMalloc(&o16d, 1024);
        hipMemcpy(d16, h16.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_sum16, dim3(1), dim3(64), 0, 0, d16, o16d);
        hipMemcpy(o16.data(), o16d, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 4; i++) for (int l = 0; l < 64; l++) {
            const float w = (float)w16[4 * (l >> 4) + i];
            if (o16[i * 64 + l] != w) { bad++; if (bad < 32) printf("sum16 out[%d] lane %d got %f want %f\n", i, l, o16[i * 64 + l], w); }
        }
        // one value per lane: quad q of row k holds term 4k + quad_term
        hipLaunchKernelGGL(k_sum16q, dim3(1), dim3(64), 0, 0, d16, o16d);
        hipMemcpy(o16.data(), o16d, 256, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) {
            const float w = (float)w16[4 * (l >> 4) + quad_term(l)];
            if (o16[l] != w) { bad++; if (bad < 40) printf("sum16q lane %d got %f want %f\n", l, o16[l], w); }
        }
    }
    printf(bad ? "FAILED (%d)\n" : "wave_ops OK\n", bad);
    return bad != 0;
}
