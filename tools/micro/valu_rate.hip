// Issue rate of VALU instruction classes on gfx950: each wave runs a long chain-free loop of one instruction kind, eight
// independent streams per lane; prints wave-instructions per SIMD cycle (1/4 = full rate for a wave64).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float a[8];
    double b[4];
    for (int i = 0; i < 8; i++) a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; i++) b[i] = 1.0 + threadIdx.x * 1e-3 + i;
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(a[1]) : "vcc");  // a defined vcc
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            if (KIND == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 3) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(a[i]));
            if (KIND == 5) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(a[(i + 1) & 7]) : "s20", "s21");
            if (KIND == 12) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (KIND == 13) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (KIND == 14) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(b[i & 3]));
            if (KIND == 15) asm volatile("v_rcp_f64 %0, %0" : "+v"(b[i & 3]));
            if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
            if (KIND == 9 && (i & 1) == 0) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(a[i + 1]));
            if (KIND == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(a[(i + 1) & 7]) : "vcc");
            if (KIND == 11) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += (float)b[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, float* d) {
    const int iters = 4096, blocks = 256 * 4;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 8 * 4;       // 4 waves per SIMD
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %.3f ms  %.2f cycles per wave instruction (at 2.4 GHz)\n", name, ms, cycles / insts_per_simd);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 256 * 4);
    run<0>("v_fma_f32", d);
    run<1>("v_rcp_f32", d);
    run<2>("v_exp_f32", d);
    run<5>("v_sqrt_f32", d);
    run<3>("v_add_f32_dpp row_ror", d);
    run<4>("v_cndmask_b32 (vcc)", d);
    run<6>("v_cndmask_b32 (sgpr pair)", d);
    run<12>("v_cndmask_b32_e64 (vcc)", d);
    run<13>("v_cndmask_b32_e32 (vcc), 2 srcs", d);
    run<14>("v_fma_f64", d);
    run<15>("v_rcp_f64", d);
    run<7>("v_mov_b32", d);
    run<8>("v_mul_f32", d);
    run<9>("v_permlane32_swap (x4, +s_nop)", d);
    run<10>("v_cmp_lt_f32 -> vcc", d);
    run<11>("v_add_f32_dpp quad_perm", d);
    return 0;
}
