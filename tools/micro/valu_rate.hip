// Issue rate of VALU instruction classes on gfx950: each wave runs a long chain-free loop of one instruction kind, eight
// independent streams per lane; prints wave-instructions per SIMD cycle (1/4 = full rate for a wave64).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* cyc) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    __shared__ float lds[1024];
    float a[8];
    double b[4];
    f2 c[4];
    f4 q[4];
    for (int i = 0; i < 4; i++) { c[i].x = 1.0f + threadIdx.x * 1e-3f + i; c[i].y = 2.0f + i; q[i] = f4{0, 0, 0, 0}; }
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = (float)i;
    __syncthreads();
    const uint32_t lds_addr = (uint32_t)(size_t)lds + (iters & 1) * 64;  // wave-uniform address
    for (int i = 0; i < 8; i++) a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 4; i++) b[i] = 1.0 + threadIdx.x * 1e-3 + i;
    asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(a[1]) : "vcc");  // a defined vcc
    const unsigned long long t0 = __builtin_readcyclecounter();  // s_memtime: shader-clock ticks
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
            // packed FP32 (two floats per lane per instruction; an instruction counts once): four independent register pairs
            if (KIND == 16 && i < 4) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(c[i]));
            if (KIND == 17 && i < 4) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(c[i]));
            if (KIND == 18 && i < 4) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(c[i]));
            // low half of src0 broadcast to both results, src2 negated: the shape of k = px * Tw - Tu for two list entries
            if (KIND == 19 && i < 4) asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(c[i]) : "v"(c[(i + 1) & 3]), "v"(c[(i + 2) & 3]));
            if (KIND == 20 && i < 4) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[1,0]" : "+v"(c[i]) : "v"(c[(i + 1) & 3]));
            // the blend loops' mix: six packed fma per transcendental
            if (KIND == 21) { if (i < 6) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(c[i & 3])); else if (i == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(a[0])); else asm volatile("v_rcp_f32 %0, %0" : "+v"(a[1])); }
            if (KIND == 22) { if (i < 6) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i])); else if (i == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(a[6])); else asm volatile("v_rcp_f32 %0, %0" : "+v"(a[7])); }
            // wave-uniform (broadcast) LDS reads, alone and beside VALU work: what feeds the blend loops their list entries
            if (KIND == 23 && i < 4) asm volatile("ds_read_b128 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "=v"(q[i]) : "v"(lds_addr + 16 * i) : "memory");
            if (KIND == 24) { if (i == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(q[0]) : "v"(lds_addr) : "memory"); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i])); if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            if (KIND == 25) { if (i < 2) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(lds_addr + 16 * i) : "memory"); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i])); if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            if (KIND == 26) { if (i < 4) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(lds_addr + 16 * i) : "memory"); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i])); if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
            if (KIND == 27) { if (i < 4) asm volatile("ds_read_b64 %0, %1" : "=v"(c[i]) : "v"(lds_addr + 8 * i) : "memory"); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i])); if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    for (int i = 0; i < 4; i++) s += (float)b[i] + c[i].x + c[i].y + q[i].x + q[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static unsigned long long* d_cyc;
// Does a VALU instruction get cheaper when whole 16-lane rows of the wave are masked off?  (If it did, clustering the
// active pixels of the blend kernels into few rows would pay.)  MASK: bit r = row r (lanes 16 r .. 16 r + 15) executes.
template <int MASK>
__global__ void __launch_bounds__(256) k_rows(float* out, int iters, unsigned long long* cyc) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    const int row = (threadIdx.x & 63) >> 4;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if ((MASK >> row) & 1) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (cyc != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *cyc = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MASK>
void run_rows(const char* name, float* d) {
    const int iters = 4096, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rows<MASK>, dim3(blocks), dim3(256), 0, 0, d, 16, (unsigned long long*)nullptr);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rows<MASK>, dim3(blocks), dim3(256), 0, 0, d, iters, (unsigned long long*)nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %.3f ms  %.2f cyc/instr at a nominal 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * 4));
}
template <int KIND>
void run(const char* name, float* d, int per_iter = 8) {
    const int iters = 4096, blocks = 256 * 4;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16, d_cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * per_iter * 4;       // 4 waves per SIMD
    const double cycles = ms * 1e-3 * 2.4e9;
    unsigned long long ticks = 0;
    hipMemcpy(&ticks, d_cyc, 8, hipMemcpyDeviceToHost);
    // the kernel's own shader-clock count (one wave of block 0) separates the issue cost from the clock the part ran at
    printf("%-44s %.3f ms  %.2f cyc/instr at a nominal 2.4 GHz | %.2f shader-clock ticks/instr, ticks/us %.0f\n", name, ms,
           cycles / insts_per_simd, (double)ticks / insts_per_simd, (double)ticks / (ms * 1e3));
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 256 * 4);
    hipMalloc(&d_cyc, 8);
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
    run<16>("v_pk_fma_f32", d, 4);
    run<17>("v_pk_mul_f32", d, 4);
    run<18>("v_pk_add_f32", d, 4);
    run<19>("v_pk_fma_f32 op_sel+neg", d, 4);
    run<20>("v_pk_mov_b32", d, 4);
    run<21>("6 v_pk_fma + exp + rcp (per instr)", d, 8);
    run<22>("6 v_fma + exp + rcp (per instr)", d, 8);
    run<23>("ds_read_b128 uniform, waited", d, 4);
    run<24>("8 v_fma + 1 ds_read_b128 (per group of 8)", d, 1);
    run<25>("8 v_fma + 2 ds_read_b128 (per group of 8)", d, 1);
    run<26>("8 v_fma + 4 ds_read_b128 (per group of 8)", d, 1);
    run<27>("8 v_fma + 4 ds_read_b64 (per group of 8)", d, 1);
    run_rows<0xF>("v_fma_f32, EXEC = all four 16-lane rows", d);
    run_rows<0x3>("v_fma_f32, EXEC = rows 0-1 (lanes 0..31)", d);
    run_rows<0x1>("v_fma_f32, EXEC = row 0 (lanes 0..15)", d);
    run_rows<0x5>("v_fma_f32, EXEC = rows 0 and 2", d);
    return 0;
}
