// Does the cost of a VALU instruction depend on WHICH lanes EXEC enables?  (If a wave64 instruction is issued as passes
// over lane groups and a pass whose lanes are all masked off is skipped, the blend kernels' visits whose blending pixels
// fit one half of the wave are cheaper than the others, and the lane <-> pixel mapping is a lever.)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/exec_rows.hip -o /tmp/exec_rows && /tmp/exec_rows
// Every variant runs the same loop of independent v_fma_f32 (8 streams per lane, 4 waves per SIMD) under a different
// lane mask; the variants are interleaved and repeated so that clock / power transients show up as spread, not as signal.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long mask, int kind) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    const int lane = threadIdx.x & 63;
    if ((mask >> lane) & 1ull) {
        if (kind == 0) {
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
            }
        } else if (kind == 1) {  // transcendental mix of the blend loops: 6 fma + exp + rcp
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int i = 0; i < 6; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[6]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(a[7]));
            }
        } else {  // one dependent chain (latency, not issue)
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[0]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 4 * 256 * 4);
    struct V { const char* name; unsigned long long mask; };
    const V vs[] = {
        {"all 64 lanes", ~0ull},
        {"lanes 0..31 (low half)", 0x00000000FFFFFFFFull},
        {"lanes 32..63 (high half)", 0xFFFFFFFF00000000ull},
        {"lanes 0..15 (row 0)", 0x000000000000FFFFull},
        {"lanes 16..31 (row 1)", 0x00000000FFFF0000ull},
        {"lanes 48..63 (row 3)", 0xFFFF000000000000ull},
        {"rows 0 and 2", 0x0000FFFF0000FFFFull},
        {"rows 0 and 1 and 2", 0x0000FFFFFFFFFFFFull},
        {"even lanes (32 lanes, both halves)", 0x5555555555555555ull},
        {"one lane per row (4 lanes)", 0x0001000100010001ull},
        {"lane 0 only", 1ull},
        {"lane 63 only", 1ull << 63},
        // how many lanes does it take?  contiguous from lane 0
        {"lanes 0..1", 0x3ull}, {"lanes 0..7", 0xFFull}, {"lanes 0..16 (17)", 0x1FFFFull}, {"lanes 0..19 (20)", 0xFFFFFull},
        {"lanes 0..23 (24)", 0xFFFFFFull}, {"lanes 0..27 (28)", 0xFFFFFFFull}, {"lanes 0..30 (31)", 0x7FFFFFFFull},
        {"lanes 0..32 (33)", 0x1FFFFFFFFull},
        // scattered
        {"every 4th lane (16 lanes, all rows)", 0x1111111111111111ull},
        {"every 3rd lane (22 lanes)", 0x9249249249249249ull},
        {"8 lanes in each half (lanes 0..7, 32..39)", 0x000000FF000000FFull},
        {"16 lanes: 0..7 and 56..63", 0xFF000000000000FFull},
        {"17 lanes: row 0 + lane 63", 0x800000000000FFFFull},
        {"row 0 + row 3 (32 lanes)", 0xFFFF00000000FFFFull},
    };
    const int nv = sizeof(vs) / sizeof(vs[0]);
    const int iters = 32768, blocks = 256 * 4, reps = 5;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kind = 0; kind < 3; kind++) {
        std::vector<std::vector<float>> ms(nv);
        // warm the clocks
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, ~0ull, kind);
        for (int r = 0; r < reps; r++)
            for (int v = 0; v < nv; v++) {
                const int vv = (r & 1) ? nv - 1 - v : v;  // alternate the order
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, vs[vv].mask, kind);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float t;
                hipEventElapsedTime(&t, e0, e1);
                ms[vv].push_back(t);
            }
        printf("%s\n", kind == 0 ? "== 8 independent v_fma_f32 per iteration" : kind == 1 ? "== 6 v_fma_f32 + v_exp_f32 + v_rcp_f32 per iteration"
                                                                                          : "== 8 DEPENDENT v_fma_f32 per iteration (one chain)");
        for (int v = 0; v < nv; v++) {
            std::sort(ms[v].begin(), ms[v].end());
            const double med = ms[v][reps / 2];
            printf("  %-38s median %.3f ms (min %.3f max %.3f)  %.2f cycles/instr at a nominal 2.4 GHz, 4 waves per SIMD\n", vs[v].name, med,
                   ms[v].front(), ms[v].back(), med * 1e-3 * 2.4e9 / ((double)iters * 8 * 4));
        }
    }
    return 0;
}
