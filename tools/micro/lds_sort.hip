// Sizing experiment (NOT part of the library): a stable radix sort of up to 256 * ITEMS (key, value) pairs inside ONE
// workgroup's LDS (four 8-bit passes, the ranking scheme of the library's radix_scatter_kernel: ballot match-any per wave
// against wave-private counters) -- the building block a sample sort of the depth keys would launch twice (chunks, then
// buckets) instead of the twelve launches of the four global passes.  Driver: tools/lds_sort_bench.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
__device__ __forceinline__ uint32_t excl_scan_256(uint32_t v, uint32_t* s4) {
    const int t = (int)threadIdx.x, lane = t & 63;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    if (lane == 63) s4[t >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (t >> 6); w++) base += s4[w];
    __syncthreads();
    return base + inc - v;
}

// chunk b = elements [b * TK, min(n, (b+1) * TK)); sorted in place (keys_out / vals_out may alias the inputs)
template <int ITEMS>
__global__ void __launch_bounds__(256) lds_sort_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                       uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, int n) {
    constexpr int TK = 256 * ITEMS;
    __shared__ uint32_t s_k[2][TK], s_v[2][TK];
    __shared__ uint32_t s_cnt[4][256];
    __shared__ uint32_t s4[4];
    const int t = (int)threadIdx.x, w = t >> 6, lane = t & 63;
    const int begin = (int)blockIdx.x * TK, m = min(TK, n - begin);
    if (m <= 0) return;
    for (int i = t; i < TK; i += 256) {
        s_k[0][i] = i < m ? keys_in[begin + i] : 0xFFFFFFFFu;
        s_v[0][i] = i < m ? vals_in[begin + i] : 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    int cur = 0;
    for (int shift = 0; shift < 32; shift += 8) {
#pragma unroll
        for (int j = 0; j < 4; j++) s_cnt[j][t] = 0;
        __syncthreads();
        // wave w ranks the contiguous quarter [w * 64 * ITEMS, (w+1) * 64 * ITEMS) of the block, 64 keys per step
        uint32_t key[ITEMS], val[ITEMS], lrank[ITEMS];
#pragma unroll
        for (int u = 0; u < ITEMS; u++) {
            const int i = w * 64 * ITEMS + 64 * u + lane;
            key[u] = s_k[cur][i];
            val[u] = s_v[cur][i];
            const uint32_t d = (key[u] >> shift) & 255u;
            uint64_t mm = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                mm &= bit ? bal : ~bal;
            }
            const uint32_t rank = (uint32_t)__popcll(mm & below);
            uint32_t old = 0;
            if (rank == 0) old = atomicAdd(&s_cnt[w][d], (uint32_t)__popcll(mm));
            old = (uint32_t)__shfl((int)old, (int)__builtin_ctzll(mm), 64);
            lrank[u] = old + rank;
        }
        __syncthreads();
        {
            const uint32_t c0 = s_cnt[0][t], c1 = s_cnt[1][t], c2 = s_cnt[2][t], c3 = s_cnt[3][t];
            const uint32_t start = excl_scan_256(c0 + c1 + c2 + c3, s4);
            s_cnt[0][t] = start; s_cnt[1][t] = start + c0; s_cnt[2][t] = start + c0 + c1; s_cnt[3][t] = start + c0 + c1 + c2;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < ITEMS; u++) {
            const uint32_t d = (key[u] >> shift) & 255u;
            const uint32_t slot = s_cnt[w][d] + lrank[u];
            s_k[cur ^ 1][slot] = key[u];
            s_v[cur ^ 1][slot] = val[u];
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int i = t; i < m; i += 256) { keys_out[begin + i] = s_k[cur][i]; vals_out[begin + i] = s_v[cur][i]; }
}
}  // namespace

extern "C" void lds_sort(int items, const void* kin, const void* vin, void* kout, void* vout, int n, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (items == 8) hipLaunchKernelGGL(lds_sort_kernel<8>, dim3((n + 2047) / 2048), dim3(256), 0, s, (const uint32_t*)kin, (const uint32_t*)vin, (uint32_t*)kout, (uint32_t*)vout, n);
    else hipLaunchKernelGGL(lds_sort_kernel<16>, dim3((n + 4095) / 4096), dim3(256), 0, s, (const uint32_t*)kin, (const uint32_t*)vin, (uint32_t*)kout, (uint32_t*)vout, n);
}
