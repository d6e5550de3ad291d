// Sizing experiment (NOT part of the library): the binning stage as "count per tile -> scan -> scatter with a cursor per tile
// -> sort every tile's list in LDS" instead of "depth sort of the Gaussians -> emit -> stable partition by tile".
// Driver: tools/tile_bins_bench.py (times the four kernels on a real frame and checks the lists against the library's).
//
//   rect[i]  = (x0 | y0 << 16, width) of the binned tile rect of Gaussian i, cnt[i] = its number of tiles (0: emits nothing)
//   key[i]   = depth bits
// Lists: entries[offset[tile] + j] = key << 32 | index, sorted ascending (depth bits, then index) = the reference's order.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int BIG = 64;  // rects with more tiles than this are expanded by a whole wave

template <bool SCATTER>
__global__ void __launch_bounds__(256) bins_kernel(int P, const uint2* __restrict__ rect, const uint32_t* __restrict__ cnt,
                                                   const uint32_t* __restrict__ key, int tiles_x, uint32_t* __restrict__ count,
                                                   const uint32_t* __restrict__ offset, uint64_t* __restrict__ entries) {
    __shared__ uint32_t s_big[256];
    __shared__ uint32_t s_nbig;
    const int t = (int)threadIdx.x, i = (int)(blockIdx.x * 256 + t);
    if (t == 0) s_nbig = 0;
    __syncthreads();
    const uint32_t c = i < P ? cnt[i] : 0u;
    if (c > (uint32_t)BIG) s_big[atomicAdd(&s_nbig, 1u)] = (uint32_t)i;
    else if (c > 0) {
        const uint2 r = rect[i];
        const uint32_t x0 = r.x & 0xFFFFu, y0 = r.x >> 16, w = r.y;
        const uint64_t e = ((uint64_t)key[i] << 32) | (uint32_t)i;
        uint32_t tx = 0, ty = 0;
        for (uint32_t k = 0; k < c; k++) {
            const uint32_t tile = (y0 + ty) * (uint32_t)tiles_x + x0 + tx;
            const uint32_t pos = atomicAdd(&count[tile], 1u);
            if (SCATTER) entries[offset[tile] + pos] = e;
            if (++tx == w) { tx = 0; ty++; }
        }
    }
    __syncthreads();
    const int nb = (int)s_nbig, wave = t >> 6, lane = t & 63;
    for (int b = wave; b < nb; b += 4) {
        const uint32_t g = s_big[b];
        const uint2 r = rect[g];
        const uint32_t x0 = r.x & 0xFFFFu, y0 = r.x >> 16, w = r.y, n = cnt[g];
        const uint64_t e = ((uint64_t)key[g] << 32) | g;
        for (uint32_t k = (uint32_t)lane; k < n; k += 64) {
            const uint32_t ty = k / w, tx = k - ty * w;
            const uint32_t tile = (y0 + ty) * (uint32_t)tiles_x + x0 + tx;
            const uint32_t pos = atomicAdd(&count[tile], 1u);
            if (SCATTER) entries[offset[tile] + pos] = e;
        }
    }
}

// one block: exclusive scan of the tile counts -> offsets (+ total), counts cleared for the scatter's cursors
__global__ void __launch_bounds__(1024) scan_kernel(int tiles, uint32_t* __restrict__ count, uint32_t* __restrict__ offset,
                                                    uint32_t* __restrict__ lengths) {
    __shared__ uint32_t s_w[16];
    const int t = (int)threadIdx.x;
    const int seg = (tiles + 1023) / 1024;
    const int b = min(tiles, t * seg), e = min(tiles, b + seg);
    uint32_t sum = 0;
    for (int i = b; i < e; i++) sum += count[i];
    uint32_t inc = sum;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d, 64); if ((t & 63) >= d) inc += o; }
    if ((t & 63) == 63) s_w[t >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (t >> 6); w++) base += s_w[w];
    uint32_t run = base + inc - sum;
    for (int i = b; i < e; i++) { const uint32_t c = count[i]; offset[i] = run; lengths[i] = c; count[i] = 0; run += c; }
    if (t == 1023) offset[tiles] = run;
}

// one block per tile: the list sorted ascending as 64-bit (depth bits, index) keys; bitonic network in LDS up to 4096 entries
constexpr int SORT_CAP = 4096;
__global__ void __launch_bounds__(256) sort_kernel(const uint32_t* __restrict__ offset, uint64_t* __restrict__ entries) {
    __shared__ uint64_t s[SORT_CAP];
    const uint32_t o = offset[blockIdx.x], n = offset[blockIdx.x + 1] - o;
    if (n < 2) return;
    uint32_t m = 2;
    while (m < n) m <<= 1;
    const int t = (int)threadIdx.x;
    uint64_t* a = entries + o;
    if (m <= (uint32_t)SORT_CAP) {
        for (uint32_t i = t; i < m; i += 256) s[i] = i < n ? a[i] : ~0ull;
        __syncthreads();
        for (uint32_t k = 2; k <= m; k <<= 1)
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t i = t; i < m; i += 256) {
                    const uint32_t p = i ^ j;
                    if (p > i) {
                        const uint64_t x = s[i], y = s[p];
                        const bool up = (i & k) == 0;
                        if ((x > y) == up) { s[i] = y; s[p] = x; }
                    }
                }
                __syncthreads();
            }
        for (uint32_t i = t; i < n; i += 256) a[i] = s[i];
    }
    // (lists deeper than SORT_CAP are left unsorted here: the driver reports how many there are -- a product version sorts
    // them through global memory, or hands the frame to the partition pipeline)
}
}  // namespace

extern "C" void tile_bins_count(int P, const void* rect, const void* cnt, const void* key, int tiles_x, void* count, void* stream) {
    hipLaunchKernelGGL(bins_kernel<false>, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, (const uint2*)rect,
                       (const uint32_t*)cnt, (const uint32_t*)key, tiles_x, (uint32_t*)count, (const uint32_t*)nullptr, (uint64_t*)nullptr);
}
extern "C" void tile_bins_scan(int tiles, void* count, void* offset, void* lengths, void* stream) {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, tiles, (uint32_t*)count, (uint32_t*)offset, (uint32_t*)lengths);
}
extern "C" void tile_bins_scatter(int P, const void* rect, const void* cnt, const void* key, int tiles_x, void* cursor,
                                  const void* offset, void* entries, void* stream) {
    hipLaunchKernelGGL(bins_kernel<true>, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, (const uint2*)rect,
                       (const uint32_t*)cnt, (const uint32_t*)key, tiles_x, (uint32_t*)cursor, (const uint32_t*)offset, (uint64_t*)entries);
}
extern "C" void tile_bins_sort(int tiles, const void* offset, void* entries, void* stream) {
    hipLaunchKernelGGL(sort_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)offset, (uint64_t*)entries);
}
