// Stream compaction of Gaussian rows: the device side of prune_points / densify_and_clone /
// densify_and_split (2dgs/scene/gaussian_model.py:510-541, 583-626).  The reference edits its six parameter tensors,
// their twelve Adam moments and three statistics with one boolean-mask indexing each (~20 gather launches plus the
// mask -> index conversions); here the mask is scanned ONCE (wave ballot + popcount per 256 rows, one single-block
// scan of the block counts) and any number of row-major [P, w] float tensors are gathered against that scan, up to
// eight per launch, with coalesced reads and writes.  Stable: kept rows stay in index order (the order the
// reference's mask indexing produces).
#include "g4s_internal.h"
#include "g4s_device.h"
#include "../../include/g4s_optim.h"

namespace g4s {

// block_count[b] = number of kept rows among rows [256 b, 256 b + 256)
__global__ void __launch_bounds__(256) compact_count_kernel(int P, const uint8_t* __restrict__ keep,
                                                            uint32_t* __restrict__ block_count) {
    __shared__ uint32_t s_w[4];
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    const bool k = i < P && keep[i] != 0;
    const uint32_t c = (uint32_t)__popcll(__ballot(k));
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

// single block: exclusive scan of the block counts in place; total -> *out_count
__global__ void __launch_bounds__(1024) compact_scan_kernel(int nblocks, uint32_t* __restrict__ block_count,
                                                            int* __restrict__ out_count) {
    __shared__ uint32_t wsum[16];
    const int t = (int)threadIdx.x;
    const int seg = (nblocks + 1023) / 1024;
    const int b = imin_(nblocks, t * seg), e = imin_(nblocks, b + seg);
    uint32_t sum = 0;
    for (int i = b; i < e; i++) sum += block_count[i];
    const uint32_t inc = wave_incl_scan_u32(sum);
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    uint32_t base = 0, all = 0;
    for (int w = 0; w < 16; w++) {
        if (w < (t >> 6)) base += wsum[w];
        all += wsum[w];
    }
    uint32_t run = base + inc - sum;
    for (int i = b; i < e; i++) {
        const uint32_t v = block_count[i];
        block_count[i] = run;
        run += v;
    }
    if (t == 0) *out_count = (int)all;
}

struct GatherArgs {
    int nseg;
    const float* src[8];
    float* dst[8];
    int width[8];
};

// rows [256 b, 256 b + 256): kept rows are listed in LDS in index order (ballot + popcount of the lower lanes +
// the wave's base), then every tensor's kept rows are copied element by element with consecutive threads on
// consecutive floats
__global__ void __launch_bounds__(256) compact_gather_kernel(int P, const uint8_t* __restrict__ keep,
                                                             const uint32_t* __restrict__ block_off, GatherArgs a,
                                                             long long dst_row0) {
    __shared__ uint32_t s_row[256];
    __shared__ uint32_t s_w[4];
    const int t = (int)threadIdx.x, w = t >> 6;
    const int i = (int)(blockIdx.x * 256 + t);
    const bool k = i < P && keep[i] != 0;
    const uint64_t m = __ballot(k);
    if (lane_id() == 0) s_w[w] = (uint32_t)__popcll(m);
    __syncthreads();
    const uint32_t wbase = (w > 0 ? s_w[0] : 0u) + (w > 1 ? s_w[1] : 0u) + (w > 2 ? s_w[2] : 0u);
    const uint32_t nkept = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    if (k) s_row[wbase + (uint32_t)__popcll(m & lanes_below_mask())] = (uint32_t)i;
    __syncthreads();
    const size_t out0 = (size_t)dst_row0 + block_off[blockIdx.x];
    for (int s = 0; s < a.nseg; s++) {
        const int wd = a.width[s];
        const float* __restrict__ src = a.src[s];
        float* __restrict__ dst = a.dst[s] + out0 * (size_t)wd;
        const uint32_t total = nkept * (uint32_t)wd;
        for (uint32_t e = (uint32_t)t; e < total; e += 256) {
            const uint32_t r = e / (uint32_t)wd, c = e - r * (uint32_t)wd;
            dst[e] = src[(size_t)s_row[r] * wd + c];
        }
    }
}

}  // namespace g4s

using namespace g4s;

extern "C" size_t g4s_compact_workspace(int P) { return align_up((size_t)((P > 0 ? P : 0) / 256 + 2) * 4) + 256; }

extern "C" int g4s_compact_scan_launch_internal(int P, const uint8_t* keep, int* out_count, char* workspace, hipStream_t s) {
    uint32_t* block_off = (uint32_t*)align_ptr(workspace);
    const int nblocks = (P + 255) / 256;
    hipLaunchKernelGGL(compact_count_kernel, dim3(nblocks), dim3(256), 0, s, P, keep, block_off);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, s, nblocks, block_off, out_count);
    return 0;
}

extern "C" int g4s_compact_gather_launch_internal(int P, const uint8_t* keep, const char* workspace, int nseg,
                                                  const float* const* src, float* const* dst, const int* widths,
                                                  long long dst_row0, hipStream_t s) {
    const uint32_t* block_off = (const uint32_t*)align_ptr((char*)workspace);
    const int nblocks = (P + 255) / 256;
    for (int s0 = 0; s0 < nseg; s0 += 8) {
        GatherArgs a{};
        a.nseg = nseg - s0 < 8 ? nseg - s0 : 8;
        for (int i = 0; i < a.nseg; i++) { a.src[i] = src[s0 + i]; a.dst[i] = dst[s0 + i]; a.width[i] = widths[s0 + i]; }
        hipLaunchKernelGGL(compact_gather_kernel, dim3(nblocks), dim3(256), 0, s, P, keep, block_off, a, dst_row0);
    }
    return 0;
}
