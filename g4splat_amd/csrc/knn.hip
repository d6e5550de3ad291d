// simple-knn replacement: mean squared distance to the 3 nearest other points (distCUDA2).
//
// Result definition: knn/simple_knn.cu:131-183 (exact 3-NN, self excluded by index, mean of the
// three best squared distances, FLT_MAX terms if fewer than three neighbours).  The search
// strategy is free as long as it is exact; this one keeps the reference's outline (Morton
// order -> boxes of 1024 consecutive points -> prune boxes by point/box distance) but
//   * never synchronises with the host (the bounding box stays on the device),
//   * gathers the Morton-ordered points once into a contiguous float4 array so that the 64
//     lanes of a wave (Morton neighbours that visit the same boxes) read the candidate
//     points as broadcast loads,
//   * sorts with the library's own wave-private radix sort (binning.hip).
#include <float.h>

#include "g4s_internal.h"
#include "g4s_device.h"

namespace g4s {

constexpr int BOX = 1024;

struct KnnLayout {
    size_t keys_a, keys_b, vals_a, vals_b, hist, bin_total, sorted, boxes, partial, bbox, bytes;
    int nboxes, nparts;
};
static KnnLayout knn_layout(size_t P) {
    KnnLayout L{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    L.nboxes = (int)((P + BOX - 1) / BOX);
    L.nparts = (int)((P + 1023) / 1024);
    L.keys_a = take(P * 4); L.keys_b = take(P * 4); L.vals_a = take(P * 4); L.vals_b = take(P * 4);
    L.hist = take((size_t)256 * (sort_blocks(P, SORT_ITEMS_U32) + 1) * 4);
    L.bin_total = take(256 * 4);
    L.sorted = take(P * 16);
    L.boxes = take((size_t)(L.nboxes ? L.nboxes : 1) * 32);
    L.partial = take((size_t)(L.nparts ? L.nparts : 1) * 32);
    L.bbox = take(32);
    L.bytes = o + 256;
    return L;
}

struct MinMax {
    float mnx, mny, mnz, mxx, mxy, mxz, pad0, pad1;
};

__device__ __forceinline__ void mm_merge(MinMax& a, const MinMax& b) {
    a.mnx = fminf(a.mnx, b.mnx); a.mny = fminf(a.mny, b.mny); a.mnz = fminf(a.mnz, b.mnz);
    a.mxx = fmaxf(a.mxx, b.mxx); a.mxy = fmaxf(a.mxy, b.mxy); a.mxz = fmaxf(a.mxz, b.mxz);
}
__device__ __forceinline__ MinMax mm_wave_reduce(MinMax m) {
    for (int off = 32; off >= 1; off >>= 1) {
        MinMax o;
        o.mnx = __shfl_xor(m.mnx, off, 64); o.mny = __shfl_xor(m.mny, off, 64); o.mnz = __shfl_xor(m.mnz, off, 64);
        o.mxx = __shfl_xor(m.mxx, off, 64); o.mxy = __shfl_xor(m.mxy, off, 64); o.mxz = __shfl_xor(m.mxz, off, 64);
        mm_merge(m, o);
    }
    return m;
}
// Block-wide (1024 threads) min/max of up to 1024 float3 items; result valid in thread 0.
__device__ __forceinline__ MinMax mm_block_reduce(MinMax m, MinMax* sm16) {
    m = mm_wave_reduce(m);
    const int w = (int)(threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) sm16[w] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i < 16; i++) mm_merge(m, sm16[i]);
    return m;
}
__device__ __forceinline__ MinMax mm_empty() {
    return MinMax{FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, 0, 0};
}

// knn/simple_knn.cu:193-200: min / max with init {0,0,0} (the box always contains the origin)
__global__ void __launch_bounds__(1024) knn_bbox_partial_kernel(int P, const float* __restrict__ pts,
                                                                MinMax* __restrict__ partial) {
    __shared__ MinMax sm[16];
    const int i = (int)(blockIdx.x * 1024 + threadIdx.x);
    MinMax m = mm_empty();
    if (i < P) {
        const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
        m = MinMax{x, y, z, x, y, z, 0, 0};
    }
    m = mm_block_reduce(m, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = m;
}
__global__ void __launch_bounds__(1024) knn_bbox_final_kernel(int nparts, const MinMax* __restrict__ partial,
                                                              MinMax* __restrict__ bbox) {
    __shared__ MinMax sm[16];
    MinMax m = mm_empty();
    for (int i = (int)threadIdx.x; i < nparts; i += 1024) mm_merge(m, partial[i]);
    m = mm_block_reduce(m, sm);
    if (threadIdx.x == 0) {
        const MinMax zero{0, 0, 0, 0, 0, 0, 0, 0};
        mm_merge(m, zero);
        *bbox = m;
    }
}

// knn/simple_knn.cu:45-70
__device__ __forceinline__ uint32_t prep_morton(uint32_t x) {
    x = (x | (x << 16)) & 0x030000FF;
    x = (x | (x << 8)) & 0x0300F00F;
    x = (x | (x << 4)) & 0x030C30C3;
    x = (x | (x << 2)) & 0x09249249;
    return x;
}
__global__ void __launch_bounds__(256) knn_morton_kernel(int P, const float* __restrict__ pts,
                                                         const MinMax* __restrict__ bbox, uint32_t* __restrict__ codes,
                                                         uint32_t* __restrict__ idx) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= P) return;
    const MinMax b = *bbox;
    const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    const uint32_t mx = prep_morton((uint32_t)(((x - b.mnx) / (b.mxx - b.mnx)) * ((1 << 10) - 1)));
    const uint32_t my = prep_morton((uint32_t)(((y - b.mny) / (b.mxy - b.mny)) * ((1 << 10) - 1)));
    const uint32_t mz = prep_morton((uint32_t)(((z - b.mnz) / (b.mxz - b.mnz)) * ((1 << 10) - 1)));
    codes[i] = mx | (my << 1) | (mz << 2);
    idx[i] = (uint32_t)i;
}

// Gather the Morton-ordered points and build the per-box AABBs (knn/simple_knn.cu:78-117).
__global__ void __launch_bounds__(1024) knn_gather_boxes_kernel(int P, const float* __restrict__ pts,
                                                                const uint32_t* __restrict__ order,
                                                                float4* __restrict__ sorted, MinMax* __restrict__ boxes) {
    __shared__ MinMax sm[16];
    const int i = (int)(blockIdx.x * BOX + threadIdx.x);
    MinMax m = mm_empty();
    if (i < P) {
        const uint32_t src = order[i];
        const float x = pts[3 * (size_t)src], y = pts[3 * (size_t)src + 1], z = pts[3 * (size_t)src + 2];
        sorted[i] = make_float4(x, y, z, __uint_as_float(src));
        m = MinMax{x, y, z, x, y, z, 0, 0};
    }
    m = mm_block_reduce(m, sm);
    if (threadIdx.x == 0) boxes[blockIdx.x] = m;
}

// knn/simple_knn.cu:119-129
__device__ __forceinline__ float dist_box_point(const MinMax& box, float px, float py, float pz) {
    float dx = 0, dy = 0, dz = 0;
    if (px < box.mnx || px > box.mxx) dx = fminf(fabsf(px - box.mnx), fabsf(px - box.mxx));
    if (py < box.mny || py > box.mxy) dy = fminf(fabsf(py - box.mny), fabsf(py - box.mxy));
    if (pz < box.mnz || pz > box.mxz) dz = fminf(fabsf(pz - box.mnz), fabsf(pz - box.mxz));
    return dx * dx + dy * dy + dz * dz;
}
// knn/simple_knn.cu:131-145 (updateKBest<3>)
__device__ __forceinline__ void update3(float rx, float ry, float rz, const float4 p, float* best) {
    const float dx = p.x - rx, dy = p.y - ry, dz = p.z - rz;
    float dist = dx * dx + dy * dy + dz * dz;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        if (best[j] > dist) {
            const float t = best[j];
            best[j] = dist;
            dist = t;
        }
    }
}

// knn/simple_knn.cu:147-183
__global__ void __launch_bounds__(256) knn_mean_dist_kernel(int P, const float4* __restrict__ sorted,
                                                            const MinMax* __restrict__ boxes, int nboxes,
                                                            float* __restrict__ dists) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= P) return;
    const float4 me = sorted[i];
    float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    for (int k = imax_(0, i - 3); k <= imin_(P - 1, i + 3); k++) {
        if (k == i) continue;
        update3(me.x, me.y, me.z, sorted[k], best);
    }
    const float reject = best[2];
    best[0] = FLT_MAX; best[1] = FLT_MAX; best[2] = FLT_MAX;
    for (int b = 0; b < nboxes; b++) {
        const MinMax box = boxes[b];
        const float d = dist_box_point(box, me.x, me.y, me.z);
        if (d > reject || d > best[2]) continue;
        const int e = imin_(P, (b + 1) * BOX);
        for (int k = b * BOX; k < e; k++) {
            if (k == i) continue;
            update3(me.x, me.y, me.z, sorted[k], best);
        }
    }
    dists[__float_as_uint(me.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace g4s

using namespace g4s;

extern "C" size_t g4s_knn_workspace(int P) { return knn_layout(P > 0 ? (size_t)P : 0).bytes; }

extern "C" int g4s_knn_launch_internal(int P, const float* points, float* meanDists, char* workspace,
                                       hipStream_t s) {
    const KnnLayout L = knn_layout((size_t)P);
    char* w = align_ptr(workspace);
    uint32_t* keys_a = (uint32_t*)(w + L.keys_a);
    uint32_t* keys_b = (uint32_t*)(w + L.keys_b);
    uint32_t* vals_a = (uint32_t*)(w + L.vals_a);
    uint32_t* vals_b = (uint32_t*)(w + L.vals_b);
    MinMax* partial = (MinMax*)(w + L.partial);
    MinMax* bbox = (MinMax*)(w + L.bbox);
    MinMax* boxes = (MinMax*)(w + L.boxes);
    float4* sorted = (float4*)(w + L.sorted);
    hipLaunchKernelGGL(knn_bbox_partial_kernel, dim3(L.nparts), dim3(1024), 0, s, P, points, partial);
    hipLaunchKernelGGL(knn_bbox_final_kernel, dim3(1), dim3(1024), 0, s, L.nparts, partial, bbox);
    hipLaunchKernelGGL(knn_morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, bbox, keys_a, vals_a);
    const int cur = radix_sort_u32_pairs(keys_a, keys_b, vals_a, vals_b, P, (uint32_t*)(w + L.hist),
                                         (uint32_t*)(w + L.bin_total), s);
    const uint32_t* order = cur ? vals_b : vals_a;
    hipLaunchKernelGGL(knn_gather_boxes_kernel, dim3(L.nboxes), dim3(1024), 0, s, P, points, order, sorted, boxes);
    hipLaunchKernelGGL(knn_mean_dist_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, sorted, boxes, L.nboxes,
                       meanDists);
    return 0;
}
