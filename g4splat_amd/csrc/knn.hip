// simple-knn replacement: mean squared distance to the 3 nearest other points (distCUDA2).
//
// WHAT is computed is the reference's (knn/simple_knn.cu:131-183, restated by brute force in
// oracle/surfel_oracle.c oracle_knn): for every point the three smallest values of
//     d(j) = (x_j - x)^2 + (y_j - y)^2 + (z_j - z)^2        (separate multiplies and adds, this association)
// over all j with a different INDEX (duplicates give 0, fewer than three other points leave FLT_MAX terms), summed
// smallest first and divided by 3.  The three smallest values are a property of the point set, so every exact search
// returns the same bits.  HOW it is searched here has nothing in common with the reference (one thread per point walking
// every 1 024-point box of a per-axis Morton order):
//
//   * the points are ordered along a Z-curve over a CUBIC lattice (one scale for the three axes: the cells of a
//     6 x 4 x 3 m room are cubes, not bricks, and the bounding boxes of consecutive runs are compact) with the library's
//     own LDS radix sort, and gathered once into a float4 array (x, y, z, index) padded to a multiple of 64;
//   * over that order sits an implicit 64-ary tree of axis-aligned boxes: a LEAF is 64 consecutive points -- what a
//     wave64 fetches with one coalesced 1-KB request --, an inner node has 64 children -- what a wave tests in ONE step,
//     one child per lane and a ballot.  Three levels cover 262 144 points per top node; the top nodes are walked 64 at a
//     time (10 M points: 39 of them);
//   * ONE WAVE ANSWERS THE 64 QUERIES OF A LEAF TOGETHER.  Its lanes are neighbours on the curve, so they want the
//     same candidates: a candidate leaf is fetched once per wave, parked in LDS and read back as broadcast
//     ds_read_b128 -- one LDS instruction per 64 (query, candidate) pairs -- and every lane keeps its own sorted
//     triple with v_min_f32 + 2 x v_med3_f32 (three independent instructions per pair, no compare / select chain);
//   * pruning is wave-level and two-stage: (1) box against box -- the query leaf's own box against a node's, compared
//     with the wave's largest third-best distance -- decides 64 nodes per step; (2) before a surviving leaf is
//     fetched, every lane measures ITS point against the leaf's box and the leaf is skipped unless some lane could
//     still improve.  Both bounds are computed with the distance's own operations in the distance's own order, so by
//     monotonicity of IEEE rounding they never exceed the distance a candidate inside the box would get: pruning on
//     `bound >= third best` is exact (an equal value changes nothing), and a cloud of coincident points costs one leaf
//     instead of the whole scene;
//   * the search starts from the wave's own leaf and its two neighbours on the curve, so the radius is already a few
//     point spacings when the tree walk begins; nothing is reset and nothing is scanned twice.
//   No host synchronisation, no allocation (caller's workspace), 11 launches of which 8 are the sort's.
#include <float.h>

#include "g4s_internal.h"
#include "g4s_device.h"

namespace g4s {

constexpr int KNN_LEAF = 64;  // points per leaf = lanes per wave
constexpr int KNN_FAN = 64;   // children per inner node = lanes per wave

struct KnnLayout {
    size_t keys_a, keys_b, vals_a, vals_b, hist, bin_total, sorted, nodes, partial, extent, bytes;
    int n0, n1, n2, nparts;  // leaves, level-1 nodes, level-2 (top) nodes
};
static KnnLayout knn_layout(size_t P) {
    KnnLayout L{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    L.n0 = (int)((P + KNN_LEAF - 1) / KNN_LEAF);
    L.n1 = (L.n0 + KNN_FAN - 1) / KNN_FAN;
    L.n2 = (L.n1 + KNN_FAN - 1) / KNN_FAN;
    L.nparts = (int)((P + 1023) / 1024);
    L.keys_a = take(P * 4); L.keys_b = take(P * 4); L.vals_a = take(P * 4); L.vals_b = take(P * 4);
    L.hist = take((size_t)256 * (sort_blocks(P, SORT_ITEMS_U32) + 1) * 4);
    L.bin_total = take(256 * 4);
    L.sorted = take((size_t)(L.n0 ? L.n0 : 1) * KNN_LEAF * 16);
    L.nodes = take((size_t)(L.n0 + L.n1 + L.n2 + 1) * 32);
    L.partial = take((size_t)(L.nparts ? L.nparts : 1) * 32);
    L.extent = take(32);
    L.bytes = o + 256;
    return L;
}

// axis-aligned box, 32 bytes (two quads): lo.xyz, hi.xyz
struct Box {
    float lx, ly, lz, hx, hy, hz, pad0, pad1;
};
__device__ __forceinline__ Box box_empty() { return Box{FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX, 0, 0}; }
__device__ __forceinline__ void box_join(Box& a, const Box& b) {
    a.lx = fminf(a.lx, b.lx); a.ly = fminf(a.ly, b.ly); a.lz = fminf(a.lz, b.lz);
    a.hx = fmaxf(a.hx, b.hx); a.hy = fmaxf(a.hy, b.hy); a.hz = fmaxf(a.hz, b.hz);
}
// union over the 64 lanes, valid in every lane
__device__ __forceinline__ Box box_wave_join(Box m) {
    for (int off = 32; off >= 1; off >>= 1) {
        Box o;
        o.lx = __shfl_xor(m.lx, off, 64); o.ly = __shfl_xor(m.ly, off, 64); o.lz = __shfl_xor(m.lz, off, 64);
        o.hx = __shfl_xor(m.hx, off, 64); o.hy = __shfl_xor(m.hy, off, 64); o.hz = __shfl_xor(m.hz, off, 64);
        box_join(m, o);
    }
    return m;
}
// union over a 1024-thread block; result valid in thread 0
__device__ __forceinline__ Box box_block_join(Box m, Box* sm16) {
    m = box_wave_join(m);
    const int w = (int)(threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) sm16[w] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i < 16; i++) box_join(m, sm16[i]);
    return m;
}

// ---- extent of the cloud (stays on the device) ------------------------------------------------------------------
__global__ void __launch_bounds__(1024) knn_extent_partial_kernel(int P, const float* __restrict__ pts, Box* __restrict__ partial) {
    __shared__ Box sm[16];
    const int i = (int)(blockIdx.x * 1024 + threadIdx.x);
    Box m = box_empty();
    if (i < P) {
        const float x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
        m = Box{x, y, z, x, y, z, 0, 0};
    }
    m = box_block_join(m, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = m;
}
__global__ void __launch_bounds__(1024) knn_extent_final_kernel(int nparts, const Box* __restrict__ partial, Box* __restrict__ extent) {
    __shared__ Box sm[16];
    Box m = box_empty();
    for (int i = (int)threadIdx.x; i < nparts; i += 1024) box_join(m, partial[i]);
    m = box_block_join(m, sm);
    if (threadIdx.x == 0) *extent = m;
}

// ---- position on the Z-curve ---------------------------------------------------------------------------------------
// bit i of a 10-bit value -> bit 3 i.  Each step doubles the gaps by adding a shifted copy (a multiplication by
// 2^k + 1: the copies never overlap, so the sum is an OR) and masking.
__device__ __forceinline__ uint32_t spread_by_3(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
// Lattice of 1024^3 CUBIC cells anchored at the cloud's lower corner, edge = longest extent / 1023.  (The order only
// has to be spatially coherent -- the result does not depend on it --, so NaN / inf coordinates simply land in cell 0.)
__global__ void __launch_bounds__(256) knn_curve_kernel(int P, const float* __restrict__ pts, const Box* __restrict__ extent,
                                                        uint32_t* __restrict__ codes, uint32_t* __restrict__ idx) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= P) return;
    const Box b = *extent;
    const float edge = fmaxf(fmaxf(b.hx - b.lx, b.hy - b.ly), b.hz - b.lz);
    const float cells = edge > 0.0f ? 1023.0f / edge : 0.0f;
    auto cell = [&](float v, float lo) {
        const float c = fminf(fmaxf((v - lo) * cells, 0.0f), 1023.0f);  // NaN -> 0
        return (uint32_t)c;
    };
    const uint32_t cx = cell(pts[3 * (size_t)i], b.lx), cy = cell(pts[3 * (size_t)i + 1], b.ly), cz = cell(pts[3 * (size_t)i + 2], b.lz);
    codes[i] = spread_by_3(cx) | (spread_by_3(cy) << 1) | (spread_by_3(cz) << 2);
    idx[i] = (uint32_t)i;
}

// ---- the tree ------------------------------------------------------------------------------------------------------
// A wave per leaf: gathers its 64 points in curve order (padding slots: NaN coordinates, which no comparison ever
// accepts) and stores the leaf's box.
__global__ void __launch_bounds__(256) knn_leaves_kernel(int P, int n0, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                         float4* __restrict__ sorted, Box* __restrict__ leaves) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);  // a wave per leaf, four per workgroup
    if ((i >> 6) >= n0) return;
    Box m = box_empty();
    float4 rec = make_float4(__uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u), __uint_as_float(0x7FC00000u),
                             __uint_as_float(0xFFFFFFFFu));
    if (i < P) {
        const uint32_t src = order[i];
        const float x = pts[3 * (size_t)src], y = pts[3 * (size_t)src + 1], z = pts[3 * (size_t)src + 2];
        rec = make_float4(x, y, z, __uint_as_float(src));
        m = Box{x, y, z, x, y, z, 0, 0};
    }
    sorted[i] = rec;
    m = box_wave_join(m);
    if ((threadIdx.x & 63) == 0) leaves[i >> 6] = m;
}
// A wave per parent: the union of its (up to) 64 children.
__global__ void __launch_bounds__(256) knn_parents_kernel(int n_child, const Box* __restrict__ child, int n_parent, Box* __restrict__ parent) {
    const int w = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= n_parent) return;
    const int c = w * KNN_FAN + (int)(threadIdx.x & 63);
    Box m = c < n_child ? child[c] : box_empty();
    m = box_wave_join(m);
    if ((threadIdx.x & 63) == 0) parent[w] = m;
}

// ---- the search ----------------------------------------------------------------------------------------------------
// Squared gap between two boxes / between a point and a box, with the operations of the distance itself in its own
// order (gap per axis by subtraction, three squares, (x^2 + y^2) + z^2): a lower bound, in floating point, of the
// distance computed for any candidate inside the box (see the header).  An empty box gives +inf (or NaN: not < r2).
__device__ __forceinline__ float gap2_box_box(const Box& a, const Box& q) {
    const float gx = fmaxf(fmaxf(a.lx - q.hx, q.lx - a.hx), 0.0f);
    const float gy = fmaxf(fmaxf(a.ly - q.hy, q.ly - a.hy), 0.0f);
    const float gz = fmaxf(fmaxf(a.lz - q.hz, q.lz - a.hz), 0.0f);
    return gx * gx + gy * gy + gz * gz;
}
__device__ __forceinline__ float gap2_box_point(float lx, float ly, float lz, float hx, float hy, float hz, float x, float y, float z) {
    const float gx = fmaxf(fmaxf(lx - x, x - hx), 0.0f);
    const float gy = fmaxf(fmaxf(ly - y, y - hy), 0.0f);
    const float gz = fmaxf(fmaxf(lz - z, z - hz), 0.0f);
    return gx * gx + gy * gy + gz * gz;
}
// sorted triple t0 <= t1 <= t2 of the smallest distances seen; insertion = min and two medians of the OLD values
struct Best3 {
    float t0, t1, t2;
};
__device__ __forceinline__ void best3_insert(Best3& b, float d) {
    const float n0 = fminf(b.t0, d);
    const float n1 = __builtin_amdgcn_fmed3f(b.t0, b.t1, d);
    const float n2 = __builtin_amdgcn_fmed3f(b.t1, b.t2, d);
    b.t0 = n0; b.t1 = n1; b.t2 = n2;
}
// All 64 candidates of a staged leaf against this lane's query.  OWN: the leaf is the wave's own, candidate k is lane
// k's point and is excluded for that lane (by index, like the reference: a duplicate elsewhere counts with distance 0).
// A candidate whose distance is NaN or inf (padding slots, non-finite input) is clamped to FLT_MAX, which inserts as a
// no-op -- exactly what the reference's `best[j] > dist` does with it.
template <bool OWN>
__device__ __forceinline__ void scan_leaf(const float4* __restrict__ cand, float qx, float qy, float qz, int lane, Best3& b) {
#pragma unroll 8
    for (int k = 0; k < KNN_LEAF; k++) {
        const float4 c = cand[k];  // wave-uniform address: broadcast read
        const float dx = c.x - qx, dy = c.y - qy, dz = c.z - qz;
        float d = fminf(dx * dx + dy * dy + dz * dz, FLT_MAX);
        if (OWN) d = k == lane ? FLT_MAX : d;
        best3_insert(b, d);
    }
}
// lane j's value of v as a wave-uniform (scalar) value; j is uniform
__device__ __forceinline__ float lane_value(float v, int j) {
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), j));
}
__device__ __forceinline__ float wave_max_nonneg(float v) {  // v >= 0: the bit patterns order like the values
    return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_u32(__float_as_uint(v))));
}

__global__ void __launch_bounds__(256) knn_search_kernel(int P, const float4* __restrict__ sorted, const Box* __restrict__ leaves,
                                                         const Box* __restrict__ mids, const Box* __restrict__ tops, int n0, int n1, int n2,
                                                         float* __restrict__ dists) {
    __shared__ float4 s_cand[4][KNN_LEAF];
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int own = (int)(blockIdx.x * 4) + wv;  // this wave's leaf
    if (own >= n0) return;
    float4* stage = s_cand[wv];
    const float4 me = sorted[(size_t)own * KNN_LEAF + lane];
    const bool valid = own * KNN_LEAF + lane < P;
    Best3 b{FLT_MAX, FLT_MAX, FLT_MAX};
    // LDS traffic of one wave is ordered by the hardware; the fences only keep the compiler from moving the broadcast
    // reads above the store that feeds them (or the next store above the last reads)
    auto fetch = [&](int leaf) {
        const float4 c = sorted[(size_t)leaf * KNN_LEAF + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        stage[lane] = c;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // the start: own leaf, then its neighbours on the curve
    fetch(own);
    scan_leaf<true>(stage, me.x, me.y, me.z, lane, b);
    const int near_lo = own > 0 ? own - 1 : own, near_hi = own + 1 < n0 ? own + 1 : own;
    if (near_lo != own) { fetch(near_lo); scan_leaf<false>(stage, me.x, me.y, me.z, lane, b); }
    if (near_hi != own) { fetch(near_hi); scan_leaf<false>(stage, me.x, me.y, me.z, lane, b); }
    float r2 = wave_max_nonneg(valid ? b.t2 : 0.0f);  // wave-uniform: nothing at or beyond it can matter to any lane

    const Box q = leaves[own];  // the 64 queries' own box (uniform)
    for (int c2 = 0; c2 < n2; c2 += KNN_FAN) {
        const float g2 = c2 + lane < n2 ? gap2_box_box(tops[c2 + lane], q) : FLT_MAX;
        uint64_t m2 = __ballot(g2 < r2);
        while (m2) {
            const int j2 = (int)__builtin_ctzll(m2);
            m2 &= m2 - 1;
            if (!(lane_value(g2, j2) < r2)) continue;  // r2 has shrunk since
            const int i1 = (c2 + j2) * KNN_FAN + lane;
            const float g1 = i1 < n1 ? gap2_box_box(mids[i1], q) : FLT_MAX;
            uint64_t m1 = __ballot(g1 < r2);
            while (m1) {
                const int j1 = (int)__builtin_ctzll(m1);
                m1 &= m1 - 1;
                if (!(lane_value(g1, j1) < r2)) continue;
                const int base0 = ((c2 + j2) * KNN_FAN + j1) * KNN_FAN;
                const int i0 = base0 + lane;
                Box lf = box_empty();
                if (i0 < n0) lf = leaves[i0];
                const bool fresh = i0 < n0 && (i0 < near_lo || i0 > near_hi);  // not one of the three leaves of the start
                uint64_t m0 = __ballot(fresh && gap2_box_box(lf, q) < r2);
                while (m0) {
                    const int j0 = (int)__builtin_ctzll(m0);
                    m0 &= m0 - 1;
                    // stage 2: could ANY lane still improve on its third best inside this leaf's box?
                    const float lx = lane_value(lf.lx, j0), ly = lane_value(lf.ly, j0), lz = lane_value(lf.lz, j0);
                    const float hx = lane_value(lf.hx, j0), hy = lane_value(lf.hy, j0), hz = lane_value(lf.hz, j0);
                    const float gp = gap2_box_point(lx, ly, lz, hx, hy, hz, me.x, me.y, me.z);
                    if (__ballot(valid && gp < b.t2) == 0ull) continue;
                    fetch(base0 + j0);
                    scan_leaf<false>(stage, me.x, me.y, me.z, lane, b);
                    r2 = wave_max_nonneg(valid ? b.t2 : 0.0f);
                }
            }
        }
    }
    if (valid) dists[__float_as_uint(me.w)] = (b.t0 + b.t1 + b.t2) / 3.0f;
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
    Box* partial = (Box*)(w + L.partial);
    Box* extent = (Box*)(w + L.extent);
    Box* leaves = (Box*)(w + L.nodes);
    Box* mids = leaves + L.n0;
    Box* tops = mids + L.n1;
    float4* sorted = (float4*)(w + L.sorted);
    hipLaunchKernelGGL(knn_extent_partial_kernel, dim3(L.nparts), dim3(1024), 0, s, P, points, partial);
    hipLaunchKernelGGL(knn_extent_final_kernel, dim3(1), dim3(1024), 0, s, L.nparts, partial, extent);
    hipLaunchKernelGGL(knn_curve_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, extent, keys_a, vals_a);
    const int cur = radix_sort_u32_pairs(keys_a, keys_b, vals_a, vals_b, P, (uint32_t*)(w + L.hist),
                                         (uint32_t*)(w + L.bin_total), s);
    const uint32_t* order = cur ? vals_b : vals_a;
    const int leaf_blocks = (L.n0 + 3) / 4;  // four leaves (waves) per workgroup
    hipLaunchKernelGGL(knn_leaves_kernel, dim3(leaf_blocks), dim3(256), 0, s, P, L.n0, points, order, sorted, leaves);
    hipLaunchKernelGGL(knn_parents_kernel, dim3((L.n1 + 3) / 4), dim3(256), 0, s, L.n0, leaves, L.n1, mids);
    hipLaunchKernelGGL(knn_parents_kernel, dim3((L.n2 + 3) / 4), dim3(256), 0, s, L.n1, mids, L.n2, tops);
    hipLaunchKernelGGL(knn_search_kernel, dim3(leaf_blocks), dim3(256), 0, s, P, sorted, leaves, mids, tops, L.n0, L.n1, L.n2,
                       meanDists);
    return 0;
}
