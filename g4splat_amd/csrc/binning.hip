// Binning: depth sort of Gaussians, instance emission, stable tile partition, tile ranges.
//
// Reference semantics (dsr/cuda_rasterizer/rasterizer_impl.cu:70-138,278-319): every visible
// Gaussian emits one instance per tile of its rect; instances are ordered by
// (tile, depth bits) with a STABLE sort, so equal depths inside a tile resolve by ascending
// Gaussian index; ranges[tile] = [start,end) of the tile's run.
//
// MI355X design (not the reference's 64-bit-key sort over all R instances):
//   1. sort the P Gaussians once by depth bits (32-bit keys, stable => ties by index),
//   2. emit instances in that order with a load-balanced expansion (coalesced 8-byte stores),
//      packed as  tile<<48 | k<<32 | gaussian   (k = instance number inside the Gaussian),
//   3. stable counting/radix partition on the tile bits only (1-2 passes instead of 6).
// A stable partition of a depth-ordered sequence yields exactly the reference's per-tile order.
// The radix passes are wave-private: one wave64 owns a contiguous chunk, ranks its 64 keys per
// step with ballot match-any and keeps its 256 running offsets in LDS -- no block barriers.
#include "g4s_internal.h"
#include "g4s_device.h"

namespace g4s {

// ---------------------------------------------------------------------------------------
// radix sort building blocks (8-bit digits)

template <typename K>
__global__ void __launch_bounds__(64) radix_hist_kernel(const K* __restrict__ keys, int n, int shift,
                                                        uint32_t* __restrict__ hist, int nchunks, int chunk_len,
                                                        const uint32_t* __restrict__ d_n) {
    __shared__ uint32_t h[256];
    const int chunk = (int)blockIdx.x;
    if (d_n != nullptr) {  // the key count only lives on the device: the grid is sized for the largest possible n
        n = (int)*d_n;
        chunk_len = sort_chunk((size_t)n);
        nchunks = sort_nchunks((size_t)n);
        if (chunk >= nchunks) return;
    }
    const int lane = (int)threadIdx.x;
    for (int i = lane; i < 256; i += 64) h[i] = 0;
    __syncthreads();
    const int begin = chunk * chunk_len;
    const int end = imin_(n, begin + chunk_len);
    for (int i = begin + lane; i < end; i += 64) {
        const uint32_t d = (uint32_t)(keys[i] >> shift) & 0xFFu;
        atomicAdd(&h[d], 1u);
    }
    __syncthreads();
    for (int i = lane; i < 256; i += 64) hist[(size_t)i * nchunks + chunk] = h[i];
}

// One block per digit row: exclusive scan of the row in place, row total to bin_total[row].
__global__ void __launch_bounds__(256) radix_scan_kernel(uint32_t* __restrict__ hist, int nchunks,
                                                         uint32_t* __restrict__ bin_total,
                                                         const uint32_t* __restrict__ d_n) {
    __shared__ uint32_t sm4[4];
    if (d_n != nullptr) nchunks = sort_nchunks((size_t)*d_n);
    uint32_t* row = hist + (size_t)blockIdx.x * nchunks;
    const int t = (int)threadIdx.x;
    const int seg = (nchunks + 255) / 256;
    const int b = imin_(nchunks, t * seg), e = imin_(nchunks, b + seg);
    uint32_t sum = 0;
    for (int i = b; i < e; i++) sum += row[i];
    uint32_t total;
    uint32_t run = block256_excl_scan_u32(sum, sm4, &total);
    for (int i = b; i < e; i++) {
        const uint32_t v = row[i];
        row[i] = run;
        run += v;
    }
    if (t == 0) bin_total[blockIdx.x] = total;
}

// One wave per chunk.  Stable: keys are processed in index order, 64 per step; inside a step
// lanes with equal digits are ranked by lane.
template <typename K, bool HAS_VAL>
__global__ void __launch_bounds__(64) radix_scatter_kernel(const K* __restrict__ keys_in, K* __restrict__ keys_out,
                                                           const uint32_t* __restrict__ vals_in,
                                                           uint32_t* __restrict__ vals_out, int n, int shift,
                                                           const uint32_t* __restrict__ hist,
                                                           const uint32_t* __restrict__ bin_total, int nchunks,
                                                           int chunk_len, const uint32_t* __restrict__ d_n) {
    __shared__ uint32_t offs[256];
    const int chunk = (int)blockIdx.x;
    if (d_n != nullptr) {
        n = (int)*d_n;
        chunk_len = sort_chunk((size_t)n);
        nchunks = sort_nchunks((size_t)n);
        if (chunk >= nchunks) return;
    }
    const int lane = (int)threadIdx.x;
    // global base of every digit = exclusive scan of bin totals + this chunk's row prefix
    {
        uint32_t t[4];
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            t[j] = bin_total[lane * 4 + j];
            s += t[j];
        }
        uint32_t ex = wave_incl_scan_u32(s) - s;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int d = lane * 4 + j;
            offs[d] = ex + hist[(size_t)d * nchunks + chunk];
            ex += t[j];
        }
    }
    __syncthreads();
    const int begin = chunk * chunk_len;
    const int end = imin_(n, begin + chunk_len);
    const uint64_t below = lanes_below_mask();
    for (int base = begin; base < end; base += 64) {
        const int i = base + lane;
        const bool valid = i < end;
        K key = 0;
        uint32_t val = 0;
        if (valid) {
            key = keys_in[i];
            if (HAS_VAL) val = vals_in[i];
        }
        const uint32_t d = (uint32_t)(key >> shift) & 0xFFu;
        // match-any over the 8 digit bits
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(m & below);
        const uint32_t cnt = (uint32_t)__popcll(m);
        uint32_t dst = 0;
        if (valid) dst = offs[d] + rank;
        __syncthreads();  // single-wave block: orders the LDS read above against the update below
        if (valid && rank == cnt - 1) offs[d] += cnt;
        __syncthreads();
        if (valid) {
            keys_out[dst] = key;
            if (HAS_VAL) vals_out[dst] = val;
        }
    }
}

// d_n == nullptr: n keys (host-known).  Otherwise n is read from *d_n by the kernels and only bounds the grid
// (n_max >= *d_n): the launches can be queued before the host knows the count.
template <typename K, bool HAS_VAL>
static void radix_pass(const K* kin, K* kout, const uint32_t* vin, uint32_t* vout, int n, int shift, uint32_t* hist,
                       uint32_t* bin_total, hipStream_t s, const uint32_t* d_n = nullptr) {
    const int chunk_len = sort_chunk((size_t)n), nchunks = d_n ? sort_nchunks_max((size_t)n) : sort_nchunks((size_t)n);
    hipLaunchKernelGGL((radix_hist_kernel<K>), dim3(nchunks), dim3(64), 0, s, kin, n, shift, hist, nchunks, chunk_len, d_n);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(256), dim3(256), 0, s, hist, nchunks, bin_total, d_n);
    hipLaunchKernelGGL((radix_scatter_kernel<K, HAS_VAL>), dim3(nchunks), dim3(64), 0, s, kin, kout, vin, vout, n,
                       shift, hist, bin_total, nchunks, chunk_len, d_n);
}

int radix_sort_u32_pairs(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int n,
                         uint32_t* hist, uint32_t* bin_total, hipStream_t s, const uint32_t* d_n) {
    if (n <= 0) return 0;
    int cur = 0;
    for (int shift = 0; shift < 32; shift += 8) {
        if (cur == 0)
            radix_pass<uint32_t, true>(keys_a, keys_b, vals_a, vals_b, n, shift, hist, bin_total, s, d_n);
        else
            radix_pass<uint32_t, true>(keys_b, keys_a, vals_b, vals_a, n, shift, hist, bin_total, s, d_n);
        cur ^= 1;
    }
    return cur;
}

int radix_sort_u64_keys(uint64_t* a, uint64_t* b, int n, int begin_bit, int end_bit, uint32_t* hist,
                        uint32_t* bin_total, hipStream_t s) {
    if (n <= 0) return 0;
    int cur = 0;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        if (cur == 0)
            radix_pass<uint64_t, false>(a, b, nullptr, nullptr, n, shift, hist, bin_total, s);
        else
            radix_pass<uint64_t, false>(b, a, nullptr, nullptr, n, shift, hist, bin_total, s);
        cur ^= 1;
    }
    return cur;
}

// ---------------------------------------------------------------------------------------
// instance counting (in depth order), emission, tile ranges

// tiles_touched gathered in depth order, 256 per block -> block sums.
__global__ void __launch_bounds__(256) count_block_sums_kernel(int P, const uint32_t* __restrict__ gidx,
                                                               const uint32_t* __restrict__ tiles_touched,
                                                               uint32_t* __restrict__ block_sums,
                                                               const uint32_t* __restrict__ d_n) {
    __shared__ uint32_t sm4[4];
    if (d_n != nullptr) {  // device-side count: blocks past the end leave
        P = (int)*d_n;
        if ((int)(blockIdx.x * 256) >= P) return;
    }
    const int r = (int)(blockIdx.x * 256 + threadIdx.x);
    uint32_t c = 0;
    if (r < P) c = tiles_touched[gidx[r]];
    uint32_t total;
    block256_excl_scan_u32(c, sm4, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Single block: exclusive scan of the block sums, grand total to *total.
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(int nblocks, const uint32_t* __restrict__ block_sums,
                                                               uint32_t* __restrict__ block_offs,
                                                               const uint32_t* __restrict__ ref_block_sums,
                                                               uint32_t* __restrict__ total,
                                                               const uint32_t* __restrict__ d_n) {
    __shared__ uint32_t wsum[16], rsum[16];
    const int t = (int)threadIdx.x;
    if (d_n != nullptr) nblocks = (int)((*d_n + 255u) / 256u);  // blocks of a device-side element count
    const int seg = (nblocks + 1023) / 1024;
    const int b = imin_(nblocks, t * seg), e = imin_(nblocks, b + seg);
    uint32_t sum = 0, ref = 0;
    for (int i = b; i < e; i++) {
        sum += block_sums[i];
        ref += ref_block_sums[i];
    }
    const uint32_t inc = wave_incl_scan_u32(sum);
    const uint32_t rinc = wave_incl_scan_u32(ref);
    if ((t & 63) == 63) {
        wsum[t >> 6] = inc;
        rsum[t >> 6] = rinc;
    }
    __syncthreads();
    uint32_t base = 0, all = 0, rall = 0;
    for (int w = 0; w < 16; w++) {
        if (w < (t >> 6)) base += wsum[w];
        all += wsum[w];
        rall += rsum[w];
    }
    uint32_t run = base + inc - sum;
    for (int i = b; i < e; i++) {
        block_offs[i] = run;
        run += block_sums[i];
    }
    if (t == 0) {
        total[0] = all;
        total[1] = rall;
    }
}

void launch_count_scan(int P, const uint32_t* gidx_sorted, const uint32_t* tiles_touched, uint32_t* block_sums,
                       uint32_t* block_offs, const uint32_t* ref_block_sums, uint32_t* total, int nblocks,
                       hipStream_t s, const uint32_t* d_n) {
    // d_n != nullptr: P / nblocks only bound the grid, the kernels read the count from the device
    hipLaunchKernelGGL(count_block_sums_kernel, dim3(nblocks), dim3(256), 0, s, P, gidx_sorted, tiles_touched,
                       block_sums, d_n);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, nblocks, block_sums, block_offs,
                       ref_block_sums, total, d_n);
}

// Load-balanced expansion: a block owns 256 consecutive depth ranks; its output range is
// contiguous, every thread produces output slots (not Gaussians), so stores are coalesced
// and the work per thread is even no matter how skewed the per-Gaussian tile counts are.
__global__ void __launch_bounds__(256) emit_kernel(int P, int tiles_x, int tiles_y,
                                                   const uint32_t* __restrict__ gidx,
                                                   const uint32_t* __restrict__ tiles_touched,
                                                   const uint32_t* __restrict__ block_offs,
                                                   const int* __restrict__ radii, float* __restrict__ rec,
                                                   uint64_t* __restrict__ entries) {
    __shared__ uint32_t sm4[4];
    __shared__ uint32_t s_off[256];   // exclusive local offsets
    __shared__ uint32_t s_idx[256];
    __shared__ uint32_t s_rect[256];  // x0 | y0<<12 | width<<24 is too narrow for big grids: use two words
    __shared__ uint32_t s_rect2[256];
    const int t = (int)threadIdx.x;
    const int r = (int)(blockIdx.x * 256 + t);
    uint32_t cnt = 0, idx = 0;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (r < P) {
        idx = gidx[r];
        cnt = tiles_touched[idx];
        if (cnt > 0) {
            const float4* rq = reinterpret_cast<const float4*>(rec) + (size_t)idx * REC_QUADS;
            const float4 q0 = rq[0], q5 = rq[5];
            int rx0, ry0, rx1, ry1;
            get_rect(q0.x, q0.y, radii[idx], tiles_x, tiles_y, rx0, ry0, rx1, ry1);
            tight_tile_rect(q5, rx0, ry0, rx1, ry1, x0, y0, x1, y1);  // same arithmetic as the preprocess
        }
    }
    uint32_t block_total;
    const uint32_t local = block256_excl_scan_u32(cnt, sm4, &block_total);
    const uint32_t base = block_offs[blockIdx.x];
    s_off[t] = local;
    s_idx[t] = idx;
    s_rect[t] = (uint32_t)x0 | ((uint32_t)y0 << 16);
    s_rect2[t] = (uint32_t)(x1 - x0);
    __syncthreads();
    for (uint32_t o = (uint32_t)t; o < block_total; o += 256) {
        // largest j with s_off[j] <= o  (zero-count ranks share an offset with their successor,
        // the search lands on the last of them, i.e. the one that owns slot o)
        int lo = 0, hi = 255;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_off[mid] <= o) lo = mid; else hi = mid - 1;
        }
        const uint32_t k = o - s_off[lo];
        const uint32_t w = s_rect2[lo];
        const uint32_t rx0 = s_rect[lo] & 0xFFFFu, ry0 = s_rect[lo] >> 16;
        const uint32_t ty = k / w, tx = k - ty * w;
        const uint64_t tile = (uint64_t)((ry0 + ty) * (uint32_t)tiles_x + rx0 + tx);
        entries[(size_t)base + o] =
            (tile << ENTRY_TILE_SHIFT) | ((uint64_t)k << ENTRY_K_SHIFT) | (uint64_t)s_idx[lo];
    }
}

__global__ void __launch_bounds__(256) grad_slots_kernel(int P, const uint32_t* __restrict__ tiles_touched,
                                                         const uint32_t* __restrict__ idx_block_offs,
                                                         float* __restrict__ rec) {
    __shared__ uint32_t sm4[4];
    const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
    const uint32_t c = idx < P ? tiles_touched[idx] : 0u;
    uint32_t total;
    const uint32_t local = block256_excl_scan_u32(c, sm4, &total);
    if (c > 0) rec[(size_t)idx * REC_FLOATS + 2] = __uint_as_float(idx_block_offs[blockIdx.x] + local);
}
void launch_grad_slots(int P, const uint32_t* tiles_touched, const uint32_t* idx_block_offs, float* rec, int nblocks,
                       hipStream_t s) {
    hipLaunchKernelGGL(grad_slots_kernel, dim3(nblocks), dim3(256), 0, s, P, tiles_touched, idx_block_offs, rec);
}

void launch_scan_totals(const uint32_t* idx_block_sums, uint32_t* idx_block_offs, const uint32_t* ref_block_sums,
                        const uint32_t* vis_block_sums, uint32_t* vis_block_offs, uint32_t* total, int nblocks,
                        hipStream_t s) {
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, nblocks, idx_block_sums, idx_block_offs,
                       ref_block_sums, total, (const uint32_t*)nullptr);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, nblocks, vis_block_sums, vis_block_offs,
                       vis_block_sums, total + 2, (const uint32_t*)nullptr);
}

// Only Gaussians that emit instances take part in the depth sort and in everything after it (typically a
// quarter of the scene): their (depth key, index) pairs are packed in index order, so the stable sort still
// resolves equal depths by ascending index.
__global__ void __launch_bounds__(256) compact_keys_kernel(int P, const uint32_t* __restrict__ tiles_touched,
                                                           const uint32_t* __restrict__ depth_keys,
                                                           const uint32_t* __restrict__ vis_block_offs,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ idx_out) {
    __shared__ uint32_t sm4[4];
    const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
    const bool emits = idx < P && tiles_touched[idx] > 0;
    uint32_t total;
    const uint32_t local = block256_excl_scan_u32(emits ? 1u : 0u, sm4, &total);
    if (emits) {
        const uint32_t dst = vis_block_offs[blockIdx.x] + local;
        keys_out[dst] = depth_keys[idx];
        idx_out[dst] = (uint32_t)idx;
    }
}
void launch_compact_keys(int P, const uint32_t* tiles_touched, const uint32_t* depth_keys, const uint32_t* vis_block_offs,
                         uint32_t* keys_out, uint32_t* idx_out, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(compact_keys_kernel, dim3(nblocks), dim3(256), 0, s, P, tiles_touched, depth_keys, vis_block_offs,
                       keys_out, idx_out);
}

void launch_emit(int P, int tiles_x, int tiles_y, const uint32_t* gidx_sorted, const uint32_t* tiles_touched,
                 const uint32_t* block_offs, const int* radii, float* rec, uint64_t* entries, int nblocks,
                 hipStream_t s) {
    hipLaunchKernelGGL(emit_kernel, dim3(nblocks), dim3(256), 0, s, P, tiles_x, tiles_y, gidx_sorted, tiles_touched,
                       block_offs, radii, rec, entries);
}

// rasterizer_impl.cu:116-138 on the packed entries (ranges pre-zeroed by the caller, :311)
__global__ void __launch_bounds__(256) tile_ranges_kernel(int R, const uint64_t* __restrict__ entries,
                                                          uint32_t* __restrict__ ranges) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= R) return;
    const uint32_t cur = entry_tile(entries[i]);
    if (i == 0) {
        ranges[2 * cur] = 0;
    } else {
        const uint32_t prev = entry_tile(entries[i - 1]);
        if (cur != prev) {
            ranges[2 * prev + 1] = (uint32_t)i;
            ranges[2 * cur] = (uint32_t)i;
        }
    }
    if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
}

// Processing order of the tiles: longest instance list first (LPT scheduling).  A tile is one
// workgroup of the blend kernels and list lengths are very uneven, so dispatching the heavy tiles
// first shortens the tail.  Single block: bucket histogram (descending) in LDS, scan, scatter; the
// order inside a bucket is arbitrary (it cannot change any result, tiles are independent).
// (An XCD-aware variant -- 8x8-tile macro-blocks assigned to the 8 XCDs round-robin, LPT inside each group,
// groups interleaved so that workgroup b lands on XCD b % 8 next to its spatial neighbours -- was measured 1 %
// SLOWER on S3: the blend kernels are VALU-bound, the splat records they share sit in the 256 MB MALL anyway,
// and per-group ordering costs more balance than the L2 locality returns.)
constexpr int ORDER_BUCKETS = 2048;
__global__ void __launch_bounds__(1024) tile_order_kernel(int tiles, const uint32_t* __restrict__ ranges,
                                                          uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[ORDER_BUCKETS];
    __shared__ uint32_t wsum[16];
    const int t = (int)threadIdx.x;
    for (int i = t; i < ORDER_BUCKETS; i += 1024) hist[i] = 0;
    __syncthreads();
    auto bucket = [](uint32_t n) {  // descending: long lists -> small bucket index
        const uint32_t b = n >> 2;
        return (uint32_t)(ORDER_BUCKETS - 1) - (b < (uint32_t)(ORDER_BUCKETS - 1) ? b : (uint32_t)(ORDER_BUCKETS - 1));
    };
    for (int i = t; i < tiles; i += 1024) atomicAdd(&hist[bucket(ranges[2 * i + 1] - ranges[2 * i])], 1u);
    __syncthreads();
    // exclusive scan of the 2048 buckets: two per thread
    const uint32_t h0 = hist[2 * t], h1 = hist[2 * t + 1];
    const uint32_t inc = wave_incl_scan_u32(h0 + h1);
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (t >> 6); w++) base += wsum[w];
    const uint32_t ex = base + inc - (h0 + h1);
    __syncthreads();
    hist[2 * t] = ex;
    hist[2 * t + 1] = ex + h0;
    __syncthreads();
    for (int i = t; i < tiles; i += 1024) order[atomicAdd(&hist[bucket(ranges[2 * i + 1] - ranges[2 * i])], 1u)] = (uint32_t)i;
}
void launch_tile_order(int tiles, const uint32_t* ranges, uint32_t* tile_order, hipStream_t s) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, tiles, ranges, tile_order);
}

void launch_tile_ranges(int R, const uint64_t* entries, uint32_t* ranges, hipStream_t s) {
    if (R <= 0) return;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, entries, ranges);
}

}  // namespace g4s
