// Binning: depth sort of Gaussians, instance emission, stable tile partition, tile ranges.
//
// Reference semantics (dsr/cuda_rasterizer/rasterizer_impl.cu:70-138,278-319): every visible
// Gaussian emits one instance per tile of its rect; instances are ordered by
// (tile, depth bits) with a STABLE sort, so equal depths inside a tile resolve by ascending
// Gaussian index; ranges[tile] = [start,end) of the tile's run.
//
// MI355X design (not the reference's 64-bit-key sort over all R instances):
//   1. sort the Gaussians that emit instances once by depth bits (32-bit keys, stable => ties by index),
//   2. emit instances in that order with a load-balanced expansion partitioned by OUTPUT slots (coalesced
//      8-byte stores), packed as  tile<<32 | gaussian  (the instance number k inside the Gaussian only picks the tile),
//   3. stable radix partition on the tile bits only (2 passes instead of 6).
// A stable partition of a depth-ordered sequence yields exactly the reference's per-tile order.
// Radix passes: a workgroup of four waves owns a block of keys, every wave ranks its contiguous share with ballot
// match-any against wave-private LDS counters (no workgroup barrier inside the ranking), and the block's keys are
// sorted by digit inside LDS before they are written out.
#include "g4s_internal.h"
#include "g4s_device.h"

namespace g4s {

// ---------------------------------------------------------------------------------------
// radix sort building blocks (8-bit digits)
//
// One pass = histogram, scan, scatter.  A workgroup of four waves owns a contiguous block of 256 * ITEMS keys;
// each wave owns a contiguous quarter of it and ranks its keys 64 at a time (ballot match-any over the digit
// bits, wave-private running counters in LDS -- no workgroup barrier inside the ranking loop).  Stability: blocks,
// waves inside a block, steps inside a wave and lanes inside a step are all ordered by key index.
//
// The scatter first sorts the block's keys by digit INSIDE LDS and then writes them out in that order: the keys of
// one digit leave as one contiguous run (256 * ITEMS / 256 keys on average), i.e. whole 128-byte lines.  Writing
// every key straight to its destination (the first design: one wave per chunk, 64 scattered 8-byte stores per
// step) left the sort bound by partial-line writes: 0.043 ms per pass over 4.4 M instances, against 0.022 now.

// BITS = digit width of the kernels' tables: 8 (256 rows; a pass may use fewer bits through `mask`) or 9 (512 rows, two
// digits per thread where a thread owns a digit: the three-pass depth sort below).  key_base (may be NULL): a device
// word subtracted from every key before its digit is taken.
template <typename K>
__device__ __forceinline__ uint32_t radix_digit(K key, uint32_t base, int shift, uint32_t mask) {
    return (uint32_t)((K)(key - (K)base) >> shift) & mask;
}

template <typename K, int ITEMS, int BITS>
__global__ void __launch_bounds__(256) radix_hist_kernel(const K* __restrict__ keys, int n, int shift, uint32_t mask,
                                                         uint32_t* __restrict__ hist, int nblocks,
                                                         const uint32_t* __restrict__ d_n,
                                                         const uint32_t* __restrict__ key_base) {
    constexpr int BINS = 1 << BITS;
    __shared__ uint32_t h[BINS];
    constexpr int TK = 256 * ITEMS;
    const int block = (int)blockIdx.x;
    if (d_n != nullptr) {  // the key count only lives on the device: the grid is sized for the largest possible n
        n = (int)*d_n;
        nblocks = (n + TK - 1) / TK;
        if (block >= nblocks) return;
    }
    const uint32_t base = key_base != nullptr ? *key_base : 0u;
    const int t = (int)threadIdx.x;
#pragma unroll
    for (int j = 0; j < BINS / 256; j++) h[t + 256 * j] = 0;
    __syncthreads();
    const int begin = block * TK;
    const int end = imin_(n, begin + TK);
    K k[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {  // all loads in flight before the first LDS atomic
        const int i = begin + 256 * u + t;
        k[u] = keys[i < end ? i : begin];
    }
#pragma unroll
    for (int u = 0; u < ITEMS; u++)
        if (begin + 256 * u + t < end) atomicAdd(&h[radix_digit(k[u], base, shift, mask)], 1u);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BINS / 256; j++) hist[(size_t)(t + 256 * j) * nblocks + block] = h[t + 256 * j];
}

// One block per digit row: exclusive scan of the row in place, row total to bin_total[row].
template <int TK>
__global__ void __launch_bounds__(256) radix_scan_kernel(uint32_t* __restrict__ hist, int nblocks,
                                                         uint32_t* __restrict__ bin_total,
                                                         const uint32_t* __restrict__ d_n) {
    __shared__ uint32_t sm4[4];
    if (d_n != nullptr) nblocks = ((int)*d_n + TK - 1) / TK;
    uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
    const int t = (int)threadIdx.x;
    const int seg = (nblocks + 255) / 256;
    const int b = imin_(nblocks, t * seg), e = imin_(nblocks, b + seg);
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

template <typename K, bool HAS_VAL, int ITEMS, int BITS>
__global__ void __launch_bounds__(256) radix_scatter_kernel(const K* __restrict__ keys_in, K* __restrict__ keys_out,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ vals_out, int n, int shift, uint32_t mask,
                                                            const uint32_t* __restrict__ hist,
                                                            const uint32_t* __restrict__ bin_total, int nblocks,
                                                            const uint32_t* __restrict__ d_n,
                                                            const uint32_t* __restrict__ key_base) {
    constexpr int TK = 256 * ITEMS;
    constexpr int BINS = 1 << BITS, PER = BINS / 256;  // digits a thread owns in the offsets step: PER * t .. PER * t + PER - 1
    __shared__ uint32_t s_cnt[4][BINS];  // wave-private digit counters, then: first local slot of (wave, digit)
    __shared__ uint32_t s_gbase[BINS];   // global position of local slot 0 of digit d, i.e. dst = s_gbase[d] + slot
    __shared__ uint32_t sm4[4];
    __shared__ K s_keys[TK];
    __shared__ uint32_t s_vals[HAS_VAL ? TK : 1];
    const int block = (int)blockIdx.x;
    if (d_n != nullptr) {
        n = (int)*d_n;
        nblocks = (n + TK - 1) / TK;
        if (block >= nblocks) return;
    }
    const uint32_t base = key_base != nullptr ? *key_base : 0u;
    const int t = (int)threadIdx.x, w = t >> 6, lane = t & 63;
    const int begin = block * TK;
    const int end = imin_(n, begin + TK);
    const int wbegin = begin + w * 64 * ITEMS;  // this wave's quarter
    K key[ITEMS];
    uint32_t val[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {
        const int i = wbegin + 64 * u + lane;
        const int ic = i < end ? i : begin;
        key[u] = keys_in[ic];
        val[u] = HAS_VAL ? vals_in[ic] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < PER; i++) s_cnt[j][t + 256 * i] = 0;
    __syncthreads();
    const uint64_t below = lanes_below_mask();
    uint32_t lrank[ITEMS];  // rank of the key among the keys of its wave with the same digit
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {
        const bool valid = wbegin + 64 * u + lane < end;
        const uint32_t d = radix_digit(key[u], base, shift, mask);
        uint64_t m = __ballot(valid);  // match-any over the digit bits
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t rank = (uint32_t)__popcll(m & below);
        uint32_t old = 0;
        if (valid && rank == 0) old = atomicAdd(&s_cnt[w][d], (uint32_t)__popcll(m));  // the group's first lane
        old = (uint32_t)__shfl((int)old, valid ? (int)__builtin_ctzll(m) : 0, 64);
        lrank[u] = old + rank;
    }
    __syncthreads();
    {   // digits PER * t ..: block-local start, first slot of every wave's share, global base of this block's run
        uint32_t c[PER][4], own[PER], tot[PER], own_sum = 0, tot_sum = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int d = PER * t + i;
#pragma unroll
            for (int j = 0; j < 4; j++) c[i][j] = s_cnt[j][d];
            own[i] = (c[i][0] + c[i][1]) + (c[i][2] + c[i][3]);
            tot[i] = bin_total[d];
            own_sum += own[i];
            tot_sum += tot[i];
        }
        uint32_t total;
        uint32_t dstart = block256_excl_scan_u32(own_sum, sm4, &total);
        uint32_t gdigit = block256_excl_scan_u32(tot_sum, sm4, &total);  // first position of digit PER * t overall
        // (the second scan's barriers: every thread has read its counters before anybody overwrites them)
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const int d = PER * t + i;
            s_cnt[0][d] = dstart;
            s_cnt[1][d] = dstart + c[i][0];
            s_cnt[2][d] = dstart + c[i][0] + c[i][1];
            s_cnt[3][d] = dstart + c[i][0] + c[i][1] + c[i][2];
            s_gbase[d] = gdigit + hist[(size_t)d * nblocks + block] - dstart;
            dstart += own[i];
            gdigit += tot[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {
        if (wbegin + 64 * u + lane < end) {
            const uint32_t d = radix_digit(key[u], base, shift, mask);
            const uint32_t slot = s_cnt[w][d] + lrank[u];
            s_keys[slot] = key[u];
            if (HAS_VAL) s_vals[slot] = val[u];
        }
    }
    __syncthreads();
    const int nvalid = end - begin;
#pragma unroll
    for (int u = 0; u < ITEMS; u++) {
        const int slot = 256 * u + t;
        if (slot < nvalid) {
            const K k = s_keys[slot];
            const uint32_t dst = s_gbase[radix_digit(k, base, shift, mask)] + (uint32_t)slot;
            keys_out[dst] = k;
            if (HAS_VAL) vals_out[dst] = s_vals[slot];
        }
    }
}

// d_n == nullptr: n keys (host-known).  Otherwise n is read from *d_n by the kernels and only bounds the grid
// (n_max >= *d_n): the launches can be queued before the host knows the count.
template <typename K, bool HAS_VAL, int ITEMS, int BITS = 8>
static void radix_pass(const K* kin, K* kout, const uint32_t* vin, uint32_t* vout, int n, int shift, int bits, uint32_t* hist,
                       uint32_t* bin_total, hipStream_t s, const uint32_t* d_n = nullptr,
                       const uint32_t* key_base = nullptr) {
    constexpr int TK = 256 * ITEMS;
    const int nblocks = (n + TK - 1) / TK;
    const uint32_t mask = (1u << bits) - 1u;  // digit width <= BITS (rows above the mask stay empty)
    hipLaunchKernelGGL((radix_hist_kernel<K, ITEMS, BITS>), dim3(nblocks), dim3(256), 0, s, kin, n, shift, mask, hist, nblocks,
                       d_n, key_base);
    hipLaunchKernelGGL(radix_scan_kernel<TK>, dim3(1 << BITS), dim3(256), 0, s, hist, nblocks, bin_total, d_n);
    hipLaunchKernelGGL((radix_scatter_kernel<K, HAS_VAL, ITEMS, BITS>), dim3(nblocks), dim3(256), 0, s, kin, kout, vin, vout,
                       n, shift, mask, hist, bin_total, nblocks, d_n, key_base);
}

int radix_sort_u32_pairs(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int n,
                         uint32_t* hist, uint32_t* bin_total, hipStream_t s, const uint32_t* d_n) {
    if (n <= 0) return 0;
    int cur = 0;
    for (int shift = 0; shift < 32; shift += 8) {
        if (cur == 0)
            radix_pass<uint32_t, true, SORT_ITEMS_U32>(keys_a, keys_b, vals_a, vals_b, n, shift, 8, hist, bin_total, s, d_n);
        else
            radix_pass<uint32_t, true, SORT_ITEMS_U32>(keys_b, keys_a, vals_b, vals_a, n, shift, 8, hist, bin_total, s, d_n);
        cur ^= 1;
    }
    return cur;
}

// The depth sort of a frame: keys are the float bits of positive depths, and the keys of one frame span far fewer than
// 32 bits -- a depth ratio of 2^16 between the farthest and the nearest emitting Gaussian is 16 exponent steps = 2^27
// key values.  Sorting (key - smallest key), which orders exactly as the keys do, therefore takes THREE passes of
// 9 bits (bits 0..26 of the difference) for every frame whose range is below 2^27, a launch-bound 3 x 3 kernels
// instead of 4 x 3.  The range is only known on the device when the passes are queued (key_min: a device word); the
// host learns it with the frame's totals and queues radix_sort_depth_top() for the remaining five bits if a frame
// ever needs it.  Returns the ping-pong index of the result, as radix_sort_u32_pairs does.
int radix_sort_depth_low(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int n, uint32_t* hist,
                         uint32_t* bin_total, hipStream_t s, const uint32_t* d_n, const uint32_t* key_min) {
    if (n <= 0) return 0;
    int cur = 0;
    for (int shift = 0; shift < DEPTH_SORT_LOW_BITS; shift += 9) {
        if (cur == 0)
            radix_pass<uint32_t, true, SORT_ITEMS_U32, 9>(keys_a, keys_b, vals_a, vals_b, n, shift, 9, hist, bin_total, s, d_n, key_min);
        else
            radix_pass<uint32_t, true, SORT_ITEMS_U32, 9>(keys_b, keys_a, vals_b, vals_a, n, shift, 9, hist, bin_total, s, d_n, key_min);
        cur ^= 1;
    }
    return cur;
}
// bits 27..31 of (key - key_min): from buffer `cur` into the other one; returns the new index
int radix_sort_depth_top(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int n, int cur,
                         uint32_t* hist, uint32_t* bin_total, hipStream_t s, const uint32_t* d_n, const uint32_t* key_min) {
    if (n <= 0) return cur;
    if (cur == 0)
        radix_pass<uint32_t, true, SORT_ITEMS_U32>(keys_a, keys_b, vals_a, vals_b, n, DEPTH_SORT_LOW_BITS, 32 - DEPTH_SORT_LOW_BITS, hist, bin_total, s, d_n, key_min);
    else
        radix_pass<uint32_t, true, SORT_ITEMS_U32>(keys_b, keys_a, vals_b, vals_a, n, DEPTH_SORT_LOW_BITS, 32 - DEPTH_SORT_LOW_BITS, hist, bin_total, s, d_n, key_min);
    return cur ^ 1;
}

// Stable partition on bits [begin_bit, end_bit): ceil(bits / 8) passes of EQUAL width (13 tile bits -> 7 + 6 rather than
// 8 + 5: with 32 bins the second pass' 64 lanes fight over too few LDS counters in the histogram).
int radix_sort_u64_keys(uint64_t* a, uint64_t* b, int n, int begin_bit, int end_bit, uint32_t* hist,
                        uint32_t* bin_total, hipStream_t s, const uint32_t* d_n) {
    if (n <= 0 || end_bit <= begin_bit) return 0;
    const int bits = end_bit - begin_bit, passes = (bits + 7) / 8, per = (bits + passes - 1) / passes;
    int cur = 0;
    for (int shift = begin_bit; shift < end_bit; shift += per) {
        const int nb = per < end_bit - shift ? per : end_bit - shift;
        if (cur == 0)
            radix_pass<uint64_t, false, SORT_ITEMS_U64>(a, b, nullptr, nullptr, n, shift, nb, hist, bin_total, s, d_n);
        else
            radix_pass<uint64_t, false, SORT_ITEMS_U64>(b, a, nullptr, nullptr, n, shift, nb, hist, bin_total, s, d_n);
        cur ^= 1;
    }
    return cur;
}

// ---------------------------------------------------------------------------------------
// instance counting (in depth order), emission, tile ranges

// tiles_touched gathered in depth order, 256 per block -> block sums; every rank also keeps its exclusive offset
// inside its block (rank_local), so that block_offs[r >> 8] + rank_local[r] is the first output slot of depth rank r.
__global__ void __launch_bounds__(256) count_block_sums_kernel(int P, const uint32_t* __restrict__ gidx,
                                                               const uint32_t* __restrict__ tiles_touched,
                                                               uint32_t* __restrict__ block_sums,
                                                               uint32_t* __restrict__ rank_local,
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
    const uint32_t local = block256_excl_scan_u32(c, sm4, &total);
    if (r < P) rank_local[r] = local;
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Single block: exclusive scan of up to two arrays of block sums (b may be NULL), the sum of a third (ref, may be
// NULL), and an optional clear of `zero_words` 32-bit words (the tile ranges, rasterizer_impl.cu:311 -- folded in
// here to save a launch).  Totals: total_a[0] = sum(a), total_a[1] = sum(ref), total_b[0] = sum(b).
__global__ void __launch_bounds__(1024) scan_block_sums_kernel(int nblocks, const uint32_t* __restrict__ a_sums,
                                                               uint32_t* __restrict__ a_offs,
                                                               const uint32_t* __restrict__ b_sums,
                                                               uint32_t* __restrict__ b_offs,
                                                               const uint32_t* __restrict__ ref_sums,
                                                               uint32_t* __restrict__ total_a, uint32_t* __restrict__ total_b,
                                                               uint32_t* __restrict__ zero_ptr, int zero_words,
                                                               const uint32_t* __restrict__ d_n, uint32_t capacity,
                                                               uint32_t* __restrict__ host_out,
                                                               uint32_t* __restrict__ status_out,
                                                               const uint32_t* __restrict__ kmin_blocks,
                                                               const uint32_t* __restrict__ kmax_blocks) {
    __shared__ uint32_t wa[16], wb[16], wr[16], wlo[16], whi[16];
    const int t = (int)threadIdx.x;
    for (int i = t; i < zero_words; i += 1024) zero_ptr[i] = 0u;
    if (d_n != nullptr) nblocks = (int)((*d_n + 255u) / 256u);  // blocks of a device-side element count
    const int seg = (nblocks + 1023) / 1024;
    const int b = imin_(nblocks, t * seg), e = imin_(nblocks, b + seg);
    uint32_t sa = 0, sb = 0, sr = 0, klo = 0xFFFFFFFFu, khi = 0u;
    for (int i = b; i < e; i++) {
        sa += a_sums[i];
        if (b_sums) sb += b_sums[i];
        if (ref_sums) sr += ref_sums[i];
        if (kmin_blocks) {  // range of the frame's depth keys
            klo = min(klo, kmin_blocks[i]);
            khi = max(khi, kmax_blocks[i]);
        }
    }
    if (kmin_blocks) {
        klo = ~wave_max_u32_full_wave(~klo);
        khi = wave_max_u32_full_wave(khi);
    }
    const uint32_t ia = wave_incl_scan_u32(sa), ib = wave_incl_scan_u32(sb), ir = wave_incl_scan_u32(sr);
    if ((t & 63) == 63) {
        wa[t >> 6] = ia;
        wb[t >> 6] = ib;
        wr[t >> 6] = ir;
        wlo[t >> 6] = klo;
        whi[t >> 6] = khi;
    }
    __syncthreads();
    uint32_t base_a = 0, base_b = 0, all_a = 0, all_b = 0, all_r = 0;
    for (int w = 0; w < 16; w++) {
        if (w < (t >> 6)) { base_a += wa[w]; base_b += wb[w]; }
        all_a += wa[w];
        all_b += wb[w];
        all_r += wr[w];
    }
    uint32_t run_a = base_a + ia - sa, run_b = base_b + ib - sb;
    for (int i = b; i < e; i++) {
        a_offs[i] = run_a;
        run_a += a_sums[i];
        if (b_sums) { b_offs[i] = run_b; run_b += b_sums[i]; }
    }
    if (t == 0) {
        total_a[0] = all_a;
        total_a[1] = all_r;
        if (kmin_blocks) {  // total[5] = smallest depth key of the frame (the sort subtracts it), total[6] = largest
            uint32_t lo = 0xFFFFFFFFu, hi = 0u;
            for (int w = 0; w < 16; w++) { lo = min(lo, wlo[w]); hi = max(hi, whi[w]); }
            if (lo > hi) lo = hi = 0u;  // (no emitting Gaussian)
            total_a[5] = lo;
            total_a[6] = hi;
            if (host_out != nullptr) { host_out[3] = lo; host_out[4] = hi; }
        }
        if (b_sums) {
            total_b[0] = all_b;
            // (presized forward) total_b[1] = instances that fit the caller's binning capacity, total_b[2] = overflow flag
            total_b[1] = all_a < capacity ? all_a : capacity;
            total_b[2] = all_a > capacity ? 1u : 0u;
        }
        if (status_out != nullptr) {  // g4s_rasterizer_forward_presized: the caller's device status words
            status_out[0] = all_r;
            status_out[1] = all_a;
            status_out[2] = all_b;
            status_out[3] = all_a > capacity ? 1u : 0u;
        }
        if (host_out != nullptr) {  // pinned, device-mapped host words: the read-back without a copy launch
            host_out[0] = all_a;
            host_out[1] = all_r;
            host_out[2] = all_b;
        }
    }
}

void launch_count_scan(int P, const uint32_t* gidx_sorted, const uint32_t* tiles_touched, uint32_t* block_sums,
                       uint32_t* block_offs, uint32_t* rank_local, uint32_t* total, int nblocks, hipStream_t s,
                       const uint32_t* d_n) {
    // d_n != nullptr: P / nblocks only bound the grid, the kernels read the count from the device
    hipLaunchKernelGGL(count_block_sums_kernel, dim3(nblocks), dim3(256), 0, s, P, gidx_sorted, tiles_touched,
                       block_sums, rank_local, d_n);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, nblocks, block_sums, block_offs,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, total, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, 0, d_n, 0xFFFFFFFFu, (uint32_t*)nullptr, (uint32_t*)nullptr,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr);
}

// Load-balanced expansion, partitioned by OUTPUT: a block owns EMIT_SLOTS consecutive instance slots, whatever
// Gaussians they belong to -- the nearest splats cover hundreds of tiles each, the far ones one or two, and a block
// per 256 depth ranks (the first design) left the few blocks of the nearest ranks running ten times longer than the
// rest.  Every depth rank emits at least one instance, so at most EMIT_SLOTS + 1 ranks reach into a block's window:
// the first one is found with three block-wide counting steps over the monotone offset arrays (no serial binary
// search over global memory), their offsets, indices and tile rects are staged in LDS, and every thread then
// produces output slots: a binary search in the LDS offsets, one coalesced 8-byte store per slot.
// The block also clears its slots of the forward's contribution masks (qhit), which saves a memset launch.
constexpr int EMIT_SLOTS = 1024;
// d_counts != NULL (g4s_rasterizer_forward_presized: the host never learns the counts): V = d_counts[0] emitting
// Gaussians, R_b = d_counts[1] instances (already clamped to the caller's capacity); the grid is sized for the
// capacity and surplus blocks leave.
__global__ void __launch_bounds__(256) emit_kernel(int V, uint32_t R_b, int tiles_x,
                                                   const uint32_t* __restrict__ gidx,
                                                   const uint32_t* __restrict__ block_offs, int nblocks_v,
                                                   const uint32_t* __restrict__ rank_local,
                                                   const uint2* __restrict__ tight_rect,
                                                   uint64_t* __restrict__ entries, uint8_t* __restrict__ qhit,
                                                   uint8_t* __restrict__ rec_flag,
                                                   const uint32_t* __restrict__ d_counts) {
    __shared__ uint32_t s_off[EMIT_SLOTS + 4];  // first slot of the staged ranks (ascending), then a sentinel
    __shared__ uint32_t s_idx[EMIT_SLOTS + 4];
    __shared__ uint32_t s_rect[EMIT_SLOTS + 4];   // x0 | y0 << 16
    __shared__ uint32_t s_rect2[EMIT_SLOTS + 4];  // rect width in tiles
    __shared__ uint32_t s_nr;
    const int t = (int)threadIdx.x;
    if (d_counts != nullptr) {
        V = (int)d_counts[0];
        R_b = d_counts[1];
        nblocks_v = (V + 255) / 256;
    }
    const uint32_t w0 = blockIdx.x * (uint32_t)EMIT_SLOTS;
    if (w0 >= R_b) return;  // (uniform)
    const uint32_t w1 = min(w0 + (uint32_t)EMIT_SLOTS, R_b);
    // contribution masks of this window (bytes [w0, w1)); w0 is a multiple of 1024, the array base 256-B aligned
    {
        // (and the validity bytes of the backward's gradient-record slots [w0, w1): the slot space has the same size)
        const uint32_t o = w0 + 4u * (uint32_t)t;
        if (o + 4u <= w1) {
            *reinterpret_cast<uint32_t*>(qhit + o) = 0u;
            *reinterpret_cast<uint32_t*>(rec_flag + o) = 0u;
        } else {
            for (uint32_t i = o; i < w1; i++) { qhit[i] = 0; rec_flag[i] = 0; }
        }
    }
    if (t == 0) s_nr = 0;
    // the 256-rank group that holds slot w0: last g with block_offs[g] <= w0 (block_offs[0] == 0)
    int lo = 0, n = nblocks_v;
    while (n > 1) {  // uniform; two rounds up to 65 536 groups
        const int S = (n + 255) / 256;
        const int j = lo + t * S;
        const int c = __syncthreads_count(t * S < n && block_offs[j] <= w0);
        lo += (c - 1) * S;
        n = imin_(S, n - (c - 1) * S);
    }
    const int g = lo;
    const uint32_t goff = block_offs[g];
    int r_first;
    {
        const int r = 256 * g + t;
        const int c = __syncthreads_count(r < V && goff + rank_local[r] <= w0);
        r_first = 256 * g + c - 1;
    }
    // stage the ranks whose first slot lies below w1
    for (int i = t; i <= EMIT_SLOTS; i += 256) {
        const int r = r_first + i;
        if (r >= V) break;
        const uint32_t off = block_offs[r >> 8] + rank_local[r];
        if (off >= w1) break;  // offsets ascend with the rank: nothing further on for this thread either
        const uint32_t idx = gidx[r];
        const uint2 tr = tight_rect[idx];  // the rect the preprocess counted (x0 | y0 << 16, width)
        s_off[i] = off;
        s_idx[i] = idx;
        s_rect[i] = tr.x;
        s_rect2[i] = tr.y;
        atomicMax(&s_nr, (uint32_t)i + 1u);
    }
    __syncthreads();
    const int nr = (int)s_nr;  // >= 1: rank r_first always qualifies
    // bisection steps for nr candidates (uniform; at most 11: 2^11 > EMIT_SLOTS + 1 -- a window of near splats holds a
    // handful of ranks, one of far ones a thousand)
    const int steps = 32 - __builtin_clz((uint32_t)nr | 1u);
    for (uint32_t o = w0 + (uint32_t)t; o < w1; o += 256) {
        // largest j < nr with s_off[j] <= o
        int a = 0, b = nr - 1;
        for (int it = 0; it < steps; it++) {
            const int mid = (a + b + 1) >> 1;
            if (s_off[mid] <= o) a = mid; else b = mid - 1;
        }
        const uint32_t k = o - s_off[a];
        const uint32_t w = s_rect2[a];
        const uint32_t rx0 = s_rect[a] & 0xFFFFu, ry0 = s_rect[a] >> 16;
        const uint32_t ty = k / w, tx = k - ty * w;
        const uint64_t tile = (uint64_t)((ry0 + ty) * (uint32_t)tiles_x + rx0 + tx);
        entries[o] = (tile << ENTRY_TILE_SHIFT) | (uint64_t)s_idx[a];
    }
}

// Index-order pass behind the totals scan, two jobs in one launch:
//  * gradient-record slots: rec[idx].inst_off = exclusive scan of tiles_touched over idx (so that the
//    per-Gaussian fold of the backward streams the record buffer sequentially);
//  * the (depth key, index) pairs of the Gaussians that emit instances (typically a quarter of the scene) are
//    packed in index order -- only they take part in the depth sort and in everything after it, and the stable
//    sort still resolves equal depths by ascending index.
__global__ void __launch_bounds__(256) slots_and_compact_kernel(int P, const uint32_t* __restrict__ tiles_touched,
                                                                const uint32_t* __restrict__ idx_block_offs,
                                                                float* __restrict__ rec,
                                                                const uint32_t* __restrict__ depth_keys,
                                                                const uint32_t* __restrict__ vis_block_offs,
                                                                uint32_t* __restrict__ keys_out,
                                                                uint32_t* __restrict__ idx_out) {
    __shared__ uint32_t sm4[4], sm4b[4];
    const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
    const uint32_t c = idx < P ? tiles_touched[idx] : 0u;
    uint32_t total;
    const uint32_t slot = block256_excl_scan_u32(c, sm4, &total);
    const uint32_t local = block256_excl_scan_u32(c > 0 ? 1u : 0u, sm4b, &total);
    if (c > 0) {
        rec[(size_t)idx * REC_FLOATS + 2] = __uint_as_float(idx_block_offs[blockIdx.x] + slot);
        const uint32_t dst = vis_block_offs[blockIdx.x] + local;
        keys_out[dst] = depth_keys[idx];
        idx_out[dst] = (uint32_t)idx;
    }
}
void launch_slots_and_compact(int P, const uint32_t* tiles_touched, const uint32_t* idx_block_offs, float* rec,
                              const uint32_t* depth_keys, const uint32_t* vis_block_offs, uint32_t* keys_out,
                              uint32_t* idx_out, int nblocks, hipStream_t s) {
    hipLaunchKernelGGL(slots_and_compact_kernel, dim3(nblocks), dim3(256), 0, s, P, tiles_touched, idx_block_offs, rec,
                       depth_keys, vis_block_offs, keys_out, idx_out);
}

void launch_scan_totals(const uint32_t* idx_block_sums, uint32_t* idx_block_offs, const uint32_t* ref_block_sums,
                        const uint32_t* vis_block_sums, uint32_t* vis_block_offs, uint32_t* total, int nblocks,
                        uint32_t* zero_ptr, int zero_words, hipStream_t s, uint32_t capacity, uint32_t* host_out,
                        uint32_t* status_out, const uint32_t* key_min_blocks, const uint32_t* key_max_blocks) {
    // total[0] = instances binned, total[1] = the reference's num_rendered, total[2] = emitting Gaussians,
    // total[3] = min(total[0], capacity), total[4] = 1 if total[0] > capacity
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3(1), dim3(1024), 0, s, nblocks, idx_block_sums, idx_block_offs,
                       vis_block_sums, vis_block_offs, ref_block_sums, total, total + 2, zero_ptr, zero_words,
                       (const uint32_t*)nullptr, capacity, host_out, status_out, key_min_blocks, key_max_blocks);
}

void launch_emit(int V, uint32_t R_b, int tiles_x, const uint32_t* gidx_sorted, const uint32_t* block_offs,
                 int nblocks_v, const uint32_t* rank_local, const uint2* tight_rect, uint64_t* entries,
                 uint8_t* qhit, uint8_t* rec_flag, hipStream_t s, const uint32_t* d_counts) {
    if (R_b == 0) return;  // (d_counts != NULL: R_b is the capacity the grid is sized for)
    hipLaunchKernelGGL(emit_kernel, dim3((R_b + EMIT_SLOTS - 1) / EMIT_SLOTS), dim3(256), 0, s, V, R_b, tiles_x,
                       gidx_sorted, block_offs, nblocks_v, rank_local, tight_rect, entries, qhit, rec_flag, d_counts);
}

// rasterizer_impl.cu:116-138 on the packed entries (ranges pre-zeroed, :311).  Four consecutive entries per thread
// (two 16-byte loads + the predecessor): a quarter of the workgroups and of the loads of the one-entry form.
__global__ void __launch_bounds__(256) tile_ranges_kernel(int R, const uint64_t* __restrict__ entries,
                                                          uint32_t* __restrict__ ranges, const uint32_t* __restrict__ d_n) {
    if (d_n != nullptr) R = (int)*d_n;
    const int i0 = (int)(blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= R) return;
    uint64_t e[4];
    if (i0 + 4 <= R) {
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(entries + i0);
        const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(entries + i0 + 2);
        e[0] = a.x; e[1] = a.y; e[2] = b.x; e[3] = b.y;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = entries[i0 + k < R ? i0 + k : R - 1];
    }
    uint32_t prev = i0 > 0 ? entry_tile(entries[i0 - 1]) : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int i = i0 + k;
        if (i >= R) break;
        const uint32_t cur = entry_tile(e[k]);
        if (i == 0) {
            ranges[2 * cur] = 0;
        } else if (cur != prev) {
            ranges[2 * prev + 1] = (uint32_t)i;
            ranges[2 * cur] = (uint32_t)i;
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
        prev = cur;
    }
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
                                                          uint32_t* __restrict__ order, uint32_t* __restrict__ zero_word) {
    __shared__ uint32_t hist[ORDER_BUCKETS];
    __shared__ uint32_t wsum[16];
    const int t = (int)threadIdx.x;
    if (t == 0 && zero_word != nullptr) *zero_word = 0u;  // (the backward's deep-tile counter: saves a memset launch)
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
void launch_tile_order(int tiles, const uint32_t* ranges, uint32_t* tile_order, hipStream_t s, uint32_t* zero_word) {
    hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, s, tiles, ranges, tile_order, zero_word);
}

void launch_tile_ranges(int R, const uint64_t* entries, uint32_t* ranges, hipStream_t s, const uint32_t* d_n) {
    if (R <= 0) return;
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 1023) / 1024), dim3(256), 0, s, R, entries, ranges, d_n);
}

}  // namespace g4s
