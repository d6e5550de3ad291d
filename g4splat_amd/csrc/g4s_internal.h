// Internal declarations shared by the translation units of libg4s_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/g4s_rasterizer.h"

namespace g4s {

constexpr int TILE = 16;           // tile edge in pixels (part of the output definition, auxiliary.h:66-76)
constexpr int REC_FLOATS = 32;     // per-Gaussian splat record, 128 B = 8 x float4 (layout below)
constexpr int REC_QUADS = REC_FLOATS / 4;
constexpr int BLEND_QUADS = 5;     // quads per staged entry in LDS: what the per-pixel arithmetic needs (layout below)
// Gradient terms per (tile, Gaussian) instance, in record order: [0..2] colour, [3..5] normal, [6..14] the T terms,
// [15] opacity, [16..17] the low-pass centre terms (mean2D).  The T terms of a REC_AFFINE splat are the moments of the adjoint of p' (the affine
// ray-splat intersection of g4s_device.h): [6..8] S = sum dL/dp', [9..11] X = sum (x - cx) dL/dp', [12..14] Y = sum
// (y - cy) dL/dp' -- the cross products of backward.cu:396-426 are linear in the pixel and are taken once per Gaussian
// by K8; those of any other splat are dL/dTu, dL/dTv, dL/dTw themselves.
constexpr int GRAD_FLOATS = 18;
constexpr int GRAD_STRIDE = 20;    // floats per stored gradient record (80 B, 16-byte aligned; the last two unused)
// Splat record (written by the forward preprocess, read by emit / blend / backward):
//   q0 = (centre.x, centre.y, bits(inst_off), bits(binned rect width | height << 16 | REC_AFFINE))
//   q1 = (normal.xyz (view space, flipped towards the camera), opacity)
//   q2 = (Tu.x, Tu.y, Tu.z, Tv.x)   q3 = (Tv.y, Tv.z, Tw.x, Tw.y)   q4 = (Tw.z, r, g, b) -- or, for REC_AFFINE splats,
//   q2 = (A'.x, A'.y, A'.z, B'.x)   q3 = (B'.y, B'.z, Dc'.x, Dc'.y)   q4 = (Dc'.z, r, g, b): the affine ray-splat
//        intersection p'(x, y) = A' (x - cx) + B' (y - cy) + Dc' (g4s_device.h: splat_affine)
//   q5 = (bits(bx0), bits(bx1), Tw.z, bits(by0 | by1 << 16)): the bounding box, in 8x8-pixel quadrants and clamped to the
//        frame, of the region where this splat can pass the 1/255 alpha test (box_quadrants below; empty: bx0 > bx1; a
//        frame is at most 131 070 quadrants across and 65 534 down); T[8] (the densification surrogate of K8 needs it
//        whatever quads 2..4 hold)
//   q6 = (ex, ey, ux, uy), q7 = (1/a^2, 1/b^2, r2, bits(binned rect x0 | y0 << 16)): the same region exactly -- the
//        union of the ellipse (centre e, unit major axis u, semi-axes a >= b) that the alpha-cutoff disk of the splat
//        projects to and of the low-pass disk |pixel - centre|^2 <= r2 -- slightly enlarged; 1/a^2 = 0: no ellipse, use
//        the box.  The binned rect lives in q0.w / q7.w: q7 shares its 64-byte line with q4, which the blend backward
//        reads anyway, so an entry's instance number (gradient-record slot - inst_off) costs it no extra memory traffic.
// The blend loops keep q0..q4 of every staged entry in LDS (BLEND_QUADS); q5..q7 only steer culling.
// The 8x8-pixel quadrants [lo, hi] (indices, clamped to the frame) that hold a pixel of the float interval [lo, hi];
// empty: lo = 1 > hi = 0.  Quadrants are all the blend forward asks the box about, and since they start at multiples of
// eight nothing is lost against pixel bounds: x0 > 8 k + 7 <=> floor(ceil(x0) / 8) > k, x1 < 8 k <=> floor(floor(x1) / 8) < k.
__host__ __device__ inline void box_quadrants(float lo, float hi, int extent, uint32_t& q0, uint32_t& q1) {
    // pixel centres sit on integer coordinates: the pixels inside [lo, hi] are ceil(lo) .. floor(hi)
    const float l = ceilf(lo), h = floorf(hi);
    const float top = (float)(extent - 1);
    q0 = 1u;
    q1 = 0u;
    if (!(l <= h) || !(h >= 0.0f) || !(l <= top)) return;  // empty (also NaN)
    q0 = (l > 0.0f ? (uint32_t)l : 0u) >> 3;
    q1 = (h < top ? (uint32_t)h : (uint32_t)(extent - 1)) >> 3;
}
constexpr uint32_t CULLED_KEY = 0xFFFFFFFFu;
// Keys per thread of the radix passes (a workgroup of 256 threads owns 256 * ITEMS consecutive keys).
#ifndef G4S_SORT_ITEMS_U32
#define G4S_SORT_ITEMS_U32 8
#endif
#ifndef G4S_SORT_ITEMS_U64
#define G4S_SORT_ITEMS_U64 16
#endif
constexpr int SORT_ITEMS_U32 = G4S_SORT_ITEMS_U32;  // depth sort of the emitting Gaussians (32-bit keys + index)
constexpr int SORT_ITEMS_U64 = G4S_SORT_ITEMS_U64;  // tile partition of the packed instances
inline size_t sort_blocks(size_t n, int items) { return (n + (size_t)256 * items - 1) / ((size_t)256 * items); }
// The depth sort takes three 9-bit passes over (key - smallest key of the frame) and a fourth one over the bits above
// only for a frame whose keys span 2^27 values or more (binning.hip: radix_sort_depth_low / _top).
constexpr int DEPTH_SORT_LOW_BITS = 27;
// Packed instance: bits 63..32 tile id, 31..0 Gaussian index -- the reference's key without the depth bits
// (rasterizer_impl.cu:102-103 keeps the tile id in the upper word too), so any frame the reference can render fits.
// Both fields sit on a 32-bit boundary on purpose: hipcc (ROCm 7.2) narrows "(e >> 24) & 0xFFFFFF" to a 3-byte load
// and then drops the mask.  The instance number k of an entry inside its Gaussian (= its gradient-record slot minus
// inst_off) is not stored: the backward recomputes it from the tile and the Gaussian's binned tile rect.
constexpr int ENTRY_TILE_SHIFT = 32;
__host__ __device__ inline uint32_t entry_idx(uint64_t e) { return (uint32_t)e; }
__host__ __device__ inline uint32_t entry_tile(uint64_t e) { return (uint32_t)(e >> ENTRY_TILE_SHIFT); }
// The rect of tiles a Gaussian is binned into, as the record keeps it: extent word = width | height << 16 (bit 31 is
// REC_AFFINE: at most 65 535 tiles across, 32 767 down), origin word = x0 | y0 << 16; row-major instance numbering.
__host__ __device__ inline uint32_t rect_extent_word(int w, int h) { return (uint32_t)w | ((uint32_t)h << 16); }
__host__ __device__ inline uint32_t rect_tiles(uint32_t extent_word) {
    return (extent_word & 0xFFFFu) * ((extent_word >> 16) & 0x7FFFu);
}
__host__ __device__ inline uint32_t instance_number(uint32_t extent_word, uint32_t origin_word, uint32_t tile_x, uint32_t tile_y) {
    return (tile_y - (origin_word >> 16)) * (extent_word & 0xFFFFu) + (tile_x - (origin_word & 0xFFFFu));
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Sub-allocation of the three caller-owned chunks.  Private between forward and backward
// (the reference's GeometryState/BinningState/ImageState, rasterizer_impl.h:21-73).
struct GeomLayout {
    size_t rec, clamped, tiles_touched, tight_rect, internal_radii, keys_a, keys_b, vals_a, vals_b, hist, bin_total,
        key_min_blocks, key_max_blocks, block_sums, block_offs, ref_block_sums, idx_block_sums, idx_block_offs, vis_block_sums, vis_block_offs, total, bytes;
    int nblocks;   // 256-wide blocks over P
};
struct BinLayout {
    size_t ent_a, ent_b, hist, bin_total, qhit, rec_flag, bytes;
};
struct ImgLayout {
    size_t ranges, final_T, n_contrib, tile_order, tile_depth, tile_order_bwd, hot_count, hot_list, bytes;
};

inline GeomLayout geom_layout(size_t P) {
    GeomLayout L{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    L.nblocks = (int)((P + 255) / 256);
    L.rec = take(P * REC_FLOATS * 4);
    L.clamped = take(P);
    L.tiles_touched = take(P * 4);
    L.tight_rect = take(P * 8);  // (x0 | y0 << 16, width) of the tile rect the Gaussian is binned into; written where tiles_touched > 0
    L.internal_radii = take(P * 4);
    L.keys_a = take(P * 4);
    L.keys_b = take(P * 4);
    L.vals_a = take(P * 4);
    L.vals_b = take(P * 4);
    // the depth sort runs over the emitting Gaussians only (n <= P): capacity for n = P
    L.hist = take((size_t)512 * (sort_blocks(P, SORT_ITEMS_U32) + 1) * 4);  // (9-bit digits: 512 rows)
    L.bin_total = take(512 * 4);
    L.key_min_blocks = take((size_t)(L.nblocks ? L.nblocks : 1) * 4);
    L.key_max_blocks = take((size_t)(L.nblocks ? L.nblocks : 1) * 4);
    L.block_sums = take((size_t)(L.nblocks ? L.nblocks : 1) * 4);
    L.block_offs = take((size_t)(L.nblocks ? L.nblocks : 1) * 4);
    L.ref_block_sums = take((size_t)(L.nblocks ? L.nblocks : 1) * 4);
    L.idx_block_sums = take((size_t)(L.nblocks ? L.nblocks : 1) * 4);
    L.idx_block_offs = take((size_t)(L.nblocks ? L.nblocks : 1) * 4);
    L.vis_block_sums = take((size_t)(L.nblocks ? L.nblocks : 1) * 4);
    L.vis_block_offs = take((size_t)(L.nblocks ? L.nblocks : 1) * 4);
    L.total = take(256);
    L.bytes = o + 256;  // slack for aligning the chunk base
    return L;
}
inline BinLayout bin_layout(size_t R) {
    BinLayout L{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    L.ent_a = take((R ? R : 1) * 8);
    L.ent_b = take((R ? R : 1) * 8);
    L.hist = take((size_t)256 * (sort_blocks(R, SORT_ITEMS_U64) + 1) * 4);
    L.bin_total = take(256 * 4);
    L.qhit = take((R ? R : 1));
    // one validity byte per gradient-record slot of the backward (cleared by emit together with qhit): bit 0 = terms
    // 0..15 written, bit 1 = the low-pass terms 16..17 written
    L.rec_flag = take((R ? R : 1));
    L.bytes = o + 256;
    return L;
}
inline ImgLayout img_layout(size_t N, size_t tiles) {
    ImgLayout L{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o = align_up(o + n); return r; };
    L.ranges = take(tiles * 8);
    L.final_T = take(N * 3 * 4);
    L.n_contrib = take(N * 2 * 4);
    L.tile_order = take(tiles * 4);
    L.tile_depth = take(tiles * 8);      // (0, deepest last contributor) per tile, written by the blend forward
    L.tile_order_bwd = take(tiles * 4);  // processing order of the backward: deepest live list first
    L.hot_count = take(256);             // the backward's deep-tile counter ...
    L.hot_list = take(tiles * 4);        // ... and list (blend_bwd_hot_kernel)
    L.bytes = o + 256;
    return L;
}
inline char* align_ptr(char* p) { return (char*)align_up((size_t)p); }

// ---- launchers (each defined next to its kernels) --------------------------------------

struct PreprocessArgs {
    int P, D, M, W, H, tiles_x, tiles_y;
    uint32_t* ref_block_sums;  // per 256-Gaussian block: sum of the reference's tiles_touched (3-sigma rect)
    uint32_t* idx_block_sums;  // per 256-Gaussian block (index order): sum of the binned tile counts
    uint32_t* vis_block_sums;  // per 256-Gaussian block: number of Gaussians that emit at least one instance
    uint32_t* key_min_blocks;  // per 256-Gaussian block: smallest / largest depth key among them (0xFFFFFFFF / 0 if none)
    uint32_t* key_max_blocks;
    const float *means3D, *scales, *rotations, *opacities, *shs, *transMat_precomp, *colors_precomp;
    const float *viewmatrix, *projmatrix, *cam_pos;
    float scale_modifier;
    bool sh_vec16;  // shs is [P,16,3] on a 16-byte aligned base
    bool no_fastpath;  // tests (option "no_fastpath"): no splat is given REC_AFFINE -- every one takes the reference's arithmetic
    const float* shs_rest;  // split layout (g4s_rasterizer_forward_split_sh): shs = [P,1,3], shs_rest = [P,M-1,3]; NULL = packed
    float* rec;
    uint8_t* clamped;
    uint32_t* tiles_touched;
    uint2* tight_rect;  // (x0 | y0 << 16, width in tiles) of the rect counted in tiles_touched; written where that is > 0
    int* radii;
    uint32_t* depth_keys;
};
void launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);

// Stable LSD radix sort.  32-bit keys with 32-bit payload (ping-pong a<->b, result index
// returned: 0 = in *_a, 1 = in *_b) over bits [0,32).
// d_n != nullptr: the count is read from *d_n on the device, `n` (>= *d_n) only sizes the launches.
int radix_sort_u32_pairs(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int n,
                         uint32_t* hist, uint32_t* bin_total, hipStream_t s, const uint32_t* d_n = nullptr);
// The depth sort: three 9-bit passes over bits [0, DEPTH_SORT_LOW_BITS) of (key - *key_min) -- complete for a frame whose
// keys span less than 2^27 values --, and the pass over the bits above for one that spans more (from buffer `cur`).
int radix_sort_depth_low(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int n, uint32_t* hist,
                         uint32_t* bin_total, hipStream_t s, const uint32_t* d_n, const uint32_t* key_min);
int radix_sort_depth_top(uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a, uint32_t* vals_b, int n, int cur,
                         uint32_t* hist, uint32_t* bin_total, hipStream_t s, const uint32_t* d_n, const uint32_t* key_min);
// 64-bit keys-only over bits [begin_bit, end_bit).
int radix_sort_u64_keys(uint64_t* a, uint64_t* b, int n, int begin_bit, int end_bit, uint32_t* hist,
                        uint32_t* bin_total, hipStream_t s, const uint32_t* d_n = nullptr);

// Exclusive scan (in depth order) of tiles_touched: block_offs[r >> 8] + rank_local[r] = first output slot of depth
// rank r; total[0] = instances binned.
void launch_count_scan(int P, const uint32_t* gidx_sorted, const uint32_t* tiles_touched, uint32_t* block_sums,
                       uint32_t* block_offs, uint32_t* rank_local, uint32_t* total, int nblocks, hipStream_t s,
                       const uint32_t* d_n = nullptr);
// Totals and block offsets that only need the preprocess' per-block partial sums (no sort): exclusive scans of
// idx_block_sums and vis_block_sums; total[0] = instances binned, total[1] = the reference's num_rendered,
// total[2] = number of emitting Gaussians.  Also clears zero_words 32-bit words at zero_ptr (the tile ranges).
void launch_scan_totals(const uint32_t* idx_block_sums, uint32_t* idx_block_offs, const uint32_t* ref_block_sums,
                        const uint32_t* vis_block_sums, uint32_t* vis_block_offs, uint32_t* total, int nblocks,
                        uint32_t* zero_ptr, int zero_words, hipStream_t s, uint32_t capacity = 0xFFFFFFFFu,
                        uint32_t* host_out = nullptr, uint32_t* status_out = nullptr,
                        const uint32_t* key_min_blocks = nullptr, const uint32_t* key_max_blocks = nullptr);
// Index-order pass: gradient-record slots (rec[idx].inst_off = exclusive scan of tiles_touched over idx) and the
// stable compaction of the emitting Gaussians' (depth key, index) pairs.
void launch_slots_and_compact(int P, const uint32_t* tiles_touched, const uint32_t* idx_block_offs, float* rec,
                              const uint32_t* depth_keys, const uint32_t* vis_block_offs, uint32_t* keys_out,
                              uint32_t* idx_out, int nblocks, hipStream_t s);
// Instances in depth order, R_b of them; also clears qhit[0, R_b) and rec_flag[0, R_b).
void launch_emit(int V, uint32_t R_b, int tiles_x, const uint32_t* gidx_sorted, const uint32_t* block_offs,
                 int nblocks_v, const uint32_t* rank_local, const uint2* tight_rect, uint64_t* entries,
                 uint8_t* qhit, uint8_t* rec_flag, hipStream_t s, const uint32_t* d_counts);
void launch_tile_ranges(int R, const uint64_t* entries, uint32_t* ranges, hipStream_t s, const uint32_t* d_n = nullptr);
// Longest-list-first processing order of the tiles (work balance of the blend kernels).
void launch_tile_order(int tiles, const uint32_t* ranges, uint32_t* tile_order, hipStream_t s, uint32_t* zero_word = nullptr);

struct BlendFwdArgs {
    int W, H, tiles_x, tiles_y;
    const uint32_t* ranges;
    const uint32_t* tile_order;
    const uint64_t* entries;
    const float* rec;
    const float* bg;
    float* final_T;
    uint32_t* n_contrib;
    uint8_t* qhit;  // per sorted instance: bit q set if some pixel of quadrant q blended it (pre-zeroed)
    uint32_t* tile_depth;  // per tile (0, number of (entry, quadrant) pairs blended): the backward's work, for its ordering
    int box_only;   // tests (option "box_only"): skip quadrants by the bounding box only
    float* out_color;
    float* out_others;
};
void launch_blend_fwd(const BlendFwdArgs& a, hipStream_t s);

struct BlendBwdArgs {
    int W, H, tiles_x, tiles_y;
    const uint32_t* ranges;
    const uint32_t* tile_order;
    const uint64_t* entries;
    const float* rec;
    const float* bg;
    const float* final_T;
    const uint32_t* n_contrib;
    const uint8_t* qhit;
    const float* dL_dpix;
    const float* dL_depths;
    float* grad_inst;  // R x GRAD_STRIDE, NOT cleared: a record is valid only where rec_flag says so
    uint8_t* rec_flag; // R bytes (binning chunk, cleared by the forward's emit): bit 0 = terms 0..15 written, bit 1 = low-pass terms 16..17 written
    // Record slots that exist (= the R both buffers were sized for).  A presized forward that overflowed its capacity
    // keeps the first `capacity` instances in depth order, but a kept instance's slot (inst_off + k, an index-order
    // scan over ALL binned instances) can lie beyond it: such a record is dropped, never written (the frame is
    // flagged invalid in the status word; memory stays safe).
    uint32_t n_slots;
    // deep tiles (more than hot_threshold live list positions) are left to blend_bwd_hot_kernel: the one-wave
    // kernel appends them to hot_list (hot_count pre-cleared), the four-wave kernel runs behind it
    int hot_threshold;  // < 0: all tiles go to the four-wave kernel (frames with too few tiles to fill the GPU one wave each)
    uint32_t* hot_count;
    uint32_t* hot_list;  // tiles entries
    // Zero-fill riding along with the one-wave kernel (which is VALU-bound and leaves HBM idle): up to two float
    // ranges (dL_dsh, or its dc / rest parts) that every workgroup clears a 1/grid share of when its tile is done.
    // K8 then writes the rows of visible Gaussians only.  zero_base == NULL: nothing to clear.
    float* zero_base[2];      // 16-byte aligned
    uint32_t zero_quads[2];   // float4 count
    uint32_t zero_tail[2];    // 0..3 floats behind the quads
};
constexpr int BWD_HOT_THRESHOLD = 2048;  // live list positions; override for tests: option "bwd_hot_threshold"
constexpr int BWD_FOUR_WAVE_MAX_TILES = 768;  // frames with at most this many tiles use the four-wave kernel throughout
void launch_blend_bwd(const BlendBwdArgs& a, hipStream_t s);

struct PreprocessBwdArgs {
    int P, D, M, W, H;  // W,H: the truncated values of backward.cu:618-619
    int frame_W, frame_H;    // the frame's real size and the forward's scale_modifier: K8 re-forms the forward's T, which
    float scale_modifier;    // the moments written by the blend backward refer to (the reference's backward ignores both)
    const float *means3D, *scales, *rotations, *shs, *transMat_precomp, *colors_precomp;
    const float *viewmatrix, *projmatrix, *campos;
    const int* radii;
    const float* rec;
    const uint8_t* clamped;
    const float* grad_inst;
    const uint8_t* rec_flag;
    uint32_t n_slots;  // record slots that exist (see BlendBwdArgs::n_slots): slot runs are clamped to them
    bool sh_vec16;  // shs and dL_dsh are [P,16,3] on 16-byte aligned bases
    const float* shs_rest;  // split layout: shs / dL_dsh are the [P,1,3] parts, shs_rest / dL_dsh_rest the [P,M-1,3] ones
    float* dL_dsh_rest;
    bool sh_prezeroed;  // dL_dsh (and dL_dsh_rest) were cleared by the blend backward: K8 writes visible rows only
    float* view_stats;  // optional [P,2]: (||dL_dmean2D.xy|| of this view, visible ? 1 : 0), written or -- accumulate -- added
    // Optional second copy of the parameter gradients, PACKED: one row of 3 M + 13 floats per Gaussian with radii > 0, in
    // index order -- [dL_dmean3D 3 | dL_dsh 3 M | dL_dopacity 1 | dL_dscale 2 | dL_drot 4 | view_stats 2 | bits(index)] --
    // the send buffer of the multi-GPU owner exchange (g4s_packed_rows, include/g4s_rasterizer.h).  Row of a Gaussian =
    // packed_block_offs[its 256-block] + the visible Gaussians before it in the block.
    float* packed_rows;
    const uint32_t* packed_block_offs;
    uint32_t packed_capacity;  // rows; a row beyond it is dropped
    bool accumulate;    // parameter gradients are ADDED to their tensors (views accumulated in place); nothing is cleared
    float *dL_dmean2D, *dL_dnormal, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dtransMat, *dL_dsh, *dL_dscale,
        *dL_drot;
};
void launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s);

}  // namespace g4s
