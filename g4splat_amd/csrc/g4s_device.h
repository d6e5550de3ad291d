// Device-side helpers for the gfx950 kernels: wave64 primitives (DPP reductions, ballots,
// scans) and the small float3 algebra of the surfel math.  Compiled with
// -ffp-contract=off: every fused multiply-add in the kernels is spelled fmaf() on purpose so
// that discrete decisions (culling, tile rects, alpha thresholds, early termination) are
// reproducible bit-for-bit against the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "g4s_internal.h"

namespace g4s {

constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ---- DPP --------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, false));
}
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

// acc += a * b / acc += a / acc -= a IN PLACE.  Written as asm so that the accumulator keeps its register: where two
// arms of a wave-uniform branch update the same loop-carried accumulators, hipcc gives one arm fresh registers and
// ends it with a chain of copies back (nine v_mov per visit in the blend backward's common arm).
__device__ __forceinline__ void acc_fma(float& acc, float a, float b) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b)); }
__device__ __forceinline__ void acc_add(float& acc, float a) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(a)); }
__device__ __forceinline__ void acc_sub(float& acc, float a) { asm volatile("v_sub_f32 %0, %0, %1" : "+v"(acc) : "v"(a)); }
__device__ __forceinline__ void acc_fnma(float& acc, float a, float b) { asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }

// Sum over the 64 lanes of a wave; the total is valid in lanes 48..63 (read it from lane 63).
// quad_perm / row_ror inside each row of 16, then row_bcast:15 and row_bcast:31 (GFX9/CDNA).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_f32<0xB1>(v);             // quad_perm:[1,0,3,2]
    v += dpp_f32<0x4E>(v);             // quad_perm:[2,3,0,1]
    v += dpp_f32<0x124>(v);            // row_ror:4
    v += dpp_f32<0x128>(v);            // row_ror:8
    v += dpp_f32<0x142, 0xA>(v);       // row_bcast:15 -> rows 1,3
    v += dpp_f32<0x143, 0xC>(v);       // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_to_lane63(uint32_t v) {
    v += dpp_u32<0xB1>(v);
    v += dpp_u32<0x4E>(v);
    v += dpp_u32<0x124>(v);
    v += dpp_u32<0x128>(v);
    v += dpp_u32<0x142, 0xA>(v);
    v += dpp_u32<0x143, 0xC>(v);
    return v;
}
// Sum of FOUR per-lane values over the 64 lanes in 10 VALU instructions (gfx950 permlane swaps):
// on return every lane of row k (lanes 16k..16k+15) holds the wave total of v_k.
//   v_permlane32_swap a,b : a' = [a.lo32, b.lo32], b' = [a.hi32, b.hi32]   => a'+b' = [a.lo+a.hi | b.lo+b.hi]
//   v_permlane16_swap p,q : p' = [p.r0, q.r0, p.r2, q.r2], q' = [p.r1, q.r1, p.r3, q.r3]
// The swaps are issued as inline asm: hipcc (ROCm 7.2) mis-compiles the
// __builtin_amdgcn_permlane{16,32}_swap pair result when two of them feed a third (it adds the
// first result to itself; caught by tests/hip_unit/wave_ops.hip).  The s_nop covers the
// VALU-write -> permlane-swap-read wait states hipcc does not insert around asm statements.
__device__ __forceinline__ float swap32_sum(float x, float y) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return x + y;
}
__device__ __forceinline__ float swap16_sum(float p, float q) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(p), "+v"(q));
    return p + q;
}
__device__ __forceinline__ float wave_sum4_to_rows(float v0, float v1, float v2, float v3) {
    const float s02 = swap32_sum(v0, v2);  // lanes 0-31: v0 partials, 32-63: v2 partials
    const float s13 = swap32_sum(v1, v3);
    float r = swap16_sum(s02, s13);        // rows: v0, v1, v2, v3 (16 partials each)
    r += dpp_f32<0xB1>(r);                 // quad_perm:[1,0,3,2]
    r += dpp_f32<0x4E>(r);                 // quad_perm:[2,3,0,1]
    r += dpp_f32<0x124>(r);                // row_ror:4
    r += dpp_f32<0x128>(r);                // row_ror:8
    asm volatile("" : "+v"(r));            // (see wave_sum16_to_rows)
    return r;
}

// Sixteen values at once: on return lane l of row k = l/16 holds, in out[i], the wave sum of v[4k+i]
// (valid in every lane of the row), so the lane that writes a row stores four consecutive terms with
// one 16-byte store.  All swaps of a stage share one s_nop and the four DPP chains interleave, so
// the 63 additions per term cost 8+4 swaps, 12 adds and 16 DPP adds for all sixteen terms.
__device__ __forceinline__ void wave_sum16_to_rows(float (&v)[16], float (&out)[4]) {
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %8\n\t"
        "v_permlane32_swap_b32 %1, %9\n\t"
        "v_permlane32_swap_b32 %2, %10\n\t"
        "v_permlane32_swap_b32 %3, %11\n\t"
        "v_permlane32_swap_b32 %4, %12\n\t"
        "v_permlane32_swap_b32 %5, %13\n\t"
        "v_permlane32_swap_b32 %6, %14\n\t"
        "v_permlane32_swap_b32 %7, %15"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
          "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    // s[j]: lanes 0-31 = partial sums of v[j], lanes 32-63 = partial sums of v[8+j]
    float s0 = v[0] + v[8], s1 = v[1] + v[9], s2 = v[2] + v[10], s3 = v[3] + v[11];
    float s4 = v[4] + v[12], s5 = v[5] + v[13], s6 = v[6] + v[14], s7 = v[7] + v[15];
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %4\n\t"
        "v_permlane16_swap_b32 %1, %5\n\t"
        "v_permlane16_swap_b32 %2, %6\n\t"
        "v_permlane16_swap_b32 %3, %7"
        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));
    // rows 0..3 of r_i: partial sums of v[i], v[4+i], v[8+i], v[12+i]
    float r0 = s0 + s4, r1 = s1 + s5, r2 = s2 + s6, r3 = s3 + s7;
    r0 += dpp_f32<0xB1>(r0); r1 += dpp_f32<0xB1>(r1); r2 += dpp_f32<0xB1>(r2); r3 += dpp_f32<0xB1>(r3);
    r0 += dpp_f32<0x4E>(r0); r1 += dpp_f32<0x4E>(r1); r2 += dpp_f32<0x4E>(r2); r3 += dpp_f32<0x4E>(r3);
    r0 += dpp_f32<0x124>(r0); r1 += dpp_f32<0x124>(r1); r2 += dpp_f32<0x124>(r2); r3 += dpp_f32<0x124>(r3);
    r0 += dpp_f32<0x128>(r0); r1 += dpp_f32<0x128>(r1); r2 += dpp_f32<0x128>(r2); r3 += dpp_f32<0x128>(r3);
    // keep the last additions here: sunk into the caller's `if (row_writer)` they can no longer be fused with their
    // DPP operand (the rotate must see all lanes) and cost three instructions each instead of one
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    // row k now holds the sums of v[4k] (r0), v[4k+1] (r1), v[4k+2] (r2), v[4k+3] (r3)
    out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3;
}

// The same sixteen sums, one value per lane: the lane's 4-lane quad q = (lane >> 2) & 3 of row k = lane >> 4 ends up with
// the wave sum of v[4k + 2 (q & 1) + (q >> 1)]  (see quad_term()).  After the swap stages the four registers are folded
// into one while they are reduced: lanes whose bit 3 is set keep r1 / r3 and hand r0 / r2 to the lane eight further (and
// vice versa), then bit 2 decides between the two survivors through a half-row mirror (which pairs the quads of a half
// row; it also reverses the lanes inside them, which the last two steps -- a sum over the quad -- do not care about).
// 11 DPP-class instructions instead of the 16 of four parallel butterflies.  `b3` / `b2` = (lane & 8) != 0 / (lane & 4)
// != 0, computed once by the caller so that they live in scalar registers.
__host__ __device__ __forceinline__ int quad_term(int lane) { return 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1); }
__device__ __forceinline__ float wave_sum16_to_quads(float (&v)[16], bool b3, bool b2) {
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %8\n\t"
        "v_permlane32_swap_b32 %1, %9\n\t"
        "v_permlane32_swap_b32 %2, %10\n\t"
        "v_permlane32_swap_b32 %3, %11\n\t"
        "v_permlane32_swap_b32 %4, %12\n\t"
        "v_permlane32_swap_b32 %5, %13\n\t"
        "v_permlane32_swap_b32 %6, %14\n\t"
        "v_permlane32_swap_b32 %7, %15"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
          "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    float s0 = v[0] + v[8], s1 = v[1] + v[9], s2 = v[2] + v[10], s3 = v[3] + v[11];
    float s4 = v[4] + v[12], s5 = v[5] + v[13], s6 = v[6] + v[14], s7 = v[7] + v[15];
    asm volatile(
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %4\n\t"
        "v_permlane16_swap_b32 %1, %5\n\t"
        "v_permlane16_swap_b32 %2, %6\n\t"
        "v_permlane16_swap_b32 %3, %7"
        : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7));
    // rows 0..3 of r_i: partial sums of v[i], v[4+i], v[8+i], v[12+i]
    const float r0 = s0 + s4, r1 = s1 + s5, r2 = s2 + s6, r3 = s3 + s7;
    const float k01 = b3 ? r1 : r0, g01 = b3 ? r0 : r1;
    const float k23 = b3 ? r3 : r2, g23 = b3 ? r2 : r3;
    const float x01 = k01 + dpp_f32<0x128>(g01);  // row_ror:8 : lane l <-> l ^ 8
    const float x23 = k23 + dpp_f32<0x128>(g23);
    const float k = b2 ? x23 : x01, g = b2 ? x01 : x23;
    float y = k + dpp_f32<0x141>(g);              // row_half_mirror: the other quad of the same half row
    y += dpp_f32<0xB1>(y);                        // quad_perm:[1,0,3,2]
    y += dpp_f32<0x4E>(y);                        // quad_perm:[2,3,0,1]
    asm volatile("" : "+v"(y));                   // (keep the last additions out of the caller's store branch)
    return y;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_sum_to_lane63(v), 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    for (int off = 32; off >= 1; off >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = v > o ? v : o;
    }
    return v;
}
// The same for a wave whose 64 lanes are all active: the DPP ladder of wave_sum_to_lane63 with max (lanes a row mask
// leaves out read 0, the identity of an unsigned max), six VALU instructions instead of six trips through the LDS crossbar.
__device__ __forceinline__ uint32_t wave_max_u32_full_wave(uint32_t v) {
    v = max(v, dpp_u32<0xB1>(v));
    v = max(v, dpp_u32<0x4E>(v));
    v = max(v, dpp_u32<0x124>(v));
    v = max(v, dpp_u32<0x128>(v));
    v = max(v, dpp_u32<0x142, 0xA>(v));
    v = max(v, dpp_u32<0x143, 0xC>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Inclusive prefix sum over the wave (lane i gets sum of lanes 0..i).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    const int l = lane_id();
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = (uint32_t)__shfl_up((int)v, off, 64);
        if (l >= off) v += o;
    }
    return v;
}

__device__ __forceinline__ uint64_t lanes_below_mask() {
    return (1ull << lane_id()) - 1ull;
}

// Exclusive scan over a 256-thread block (4 waves).  `smem4` = 4 words of LDS.  Returns the
// exclusive prefix of `v`; *block_total = sum over the block.  Contains two barriers.
__device__ __forceinline__ uint32_t block256_excl_scan_u32(uint32_t v, uint32_t* smem4, uint32_t* block_total) {
    const int w = (int)(threadIdx.x >> 6);
    uint32_t inc = wave_incl_scan_u32(v);
    __syncthreads();
    if (lane_id() == 63) smem4[w] = inc;
    __syncthreads();
    uint32_t s0 = smem4[0], s1 = smem4[1], s2 = smem4[2], s3 = smem4[3];
    uint32_t base = (w > 0 ? s0 : 0) + (w > 1 ? s1 : 0) + (w > 2 ? s2 : 0);
    *block_total = s0 + s1 + s2 + s3;
    return base + inc - v;
}

// ---- float -> int with saturation (v_cvt_i32_f32 already saturates, NaN -> 0) -----------
__device__ __forceinline__ int sat_int(float v) { return (int)v; }

// ---- tiny vector algebra ------------------------------------------------------------------
struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 mk3(float x, float y, float z) { return F3{x, y, z}; }

// auxiliary.h:78-86 (transformPoint4x3), :99-107 (transformVec4x3), :109-117 (…Transpose)
__device__ __forceinline__ F3 xform_point_4x3(F3 p, const float* m) {
    return mk3(m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12], m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
               m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]);
}
__device__ __forceinline__ F3 xform_vec_4x3(F3 p, const float* m) {
    return mk3(m[0] * p.x + m[4] * p.y + m[8] * p.z, m[1] * p.x + m[5] * p.y + m[9] * p.z,
               m[2] * p.x + m[6] * p.y + m[10] * p.z);
}
__device__ __forceinline__ F3 xform_vec_4x3_T(F3 p, const float* m) {
    return mk3(m[0] * p.x + m[1] * p.y + m[2] * p.z, m[4] * p.x + m[5] * p.y + m[6] * p.z,
               m[8] * p.x + m[9] * p.y + m[10] * p.z);
}

// Unit quaternion (stored w,x,y,z) -> rotation, column-major R[c*3+r] (auxiliary.h:212-234).
// rsqrtf of the reference is evaluated as an exact 1/sqrt so CPU and GPU agree bitwise.
__device__ __forceinline__ void quat_to_rotmat(const float4 q, float* R) {
    const float s = 1.0f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    const float w = q.x * s, x = q.y * s, y = q.z * s, z = q.w * s;
    R[0] = 1.f - 2.f * (y * y + z * z);
    R[1] = 2.f * (x * y + w * z);
    R[2] = 2.f * (x * z - w * y);
    R[3] = 2.f * (x * y - w * z);
    R[4] = 1.f - 2.f * (x * x + z * z);
    R[5] = 2.f * (y * z + w * x);
    R[6] = 2.f * (x * z + w * y);
    R[7] = 2.f * (y * z - w * x);
    R[8] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ int imin_(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax_(int a, int b) { return a > b ? a : b; }

// auxiliary.h:66-76 (getRect)
__device__ __forceinline__ void get_rect(float px, float py, int max_radius, int gx, int gy, int& x0, int& y0,
                                         int& x1, int& y1) {
    x0 = imin_(gx, imax_(0, sat_int((px - max_radius) / TILE)));
    y0 = imin_(gy, imax_(0, sat_int((py - max_radius) / TILE)));
    x1 = imin_(gx, imax_(0, sat_int((px + max_radius + TILE - 1) / TILE)));
    y1 = imin_(gy, imax_(0, sat_int((py + max_radius + TILE - 1) / TILE)));
}

// Tiles a splat is binned into: the reference's 3-sigma square rect [x0,x1) x [y0,y1) (getRect)
// intersected with the tiles its alpha-cutoff box (record quad 5) touches.  Pixel centres sit on
// integer coordinates, tile t covers pixels [16 t, 16 t + 15].
__device__ __forceinline__ void tight_tile_rect(const float4 box, int x0, int y0, int x1, int y1, int& tx0,
                                                int& ty0, int& tx1, int& ty1) {
    const int LIM = 1 << 20;
    const int bx0 = imax_(-LIM, imin_(LIM, sat_int(floorf(box.x * (1.0f / TILE)))));
    const int by0 = imax_(-LIM, imin_(LIM, sat_int(floorf(box.y * (1.0f / TILE)))));
    const int bx1 = imax_(-LIM, imin_(LIM, sat_int(floorf(box.z * (1.0f / TILE))))) + 1;
    const int by1 = imax_(-LIM, imin_(LIM, sat_int(floorf(box.w * (1.0f / TILE))))) + 1;
    tx0 = imax_(x0, bx0);
    ty0 = imax_(y0, by0);
    tx1 = imax_(tx0, imin_(x1, bx1));
    ty1 = imax_(ty0, imin_(y1, by1));
    if (!(box.x <= box.z) || !(box.y <= box.w)) {  // empty box
        tx1 = tx0;
        ty1 = ty0;
    }
}

constexpr float NEAR_N = 0.2f;    // auxiliary.h:37
constexpr float FAR_N = 100.0f;   // auxiliary.h:38
constexpr float FILTER_INV_SQUARE = 2.0f;  // auxiliary.h:39

// Ray/splat evaluation for one (pixel, splat) pair -- forward.cu:351-383 / backward.cu:267-301.
//
// The reference forms k = x Tw - Tu, l = y Tw - Tv, p = k x l per (pixel, splat) pair (forward.cu:355-357: twelve
// multiply-adds) and the interpolated depth s.x Tw.x + s.y Tw.y + Tw.z from s = p.xy / p.z.  p is AFFINE in the pixel,
//     p(x, y) = x A + y B + D,     A = Tv x Tw,  B = Tw x Tu,  D = Tu x Tv,
// and the depth is (p . Tw) / p.z = det(T) / p.z  (p . Tw = det[k; l; Tw] = det[Tu; Tv; Tw]).  For splats that qualify
// (REC_AFFINE, below) the forward preprocess therefore stores the three vectors of
//     p'(x, y) = (A (x - cx) + B (y - cy) + Dc) / det(T),     Dc = p(cx, cy),
// expanded about the splat's own 3-sigma centre c (no cancellation between frame-sized terms: |x - cx| is a few
// sigma wherever the splat matters) and evaluated in double from the single-precision T (splat_affine).  A pixel then
// costs two subtractions and six FMAs instead of twelve, s = p'.xy / p'.z is unchanged (homogeneous), and
//     depth = 1 / p'.z,   1 / depth = p'.z
// come for free -- the reciprocal of the depth that the distortion terms need is no longer a second v_rcp_f32 -- and the
// backward accumulates the moments of dL/dp' instead of dL/dTu, dL/dTv, dL/dTw (blend.hip, preprocess.hip: K8).
//
// REC_AFFINE (bit 31 of the record's tile-count word, q0.w) is a certificate, decided once per splat by the forward
// preprocess: (a) the affine form is as accurate as the reference's own arithmetic to AFFINE_TOL in alpha -- a
// first-order bound on the rounding of p' over the splat's alpha-cutoff box; it fails for splats seen nearly edge-on,
// whose apparent aspect ratio exceeds ~15, and for a singular T -- (b) at every pixel where the low-pass exponent rho2d
// could still pass the alpha test the 3-D exponent is the smaller one and is not in the tie band, so rho = rho3d and
// the low-pass arithmetic is dead, (c) the interpolated depth stays above the near plane wherever alpha can pass.
// Every other splat keeps T in its record and takes the general path: the reference's arithmetic, operation for
// operation (the fmaf placement there is the contract shared with oracle/surfel_oracle.c eval_pair).
struct PairEval {
    float sx, sy, inv_pz, rho3d, rho2d, depth, inv_depth, G, alpha, dx, dy;
    bool in3d;  // the 3-D exponent was the smaller one (always, on the affine path)
};
constexpr uint32_t REC_AFFINE = 0x80000000u;
#ifndef G4S_AFFINE_TOL
#define G4S_AFFINE_TOL 2e-5f
#endif
constexpr float AFFINE_TOL = G4S_AFFINE_TOL;  // bound on the relative error of alpha that the affine form may add (see splat_affine; swept in profiles/r06_affine_tol.txt)

// The nine coefficients of p' (record quads 2..4) and whether the splat qualifies by (a).
struct SplatAffine {
    float A[3], B[3], Dc[3];
    bool ok;
};
// ex, ey: how far from the centre, in pixels, the splat is ever evaluated with a chance to pass the alpha test (its
// alpha-cutoff box clipped to the frame); smax: |s| there (sqrt of the cutoff exponent).
__device__ __forceinline__ void splat_affine(const float* T, float cx, float cy, float ex, float ey, float smax, SplatAffine& o) {
    const double Tu[3] = {T[0], T[1], T[2]}, Tv[3] = {T[3], T[4], T[5]}, Tw[3] = {T[6], T[7], T[8]};
    auto cross = [](const double* a, const double* b, double* c) {
        c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
    };
    double A[3], B[3], D[3], Dc[3];
    cross(Tv, Tw, A);
    cross(Tw, Tu, B);
    cross(Tu, Tv, D);
    const double det = Tu[0] * A[0] + Tu[1] * A[1] + Tu[2] * A[2];
#pragma unroll
    for (int i = 0; i < 3; i++) Dc[i] = (double)cx * A[i] + (double)cy * B[i] + D[i];
    const double g = 1.0 / det;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        o.A[i] = (float)(g * A[i]);
        o.B[i] = (float)(g * B[i]);
        o.Dc[i] = (float)(g * Dc[i]);
    }
    // First-order bound on what single precision does to p'_c = A'_c dx + B'_c dy + Dc'_c over |dx| <= ex, |dy| <= ey:
    // the stored coefficients, dx / dy and the two FMAs each round once -> E_c <= 3 u (|A'_c| ex + |B'_c| ey + |Dc'_c|).
    // s = p'.xy / p'.z, alpha ~ exp(-|s|^2 / 2):  |d alpha / alpha| <= smax (E_x + E_y + 2 smax E_z) / min |p'.z|.
    const float u3 = 3.0f * 5.9604645e-8f;
    const float Ex = fmaf(fabsf(o.A[0]), ex, fmaf(fabsf(o.B[0]), ey, fabsf(o.Dc[0])));
    const float Ey = fmaf(fabsf(o.A[1]), ex, fmaf(fabsf(o.B[1]), ey, fabsf(o.Dc[1])));
    const float vz = fmaf(fabsf(o.A[2]), ex, fabsf(o.B[2]) * ey);
    const float Ez = vz + fabsf(o.Dc[2]);
    const float pzmin = fabsf(o.Dc[2]) - vz;
    const float bound = u3 * smax * (Ex + Ey + 2.0f * smax * Ez);
    // (NaN / inf anywhere -- det == 0, overflow of p / det -- fails the comparisons)
    o.ok = pzmin > 0.5f * fabsf(o.Dc[2]) && bound <= AFFINE_TOL * pzmin && fabsf(o.Dc[2]) < 1e30f && Ex < 1e30f && Ey < 1e30f;
}

// The ray-splat intersection of one pixel with an AFFINE splat up to the two candidate exponents (forward.cu:349-372):
//   rho3d = |s|^2 with s = p.xy / p.z;   rho2d = FilterInvSquare |centre - pixel|^2.
// Returns false if p.z == 0.  Same arithmetic as eval_pair's affine path: parts (b) of the REC_AFFINE certificate are
// evaluated with it.  `tie` = the two exponents are within 1e-5 relative of each other (then (b) fails).
__device__ __forceinline__ bool eval_rho_affine(float pxf, float pyf, float cx, float cy, const SplatAffine& f, PairEval& e,
                                                bool& tie) {
    e.dx = pxf - cx;
    e.dy = pyf - cy;
    const float ppx = fmaf(f.A[0], e.dx, fmaf(f.B[0], e.dy, f.Dc[0]));
    const float ppy = fmaf(f.A[1], e.dx, fmaf(f.B[1], e.dy, f.Dc[1]));
    const float ppz = fmaf(f.A[2], e.dx, fmaf(f.B[2], e.dy, f.Dc[2]));
    tie = false;
    if (ppz == 0.0f) return false;
    e.rho2d = FILTER_INV_SQUARE * fmaf(e.dx, e.dx, e.dy * e.dy);
    const float inv = __builtin_amdgcn_rcpf(ppz);
    e.sx = ppx * inv;
    e.sy = ppy * inv;
    e.rho3d = fmaf(e.sx, e.sx, e.sy * e.sy);
    tie = fabsf(e.rho3d - e.rho2d) <= 1e-5f * e.rho2d;
    return true;
}

// One pixel against one splat: forward.cu:349-393 up to alpha.  `affine` (wave-uniform) = the splat's record carries
// REC_AFFINE; c0..c8 are then A', B', Dc', otherwise Tu, Tv, Tw (record quads 2..4).  The two kinds of splat share
// everything after p: the general extras sit inside one uniform branch that only refines values the common code has
// already produced.
// (Flag: bool, or a 0 / 1 word the caller keeps in an SGPR)
template <typename Flag>
__device__ __forceinline__ bool eval_pair(Flag affine, float pxf, float pyf, float cx, float cy, float c0, float c1,
                                          float c2, float c3, float c4, float c5, float c6, float c7, float c8,
                                          float opa, PairEval& e) {
    // No early exits: the three rejections (p.z == 0, depth < near, alpha < 1/255) are ANDed into the returned
    // predicate at the end.  Every lane runs every instruction anyway (a VALU instruction costs the same whatever EXEC
    // says); an early `return false` only buys an EXEC region -- s_and_saveexec, a branch, the restore -- around the
    // rest, and those scalar instructions are what the blend kernels' waves are short of (-3 % blend_bwd, -6 % blend_fwd,
    // profiles/r04_ab_blend_bwd.txt).  A lane with p.z == 0 computes with inf / NaN and is rejected by `ok`.
    // The affine form is the main line; a splat without REC_AFFINE redoes the evaluation the reference's way inside ONE
    // uniform region, so that the common case passes a single (taken) branch here.  Written as an if / else the two forms
    // cost every visit four branches (hipcc lowers the wave-uniform bool to a lane mask and a diamond to two
    // conditional branches): blend_bwd 0.79 -> 0.92 ms, all of what the affine form had gained.
    e.dx = pxf - cx;
    e.dy = pyf - cy;
    float ppx = fmaf(c0, e.dx, fmaf(c3, e.dy, c6));
    float ppy = fmaf(c1, e.dx, fmaf(c4, e.dy, c7));
    float ppz = fmaf(c2, e.dx, fmaf(c5, e.dy, c8));
    // s = p.xy / p.z through one v_rcp_f32 (1 ulp) instead of two IEEE divisions (~22 instructions); the only discrete
    // decision that depends on it is `rho3d <= rho2d` on the general path: when the two are within 1e-5 relative of each
    // other the exact quotient is recomputed, so the branch taken is the oracle's.
    float inv = __builtin_amdgcn_rcpf(ppz);
    e.sx = ppx * inv;
    e.sy = ppy * inv;
    e.rho3d = fmaf(e.sx, e.sx, e.sy * e.sy);
    e.in3d = true;
    float rho = e.rho3d;
    e.depth = inv;      // affine: det(T) / p.z with p pre-divided by det(T)
    e.inv_depth = ppz;
    // forward.cu:354 `if (p.z == 0.0) continue;` -- for the affine form: a NORMAL p'.z.  1 / denormal is +-inf here (one
    // v_rcp_f32): s is then +-inf (rho = inf, G = 0, alpha = 0) or, where p'.x is exactly 0 on the same pixel, 0 x inf = NaN,
    // which fminf(0.99, NaN) would turn into alpha = 0.99.  Neither needs a test of its own: the alpha test at the end is an
    // ORDERED compare of opa * G, before the clamp, and rejects 0 and NaN alike -- a v_cmp and an s_and less per visit than
    // a predicate on p'.z ANDed in.  (|p'.z| < 1e-38 means a depth beyond 1e38: the horizon of the splat's plane.)
    if (__builtin_expect(!affine, 0)) {  // forward.cu:355-378 with (c0..c2) = Tu, (c3..c5) = Tv, (c6..c8) = Tw  (out of line: the common path falls through)
        const float kx = fmaf(pxf, c6, -c0), ky = fmaf(pxf, c7, -c1), kz = fmaf(pxf, c8, -c2);
        const float lx = fmaf(pyf, c6, -c3), ly = fmaf(pyf, c7, -c4), lz = fmaf(pyf, c8, -c5);
        ppx = fmaf(ky, lz, -(kz * ly));
        ppy = fmaf(kz, lx, -(kx * lz));
        ppz = fmaf(kx, ly, -(ky * lx));
        inv = __builtin_amdgcn_rcpf(ppz);
        e.sx = ppx * inv;
        e.sy = ppy * inv;
        e.rho3d = fmaf(e.sx, e.sx, e.sy * e.sy);
        e.rho2d = FILTER_INV_SQUARE * fmaf(e.dx, e.dx, e.dy * e.dy);
        float d3 = fmaf(e.sx, c6, e.sy * c7) + c8;  // forward.cu:373
        if (fabsf(e.rho3d - e.rho2d) <= 1e-5f * e.rho2d) {  // too close to call: exact quotient
            e.sx = ppx / ppz;
            e.sy = ppy / ppz;
            e.rho3d = fmaf(e.sx, e.sx, e.sy * e.sy);
            d3 = fmaf(e.sx, c6, e.sy * c7) + c8;
        }
        // rho = min(rho3d, rho2d) and the depth select share one compare.  (NaN rho3d -- 0 * inf when p.z is
        // denormal -- takes the rho2d side in both, like fminf.)
        e.in3d = e.rho3d <= e.rho2d;
        rho = e.in3d ? e.rho3d : e.rho2d;
        e.depth = e.in3d ? d3 : c8;
        e.inv_depth = __builtin_amdgcn_rcpf(e.depth);
        // forward.cu:378 `if (depth < near_n) continue;` (REC_AFFINE certifies depth >= near wherever the splat can pass)
        // (the reference's two rejections of this path enter the alpha test the same way: an infinite exponent is G = 0)
        if (!(ppz != 0.0f && !(e.depth < NEAR_N))) rho = __builtin_inff();
        // 1 / p.z for the general gradient block.  Set on this path only: an affine splat's 1 / p'.z IS its depth, and a
        // second name for it that the other arm redefines costs the common arm a register copy at the join.
        e.inv_pz = inv;
    }
    // forward.cu:383-385 `power = -0.5 rho; if (power > 0) continue;` can never fire (rho is a sum of squares),
    // and exp(power) = exp2(rho * (-0.5 log2 e)): scaling by -0.5 is exact, so folding it into the constant
    // rounds exactly like (-0.5f * rho) * log2e.  One v_exp_f32: rho in [0, 11.2] for anything that can pass,
    // error ~3e-7 relative.
    e.G = __builtin_amdgcn_exp2f(rho * (-0.5f * 1.4426950408889634f));
    const float a = opa * e.G;
    // (v_min_f32 by hand: behind the branch on the test below hipcc no longer knows that `a` is a canonical number and
    // puts a v_max_f32 a, a in front of fminf's v_min)
    asm("v_min_f32 %0, 0x3f7d70a4, %1" : "=v"(e.alpha) : "v"(a));  // fminf(0.99f, a)
    return a >= 1.0f / 255.0f;  // forward.cu:391 `if (alpha < 1.0f / 255.0f) continue;` (0.99 > 1 / 255: the clamp never decides it)
}

}  // namespace g4s
