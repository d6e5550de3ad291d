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

constexpr float NEAR_N = 0.2f;    // auxiliary.h:37
constexpr float FAR_N = 100.0f;   // auxiliary.h:38
constexpr float FILTER_INV_SQUARE = 2.0f;  // auxiliary.h:39

// Ray/splat evaluation for one (pixel, splat) pair -- forward.cu:351-383 / backward.cu:267-301.
// The fmaf placement is the contract shared with the oracle (oracle/surfel_oracle.c eval_pair).
struct PairEval {
    float sx, sy, pz, kx, ky, kz, lx, ly, lz, rho3d, rho2d, depth, G, alpha, dx, dy;
};
__device__ __forceinline__ bool eval_pair(float pxf, float pyf, float cx, float cy, float Tux, float Tuy, float Tuz,
                                          float Tvx, float Tvy, float Tvz, float Twx, float Twy, float Twz,
                                          float opa, PairEval& e) {
    e.kx = fmaf(pxf, Twx, -Tux);
    e.ky = fmaf(pxf, Twy, -Tuy);
    e.kz = fmaf(pxf, Twz, -Tuz);
    e.lx = fmaf(pyf, Twx, -Tvx);
    e.ly = fmaf(pyf, Twy, -Tvy);
    e.lz = fmaf(pyf, Twz, -Tvz);
    const float ppx = fmaf(e.ky, e.lz, -(e.kz * e.ly));
    const float ppy = fmaf(e.kz, e.lx, -(e.kx * e.lz));
    const float ppz = fmaf(e.kx, e.ly, -(e.ky * e.lx));
    if (ppz == 0.0f) return false;
    e.pz = ppz;
    e.sx = ppx / ppz;
    e.sy = ppy / ppz;
    e.rho3d = fmaf(e.sx, e.sx, e.sy * e.sy);
    e.dx = cx - pxf;
    e.dy = cy - pyf;
    e.rho2d = FILTER_INV_SQUARE * fmaf(e.dx, e.dx, e.dy * e.dy);
    const float rho = fminf(e.rho3d, e.rho2d);
    e.depth = (e.rho3d <= e.rho2d) ? fmaf(e.sx, Twx, e.sy * Twy) + Twz : Twz;
    if (e.depth < NEAR_N) return false;
    const float power = -0.5f * rho;
    if (power > 0.0f) return false;
    e.G = expf(power);
    e.alpha = fminf(0.99f, opa * e.G);
    if (e.alpha < 1.0f / 255.0f) return false;
    return true;
}

}  // namespace g4s
