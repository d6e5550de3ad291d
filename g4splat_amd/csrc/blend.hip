// Front-to-back alpha blending (K6) and its backward (K7) over 16x16 tiles.
//
// Reference semantics: dsr/cuda_rasterizer/forward.cu:258-443, backward.cu:143-440.
//
// MI355X design:
//   * a tile's sorted instance list is staged through LDS: the five 16-byte quads of the splat record that the
//     per-pixel arithmetic needs, SoA in LDS so the inner loop reads them with conflict-free broadcast
//     ds_read_b128 (the three culling quads are consumed by the staging thread itself);
//   * pixels are grouped in wave64-sized 8x8 quadrants; which quadrants a splat can reach at all (its
//     alpha-cutoff region: bounding box, then exact ellipse + low-pass disk) is decided once per staged
//     entry, and the quadrant waves visit only those entries;
//   * forward: 4 waves per tile, one quadrant each; a saturated pixel carries Tt = 0, a quadrant stops as
//     soon as its 64 pixels are saturated; per instance the forward records which quadrants blended it
//     (qhit, one byte);
//   * backward: no float atomics anywhere.  ONE wave owns a whole tile, every lane owns four pixels
//     (the same position in each quadrant); qhit is an exact work mask, so entries and quadrants without
//     contribution are never touched.  The sum over the tile's 256 pixels is 3 in-register adds plus one
//     wave reduction built from gfx950's v_permlane32_swap / v_permlane16_swap (sixteen terms at once,
//     g4s_device.h), written as ONE 80-byte record per (tile, Gaussian) instance at a slot reserved for that
//     Gaussian (inst_off + k) and flagged valid in a byte array.  The per-Gaussian kernel (preprocess.hip,
//     K8) folds a Gaussian's contiguous records in a fixed order => bit-reproducible gradients, no
//     workgroup barriers in the hot loop.
#include "g4s_internal.h"
#include "g4s_device.h"

// visits of the blend forward per test of "is the quadrant saturated?" (1, 2, 4 or 8; profiles/r06_ab_bitclear.txt)
#ifndef G4S_FWD_PAIRS
#define G4S_FWD_PAIRS 4
#endif

namespace g4s {

// Which of the tile's four 8x8-pixel quadrants -- columns 2 tile_x + {0, 1}, rows 2 tile_y + {0, 1} -- the splat's
// alpha-cutoff box reaches (record quad 5: quadrant bounds, g4s_internal.h): bit q = quadrant (q & 1, q >> 1).
__device__ __forceinline__ uint32_t quads_in_box(const float4 q5, uint32_t kx0, uint32_t ky0) {
    const uint32_t x0 = __float_as_uint(q5.x), x1 = __float_as_uint(q5.y), yw = __float_as_uint(q5.w);
    const uint32_t y0 = yw & 0xFFFFu, y1 = yw >> 16;
    const uint32_t cols = (kx0 >= x0 && kx0 <= x1 ? 1u : 0u) | (kx0 + 1 >= x0 && kx0 + 1 <= x1 ? 2u : 0u);
    const uint32_t rows = (ky0 >= y0 && ky0 <= y1 ? 1u : 0u) | (ky0 + 1 >= y0 && ky0 + 1 <= y1 ? 2u : 0u);
    return ((rows & 1u) ? cols : 0u) | ((rows & 2u) ? cols << 2 : 0u);
}

// true if no pixel of the quadrant can pass the alpha test: the quadrant rectangle [qx, qx+7] x [qy, qy+7] meets
// neither the low-pass disk around the splat centre nor the (enlarged) cutoff ellipse of record quads 6..7
// (centre e, unit major axis u, 1/a^2, 1/b^2).  The minimum of the convex form over a rectangle that does not
// contain the centre lies on its boundary: four 1-D minimisations with clamping; the form itself is evaluated in
// the ellipse's eigenframe, where it is a sum of two squares (no cancellation for needles).
__device__ __forceinline__ bool quad_misses_region(const float4 q0, const float4 q6, const float4 q7, float qx, float qy) {
    const float x1 = qx + 7.0f, y1 = qy + 7.0f;
    // low-pass disk
    const float ddx = fmaxf(fmaxf(qx - q0.x, q0.x - x1), 0.0f), ddy = fmaxf(fmaxf(qy - q0.y, q0.y - y1), 0.0f);
    if (ddx * ddx + ddy * ddy <= q7.z) return false;
    if (q7.x == 0.0f) return false;  // no ellipse (1 / a^2 = 0): the bounding box decided
    const float ux = q6.z, uy = q6.w, ia = q7.x, ib = q7.y;
    const float ax0 = qx - q6.x, ax1 = x1 - q6.x, ay0 = qy - q6.y, ay1 = y1 - q6.y;  // rectangle relative to e
    if (ax0 <= 0.0f && ax1 >= 0.0f && ay0 <= 0.0f && ay1 >= 0.0f) return false;      // centre inside
    // M = ia u u^T + ib v v^T, v = (-uy, ux): all three entries are sums of non-negative terms except m12
    const float m11 = ia * ux * ux + ib * uy * uy, m22 = ia * uy * uy + ib * ux * ux, m12 = (ia - ib) * ux * uy;
    const float r22 = m12 / m22, r11 = m12 / m11;
    auto form = [&](float dx, float dy) {
        const float t1 = dx * ux + dy * uy, t2 = dy * ux - dx * uy;
        return ia * t1 * t1 + ib * t2 * t2;
    };
    auto clampf = [](float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); };
    const float f0 = form(ax0, clampf(-r22 * ax0, ay0, ay1));  // edge x = x0: dy* = -m12 dx / m22
    const float f1 = form(ax1, clampf(-r22 * ax1, ay0, ay1));
    const float f2 = form(clampf(-r11 * ay0, ax0, ax1), ay0);  // edge y = y0: dx* = -m12 dy / m11
    const float f3 = form(clampf(-r11 * ay1, ax0, ax1), ay1);
    return fminf(fminf(f0, f1), fminf(f2, f3)) > 1.0f;
}

// ---------------------------------------------------------------------------------------
// K6 forward

constexpr int FWD_BATCH = 256;

struct FwdPixel {
    // Tt = running transmittance while the pixel is live; once it is saturated (forward.cu:327,399 `done`) Tt = -T, the
    // NEGATED transmittance it had when it stopped (the value the reference leaves in T), and a pixel outside the image
    // starts at 0: a dead pixel then fails `test_T >= 1e-4` by itself (test_T <= 0), so the loop needs no per-lane `done`
    // predicate, "all 64 pixels dead" is one v_cmp of Tt against 0, saturating is ONE select (Tt = stop ? -|T| : test_T:
    // the sign and magnitude operand modifiers are free), and the final transmittance is |Tt| either way.  (Until round 6 a
    // dead pixel carried Tt = 0 and a second register kept its last T: one more compare, one more select and the join
    // copies of a two-armed branch per visit.)
    float Tt;
    uint32_t last_contributor, median_contributor;
    float C0, C1, C2, N0, N1, N2, Dd, M1, M2, distortion, median_depth;
};
constexpr float FWD_MSCALE = FAR_N / (FAR_N - NEAR_N);
constexpr float FWD_DMD_K = (FAR_N * NEAR_N) / (FAR_N - NEAR_N);  // m(depth) = mscale - dmd_k / depth

// one pixel against one staged splat (forward.cu:349-433).  `nolp` (wave-uniform) = the splat is REC_AFFINE (see eval_pair).
__device__ __forceinline__ void fwd_visit(FwdPixel& p, bool nolp, float pxf, float pyf, const float4 q0, const float4 q1,
                                          const float4 q2, const float4 q3, const float4 q4, uint32_t contributor) {
    PairEval e;
    if (eval_pair(nolp, pxf, pyf, q0.x, q0.y, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q1.w, e)) {
        const float alpha = e.alpha, depth = e.depth;
        const float T = p.Tt;
        const float test_T = T * (1 - alpha);
        // Both arms update Tt (and the blending arm last_contributor) IN PLACE, as asm: written as assignments, the two
        // definitions meet in fresh registers and every visit pays four copies at the join.
        if (test_T < 0.0001f) {
            asm volatile("v_or_b32 %0, 0x80000000, %0" : "+v"(p.Tt));  // dead from here on: -|T|
        } else {
            const float w = alpha * T;
            const float A = 1 - T;
            const float md = fmaf(-FWD_DMD_K, e.inv_depth, FWD_MSCALE);  // 1 / depth = p'.z: no second reciprocal
            const float md2 = md * md;
            p.distortion = fmaf(fmaf(-(md + md), p.M1, fmaf(md2, A, p.M2)), w, p.distortion);
            p.Dd = fmaf(depth, w, p.Dd);
            p.M1 = fmaf(md, w, p.M1);
            p.M2 = fmaf(md2, w, p.M2);
            // (last_contributor first, in place; the median select then reads it instead of a second copy of `contributor`)
            asm volatile("v_mov_b32 %0, %1" : "+v"(p.last_contributor) : "s"(contributor));
            if (T > 0.5f) {
                p.median_depth = depth;
                p.median_contributor = p.last_contributor;
            }
            p.N0 = fmaf(q1.x, w, p.N0);
            p.N1 = fmaf(q1.y, w, p.N1);
            p.N2 = fmaf(q1.z, w, p.N2);
            p.C0 = fmaf(q4.y, w, p.C0);
            p.C1 = fmaf(q4.z, w, p.C1);
            p.C2 = fmaf(q4.w, w, p.C2);
            asm volatile("v_mov_b32 %0, %1" : "+v"(p.Tt) : "v"(test_T));
        }
    }
}

// 7 waves per SIMD (72 VGPRs, three 4-byte spills outside the entry loop; 7 x 20.6 KB of LDS per CU): measured
// 0.502 -> 0.478 ms against the compiler's own choice of 78 VGPRs / 6 waves (profiles/r03_ab_forward_occupancy.txt).
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) blend_fwd_kernel(BlendFwdArgs a) {
    __shared__ float4 s_rec[BLEND_QUADS][FWD_BATCH];
    __shared__ uint32_t s_rel[FWD_BATCH];  // bit q: the entry's alpha-cutoff region can reach quadrant q
    __shared__ unsigned long long s_hit[FWD_BATCH / 64][4];  // [group][quadrant]: entries some pixel blended
    const int tile = (int)a.tile_order[blockIdx.x];
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const int qx = tile_x * TILE + (wv & 1) * 8, qy = tile_y * TILE + (wv >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const bool inside = px < a.W && py < a.H;
    const float pxf = (float)px, pyf = (float)py;
    const size_t N = (size_t)a.W * a.H;
    const size_t pix_id = (size_t)a.W * py + px;

    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int n = (int)(r1 - r0);

    FwdPixel st{};
    st.Tt = inside ? 1.0f : 0.0f;
    uint32_t n_blended = 0;  // wave-uniform: (entry, this quadrant) pairs some pixel blended = the backward's visits
    uint32_t n_entries = 0;  // wave-uniform: staged entries (of this wave's share of every batch) that some pixel blended

    for (int b0 = 0; b0 < n; b0 += FWD_BATCH) {
        // end if the entire tile is saturated (forward.cu:327)
        if (__syncthreads_count(!(st.Tt > 0.0f)) == 256) break;
        const int m = imin_(FWD_BATCH, n - b0);
        if ((int)threadIdx.x < m) {
            const uint64_t e = a.entries[r0 + b0 + threadIdx.x];
            const float4* r = reinterpret_cast<const float4*>(a.rec) + (size_t)entry_idx(e) * REC_QUADS;
            float4 rq[REC_QUADS];
#pragma unroll
            for (int i = 0; i < REC_QUADS; i++) rq[i] = r[i];
#pragma unroll
            for (int i = 0; i < BLEND_QUADS; i++) s_rec[i][threadIdx.x] = rq[i];
            // Which of the tile's four quadrants can this splat reach at all?  Decided ONCE per staged entry (not
            // once per quadrant wave): bounding box first, then the exact region (low-pass disk + cutoff ellipse).
            const uint32_t inbox = quads_in_box(rq[5], 2u * (uint32_t)tile_x, 2u * (uint32_t)tile_y);
            uint32_t relmask = 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int bxi = tile_x * TILE + (q & 1) * 8, byi = tile_y * TILE + (q >> 1) * 8;
                bool rel = (inbox >> q) & 1u;
                if (rel && !a.box_only) rel = !quad_misses_region(rq[0], rq[6], rq[7], (float)bxi, (float)byi);
                relmask |= rel ? (1u << q) : 0u;
            }
            // bit 4: quads 2..4 of this splat's record hold the affine form of the ray-splat intersection (REC_AFFINE)
            if (__float_as_uint(rq[0].w) & REC_AFFINE) relmask |= 16u;
            s_rel[threadIdx.x] = relmask;
        }
        if (threadIdx.x < (FWD_BATCH / 64) * 4) (&s_hit[0][0])[threadIdx.x] = 0ull;
        __syncthreads();
        if (__ballot(st.Tt > 0.0f) != 0ull) {
        // Each group of 64 staged entries is filtered for this wave's quadrant with one bit test per lane + a
        // ballot; only entries whose region touches the quadrant are visited (scalar bit scan), so a rejected
        // entry costs ~1/64 of a loop iteration.
        for (int g0 = 0; g0 < m; g0 += 64) {
            const int jl = g0 + lane;
            const uint32_t rbits = s_rel[jl < FWD_BATCH ? jl : 0];
            const bool rel = jl < m && ((rbits >> wv) & 1u);
            uint64_t todo = __ballot(rel);
            const uint64_t nolp_mask = __ballot((rbits & 16u) != 0);  // wave-uniform, one bit per staged entry
            uint64_t hit = 0;  // wave-uniform: entries of this group blended by some pixel of this quadrant
            const uint32_t contributor0 = (uint32_t)(b0 + g0 + 1);
            uint32_t rec_base = (uint32_t)g0 * 16u;  // byte offset of the group's first record inside a quad's row
            asm volatile("" : "+v"(rec_base));       // (a vector register on purpose, see the visit)
            // One visit: the next staged entry of this group that can reach the quadrant.  (A macro, not a lambda: with `todo`
            // and `hit` captured by reference hipcc keeps copies of them and the loop grows by eight instructions.)
            // The visited bit is cleared with ONE s_andn2_b64 against the mask the hit / nolp tests form anyway; `todo &= todo - 1`
            // is s_add_u32, s_addc_u32, s_and_b64.  Every instruction of this loop, scalar ones included, is ~1.2 % of the kernel
            // (the waves issue at the SIMD's limit for their instruction COUNT, LAB_NOTES section 6): blend_fwd 0.414 -> 0.400 ms
            // (profiles/r06_ab_bitclear.txt).
#define G4S_FWD_VISIT()                                                                                                          \
    do {                                                                                                                         \
        const int bit = (int)__builtin_ctzll(todo);                                                                              \
        todo &= ~(1ull << bit);                                                                                                  \
        const uint32_t contributor = contributor0 + (uint32_t)bit; /* the reference's 1-based list position (forward.cu:349) */  \
        /* the record's LDS address in ONE instruction (v_lshl_add_u32 from the group's base, kept in a VGPR) */                 \
        const float4* rq_ = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(&s_rec[0][0]) + (rec_base + ((uint32_t)bit << 4))); \
        const float4 q0 = rq_[0], q1 = rq_[FWD_BATCH], q2 = rq_[2 * FWD_BATCH], q3 = rq_[3 * FWD_BATCH], q4 = rq_[4 * FWD_BATCH];   \
        const bool nolp = (nolp_mask >> bit) & 1ull; /* scalar */                                                                \
        fwd_visit(st, nolp, pxf, pyf, q0, q1, q2, q3, q4, contributor);                                                          \
        /* blended by some pixel <=> some pixel's last_contributor is this entry */                                             \
        if (__ballot(st.last_contributor == contributor) != 0ull) {                                                              \
            hit |= 1ull << bit;                                                                                                  \
            asm volatile(""); /* keeps this a branch around ONE s_or_b64: as a select it is two s_cselect_b32 + the s_or */        \
        }                                                                                                                        \
    } while (0)
#if G4S_FWD_PAIRS >= 2
            // Several visits per test of "is the quadrant saturated?" (a compare, a scalar test, a select, a mask and a branch
            // per visit otherwise): at worst G4S_FWD_PAIRS - 1 visits run on a quadrant that has just saturated -- they blend
            // nothing (Tt <= 0 in every lane) and change nothing.  Per visit of 1 / 2 / 4 / 8: blend_fwd 0.401 / 0.394 / 0.388 /
            // 0.388 ms (bit-identical; profiles/r06_ab_bitclear.txt).
            while (todo) {
                G4S_FWD_VISIT();
                if (todo == 0ull) break;
                G4S_FWD_VISIT();
#if G4S_FWD_PAIRS >= 4
                if (todo == 0ull) break;
                G4S_FWD_VISIT();
                if (todo == 0ull) break;
                G4S_FWD_VISIT();
#endif
#if G4S_FWD_PAIRS >= 8
                if (todo == 0ull) break;
                G4S_FWD_VISIT();
                if (todo == 0ull) break;
                G4S_FWD_VISIT();
                if (todo == 0ull) break;
                G4S_FWD_VISIT();
                if (todo == 0ull) break;
                G4S_FWD_VISIT();
#endif
                if (__ballot(st.Tt > 0.0f) == 0ull) break;
            }
            const bool live = __ballot(st.Tt > 0.0f) != 0ull;
#else
            bool live = true;  // wave-uniform: some pixel of the quadrant is not saturated yet
            while (todo) {
                G4S_FWD_VISIT();
                live = __ballot(st.Tt > 0.0f) != 0ull;
                if (!live) break;
            }
#endif
#undef G4S_FWD_VISIT
            if (lane == 0 && hit) s_hit[g0 >> 6][wv] = hit;
            n_blended += (uint32_t)__builtin_popcountll(hit);
            if (!live) break;
        }
        }
        // contribution mask for the backward: one byte per staged entry, bit q = quadrant q blended it
        __syncthreads();
        uint32_t nib = 0;
        if ((int)threadIdx.x < m) {
            const int g = (int)threadIdx.x >> 6, b = (int)threadIdx.x & 63;
            nib = (uint32_t)((s_hit[g][0] >> b) & 1ull) | ((uint32_t)((s_hit[g][1] >> b) & 1ull) << 1) |
                  ((uint32_t)((s_hit[g][2] >> b) & 1ull) << 2) | ((uint32_t)((s_hit[g][3] >> b) & 1ull) << 3);
            if (nib) a.qhit[r0 + b0 + threadIdx.x] = (uint8_t)nib;
        }
        n_entries += (uint32_t)__popcll(__ballot(nib != 0));  // (this wave's 64 staged entries)
    }
    // the backward's work in this tile, for its longest-first ordering (tile_order_kernel reads (start, end) pairs)
    __syncthreads();
    // the backward's cost of a tile ~ 111 instructions per visit + 78 per contributing entry (reduction, record): the
    // ordering key is visits + 3/4 entries
    if (lane == 0) (&s_hit[0][0])[wv] = n_blended + ((3u * n_entries) >> 2);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long* v = &s_hit[0][0];
        a.tile_depth[2 * tile] = 0u;
        a.tile_depth[2 * tile + 1] = (uint32_t)((v[0] + v[1]) + (v[2] + v[3]));
    }
    const float T = fabsf(st.Tt);
    if (inside) {
        a.final_T[pix_id] = T;
        a.final_T[pix_id + N] = st.M1;
        a.final_T[pix_id + 2 * N] = st.M2;
        a.n_contrib[pix_id] = st.last_contributor;
        a.n_contrib[pix_id + N] = st.median_contributor;
        a.out_color[pix_id] = st.C0 + T * a.bg[0];
        a.out_color[pix_id + N] = st.C1 + T * a.bg[1];
        a.out_color[pix_id + 2 * N] = st.C2 + T * a.bg[2];
        a.out_others[pix_id + 0 * N] = st.Dd;            // DEPTH_OFFSET
        a.out_others[pix_id + 1 * N] = 1 - T;         // ALPHA_OFFSET
        a.out_others[pix_id + 2 * N] = st.N0;            // NORMAL_OFFSET..+2
        a.out_others[pix_id + 3 * N] = st.N1;
        a.out_others[pix_id + 4 * N] = st.N2;
        a.out_others[pix_id + 5 * N] = st.median_depth;  // MIDDEPTH_OFFSET
        a.out_others[pix_id + 6 * N] = st.distortion;    // DISTORTION_OFFSET
    }
}

void launch_blend_fwd(const BlendFwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(blend_fwd_kernel, dim3(a.tiles_x * a.tiles_y), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------
// K7 backward

constexpr int BWD_BATCH = 64;

struct BwdPixel {
    // constants.  The distortion terms only ever appear multiplied by dL_dreg, so they are kept as
    //   A2 = (1 - T_final) dL_dreg,  D2 = 2 final_D dL_dreg,  C2 = final_D2 dL_dreg
    // (backward.cu:342-359: dL_dweight = m^2 A2 - m D2 + C2,  dL_dmd = w (2 m A2 - D2)),
    // and the background term (backward.cu:384-387) as nTfbg = -T_final <bg, dL_dpix>.
    float A2, D2, C2, nTfbg;
    uint32_t last_c, median_c;
    float dpx0, dpx1, dpx2, dL_ddepth, dL_daccum, dn0, dn1, dn2, dL_dmedian;
    // running state (back to front)
    float T, V_rec, last_dL_dT;
};

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// Occupancy is what this kernel responds to (LAB_NOTES.md): at 158 VGPRs three waves fit a SIMD, at <= 128 four
// do.  Of the 20 per-pixel values a lane carries for each of its four pixels, six are read once per visit (or
// less) and never written: the three distortion constants, the background term, the median position and its
// cotangent.  They are parked in LDS (24 B per pixel, read back with one ds_read_b128 per visit plus one
// ds_read_b64 under the median branch), and the V_rec recurrence is advanced at the end of a visit instead of at the
// start of the next one (two more registers per pixel).  That is 128 VGPRs with one 4-byte spill in the per-batch
// prologue.  LDS then decides the occupancy: 16 single-wave workgroups per CU need <= 10 240 B each, which is why this
// kernel stages 48 list entries per batch (3 840 B + 192 B slots + 6 144 B parked = 10 176 B).  Measured on S3:
// 1.106 -> 1.046 ms; with 64-entry batches (14 workgroups per CU) it was slower than three waves (1.135 ms).
constexpr int BWD1_BATCH = 48;  // list entries staged per batch by the one-wave kernel (see above)
struct BwdPixelLite {  // what stays in registers per pixel; the rest of BwdPixel is parked in LDS
    uint32_t last_c;
    float dpx0, dpx1, dpx2, dL_ddepth, dL_daccum, dn0, dn1, dn2;
    float T, V_rec, last_dL_dT;
};
__device__ __forceinline__ void blend_bwd_tile(const BlendBwdArgs& a) {
    __shared__ float4 s_rec[BLEND_QUADS][BWD1_BATCH];  // the culling quads are not needed here
    __shared__ float4 s_cst[256];   // per pixel (quadrant * 64 + lane): A2, D2, C2, nTfbg
    __shared__ float2 s_med[256];   // per pixel: bits(median_c), dL_dmedian

    const int tile = (int)a.tile_order[blockIdx.x];
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int lane = (int)threadIdx.x;
    const size_t N = (size_t)a.W * a.H;

    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int n = (int)(r1 - r0);
    if (n == 0) return;

    const int tx0 = tile_x * TILE, ty0 = tile_y * TILE;
    const int px0 = tx0 + (lane & 7), py0 = ty0 + (lane >> 3);
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
    BwdPixelLite p[4];
    uint32_t max_last = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int px = px0 + (q & 1) * 8, py = py0 + (q >> 1) * 8;
        BwdPixelLite& x = p[q];
        x = BwdPixelLite{};
        uint32_t median_c = 0; float dL_dmedian = 0;
        float T_final = 0, final_D = 0, final_D2 = 0, dL_dreg = 0;
        if (px < a.W && py < a.H) {
            const size_t pix_id = (size_t)a.W * py + px;
            T_final = a.final_T[pix_id];
            final_D = a.final_T[pix_id + N];
            final_D2 = a.final_T[pix_id + 2 * N];
            x.last_c = a.n_contrib[pix_id];
            median_c = a.n_contrib[pix_id + N];
            x.dpx0 = a.dL_dpix[pix_id];
            x.dpx1 = a.dL_dpix[pix_id + N];
            x.dpx2 = a.dL_dpix[pix_id + 2 * N];
            x.dL_ddepth = a.dL_depths[pix_id + 0 * N];
            x.dL_daccum = a.dL_depths[pix_id + 1 * N];
            x.dn0 = a.dL_depths[pix_id + 2 * N];
            x.dn1 = a.dL_depths[pix_id + 3 * N];
            x.dn2 = a.dL_depths[pix_id + 4 * N];
            dL_dmedian = a.dL_depths[pix_id + 5 * N];
            dL_dreg = a.dL_depths[pix_id + 6 * N];
        }
        s_cst[q * 64 + lane] = make_float4((1 - T_final) * dL_dreg, 2.0f * final_D * dL_dreg, final_D2 * dL_dreg,
                                           -T_final * ((bg0 * x.dpx0 + bg1 * x.dpx1) + bg2 * x.dpx2));
        s_med[q * 64 + lane] = make_float2(__uint_as_float(median_c), dL_dmedian);
        x.T = T_final;
        max_last = max(max_last, x.last_c);  // pixels outside the image keep last_c = 0: never active
    }
    const float mscale = FAR_N / (FAR_N - NEAR_N);
    const float dmd_k = (FAR_N * NEAR_N) / (FAR_N - NEAR_N);  // m = mscale - dmd_k / depth, dm/ddepth = dmd_k / depth^2
    const bool row_writer = (lane & 15) == 15;
    const int row = lane >> 4;
    // the reduction of an entry's sixteen common terms leaves term 4 row + quad_term(lane) in every lane of a quad
    const bool lane_b3 = (lane & 8) != 0, lane_b2 = (lane & 4) != 0, quad_writer = (lane & 3) == 0;
    const int term_of_lane = 4 * row + quad_term(lane);

    // list positions >= the tile's max last_contributor cannot contribute anywhere
    // (readfirstlane: the two maxima are wave-uniform, and what derives from them -- hi, m, pos, the median test --
    // belongs in SGPRs: scalar compares and branches instead of vector compares inside EXEC regions)
    const int n_live = __builtin_amdgcn_readfirstlane((int)wave_max_u32(max_last));
    if (n_live > a.hot_threshold) {  // a deep tile would be this kernel's tail: four waves take it (blend_bwd_hot_kernel)
        if (lane == 0) a.hot_list[atomicAdd(a.hot_count, 1u)] = (uint32_t)tile;
        return;
    }
    // (records of instances that receive no contribution are never written; rec_flag tells the fold which are)

    // batches from the back of the live range; lane t stages list position hi-1-t
    for (int hi = n_live; hi > 0; hi -= BWD1_BATCH) {
        const int m = imin_(BWD1_BATCH, hi);
        __syncthreads();
        // bit q of qmask: some pixel of quadrant q blended the entry this lane stages -- recorded by
        // the forward (qhit).  A pixel is active here iff it blended the entry there (same eval, and
        // every passing entry below last_contributor was blended), so this mask is exact: entries and
        // quadrants without contribution are never touched.
        uint32_t qmask = 0;
        uint32_t slot_l = 0;  // gradient-record slot of the entry this lane stages: taken with one v_readlane per entry
        bool nolp_l = false;
        if (lane < m) {
            const uint32_t pos_l = (uint32_t)(hi - 1 - lane);
            qmask = a.qhit[r0 + pos_l];
        }
        if (qmask != 0) {
            const uint32_t pos_l = (uint32_t)(hi - 1 - lane);
            const uint64_t e = a.entries[r0 + pos_l];
            const float4* r = reinterpret_cast<const float4*>(a.rec) + (size_t)entry_idx(e) * REC_QUADS;
            float4 rq[BLEND_QUADS];
#pragma unroll
            for (int i = 0; i < BLEND_QUADS; i++) rq[i] = r[i];  // all loads in flight before any LDS store
            const float4 q0 = rq[0];
            nolp_l = (__float_as_uint(q0.w) & REC_AFFINE) != 0;
#pragma unroll
            for (int i = 0; i < BLEND_QUADS; i++) s_rec[i][lane] = rq[i];
            // (the rect's origin word sits in q7.w, in the 64-byte line q4 was just read from)
            slot_l = __float_as_uint(q0.z) + instance_number(__float_as_uint(q0.w), __float_as_uint(reinterpret_cast<const float*>(r)[31]),
                                                                  (uint32_t)tile_x, (uint32_t)tile_y);
        }
        __syncthreads();

        uint64_t todo = __ballot(qmask != 0);
        const uint64_t nolp_mask = __ballot(nolp_l);  // wave-uniform: REC_AFFINE entries
        while (todo) {
            const int j = (int)__builtin_ctzll(todo);
            todo &= ~(1ull << j);  // (one s_andn2_b64 instead of the three instructions of `todo & (todo - 1)`, see the forward)
            const uint32_t qm = (uint32_t)__builtin_amdgcn_readlane((int)qmask, j);  // wave-uniform
            // bit j of the mask as a 0 / 1 word in an SGPR, taken by hand: from `(nolp_mask >> j) & 1` hipcc builds a lane
            // mask, turns it into a 0 / 1 VGPR and compares that again (v_cndmask + v_cmp + five scalar instructions per entry)
            uint32_t nolp_s, general;
            asm volatile("s_bitcmp1_b64 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(nolp_s) : "s"(nolp_mask), "s"(j) : "scc");
            // == !nolp, in an SGPR and opaque to the optimiser (see the gradient block)
            asm volatile("s_xor_b32 %0, %1, 1" : "=s"(general) : "s"(nolp_s) : "scc");
            const uint32_t pos = (uint32_t)(hi - 1 - j);  // 0-based list position == backward `contributor`
            const float4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j], q3 = s_rec[3][j], q4 = s_rec[4][j];
            // the accumulators are plain floats zeroed one by one, and the zero is pinned here: left alone, the
            // compiler sinks the initialisation into both arms of the first quadrant's branch and joins them with a
            // 16-deep copy chain (33 moves).  (Nine 64-bit register pairs zeroed with v_mov_b64 looked cheaper but every
            // half had to be copied out again in front of the permlane swaps of the reduction: 9 + 15 moves instead of 18.)
            // record order (g4s_internal.h): gc* colour, gn* normal, gt0..gt8 the T terms (moments S, X, Y of dL/dp' for a
            // REC_AFFINE splat, dL/dTu, dL/dTv, dL/dTw otherwise), gop opacity; glp* = the low-pass branch's centre terms.
            // Separate scalars, not an array: an array handed to the reduction by reference becomes one 512-bit register
            // tuple, and every arm of the branches below then ends in a copy of all sixteen registers.
            float gc0 = 0.0f, gc1 = 0.0f, gc2 = 0.0f, gn0 = 0.0f, gn1 = 0.0f, gn2 = 0.0f, gt0 = 0.0f, gt1 = 0.0f, gt2 = 0.0f, gt3 = 0.0f,
                  gt4 = 0.0f, gt5 = 0.0f, gt6 = 0.0f, gt7 = 0.0f, gt8 = 0.0f, gop = 0.0f, glp0 = 0.0f, glp1 = 0.0f;
            asm volatile("" : "+v"(gc0), "+v"(gc1), "+v"(gc2), "+v"(gn0), "+v"(gn1), "+v"(gn2));
            asm volatile("" : "+v"(gt0), "+v"(gt1), "+v"(gt2), "+v"(gt3), "+v"(gt4), "+v"(gt5), "+v"(gt6), "+v"(gt7), "+v"(gt8));
            asm volatile("" : "+v"(gop), "+v"(glp0), "+v"(glp1));
            // "some pixel took the low-pass branch" as a REGISTER a pixel writes when it does (non-affine entries only), not as a
            // bool: a bool set in a divergent arm and carried across the four visits is a lane mask that every visit merges
            // (s_mov, s_andn2, s_and, s_or -- four scalar instructions on the common path for a flag only the rare path sets)
            float lowpass = 0.0f;
            asm volatile("" : "+v"(lowpass));
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (!((qm >> q) & 1u)) continue;  // scalar branch
                BwdPixelLite& x = p[q];
                const float4 cst = s_cst[q * 64 + lane];
                // (the median position and its cotangent: read with cst on every visit and SELECTED below.  A scalar branch
                // around the read for entries behind the tile's deepest median was three instructions when not taken and
                // eight when taken -- half the visits --, the read + compare + select are four: blend_bwd 0.773 -> 0.762 ms)
                const float2 md = s_med[q * 64 + lane];
                const float pxf = (float)(px0 + (q & 1) * 8), pyf = (float)(py0 + (q >> 1) * 8);
                PairEval e;
                // eval_pair runs on all 64 lanes, not under `pos < last_c`: a VALU instruction costs the same whatever
                // EXEC says, and the mask region around it (s_and_saveexec, branch, restore) is pure overhead in a
                // kernel whose waves are short of issue slots: -3.2 % (profiles/r04_ab_blend_bwd.txt)
                // (the flag re-read behind an empty asm at every use: left to itself hipcc forms ONE `nolp_s == 0` for the
                // four visits, keeps it as a 0 / 1 VGPR across their branches and compares that again in front of each)
                asm volatile("" : "+s"(nolp_s));
                const bool pass = eval_pair(nolp_s, pxf, pyf, q0.x, q0.y, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q1.w, e);
                const bool act = pass && pos < x.last_c;
                asm volatile("" : "+s"(nolp_s));  // (again, for the test in the gradient block)
                // (qhit is exact, so some pixel of the quadrant is active; no wave vote needed)
                if (act) {
                    const float G = e.G, alpha = e.alpha, c_d = e.depth;
                    const float inv1ma = fast_rcp(1.f - alpha);
                    x.T = x.T * inv1ma;  // T / (1 - alpha), backward.cu:316
                    const float T = x.T;
                    const float w = alpha * T;
                    // The colour / depth / alpha / normal "accum_rec" recurrences (backward.cu:328,362-371)
                    // share their coefficients, so they fold into one scalar recurrence on
                    // v = <c,dL_dpix> + c_d*dL_ddepth + dL_daccum + <n,dL_dnormal>.
                    const float v = fmaf(q4.y, x.dpx0, fmaf(q4.z, x.dpx1, q4.w * x.dpx2)) +
                                    fmaf(c_d, x.dL_ddepth, x.dL_daccum) +
                                    fmaf(q1.x, x.dn0, fmaf(q1.y, x.dn1, q1.z * x.dn2));
                    // V_rec <- a v + (1 - a) V_rec is applied right here, for the next (shallower) entry: the same
                    // operation on the same operands as the reference's deferred update (backward.cu:328), minus the
                    // two registers per pixel that would carry last_alpha / last_v across the visit
                    const float v_minus_rec = v - x.V_rec;
                    float dL_dalpha = v_minus_rec;
                    x.V_rec = fmaf(alpha, v_minus_rec, x.V_rec);
                    gc0 = fmaf(w, x.dpx0, gc0);
                    gc1 = fmaf(w, x.dpx1, gc1);
                    gc2 = fmaf(w, x.dpx2, gc2);
                    gn0 = fmaf(w, x.dn0, gn0);
                    gn1 = fmaf(w, x.dn1, gn1);
                    gn2 = fmaf(w, x.dn2, gn2);

                    const float inv_cd = e.inv_depth;  // = p'.z on the fast path: no reciprocal
                    const float m_d = fmaf(-dmd_k, inv_cd, mscale);
                    const float dmd_dd = dmd_k * inv_cd * inv_cd;
                    float dL_dz = (pos + 1 == __float_as_uint(md.x)) ? md.y : 0.0f;
                    const float dL_dweight = fmaf(m_d, fmaf(m_d, cst.x, -cst.y), cst.z);
                    const float dwt = dL_dweight - x.last_dL_dT;
                    dL_dalpha += dwt;
                    x.last_dL_dT = fmaf(alpha, dwt, x.last_dL_dT);  // a dL_dweight + (1 - a) last_dL_dT
                    const float dL_dmd = w * fmaf(m_d + m_d, cst.x, -cst.y);
                    dL_dz = fmaf(dL_dmd, dmd_dd, dL_dz);

                    dL_dalpha *= T;
                    dL_dalpha = fmaf(cst.w, inv1ma, dL_dalpha);
                    const float dL_dG = q1.w * dL_dalpha;  // not gated by the 0.99 clamp (backward.cu:390)
                    dL_dz = fmaf(w, x.dL_ddepth, dL_dz);

                    if (nolp_s) {  // scalar branch: REC_AFFINE splat (always in3d)
                        // backward.cu:396-426 in the affine form: G = exp(-|s|^2 / 2), s = p'.xy / p'.z, depth = 1 / p'.z
                        //   dL/dp'.xy = dL/ds / p'.z,   dL/dp'.z = -(s . dL/ds + depth dL/ddepth) / p'.z
                        // and the only thing accumulated for T are the moments S, X, Y of dL/dp' (the reference's dL/dk =
                        // l x dL/dp, dL/dl = dL/dp x k and its 18 accumulations per pixel are linear in the pixel: K8 takes
                        // the cross products once per Gaussian)
                        const float mGi = (dL_dG * -G) * e.depth;  // (e.depth: 1 / p'.z, see eval_pair)
                        const float dpx_ = mGi * e.sx, dpy_ = mGi * e.sy;
                        const float ndpz = fmaf(dpx_, e.sx, fmaf(dpy_, e.sy, (dL_dz * e.depth) * e.depth));  // -dL/dp'.z (depth = 1 / p'.z)
                        acc_add(gt0, dpx_);  // (in place: see acc_fma)
                        acc_add(gt1, dpy_);
                        acc_sub(gt2, ndpz);
                        acc_fma(gt3, e.dx, dpx_);
                        acc_fma(gt4, e.dx, dpy_);
                        acc_fnma(gt5, e.dx, ndpz);
                        acc_fma(gt6, e.dy, dpx_);
                        acc_fma(gt7, e.dy, dpy_);
                        acc_fnma(gt8, e.dy, ndpz);
                    }
                    // (two regions in a row on two conditions the compiler cannot relate, not if / else: under one uniform
                    // test the divergent branch below makes hipcc structurize the whole nest with flag registers and copy
                    // the accumulators in and out of the arm above)
                    if (general) {
                    if (e.in3d) {  // general path, backward.cu:396-426 as written: gt0..gt8 = dL/dTu, dL/dTv, dL/dTw
                        const float mG = dL_dG * -G;
                        const float dL_dsx = fmaf(mG, e.sx, dL_dz * q3.z);
                        const float dL_dsy = fmaf(mG, e.sy, dL_dz * q3.w);
                        const float inv_pz = e.inv_pz;  // the v_rcp_f32 of eval_pair
                        const float dpx_ = dL_dsx * inv_pz, dpy_ = dL_dsy * inv_pz;
                        const float dpz_ = -fmaf(dpx_, e.sx, dpy_ * e.sy);
                        // dL_dk = cross(l, dL_dp), dL_dl = cross(dL_dp, k); k and l are formed again here (bit-identical to
                        // eval_pair's) rather than carried across the visit: six registers the affine path never needs
                        const float kx = fmaf(pxf, q3.z, -q2.x), ky = fmaf(pxf, q3.w, -q2.y), kz = fmaf(pxf, q4.x, -q2.z);
                        const float lx = fmaf(pyf, q3.z, -q2.w), ly = fmaf(pyf, q3.w, -q3.x), lz = fmaf(pyf, q4.x, -q3.y);
                        const float dkx = fmaf(ly, dpz_, -(lz * dpy_)), dky = fmaf(lz, dpx_, -(lx * dpz_)),
                                    dkz = fmaf(lx, dpy_, -(ly * dpx_));
                        const float dlx = fmaf(dpy_, kz, -(dpz_ * ky)), dly = fmaf(dpz_, kx, -(dpx_ * kz)),
                                    dlz = fmaf(dpx_, ky, -(dpy_ * kx));
                        gt0 -= dkx;
                        gt1 -= dky;
                        gt2 -= dkz;
                        gt3 -= dlx;
                        gt4 -= dly;
                        gt5 -= dlz;
                        gt6 += fmaf(pxf, dkx, fmaf(pyf, dlx, dL_dz * e.sx));
                        gt7 += fmaf(pxf, dky, fmaf(pyf, dly, dL_dz * e.sy));
                        gt8 += fmaf(pxf, dkz, fmaf(pyf, dlz, dL_dz));
                    } else {
                        // backward.cu:427-434 (d = centre - pixel = -e.dx)
                        const float c2 = dL_dG * (G * FILTER_INV_SQUARE);
                        glp0 = fmaf(c2, e.dx, glp0);
                        glp1 = fmaf(c2, e.dy, glp1);
                        gt8 += dL_dz;  // depth = Tw.z here, and on the general path term 14 is dL/dTw.z
                        asm volatile("v_mov_b32 %0, 1.0" : "+v"(lowpass));
                    }
                    }
                    gop = fmaf(G, dL_dalpha, gop);
                }
            }
            // 256 pixels -> 1: the four pixels of a lane were summed in registers above; the 64 lanes are
            // summed sixteen terms at a time, after which row k holds terms 4k..4k+3 and its last lane writes
            // them with one 16-byte store.
            // Record layout (g4s_internal.h): floats 0..14 = colour, normal, the moments S / X / Y of dL/dp', 15 = opacity
            // term, 16..17 = the low-pass centre terms -- those are non-zero only when some pixel took the 2-D filter
            // branch (rare for splats wider than a pixel), so their group is reduced and stored only then.  The record
            // buffer is not cleared: rec_flag[slot] (pre-cleared, one byte) says which parts are valid.
            {
                const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)slot_l, j);
                float t[16] = {gc0, gc1, gc2, gn0, gn1, gn2, gt0, gt1, gt2, gt3, gt4, gt5, gt6, gt7, gt8, gop};
                const float y = wave_sum16_to_quads(t, lane_b3, lane_b2);
                // ONE uniform branch around the stores (false only in a frame that overflowed its presized capacity) instead of
                // the test ANDed into each store's lane mask; and the validity byte goes out with the sixteen terms, from the
                // same sixteen lanes (one address, one value), not from lane 0 in an EXEC region of its own
                if (slot < a.n_slots) {
                    float* rec = a.grad_inst + (size_t)slot * GRAD_STRIDE;
                    // the validity byte as a word in an SGPR, by hand: 1, or 3 when some pixel took the low-pass branch (as a
                    // bool hipcc makes two lane masks of it and selects the byte from one of them inside the store's region)
                    const uint64_t lp_lanes = __ballot(lowpass != 0.0f);
                    uint32_t valid;
                    asm volatile("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, 3, 1" : "=s"(valid) : "s"(lp_lanes) : "scc");
                    if (quad_writer) {
                        rec[term_of_lane] = y;  // sixteen lanes, sixteen consecutive floats
                        a.rec_flag[slot] = (uint8_t)valid;
                    }
                    if (valid & 2u) {
                        const float r4 = wave_sum4_to_rows(glp0, glp1, 0.0f, 0.0f);
                        if (row_writer) rec[16 + row] = r4;
                    }
                }
            }
        }
    }
}

// The kernel proper: the tile, then this workgroup's share of the zero-fill (BlendBwdArgs::zero_*).  The tile work is
// VALU-bound and HBM is nearly idle under it; the stores of a finished wave drain while the others compute, which
// takes 207 MB of zero rows (72 % of dL_dsh at the metric size) out of the HBM-bound K8 behind this kernel.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) blend_bwd_kernel(BlendBwdArgs a) {
    blend_bwd_tile(a);
    const uint32_t lane = threadIdx.x, first = blockIdx.x * 64 + lane, stride = gridDim.x * 64;
#pragma unroll
    for (int z = 0; z < 2; z++) {
        float4* b = reinterpret_cast<float4*>(a.zero_base[z]);
        if (b == nullptr) continue;
        const uint32_t nq = a.zero_quads[z];
        for (uint32_t i = first; i < nq; i += stride) b[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (blockIdx.x == 0 && lane < a.zero_tail[z]) reinterpret_cast<float*>(b + nq)[lane] = 0.0f;
    }
}

// ---------------------------------------------------------------------------------------
// K7 backward for deep tiles: four waves per tile, one 8x8 quadrant (one pixel per lane) each.
//
// The one-wave kernel above is the efficient form (the lane's four pixels share one reduction per entry) but its
// time per tile grows with the list: a tile whose last contributor sits at position 10^4 keeps a single wave busy
// for milliseconds while the rest of the GPU idles.  Such tiles are collected in hot_list and handled here with
// four times the lanes: every wave walks the same staged batch for its own quadrant (qhit bit), reduces its 64
// pixels with the same permlane tree, and parks the partial sums in LDS; after the batch the four partials of an
// entry are added in a fixed order (quadrant 0..3) and stored as the entry's gradient record.  Deterministic; the
// sums are associated differently from the one-wave kernel's, so the two agree to rounding, not bit for bit.
constexpr int HOT_PART = 20;  // floats per partial: 16 common terms, 2 low-pass terms, 2 unused

__global__ void __launch_bounds__(256) blend_bwd_hot_kernel(BlendBwdArgs a) {
    __shared__ float4 s_rec[BLEND_QUADS][BWD_BATCH];
    __shared__ uint32_t s_slot[BWD_BATCH];
    __shared__ uint32_t s_q[BWD_BATCH];      // bits 0..3 qhit, bit 4 = low-pass exponent never matters
    __shared__ float s_part[4][BWD_BATCH][HOT_PART];
    __shared__ unsigned long long s_lp[4];   // per wave: entries of the batch whose low-pass terms it wrote
    __shared__ uint32_t s_max[2][4];

    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    const size_t N = (size_t)a.W * a.H;
    // hot_threshold < 0: every tile of the frame is handled here, in tile_order (the one-wave kernel is not launched)
    const bool all_tiles = a.hot_threshold < 0;
    const uint32_t n_hot = all_tiles ? (uint32_t)(a.tiles_x * a.tiles_y) : *a.hot_count;
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
    const float mscale = FAR_N / (FAR_N - NEAR_N);
    const float dmd_k = (FAR_N * NEAR_N) / (FAR_N - NEAR_N);
    const bool row_writer = (lane & 15) == 15;
    const int row = lane >> 4;

    for (uint32_t h = blockIdx.x; h < n_hot; h += gridDim.x) {
        const int tile = (int)(all_tiles ? a.tile_order[h] : a.hot_list[h]);
        const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
        const uint32_t r0 = a.ranges[2 * tile];
        const int px = tile_x * TILE + (wv & 1) * 8 + (lane & 7), py = tile_y * TILE + (wv >> 1) * 8 + (lane >> 3);
        const float pxf = (float)px, pyf = (float)py;
        BwdPixel x{};
        {
            float T_final = 0, final_D = 0, final_D2 = 0, dL_dreg = 0;
            if (px < a.W && py < a.H) {
                const size_t pix_id = (size_t)a.W * py + px;
                T_final = a.final_T[pix_id];
                final_D = a.final_T[pix_id + N];
                final_D2 = a.final_T[pix_id + 2 * N];
                x.last_c = a.n_contrib[pix_id];
                x.median_c = a.n_contrib[pix_id + N];
                x.dpx0 = a.dL_dpix[pix_id];
                x.dpx1 = a.dL_dpix[pix_id + N];
                x.dpx2 = a.dL_dpix[pix_id + 2 * N];
                x.dL_ddepth = a.dL_depths[pix_id + 0 * N];
                x.dL_daccum = a.dL_depths[pix_id + 1 * N];
                x.dn0 = a.dL_depths[pix_id + 2 * N];
                x.dn1 = a.dL_depths[pix_id + 3 * N];
                x.dn2 = a.dL_depths[pix_id + 4 * N];
                x.dL_dmedian = a.dL_depths[pix_id + 5 * N];
                dL_dreg = a.dL_depths[pix_id + 6 * N];
            }
            x.A2 = (1 - T_final) * dL_dreg;
            x.D2 = 2.0f * final_D * dL_dreg;
            x.C2 = final_D2 * dL_dreg;
            x.nTfbg = -T_final * ((bg0 * x.dpx0 + bg1 * x.dpx1) + bg2 * x.dpx2);
            x.T = T_final;
        }
        __syncthreads();  // the previous tile's last combine has read s_max / s_part
        {
            const uint32_t ml = wave_max_u32(x.last_c), mm = wave_max_u32(x.median_c);
            if (lane == 0) { s_max[0][wv] = ml; s_max[1][wv] = mm; }
        }
        __syncthreads();
        const int n_live = (int)max(max(s_max[0][0], s_max[0][1]), max(s_max[0][2], s_max[0][3]));
        const uint32_t tile_max_median = max(max(s_max[1][0], s_max[1][1]), max(s_max[1][2], s_max[1][3]));

        for (int hi = n_live; hi > 0; hi -= BWD_BATCH) {
            const int m = imin_(BWD_BATCH, hi);
            __syncthreads();  // the previous batch's combine is done with the LDS buffers
            if (wv == 0) {    // wave 0 stages list positions hi-1 .. hi-m (lane t <-> position hi-1-t)
                uint32_t qmask = 0;
                if (lane < m) qmask = a.qhit[r0 + (uint32_t)(hi - 1 - lane)];
                if (qmask != 0) {
                    const uint64_t e = a.entries[r0 + (uint32_t)(hi - 1 - lane)];
                    const float4* r = reinterpret_cast<const float4*>(a.rec) + (size_t)entry_idx(e) * REC_QUADS;
                    float4 rq[BLEND_QUADS];
#pragma unroll
                    for (int i = 0; i < BLEND_QUADS; i++) rq[i] = r[i];
                    if (__float_as_uint(rq[0].w) & REC_AFFINE) qmask |= 16u;
#pragma unroll
                    for (int i = 0; i < BLEND_QUADS; i++) s_rec[i][lane] = rq[i];
                    s_slot[lane] = __float_as_uint(rq[0].z) + instance_number(__float_as_uint(rq[0].w), __float_as_uint(reinterpret_cast<const float*>(r)[31]),
                                                                              (uint32_t)tile_x, (uint32_t)tile_y);
                }
                s_q[lane] = qmask;
            }
            __syncthreads();

            const uint32_t qbits = s_q[lane];
            uint64_t todo = __ballot(((qbits >> wv) & 1u) != 0);
            const uint64_t nolp_mask = __ballot((qbits & 16u) != 0);
            uint64_t lp_mask = 0;  // wave-uniform
            while (todo) {
                const int j = (int)__builtin_ctzll(todo);
                todo &= ~(1ull << j);
                const bool nolp = (nolp_mask >> j) & 1ull;
                const uint32_t general = (uint32_t)__builtin_amdgcn_readfirstlane(nolp ? 0 : 1);  // (see the one-wave kernel)
                const uint32_t pos = (uint32_t)(hi - 1 - j);
                const float4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j], q3 = s_rec[3][j], q4 = s_rec[4][j];
                float g[18];  // record order: 0..15 the common terms, 16..17 the low-pass centre terms
#pragma unroll
                for (int i = 0; i < 18; i++) {
                    g[i] = 0.0f;
                    asm volatile("" : "+v"(g[i]));  // pinned, and accumulated into below (see the one-wave kernel)
                }
                bool lowpass = false;
                PairEval e;
                const bool pass = eval_pair(nolp, pxf, pyf, q0.x, q0.y, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q1.w, e);
                const bool act = pass && pos < x.last_c;
                if (act) {  // the same arithmetic, in the same order, as the one-wave kernel's visit
                    const float G = e.G, alpha = e.alpha, c_d = e.depth;
                    const float inv1ma = fast_rcp(1.f - alpha);
                    x.T = x.T * inv1ma;
                    const float T = x.T;
                    const float w = alpha * T;
                    const float v = fmaf(q4.y, x.dpx0, fmaf(q4.z, x.dpx1, q4.w * x.dpx2)) +
                                    fmaf(c_d, x.dL_ddepth, x.dL_daccum) +
                                    fmaf(q1.x, x.dn0, fmaf(q1.y, x.dn1, q1.z * x.dn2));
                    const float v_minus_rec = v - x.V_rec;
                    float dL_dalpha = v_minus_rec;
                    x.V_rec = fmaf(alpha, v_minus_rec, x.V_rec);
                    g[0] = fmaf(w, x.dpx0, g[0]); g[1] = fmaf(w, x.dpx1, g[1]); g[2] = fmaf(w, x.dpx2, g[2]);
                    g[3] = fmaf(w, x.dn0, g[3]); g[4] = fmaf(w, x.dn1, g[4]); g[5] = fmaf(w, x.dn2, g[5]);
                    const float inv_cd = e.inv_depth;
                    const float m_d = fmaf(-dmd_k, inv_cd, mscale);
                    const float dmd_dd = dmd_k * inv_cd * inv_cd;
                    float dL_dz = 0.0f;
                    if (pos < tile_max_median) dL_dz = (pos + 1 == x.median_c) ? x.dL_dmedian : 0.0f;
                    const float dL_dweight = fmaf(m_d, fmaf(m_d, x.A2, -x.D2), x.C2);
                    const float dwt = dL_dweight - x.last_dL_dT;
                    dL_dalpha += dwt;
                    x.last_dL_dT = fmaf(alpha, dwt, x.last_dL_dT);
                    const float dL_dmd = w * fmaf(m_d + m_d, x.A2, -x.D2);
                    dL_dz = fmaf(dL_dmd, dmd_dd, dL_dz);
                    dL_dalpha *= T;
                    dL_dalpha = fmaf(x.nTfbg, inv1ma, dL_dalpha);
                    const float dL_dG = q1.w * dL_dalpha;
                    dL_dz = fmaf(w, x.dL_ddepth, dL_dz);
                    if (nolp) {
                        const float mGi = (dL_dG * -G) * e.depth;  // (e.depth: 1 / p'.z, see eval_pair)
                        const float dpx_ = mGi * e.sx, dpy_ = mGi * e.sy;
                        const float ndpz = fmaf(dpx_, e.sx, fmaf(dpy_, e.sy, (dL_dz * e.depth) * e.depth));
                        acc_add(g[6], dpx_); acc_add(g[7], dpy_); acc_sub(g[8], ndpz);
                        acc_fma(g[9], e.dx, dpx_); acc_fma(g[10], e.dx, dpy_); acc_fnma(g[11], e.dx, ndpz);
                        acc_fma(g[12], e.dy, dpx_); acc_fma(g[13], e.dy, dpy_); acc_fnma(g[14], e.dy, ndpz);
                    }
                    if (general) {
                    if (e.in3d) {
                        const float mG = dL_dG * -G;
                        const float dL_dsx = fmaf(mG, e.sx, dL_dz * q3.z);
                        const float dL_dsy = fmaf(mG, e.sy, dL_dz * q3.w);
                        const float inv_pz = e.inv_pz;
                        const float dpx_ = dL_dsx * inv_pz, dpy_ = dL_dsy * inv_pz;
                        const float dpz_ = -fmaf(dpx_, e.sx, dpy_ * e.sy);
                        const float kx = fmaf(pxf, q3.z, -q2.x), ky = fmaf(pxf, q3.w, -q2.y), kz = fmaf(pxf, q4.x, -q2.z);
                        const float lx = fmaf(pyf, q3.z, -q2.w), ly = fmaf(pyf, q3.w, -q3.x), lz = fmaf(pyf, q4.x, -q3.y);
                        const float dkx = fmaf(ly, dpz_, -(lz * dpy_)), dky = fmaf(lz, dpx_, -(lx * dpz_)),
                                    dkz = fmaf(lx, dpy_, -(ly * dpx_));
                        const float dlx = fmaf(dpy_, kz, -(dpz_ * ky)), dly = fmaf(dpz_, kx, -(dpx_ * kz)),
                                    dlz = fmaf(dpx_, ky, -(dpy_ * kx));
                        g[6] -= dkx; g[7] -= dky; g[8] -= dkz;
                        g[9] -= dlx; g[10] -= dly; g[11] -= dlz;
                        g[12] += fmaf(pxf, dkx, fmaf(pyf, dlx, dL_dz * e.sx));
                        g[13] += fmaf(pxf, dky, fmaf(pyf, dly, dL_dz * e.sy));
                        g[14] += fmaf(pxf, dkz, fmaf(pyf, dlz, dL_dz));
                    } else {
                        const float c2 = dL_dG * (G * FILTER_INV_SQUARE);
                        g[16] = fmaf(c2, e.dx, g[16]);
                        g[17] = fmaf(c2, e.dy, g[17]);
                        g[14] += dL_dz;
                        lowpass = true;
                    }
                    }
                    g[15] = fmaf(G, dL_dalpha, g[15]);
                }
                float t[16] = {g[0], g[1], g[2],  g[3],  g[4],  g[5],  g[6],  g[7],
                               g[8], g[9], g[10], g[11], g[12], g[13], g[14], g[15]};
                float r[4];
                wave_sum16_to_rows(t, r);
                float* part = &s_part[wv][j][0];
                if (row_writer) *reinterpret_cast<float4*>(part + 4 * row) = make_float4(r[0], r[1], r[2], r[3]);
                if (__any(lowpass)) {
                    const float r4 = wave_sum4_to_rows(g[16], g[17], 0.0f, 0.0f);
                    if (row_writer && row < 2) part[16 + row] = r4;
                    lp_mask |= 1ull << j;
                }
            }
            if (lane == 0) s_lp[wv] = lp_mask;
            __syncthreads();

            // combine: entry e, quad c (five quads per record); partials added in quadrant order
            for (int it = (int)threadIdx.x; it < BWD_BATCH * 5; it += 256) {
                const int e = it / 5, c = it - 5 * e;
                const uint32_t qm = s_q[e] & 15u;
                if (qm == 0) continue;
                const uint32_t slot = s_slot[e];
                if (slot >= a.n_slots) continue;  // (a frame that overflowed its presized capacity, see n_slots)
                float* rec = a.grad_inst + (size_t)slot * GRAD_STRIDE;
                if (c < 4) {
                    float4 acc = make_float4(0, 0, 0, 0);
                    bool first = true;
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        if (!((qm >> w) & 1u)) continue;
                        const float4 v = *reinterpret_cast<const float4*>(&s_part[w][e][4 * c]);
                        if (first) { acc = v; first = false; }
                        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
                    }
                    *reinterpret_cast<float4*>(rec + 4 * c) = acc;
                } else {
                    float ax = 0.0f, ay = 0.0f;
                    bool any_lp = false;
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        if (((qm >> w) & 1u) && ((s_lp[w] >> e) & 1ull)) {
                            ax = any_lp ? ax + s_part[w][e][16] : s_part[w][e][16];
                            ay = any_lp ? ay + s_part[w][e][17] : s_part[w][e][17];
                            any_lp = true;
                        }
                    }
                    if (any_lp) { rec[16] = ax; rec[17] = ay; }
                    a.rec_flag[slot] = any_lp ? 3 : 1;
                }
            }
        }
    }
}

void launch_blend_bwd(const BlendBwdArgs& a, hipStream_t s) {
    const int tiles = a.tiles_x * a.tiles_y;
    // (up to four resident workgroups per CU; each loops over its share of the list)
    const dim3 hot_grid(tiles < 1024 ? tiles : 1024);
    if (a.hot_threshold < 0) {  // small frame: four waves per tile for all of them
        hipLaunchKernelGGL(blend_bwd_hot_kernel, hot_grid, dim3(256), 0, s, a);
        return;
    }
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(tiles), dim3(64), 0, s, a);
    // deep tiles the kernel above handed over (none in ordinary frames: the workgroups read hot_count and leave)
    hipLaunchKernelGGL(blend_bwd_hot_kernel, hot_grid, dim3(256), 0, s, a);
}

}  // namespace g4s
