// Front-to-back alpha blending (K6) and its backward (K7): one 16x16 tile per 256-thread
// workgroup = four wave64s, each wave owning an 8x8 pixel quadrant.
//
// Reference semantics: dsr/cuda_rasterizer/forward.cu:258-443, backward.cu:143-440.
//
// MI355X design:
//   * the tile's sorted instance list is staged through LDS in batches of 256 splat records
//     (80 B each, fetched as five 16-byte quads per thread, stored SoA so that the inner loop
//     reads them with conflict-free broadcast ds_read_b128);
//   * wave64 quadrants (8x8) instead of 16x2 warp strips: a quadrant stops as soon as its 64
//     pixels are saturated / skips a splat as soon as no lane passes the alpha test;
//   * backward: no global float atomics.  The 18 per-(pixel,splat) gradient terms are summed
//     across the 64 lanes with DPP row operations, across the 4 waves in LDS, and stored as
//     ONE 72-byte record per (tile, Gaussian) instance at a slot reserved for that Gaussian
//     (inst_off + k).  The per-Gaussian kernel (preprocess.hip, K8) folds a Gaussian's
//     contiguous records in a fixed order => bit-reproducible gradients.
#include "g4s_internal.h"
#include "g4s_device.h"

namespace g4s {

constexpr int BATCH = 256;

struct PixelCoord {
    int px, py;
    bool inside;
};
__device__ __forceinline__ PixelCoord pixel_of_thread(int tile_x, int tile_y, int W, int H) {
    const int w = (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63);
    PixelCoord c;
    c.px = tile_x * TILE + (w & 1) * 8 + (l & 7);
    c.py = tile_y * TILE + (w >> 1) * 8 + (l >> 3);
    c.inside = c.px < W && c.py < H;
    return c;
}

// Stage one batch of splat records: thread t fetches list entry `pos` (if valid).
__device__ __forceinline__ void stage_record(const uint64_t* __restrict__ entries, const float* __restrict__ rec,
                                             uint32_t pos, bool valid, float4 (*s_rec)[BATCH], uint32_t* s_slot) {
    const int t = (int)threadIdx.x;
    if (valid) {
        const uint64_t e = entries[pos];
        const uint32_t idx = entry_idx(e);
        const float4* r = reinterpret_cast<const float4*>(rec) + (size_t)idx * 5;
        const float4 q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3], q4 = r[4];
        s_rec[0][t] = q0;
        s_rec[1][t] = q1;
        s_rec[2][t] = q2;
        s_rec[3][t] = q3;
        s_rec[4][t] = q4;
        if (s_slot) s_slot[t] = __float_as_uint(q0.z) + entry_k(e);  // inst_off + k
    }
}

// ---------------------------------------------------------------------------------------
// K6 forward

__global__ void __launch_bounds__(256) blend_fwd_kernel(BlendFwdArgs a) {
    __shared__ float4 s_rec[5][BATCH];
    const int tile = (int)blockIdx.x;
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const PixelCoord pc = pixel_of_thread(tile_x, tile_y, a.W, a.H);
    const float pxf = (float)pc.px, pyf = (float)pc.py;
    const size_t N = (size_t)a.W * a.H;
    const size_t pix_id = (size_t)a.W * pc.py + pc.px;

    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int n = (int)(r1 - r0);
    bool done = !pc.inside;

    float T = 1.0f;
    uint32_t contributor = 0, last_contributor = 0, median_contributor = 0;
    float C0 = 0, C1 = 0, C2 = 0, N0 = 0, N1 = 0, N2 = 0;
    float Dd = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
    const float mscale = FAR_N / (FAR_N - NEAR_N);

    for (int b0 = 0; b0 < n; b0 += BATCH) {
        // end if the entire tile is saturated (forward.cu:327)
        if (__syncthreads_count(done) == 256) break;
        const int m = imin_(BATCH, n - b0);
        stage_record(a.entries, a.rec, r0 + b0 + threadIdx.x, (int)threadIdx.x < m, s_rec, nullptr);
        __syncthreads();
        if (__all(done)) continue;  // this quadrant is finished; keep taking part in the staging
        for (int j = 0; j < m; j++) {
            if (done) {
                if (__all(done)) break;
                continue;
            }
            contributor = (uint32_t)(b0 + j + 1);
            const float4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j], q3 = s_rec[3][j], q4 = s_rec[4][j];
            PairEval e;
            if (!eval_pair(pxf, pyf, q0.x, q0.y, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q1.w, e))
                continue;
            const float alpha = e.alpha, depth = e.depth;
            const float test_T = T * (1 - alpha);
            if (test_T < 0.0001f) {
                done = true;
                continue;
            }
            const float w = alpha * T;
            const float A = 1 - T;
            const float md = mscale * (1 - NEAR_N / depth);
            distortion += (md * md * A + M2 - 2 * md * M1) * w;
            Dd += depth * w;
            M1 += md * w;
            M2 += md * md * w;
            if (T > 0.5f) {
                median_depth = depth;
                median_contributor = contributor;
            }
            N0 += q1.x * w;
            N1 += q1.y * w;
            N2 += q1.z * w;
            C0 += q4.y * w;
            C1 += q4.z * w;
            C2 += q4.w * w;
            T = test_T;
            last_contributor = contributor;
        }
    }
    if (pc.inside) {
        a.final_T[pix_id] = T;
        a.final_T[pix_id + N] = M1;
        a.final_T[pix_id + 2 * N] = M2;
        a.n_contrib[pix_id] = last_contributor;
        a.n_contrib[pix_id + N] = median_contributor;
        a.out_color[pix_id] = C0 + T * a.bg[0];
        a.out_color[pix_id + N] = C1 + T * a.bg[1];
        a.out_color[pix_id + 2 * N] = C2 + T * a.bg[2];
        a.out_others[pix_id + 0 * N] = Dd;          // DEPTH_OFFSET
        a.out_others[pix_id + 1 * N] = 1 - T;       // ALPHA_OFFSET
        a.out_others[pix_id + 2 * N] = N0;          // NORMAL_OFFSET..+2
        a.out_others[pix_id + 3 * N] = N1;
        a.out_others[pix_id + 4 * N] = N2;
        a.out_others[pix_id + 5 * N] = median_depth;  // MIDDEPTH_OFFSET
        a.out_others[pix_id + 6 * N] = distortion;    // DISTORTION_OFFSET
    }
}

void launch_blend_fwd(const BlendFwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(blend_fwd_kernel, dim3(a.tiles_x * a.tiles_y), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------
// K7 backward

__global__ void __launch_bounds__(256) blend_bwd_kernel(BlendBwdArgs a) {
    __shared__ float4 s_rec[5][BATCH];
    __shared__ uint32_t s_slot[BATCH];
    __shared__ float s_grad[BATCH * GRAD_FLOATS];
    __shared__ uint32_t s_maxc[4];

    const int tile = (int)blockIdx.x;
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const PixelCoord pc = pixel_of_thread(tile_x, tile_y, a.W, a.H);
    const float pxf = (float)pc.px, pyf = (float)pc.py;
    const size_t N = (size_t)a.W * a.H;
    const size_t pix_id = (size_t)a.W * pc.py + pc.px;
    const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);

    const uint32_t r0 = a.ranges[2 * tile], r1 = a.ranges[2 * tile + 1];
    const int n = (int)(r1 - r0);
    if (n == 0) return;

    // per-pixel constants
    float T_final = 0, final_D = 0, final_D2 = 0;
    uint32_t last_contributor = 0, median_contributor = 0;
    float dpx0 = 0, dpx1 = 0, dpx2 = 0, dL_ddepth = 0, dL_daccum = 0, dL_dreg = 0, dn0 = 0, dn1 = 0, dn2 = 0,
          dL_dmedian = 0;
    if (pc.inside) {
        T_final = a.final_T[pix_id];
        final_D = a.final_T[pix_id + N];
        final_D2 = a.final_T[pix_id + 2 * N];
        last_contributor = a.n_contrib[pix_id];
        median_contributor = a.n_contrib[pix_id + N];
        dpx0 = a.dL_dpix[pix_id];
        dpx1 = a.dL_dpix[pix_id + N];
        dpx2 = a.dL_dpix[pix_id + 2 * N];
        dL_ddepth = a.dL_depths[pix_id + 0 * N];
        dL_daccum = a.dL_depths[pix_id + 1 * N];
        dn0 = a.dL_depths[pix_id + 2 * N];
        dn1 = a.dL_depths[pix_id + 3 * N];
        dn2 = a.dL_depths[pix_id + 4 * N];
        dL_dmedian = a.dL_depths[pix_id + 5 * N];
        dL_dreg = a.dL_depths[pix_id + 6 * N];
    }
    const float final_A = 1 - T_final;
    const float bg_dot_dpixel = (a.bg[0] * dpx0 + a.bg[1] * dpx1) + a.bg[2] * dpx2;
    const float mscale = FAR_N / (FAR_N - NEAR_N);
    const float dmd_k = (FAR_N * NEAR_N) / (FAR_N - NEAR_N);

    // entries at list positions >= max(last_contributor) over the tile contribute nothing
    {
        const uint32_t m = wave_max_u32(last_contributor);
        if (lane == 0) s_maxc[wv] = m;
    }
    __syncthreads();
    const int n_live = (int)max(max(s_maxc[0], s_maxc[1]), max(s_maxc[2], s_maxc[3]));

    // zero gradient records for the dead tail [n_live, n)
    for (int i = n_live * GRAD_FLOATS / 2 + (int)threadIdx.x; i < n * GRAD_FLOATS / 2; i += 256) {
        const int j = i / (GRAD_FLOATS / 2), q = i - j * (GRAD_FLOATS / 2);
        const uint64_t e = a.entries[r0 + j];
        const uint32_t idx = entry_idx(e);
        const uint32_t slot = __float_as_uint(a.rec[(size_t)idx * REC_FLOATS + 2]) + entry_k(e);
        reinterpret_cast<float2*>(a.grad_inst + (size_t)slot * GRAD_FLOATS)[q] = make_float2(0.f, 0.f);
    }

    // running per-pixel state (back to front)
    float T = T_final;
    float last_alpha = 0, last_v = 0, V_rec = 0, last_dL_dT = 0;

    // batches run from the back of the live range: batch b covers list positions
    // [hi - m, hi), thread t of the staging handles position hi - 1 - t (reverse order)
    for (int hi = n_live; hi > 0; hi -= BATCH) {
        const int m = imin_(BATCH, hi);
        __syncthreads();  // previous batch fully consumed / written out
        stage_record(a.entries, a.rec, r0 + (uint32_t)(hi - 1 - (int)threadIdx.x), (int)threadIdx.x < m, s_rec, s_slot);
        for (int i = (int)threadIdx.x; i < m * GRAD_FLOATS; i += 256) s_grad[i] = 0.0f;
        __syncthreads();

        for (int j = 0; j < m; j++) {
            const uint32_t pos = (uint32_t)(hi - 1 - j);  // 0-based list position == backward `contributor`
            const float4 q0 = s_rec[0][j], q1 = s_rec[1][j], q2 = s_rec[2][j], q3 = s_rec[3][j], q4 = s_rec[4][j];
            PairEval e;
            bool act = pc.inside && pos < last_contributor;
            if (act)
                act = eval_pair(pxf, pyf, q0.x, q0.y, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w, q4.x, q1.w, e);
            if (!__any(act)) continue;  // wave-uniform: nothing to reduce for this splat

            float g[GRAD_FLOATS];
#pragma unroll
            for (int i = 0; i < GRAD_FLOATS; i++) g[i] = 0.0f;
            if (act) {
                const float G = e.G, alpha = e.alpha, c_d = e.depth;
                const float inv1ma = 1.0f / (1.f - alpha);
                T = T * inv1ma;                       // T / (1 - alpha), backward.cu:316
                const float w = alpha * T;
                // colour / depth / alpha / normal "accum_rec" recurrences (backward.cu:328,362-371)
                // share their coefficients, so they are folded into one scalar recurrence on
                // v = <c,dL_dpix> + c_d*dL_ddepth + dL_daccum + <n,dL_dnormal>.
                const float v = ((q4.y * dpx0 + q4.z * dpx1) + q4.w * dpx2) + c_d * dL_ddepth + dL_daccum +
                                ((q1.x * dn0 + q1.y * dn1) + q1.z * dn2);
                V_rec = last_alpha * last_v + (1.f - last_alpha) * V_rec;
                last_v = v;
                float dL_dalpha = v - V_rec;
                g[0] = w * dpx0;
                g[1] = w * dpx1;
                g[2] = w * dpx2;
                g[3] = w * dn0;
                g[4] = w * dn1;
                g[5] = w * dn2;

                const float inv_cd = 1.0f / c_d;
                const float m_d = mscale * (1 - NEAR_N * inv_cd);
                const float dmd_dd = dmd_k * inv_cd * inv_cd;
                float dL_dz = (pos + 1 == median_contributor) ? dL_dmedian : 0.0f;
                const float dL_dweight = (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
                dL_dalpha += dL_dweight - last_dL_dT;
                last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
                const float dL_dmd = 2.0f * w * (m_d * final_A - final_D) * dL_dreg;
                dL_dz += dL_dmd * dmd_dd;

                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final * inv1ma) * bg_dot_dpixel;
                const float dL_dG = q1.w * dL_dalpha;  // not gated by the 0.99 clamp (backward.cu:390)
                dL_dz += w * dL_ddepth;

                if (e.rho3d <= e.rho2d) {
                    const float dL_dsx = dL_dG * -G * e.sx + dL_dz * q3.z;
                    const float dL_dsy = dL_dG * -G * e.sy + dL_dz * q3.w;
                    const float inv_pz = 1.0f / e.pz;
                    const float dpx_ = dL_dsx * inv_pz, dpy_ = dL_dsy * inv_pz;
                    const float dpz_ = -(dpx_ * e.sx + dpy_ * e.sy);
                    // dL_dk = cross(l, dL_dp), dL_dl = cross(dL_dp, k)
                    const float dkx = e.ly * dpz_ - e.lz * dpy_, dky = e.lz * dpx_ - e.lx * dpz_,
                                dkz = e.lx * dpy_ - e.ly * dpx_;
                    const float dlx = dpy_ * e.kz - dpz_ * e.ky, dly = dpz_ * e.kx - dpx_ * e.kz,
                                dlz = dpx_ * e.ky - dpy_ * e.kx;
                    g[6] = -dkx;
                    g[7] = -dky;
                    g[8] = -dkz;
                    g[9] = -dlx;
                    g[10] = -dly;
                    g[11] = -dlz;
                    g[12] = pxf * dkx + pyf * dlx + dL_dz * e.sx;
                    g[13] = pxf * dky + pyf * dly + dL_dz * e.sy;
                    g[14] = pxf * dkz + pyf * dlz + dL_dz;
                } else {
                    g[15] = dL_dG * (-G * FILTER_INV_SQUARE * e.dx);
                    g[16] = dL_dG * (-G * FILTER_INV_SQUARE * e.dy);
                    g[14] = dL_dz;
                }
                g[17] = G * dL_dalpha;
            }
            // 64 lanes -> 1 with DPP, 4 waves -> 1 in LDS
#pragma unroll
            for (int i = 0; i < GRAD_FLOATS; i++) g[i] = wave_sum_to_lane63(g[i]);
            if (lane == 63) {
#pragma unroll
                for (int i = 0; i < GRAD_FLOATS; i++) atomicAdd(&s_grad[j * GRAD_FLOATS + i], g[i]);
            }
        }
        __syncthreads();
        // one 72-byte record per instance
        for (int i = (int)threadIdx.x; i < m * (GRAD_FLOATS / 2); i += 256) {
            const int j = i / (GRAD_FLOATS / 2), q = i - j * (GRAD_FLOATS / 2);
            reinterpret_cast<float2*>(a.grad_inst + (size_t)s_slot[j] * GRAD_FLOATS)[q] =
                make_float2(s_grad[j * GRAD_FLOATS + 2 * q], s_grad[j * GRAD_FLOATS + 2 * q + 1]);
        }
    }
}

void launch_blend_bwd(const BlendBwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(a.tiles_x * a.tiles_y), dim3(256), 0, s, a);
}

}  // namespace g4s
