/*
 * surfel_oracle.c -- CPU restatement of the reference 2D-Gaussian (surfel) rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under g4splat_amd/ may import, link or call this
 * file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / reported CPU baseline -- never as the thing measured or shipped.
 *
 * PARITY UNPINNED: the reference (DaLi-Jack/G4Splat, submodule diff-surfel-rasterization)
 * ships no tests, golden vectors or known-answer fixtures for this path, it has no CPU
 * fallback, and its CUDA sources cannot be built or run in this project (no nvcc, no
 * CUDA runtime).  This restatement is therefore pinned only by (a) an independent
 * pure-PyTorch autograd renderer (oracle/torch_ref.py), (b) golden vectors produced by
 * importing the reference's own Python helpers (eval_sh, getProjectionMatrix,
 * getWorld2View2, build_scaling_rotation formulation of T; tests/golden/make_golden.py)
 * and (c) explicit quirk assertions in tests/.
 *
 * Every function cites the reference file:line it follows.  Prefixes:
 *   dsr/ = 2d-gaussian-splatting/submodules/diff-surfel-rasterization/
 *   knn/ = 2d-gaussian-splatting/submodules/simple-knn/
 *
 * Arithmetic policy: single precision, operations in the order the reference source
 * writes them, compiled with -ffp-contract=off.  Where this file spells fmaf() explicitly
 * (per-pixel ray/splat geometry) it fixes ONE valid contraction of the reference
 * expression (nvcc contracts a*b+c under its default -fmad=true; which operand pairs it
 * picks is not knowable here) so the HIP kernels can use FMA and still agree bit-for-bit
 * on every discrete decision.  rsqrtf (an approximate instruction on the GPU) is restated
 * as 1/sqrtf.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <stdio.h>

#define BLOCK_X 16 /* dsr/cuda_rasterizer/config.h:16 */
#define BLOCK_Y 16 /* dsr/cuda_rasterizer/config.h:17 */

/* dsr/cuda_rasterizer/auxiliary.h:23-27 : channel map of out_others */
#define DEPTH_OFFSET 0
#define ALPHA_OFFSET 1
#define NORMAL_OFFSET 2
#define MIDDEPTH_OFFSET 5
#define DISTORTION_OFFSET 6

/* dsr/cuda_rasterizer/auxiliary.h:37-39 */
static const float near_n = 0.2f;
static const float far_n = 100.0f;
static const float FilterInvSquare = 2.0f;

/* dsr/cuda_rasterizer/auxiliary.h:42-59 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

typedef struct { float x, y, z; } f3;

typedef struct OracleState {
    int P, W, H, R, tiles_x, tiles_y;
    /* GeometryState, dsr/cuda_rasterizer/rasterizer_impl.h:33-48 */
    float *depths;          /* P */
    uint8_t *clamped;       /* 3P */
    int *radii;             /* P */
    float *means2D;         /* 2P */
    float *transMat;        /* 9P */
    float *normal_opacity;  /* 4P */
    float *rgb;             /* 3P */
    uint32_t *tiles_touched;/* P */
    uint32_t *point_offsets;/* P (inclusive scan) */
    /* BinningState, rasterizer_impl.h:59-69 */
    uint64_t *keys;         /* R sorted */
    uint32_t *point_list;   /* R sorted */
    /* ImageState, rasterizer_impl.h:50-57 */
    uint32_t *ranges;       /* 2*tiles */
    float *final_T;         /* 3N : T, M1, M2 */
    uint32_t *n_contrib;    /* 2N : last, median */
    /* raw blend-backward accumulators kept for stage-wise comparison */
    float *dL_dtransMat_raw; /* 9P, value after blend backward and before the mean2D fold */
    float *dL_dnormal3D;     /* 3P */
    float *dL_dmean2D_raw;   /* 3P, low-pass branch accumulations before the surrogate overwrite */
} OracleState;

OracleState *oracle_state_new(void) { return (OracleState *)calloc(1, sizeof(OracleState)); }

static void state_clear(OracleState *s) {
    free(s->depths); free(s->clamped); free(s->radii); free(s->means2D); free(s->transMat);
    free(s->normal_opacity); free(s->rgb); free(s->tiles_touched); free(s->point_offsets);
    free(s->keys); free(s->point_list); free(s->ranges); free(s->final_T); free(s->n_contrib);
    free(s->dL_dtransMat_raw); free(s->dL_dnormal3D); free(s->dL_dmean2D_raw);
    memset(s, 0, sizeof(*s));
}
void oracle_state_free(OracleState *s) { if (s) { state_clear(s); free(s); } }

/* accessors used by the ctypes wrapper */
int oracle_state_R(const OracleState *s) { return s->R; }
const float *oracle_state_depths(const OracleState *s) { return s->depths; }
const uint8_t *oracle_state_clamped(const OracleState *s) { return s->clamped; }
const float *oracle_state_means2D(const OracleState *s) { return s->means2D; }
const float *oracle_state_transMat(const OracleState *s) { return s->transMat; }
const float *oracle_state_normal_opacity(const OracleState *s) { return s->normal_opacity; }
const float *oracle_state_rgb(const OracleState *s) { return s->rgb; }
const uint32_t *oracle_state_tiles_touched(const OracleState *s) { return s->tiles_touched; }
const uint32_t *oracle_state_point_offsets(const OracleState *s) { return s->point_offsets; }
const uint64_t *oracle_state_keys(const OracleState *s) { return s->keys; }
const uint32_t *oracle_state_point_list(const OracleState *s) { return s->point_list; }
const uint32_t *oracle_state_ranges(const OracleState *s) { return s->ranges; }
const float *oracle_state_final_T(const OracleState *s) { return s->final_T; }
const uint32_t *oracle_state_n_contrib(const OracleState *s) { return s->n_contrib; }
const float *oracle_state_dL_dtransMat_raw(const OracleState *s) { return s->dL_dtransMat_raw; }
const float *oracle_state_dL_dnormal3D(const OracleState *s) { return s->dL_dnormal3D; }
const float *oracle_state_dL_dmean2D_raw(const OracleState *s) { return s->dL_dmean2D_raw; }

/* ------------------------------------------------------------------------------------ */
/* small helpers                                                                          */

/* float -> int conversion with the GPU's semantics (cvt.rzi.s32.f32 / v_cvt_i32_f32:
 * saturating, NaN -> 0); plain C casts are undefined out of range. */
static inline int sat_int(float v) {
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* dsr/cuda_rasterizer/auxiliary.h:78-86 */
static inline f3 transformPoint4x3(f3 p, const float *m) {
    f3 r = {m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
            m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
            m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14]};
    return r;
}
/* dsr/cuda_rasterizer/auxiliary.h:99-107 */
static inline f3 transformVec4x3(f3 p, const float *m) {
    f3 r = {m[0] * p.x + m[4] * p.y + m[8] * p.z,
            m[1] * p.x + m[5] * p.y + m[9] * p.z,
            m[2] * p.x + m[6] * p.y + m[10] * p.z};
    return r;
}
/* dsr/cuda_rasterizer/auxiliary.h:109-117 */
static inline f3 transformVec4x3Transpose(f3 p, const float *m) {
    f3 r = {m[0] * p.x + m[1] * p.y + m[2] * p.z,
            m[4] * p.x + m[5] * p.y + m[6] * p.z,
            m[8] * p.x + m[9] * p.y + m[10] * p.z};
    return r;
}

/* dsr/cuda_rasterizer/auxiliary.h:184-209 : near-plane test only (the NDC test is
 * commented out in the reference); prefiltered && culled traps in the reference. */
static inline int in_frustum(int idx, const float *orig_points, const float *viewmatrix, f3 *p_view) {
    f3 p = {orig_points[3 * idx], orig_points[3 * idx + 1], orig_points[3 * idx + 2]};
    *p_view = transformPoint4x3(p, viewmatrix);
    return p_view->z > 0.2f;
}

/* dsr/cuda_rasterizer/auxiliary.h:212-234 ; rot is stored (w,x,y,z); rsqrtf -> 1/sqrtf.
 * R is returned column-major like glm: R[c*3+r]. */
static inline void quat_to_rotmat(const float *q, float *R) {
    float s = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
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

/* dsr/cuda_rasterizer/auxiliary.h:237-281 ; v_R column-major v_R[c*3+r]; output stored
 * (w,x,y,z).  No normalisation Jacobian (quirk). */
static inline void quat_to_rotmat_vjp(const float *q, const float *v_R, float *v_q) {
    float s = 1.0f / sqrtf(q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    float w = q[0] * s, x = q[1] * s, y = q[2] * s, z = q[3] * s;
#define VR(c, r) v_R[(c) * 3 + (r)]
    v_q[0] = 2.f * (x * (VR(1, 2) - VR(2, 1)) + y * (VR(2, 0) - VR(0, 2)) + z * (VR(0, 1) - VR(1, 0)));
    v_q[1] = 2.f * (-2.f * x * (VR(1, 1) + VR(2, 2)) + y * (VR(0, 1) + VR(1, 0)) +
                    z * (VR(0, 2) + VR(2, 0)) + w * (VR(1, 2) - VR(2, 1)));
    v_q[2] = 2.f * (x * (VR(0, 1) + VR(1, 0)) - 2.f * y * (VR(0, 0) + VR(2, 2)) +
                    z * (VR(1, 2) + VR(2, 1)) + w * (VR(2, 0) - VR(0, 2)));
    v_q[3] = 2.f * (x * (VR(0, 2) + VR(2, 0)) + y * (VR(1, 2) + VR(2, 1)) -
                    2.f * z * (VR(0, 0) + VR(1, 1)) + w * (VR(0, 1) - VR(1, 0)));
#undef VR
}

/* dsr/cuda_rasterizer/auxiliary.h:66-76 */
static inline void getRect(float px, float py, int max_radius, int gx, int gy, int *rmin, int *rmax) {
    rmin[0] = imin(gx, imax(0, sat_int((px - max_radius) / BLOCK_X)));
    rmin[1] = imin(gy, imax(0, sat_int((py - max_radius) / BLOCK_Y)));
    rmax[0] = imin(gx, imax(0, sat_int((px + max_radius + BLOCK_X - 1) / BLOCK_X)));
    rmax[1] = imin(gy, imax(0, sat_int((py + max_radius + BLOCK_Y - 1) / BLOCK_Y)));
}

/* dsr/cuda_rasterizer/forward.cu:75-115 (glm products expanded in glm's own summation order,
 * third_party/glm/glm/detail/type_mat4x3.inl:506-560; exact-zero terms dropped).
 * T is returned as the three stored rows Tu,Tv,Tw = glm columns T[0],T[1],T[2]. */
static inline void compute_transmat(f3 p, float sx, float sy, float mod, const float *rot,
                                    const float *proj, const float *view, int W, int H,
                                    float *T /*9*/, f3 *normal) {
    float R[9];
    quat_to_rotmat(rot, R);
    float s0 = mod * sx, s1 = mod * sy; /* auxiliary.h:284-291 */
    float M[3][4] = {{R[0] * s0, R[1] * s0, R[2] * s0, 0.0f},
                     {R[3] * s1, R[4] * s1, R[5] * s1, 0.0f},
                     {p.x, p.y, p.z, 1.0f}};
    float hw = (float)((float)W / 2.0), cw = (float)((float)(W - 1) / 2.0);
    float hh = (float)((float)H / 2.0), ch = (float)((float)(H - 1) / 2.0);
    for (int r = 0; r < 3; r++) {
        float c[4];
        for (int j = 0; j < 4; j++)
            c[j] = M[r][0] * proj[j] + M[r][1] * proj[4 + j] + M[r][2] * proj[8 + j] + M[r][3] * proj[12 + j];
        T[0 * 3 + r] = c[0] * hw + c[3] * cw;
        T[1 * 3 + r] = c[1] * hh + c[3] * ch;
        T[2 * 3 + r] = c[3];
    }
    f3 l2 = {R[6], R[7], R[8]};
    *normal = transformVec4x3(l2, view);
}

/* dsr/cuda_rasterizer/forward.cu:119-147 */
static inline int compute_aabb(const float *T, float cutoff, float *pt, float *ext) {
    const float *T0 = T, *T1 = T + 3, *T3 = T + 6;
    float t[3] = {cutoff * cutoff, cutoff * cutoff, -1.0f};
    float distance = (T3[0] * T3[0] * t[0] + T3[1] * T3[1] * t[1]) + T3[2] * T3[2] * t[2];
    float f[3] = {(1 / distance) * t[0], (1 / distance) * t[1], (1 / distance) * t[2]};
    if (distance == 0.0f) return 0;
    pt[0] = (f[0] * T0[0] * T3[0] + f[1] * T0[1] * T3[1]) + f[2] * T0[2] * T3[2];
    pt[1] = (f[0] * T1[0] * T3[0] + f[1] * T1[1] * T3[1]) + f[2] * T1[2] * T3[2];
    float tmp0 = (f[0] * T0[0] * T0[0] + f[1] * T0[1] * T0[1]) + f[2] * T0[2] * T0[2];
    float tmp1 = (f[0] * T1[0] * T1[0] + f[1] * T1[1] * T1[1]) + f[2] * T1[2] * T1[2];
    float h0 = pt[0] * pt[0] - tmp0, h1 = pt[1] * pt[1] - tmp1;
    ext[0] = sqrtf(fmaxf(1e-4f, h0));
    ext[1] = sqrtf(fmaxf(1e-4f, h1));
    return 1;
}

/* dsr/cuda_rasterizer/forward.cu:20-71 */
static inline void computeColorFromSH(int idx, int deg, int max_coeffs, const float *means, const float *campos,
                                      const float *shs, uint8_t *clamped, float *out) {
    float dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    const float *sh = shs + (size_t)idx * max_coeffs * 3;
    float res[3];
    for (int c = 0; c < 3; c++) {
#define S(i) sh[(i) * 3 + c]
        float r = SH_C0 * S(0);
        if (deg > 0) {
            r = r - SH_C1 * y * S(1) + SH_C1 * z * S(2) - SH_C1 * x * S(3);
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * S(4) + SH_C2[1] * yz * S(5) + SH_C2[2] * (2.0f * zz - xx - yy) * S(6) +
                    SH_C2[3] * xz * S(7) + SH_C2[4] * (xx - yy) * S(8);
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * S(9) + SH_C3[1] * xy * z * S(10) +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * S(11) +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * S(12) +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * S(13) + SH_C3[5] * z * (xx - yy) * S(14) +
                        SH_C3[6] * x * (xx - 3.0f * yy) * S(15);
                }
            }
        }
#undef S
        r += 0.5f;
        res[c] = r;
    }
    for (int c = 0; c < 3; c++) {
        clamped[3 * idx + c] = (res[c] < 0);
        out[c] = fmaxf(res[c], 0.0f);
    }
}

/* dsr/cuda_rasterizer/rasterizer_impl.cu:35-50 */
static uint32_t getHigherMsb(uint32_t n) {
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step; else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}
uint32_t oracle_get_higher_msb(uint32_t n) { return getHigherMsb(n); }

/* Stable LSD radix sort of (u64 key, u32 value) over bits [0,end_bit): restates the
 * semantics of cub::DeviceRadixSort::SortPairs used at rasterizer_impl.cu:304-309. */
static void stable_sort_pairs(uint64_t *keys, uint32_t *vals, size_t n, int end_bit) {
    if (n == 0) return;
    uint64_t *k2 = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint32_t *v2 = (uint32_t *)malloc(n * sizeof(uint32_t));
    size_t *cnt = (size_t *)malloc(65537 * sizeof(size_t));
    uint64_t *ka = keys, *kb = k2; uint32_t *va = vals, *vb = v2;
    for (int shift = 0; shift < end_bit; shift += 16) {
        int bits = end_bit - shift < 16 ? end_bit - shift : 16;
        uint64_t mask = ((uint64_t)1 << bits) - 1;
        memset(cnt, 0, 65537 * sizeof(size_t));
        for (size_t i = 0; i < n; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (size_t i = 0; i < 65536; i++) cnt[i + 1] += cnt[i];
        for (size_t i = 0; i < n; i++) {
            size_t d = cnt[(ka[i] >> shift) & mask]++;
            kb[d] = ka[i]; vb[d] = va[i];
        }
        uint64_t *tk = ka; ka = kb; kb = tk; uint32_t *tv = va; va = vb; vb = tv;
    }
    if (ka != keys) { memcpy(keys, ka, n * sizeof(uint64_t)); memcpy(vals, va, n * sizeof(uint32_t)); }
    free(k2); free(v2); free(cnt);
}

/* ------------------------------------------------------------------------------------ */
/* forward                                                                                */

/* dsr/cuda_rasterizer/rasterizer_impl.cu:54-66,141-153 */
void oracle_mark_visible(int P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                         uint8_t *present) {
    (void)projmatrix;
    for (int i = 0; i < P; i++) { f3 pv; present[i] = (uint8_t)in_frustum(i, means3D, viewmatrix, &pv); }
}

/* dsr/cuda_rasterizer/forward.cu:150-253 (preprocessCUDA) */
static void preprocess_fwd(OracleState *s, int D, int M, const float *means3D, const float *scales,
                           float scale_modifier, const float *rotations, const float *opacities,
                           const float *shs, const float *transMat_precomp, const float *colors_precomp,
                           const float *viewmatrix, const float *projmatrix, const float *campos, int *radii) {
    const int P = s->P, W = s->W, H = s->H, gx = s->tiles_x, gy = s->tiles_y;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        radii[idx] = 0;
        s->tiles_touched[idx] = 0;
        f3 p_view;
        if (!in_frustum(idx, means3D, viewmatrix, &p_view)) continue;
        float T[9]; f3 normal;
        if (transMat_precomp == NULL) {
            f3 p = {means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]};
            compute_transmat(p, scales[2 * idx], scales[2 * idx + 1], scale_modifier, rotations + 4 * idx,
                             projmatrix, viewmatrix, W, H, T, &normal);
            memcpy(s->transMat + 9 * (size_t)idx, T, sizeof(T)); /* written even if culled below */
        } else {
            memcpy(T, transMat_precomp + 9 * (size_t)idx, sizeof(T));
            normal.x = 0.0f; normal.y = 0.0f; normal.z = 1.0f;
        }
        /* DUAL_VISIABLE, forward.cu:211-216 */
        float cosv = -((p_view.x * normal.x + p_view.y * normal.y) + p_view.z * normal.z);
        if (cosv == 0) continue;
        float multiplier = cosv > 0 ? 1.0f : -1.0f;
        normal.x = multiplier * normal.x; normal.y = multiplier * normal.y; normal.z = multiplier * normal.z;

        float pt[2], ext[2];
        if (!compute_aabb(T, 3.0f, pt, ext)) continue;
        float radius = ceilf(fmaxf(ext[0], ext[1]));
        int rmin[2], rmax[2];
        getRect(pt[0], pt[1], sat_int(radius), gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;

        if (colors_precomp == NULL)
            computeColorFromSH(idx, D, M, means3D, campos, shs, s->clamped, s->rgb + 3 * (size_t)idx);
        s->depths[idx] = p_view.z;
        radii[idx] = sat_int(radius);
        s->means2D[2 * idx] = pt[0]; s->means2D[2 * idx + 1] = pt[1];
        float *no = s->normal_opacity + 4 * (size_t)idx;
        no[0] = normal.x; no[1] = normal.y; no[2] = normal.z; no[3] = opacities[idx];
        s->tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
}

/* per-pixel ray/splat evaluation shared by forward and backward (forward.cu:346-381,
 * backward.cu:267-301).  Returns 0 if the pair is skipped. */
typedef struct { float sx, sy, pz, kx, ky, kz, lx, ly, lz, rho3d, rho2d, depth, G, alpha, dx, dy; } PairEval;

static inline int eval_pair(float pxf, float pyf, const float *xy, const float *Tm, float opa, PairEval *e) {
    const float *Tu = Tm, *Tv = Tm + 3, *Tw = Tm + 6;
    /* k = pix.x * Tw - Tu ; l = pix.y * Tw - Tv  (forward.cu:355-356) */
    e->kx = fmaf(pxf, Tw[0], -Tu[0]); e->ky = fmaf(pxf, Tw[1], -Tu[1]); e->kz = fmaf(pxf, Tw[2], -Tu[2]);
    e->lx = fmaf(pyf, Tw[0], -Tv[0]); e->ly = fmaf(pyf, Tw[1], -Tv[1]); e->lz = fmaf(pyf, Tw[2], -Tv[2]);
    /* p = cross(k, l)  (auxiliary.h:154) */
    float ppx = fmaf(e->ky, e->lz, -(e->kz * e->ly));
    float ppy = fmaf(e->kz, e->lx, -(e->kx * e->lz));
    float ppz = fmaf(e->kx, e->ly, -(e->ky * e->lx));
    if (ppz == 0.0f) return 0;
    e->pz = ppz;
    e->sx = ppx / ppz; e->sy = ppy / ppz;
    e->rho3d = fmaf(e->sx, e->sx, e->sy * e->sy);
    e->dx = xy[0] - pxf; e->dy = xy[1] - pyf;
    e->rho2d = FilterInvSquare * fmaf(e->dx, e->dx, e->dy * e->dy);
    float rho = fminf(e->rho3d, e->rho2d);
    e->depth = (e->rho3d <= e->rho2d) ? fmaf(e->sx, Tw[0], e->sy * Tw[1]) + Tw[2] : Tw[2];
    if (e->depth < near_n) return 0;
    float power = -0.5f * rho;
    if (power > 0.0f) return 0;
    e->G = expf(power);
    e->alpha = fminf(0.99f, opa * e->G);
    if (e->alpha < 1.0f / 255.0f) return 0;
    return 1;
}

/* dsr/cuda_rasterizer/forward.cu:258-443 (renderCUDA), one pixel at a time: the block-level
 * early exit and shared-memory batching of the reference only skip work. */
static void render_fwd(OracleState *s, const float *features, const float *transMats, const float *bg,
                       float *out_color, float *out_others) {
    const int W = s->W, H = s->H, gx = s->tiles_x, gy = s->tiles_y;
    const size_t N = (size_t)W * H;
    const float mscale = far_n / (far_n - near_n);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
            int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (px >= W || py >= H) continue;
            size_t pix_id = (size_t)W * py + px;
            float pxf = (float)px, pyf = (float)py;
            float T = 1.0f; uint32_t contributor = 0, last_contributor = 0;
            float C[3] = {0, 0, 0}, Nn[3] = {0, 0, 0};
            float Dd = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
            float median_contributor = -1; /* forward.cu:318 */
            for (uint32_t i = r0; i < r1; i++) {
                contributor++;
                uint32_t id = s->point_list[i];
                const float *no = s->normal_opacity + 4 * (size_t)id;
                PairEval e;
                if (!eval_pair(pxf, pyf, s->means2D + 2 * (size_t)id, transMats + 9 * (size_t)id, no[3], &e)) continue;
                float alpha = e.alpha, depth = e.depth;
                float test_T = T * (1 - alpha);
                if (test_T < 0.0001f) break; /* done = true */
                float w = alpha * T;
                float A = 1 - T;
                float m = mscale * (1 - near_n / depth);
                distortion += (m * m * A + M2 - 2 * m * M1) * w;
                Dd += depth * w;
                M1 += m * w;
                M2 += m * m * w;
                if (T > 0.5f) { median_depth = depth; median_contributor = (float)contributor; }
                for (int ch = 0; ch < 3; ch++) Nn[ch] += no[ch] * w;
                for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * (size_t)id + ch] * w;
                T = test_T;
                last_contributor = contributor;
            }
            s->final_T[pix_id] = T;
            s->n_contrib[pix_id] = last_contributor;
            for (int ch = 0; ch < 3; ch++) out_color[ch * N + pix_id] = C[ch] + T * bg[ch];
            /* float -> uint32 of -1.0f saturates to 0 on the GPU (cvt.rzi.u32.f32) */
            s->n_contrib[pix_id + N] = median_contributor < 0 ? 0u : (uint32_t)median_contributor;
            s->final_T[pix_id + N] = M1;
            s->final_T[pix_id + 2 * N] = M2;
            out_others[pix_id + DEPTH_OFFSET * N] = Dd;
            out_others[pix_id + ALPHA_OFFSET * N] = 1 - T;
            for (int ch = 0; ch < 3; ch++) out_others[pix_id + (NORMAL_OFFSET + ch) * N] = Nn[ch];
            out_others[pix_id + MIDDEPTH_OFFSET * N] = median_depth;
            out_others[pix_id + DISTORTION_OFFSET * N] = distortion;
        }
    }
}

/* dsr/cuda_rasterizer/rasterizer_impl.cu:198-342 (Rasterizer::forward) and
 * dsr/rasterize_points.cu:39-134 (output allocation / zero-init).  Returns num_rendered. */
int oracle_forward(OracleState *s, int P, int D, int M, const float *background, int width, int height,
                   const float *means3D, const float *shs, const float *colors_precomp, const float *opacities,
                   const float *scales, float scale_modifier, const float *rotations,
                   const float *transMat_precomp, const float *viewmatrix, const float *projmatrix,
                   const float *cam_pos, float tan_fovx, float tan_fovy, int prefiltered, float *out_color,
                   float *out_others, int *radii) {
    (void)tan_fovx; (void)tan_fovy; (void)prefiltered;
    state_clear(s);
    const size_t N = (size_t)width * height;
    memset(out_color, 0, 3 * N * sizeof(float));   /* rasterize_points.cu:85 */
    memset(out_others, 0, 7 * N * sizeof(float));  /* rasterize_points.cu:86 */
    for (int i = 0; i < P; i++) radii[i] = 0;      /* rasterize_points.cu:87 */
    s->P = P; s->W = width; s->H = height;
    s->tiles_x = (width + BLOCK_X - 1) / BLOCK_X; s->tiles_y = (height + BLOCK_Y - 1) / BLOCK_Y;
    if (P == 0) return 0;                          /* rasterize_points.cu:99 */
    const int tiles = s->tiles_x * s->tiles_y;
    s->depths = (float *)calloc(P, sizeof(float));
    s->clamped = (uint8_t *)calloc(3 * (size_t)P, 1);
    s->radii = (int *)calloc(P, sizeof(int));
    s->means2D = (float *)calloc(2 * (size_t)P, sizeof(float));
    s->transMat = (float *)calloc(9 * (size_t)P, sizeof(float));
    s->normal_opacity = (float *)calloc(4 * (size_t)P, sizeof(float));
    s->rgb = (float *)calloc(3 * (size_t)P, sizeof(float));
    s->tiles_touched = (uint32_t *)calloc(P, sizeof(uint32_t));
    s->point_offsets = (uint32_t *)calloc(P, sizeof(uint32_t));
    s->ranges = (uint32_t *)calloc(2 * (size_t)tiles, sizeof(uint32_t));
    s->final_T = (float *)calloc(3 * N, sizeof(float));
    s->n_contrib = (uint32_t *)calloc(2 * N, sizeof(uint32_t));

    preprocess_fwd(s, D, M, means3D, scales, scale_modifier, rotations, opacities, shs, transMat_precomp,
                   colors_precomp, viewmatrix, projmatrix, cam_pos, radii);
    memcpy(s->radii, radii, P * sizeof(int));

    /* InclusiveSum, rasterizer_impl.cu:278 */
    uint32_t acc = 0;
    for (int i = 0; i < P; i++) { acc += s->tiles_touched[i]; s->point_offsets[i] = acc; }
    const int R = (int)acc;
    s->R = R;
    s->keys = (uint64_t *)malloc((R ? R : 1) * sizeof(uint64_t));
    s->point_list = (uint32_t *)malloc((R ? R : 1) * sizeof(uint32_t));

    /* duplicateWithKeys, rasterizer_impl.cu:70-111 */
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (radii[idx] > 0) {
            uint32_t off = (idx == 0) ? 0 : s->point_offsets[idx - 1];
            int rmin[2], rmax[2];
            getRect(s->means2D[2 * idx], s->means2D[2 * idx + 1], radii[idx], s->tiles_x, s->tiles_y, rmin, rmax);
            uint32_t dbits; memcpy(&dbits, &s->depths[idx], 4);
            for (int y = rmin[1]; y < rmax[1]; y++) for (int x = rmin[0]; x < rmax[0]; x++) {
                uint64_t key = (uint64_t)(y * s->tiles_x + x);
                key <<= 32; key |= dbits;
                s->keys[off] = key; s->point_list[off] = (uint32_t)idx; off++;
            }
        }
    }
    int bit = (int)getHigherMsb((uint32_t)tiles);
    stable_sort_pairs(s->keys, s->point_list, (size_t)R, 32 + bit); /* rasterizer_impl.cu:304-309 */

    /* identifyTileRanges, rasterizer_impl.cu:116-138 (ranges zeroed at :311) */
    for (int i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(s->keys[i] >> 32);
        if (i == 0) s->ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(s->keys[i - 1] >> 32);
            if (cur != prev) { s->ranges[2 * prev + 1] = (uint32_t)i; s->ranges[2 * cur] = (uint32_t)i; }
        }
        if (i == R - 1) s->ranges[2 * cur + 1] = (uint32_t)R;
    }
    const float *feature_ptr = colors_precomp ? colors_precomp : s->rgb;
    const float *transMat_ptr = transMat_precomp ? transMat_precomp : s->transMat;
    render_fwd(s, feature_ptr, transMat_ptr, background, out_color, out_others);
    return R;
}

/* ------------------------------------------------------------------------------------ */
/* backward                                                                               */

/* dsr/cuda_rasterizer/backward.cu:143-440 (renderCUDA backward).  The reference sums the
 * per-pair terms with float atomicAdd in a non-deterministic order; this restatement sums
 * the same float terms into double accumulators (order-independent to ~1e-16). */
static void render_bwd(const OracleState *s, const float *bg, const float *colors, const float *transMats,
                       const float *dL_dpixels, const float *dL_depths, double *dL_dtransMat, double *dL_dmean2D,
                       double *dL_dnormal3D, double *dL_dopacity, double *dL_dcolors) {
    const int W = s->W, H = s->H, gx = s->tiles_x, gy = s->tiles_y;
    const size_t N = (size_t)W * H;
    const float mscale = far_n / (far_n - near_n);
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
            int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (px >= W || py >= H) continue;
            size_t pix_id = (size_t)W * py + px;
            float pxf = (float)px, pyf = (float)py;
            const float T_final = s->final_T[pix_id];
            float T = T_final;
            uint32_t contributor = r1 - r0;
            const int last_contributor = (int)s->n_contrib[pix_id];
            float accum_rec[3] = {0, 0, 0}, dL_dpixel[3];
            const int median_contributor = (int)s->n_contrib[pix_id + N];
            float dL_ddepth = dL_depths[DEPTH_OFFSET * N + pix_id];
            float dL_daccum = dL_depths[ALPHA_OFFSET * N + pix_id];
            float dL_dreg = dL_depths[DISTORTION_OFFSET * N + pix_id];
            float dL_dnormal2D[3];
            for (int i = 0; i < 3; i++) dL_dnormal2D[i] = dL_depths[(NORMAL_OFFSET + i) * N + pix_id];
            float dL_dmedian_depth = dL_depths[MIDDEPTH_OFFSET * N + pix_id];
            float last_depth = 0, last_normal[3] = {0, 0, 0};
            float accum_depth_rec = 0, accum_alpha_rec = 0, accum_normal_rec[3] = {0, 0, 0};
            const float final_D = s->final_T[pix_id + N];
            const float final_D2 = s->final_T[pix_id + 2 * N];
            const float final_A = 1 - T_final;
            float last_dL_dT = 0;
            for (int i = 0; i < 3; i++) dL_dpixel[i] = dL_dpixels[i * N + pix_id];
            float last_alpha = 0, last_color[3] = {0, 0, 0};

            for (uint32_t it = 0; it < r1 - r0; it++) {
                contributor--;
                if (contributor >= (uint32_t)last_contributor) continue;
                uint32_t id = s->point_list[r1 - it - 1];
                const float *Tm = transMats + 9 * (size_t)id;
                const float *Tw = Tm + 6;
                const float *no = s->normal_opacity + 4 * (size_t)id;
                PairEval e;
                if (!eval_pair(pxf, pyf, s->means2D + 2 * (size_t)id, Tm, no[3], &e)) continue;
                const float G = e.G, alpha = e.alpha, c_d = e.depth;
                T = T / (1.f - alpha);
                const float dchannel_dcolor = alpha * T;
                float dL_dalpha = 0.0f;
                for (int ch = 0; ch < 3; ch++) {
                    const float c = colors[3 * (size_t)id + ch];
                    accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                    last_color[ch] = c;
                    const float dL_dchannel = dL_dpixel[ch];
                    dL_dalpha += (c - accum_rec[ch]) * dL_dchannel;
#pragma omp atomic
                    dL_dcolors[3 * (size_t)id + ch] += (double)(dchannel_dcolor * dL_dchannel);
                }
                float dL_dz = 0.0f, dL_dweight = 0;
                const float m_d = mscale * (1 - near_n / c_d);
                const float dmd_dd = (far_n * near_n) / ((far_n - near_n) * c_d * c_d);
                if (contributor == (uint32_t)(median_contributor - 1)) dL_dz += dL_dmedian_depth;
                dL_dweight += (final_D2 + m_d * m_d * final_A - 2 * m_d * final_D) * dL_dreg;
                dL_dalpha += dL_dweight - last_dL_dT;
                last_dL_dT = dL_dweight * alpha + (1 - alpha) * last_dL_dT;
                const float dL_dmd = 2.0f * (T * alpha) * (m_d * final_A - final_D) * dL_dreg;
                dL_dz += dL_dmd * dmd_dd;

                accum_depth_rec = last_alpha * last_depth + (1.f - last_alpha) * accum_depth_rec;
                last_depth = c_d;
                dL_dalpha += (c_d - accum_depth_rec) * dL_ddepth;
                /* backward.cu:366 mixes in a double literal (last_alpha * 1.0) */
                accum_alpha_rec = (float)((double)last_alpha * 1.0 + (double)((1.f - last_alpha) * accum_alpha_rec));
                dL_dalpha += (1 - accum_alpha_rec) * dL_daccum;
                for (int ch = 0; ch < 3; ch++) {
                    accum_normal_rec[ch] = last_alpha * last_normal[ch] + (1.f - last_alpha) * accum_normal_rec[ch];
                    last_normal[ch] = no[ch];
                    dL_dalpha += (no[ch] - accum_normal_rec[ch]) * dL_dnormal2D[ch];
#pragma omp atomic
                    dL_dnormal3D[3 * (size_t)id + ch] += (double)(alpha * T * dL_dnormal2D[ch]);
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                float bg_dot_dpixel = 0;
                for (int i = 0; i < 3; i++) bg_dot_dpixel += bg[i] * dL_dpixel[i];
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;
                const float dL_dG = no[3] * dL_dalpha; /* no gating when alpha was clamped (quirk) */
                dL_dz += alpha * T * dL_ddepth;

                if (e.rho3d <= e.rho2d) {
                    const float dL_dsx = dL_dG * -G * e.sx + dL_dz * Tw[0];
                    const float dL_dsy = dL_dG * -G * e.sy + dL_dz * Tw[1];
                    const float dsx_pz = dL_dsx / e.pz, dsy_pz = dL_dsy / e.pz;
                    const float dpx = dsx_pz, dpy = dsy_pz, dpz = -(dsx_pz * e.sx + dsy_pz * e.sy);
                    /* dL_dk = cross(l, dL_dp) ; dL_dl = cross(dL_dp, k) */
                    const float dkx = e.ly * dpz - e.lz * dpy, dky = e.lz * dpx - e.lx * dpz, dkz = e.lx * dpy - e.ly * dpx;
                    const float dlx = dpy * e.kz - dpz * e.ky, dly = dpz * e.kx - dpx * e.kz, dlz = dpx * e.ky - dpy * e.kx;
                    const float g[9] = {-dkx, -dky, -dkz, -dlx, -dly, -dlz,
                                        pxf * dkx + pyf * dlx + dL_dz * e.sx,
                                        pxf * dky + pyf * dly + dL_dz * e.sy,
                                        pxf * dkz + pyf * dlz + dL_dz * 1.0f};
                    for (int q = 0; q < 9; q++) {
#pragma omp atomic
                        dL_dtransMat[9 * (size_t)id + q] += (double)g[q];
                    }
                } else {
                    const float dG_ddelx = -G * FilterInvSquare * e.dx;
                    const float dG_ddely = -G * FilterInvSquare * e.dy;
#pragma omp atomic
                    dL_dmean2D[3 * (size_t)id + 0] += (double)(dL_dG * dG_ddelx);
#pragma omp atomic
                    dL_dmean2D[3 * (size_t)id + 1] += (double)(dL_dG * dG_ddely);
#pragma omp atomic
                    dL_dtransMat[9 * (size_t)id + 8] += (double)dL_dz;
                }
#pragma omp atomic
                dL_dopacity[id] += (double)(G * dL_dalpha);
            }
        }
    }
}

/* dsr/cuda_rasterizer/auxiliary.h:130-140 */
static inline f3 dnormvdv3(f3 v, f3 dv) {
    float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    f3 r;
    r.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;
    r.y = (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32;
    r.z = (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32;
    return r;
}

/* dsr/cuda_rasterizer/backward.cu:20-139 (SH backward); adds into dL_dmeans */
static void sh_backward(int idx, int deg, int max_coeffs, const float *means, const float *campos,
                        const float *shs, const uint8_t *clamped, const float *dL_dcolor, float *dL_dmeans,
                        float *dL_dshs) {
    f3 dir_orig = {means[3 * idx] - campos[0], means[3 * idx + 1] - campos[1], means[3 * idx + 2] - campos[2]};
    float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    const float *sh = shs + (size_t)idx * max_coeffs * 3;
    float *dsh = dL_dshs + (size_t)idx * max_coeffs * 3;
    float dRGB[3];
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[3 * (size_t)idx + c] * (clamped[3 * idx + c] ? 0.0f : 1.0f);
    float dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
#define SH(i, c) sh[(i) * 3 + (c)]
#define DSH(i, v) for (int c = 0; c < 3; c++) dsh[(i) * 3 + c] = (v) * dRGB[c]
    DSH(0, SH_C0);
    if (deg > 0) {
        float d1 = -SH_C1 * y, d2 = SH_C1 * z, d3 = -SH_C1 * x;
        DSH(1, d1); DSH(2, d2); DSH(3, d3);
        for (int c = 0; c < 3; c++) { dx[c] = -SH_C1 * SH(3, c); dy[c] = -SH_C1 * SH(1, c); dz[c] = SH_C1 * SH(2, c); }
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            float d4 = SH_C2[0] * xy, d5 = SH_C2[1] * yz, d6 = SH_C2[2] * (2.f * zz - xx - yy);
            float d7 = SH_C2[3] * xz, d8 = SH_C2[4] * (xx - yy);
            DSH(4, d4); DSH(5, d5); DSH(6, d6); DSH(7, d7); DSH(8, d8);
            for (int c = 0; c < 3; c++) {
                dx[c] += SH_C2[0] * y * SH(4, c) + SH_C2[2] * 2.f * -x * SH(6, c) + SH_C2[3] * z * SH(7, c) + SH_C2[4] * 2.f * x * SH(8, c);
                dy[c] += SH_C2[0] * x * SH(4, c) + SH_C2[1] * z * SH(5, c) + SH_C2[2] * 2.f * -y * SH(6, c) + SH_C2[4] * 2.f * -y * SH(8, c);
                dz[c] += SH_C2[1] * y * SH(5, c) + SH_C2[2] * 2.f * 2.f * z * SH(6, c) + SH_C2[3] * x * SH(7, c);
            }
            if (deg > 2) {
                float d9 = SH_C3[0] * y * (3.f * xx - yy), d10 = SH_C3[1] * xy * z;
                float d11 = SH_C3[2] * y * (4.f * zz - xx - yy), d12 = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                float d13 = SH_C3[4] * x * (4.f * zz - xx - yy), d14 = SH_C3[5] * z * (xx - yy);
                float d15 = SH_C3[6] * x * (xx - 3.f * yy);
                DSH(9, d9); DSH(10, d10); DSH(11, d11); DSH(12, d12); DSH(13, d13); DSH(14, d14); DSH(15, d15);
                for (int c = 0; c < 3; c++) {
                    dx[c] += (SH_C3[0] * SH(9, c) * 3.f * 2.f * xy + SH_C3[1] * SH(10, c) * yz +
                              SH_C3[2] * SH(11, c) * -2.f * xy + SH_C3[3] * SH(12, c) * -3.f * 2.f * xz +
                              SH_C3[4] * SH(13, c) * (-3.f * xx + 4.f * zz - yy) + SH_C3[5] * SH(14, c) * 2.f * xz +
                              SH_C3[6] * SH(15, c) * 3.f * (xx - yy));
                    dy[c] += (SH_C3[0] * SH(9, c) * 3.f * (xx - yy) + SH_C3[1] * SH(10, c) * xz +
                              SH_C3[2] * SH(11, c) * (-3.f * yy + 4.f * zz - xx) + SH_C3[3] * SH(12, c) * -3.f * 2.f * yz +
                              SH_C3[4] * SH(13, c) * -2.f * xy + SH_C3[5] * SH(14, c) * -2.f * yz +
                              SH_C3[6] * SH(15, c) * -3.f * 2.f * xy);
                    dz[c] += (SH_C3[1] * SH(10, c) * xy + SH_C3[2] * SH(11, c) * 4.f * 2.f * yz +
                              SH_C3[3] * SH(12, c) * 3.f * (2.f * zz - xx - yy) + SH_C3[4] * SH(13, c) * 4.f * 2.f * xz +
                              SH_C3[5] * SH(14, c) * (xx - yy));
                }
            }
        }
    }
#undef SH
#undef DSH
    /* glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z */
    f3 dL_ddir = {dx[0] * dRGB[0] + dx[1] * dRGB[1] + dx[2] * dRGB[2],
                  dy[0] * dRGB[0] + dy[1] * dRGB[1] + dy[2] * dRGB[2],
                  dz[0] * dRGB[0] + dz[1] * dRGB[1] + dz[2] * dRGB[2]};
    f3 dm = dnormvdv3(dir_orig, dL_ddir);
    dL_dmeans[3 * idx] += dm.x; dL_dmeans[3 * idx + 1] += dm.y; dL_dmeans[3 * idx + 2] += dm.z;
}

/* dsr/cuda_rasterizer/backward.cu:443-584 (compute_transmat_aabb), glm products expanded in
 * glm's summation order (type_mat4x4.inl / type_mat4x3.inl / type_mat3x4.inl). */
static void transmat_backward(int idx, const float *Ts_precomp, const float *means3D, const float *scales,
                              const float *rots, const float *proj, const float *view, int W, int H,
                              const float *dL_dnormals, const float *dL_dmean2Ds, float *dL_dTs,
                              float *dL_dmeans, float *dL_dscales, float *dL_drots) {
    float T[9]; /* T[c*3+r] : column c = stored row (Tu,Tv,Tw) */
    f3 normal = {0, 0, 0}, p_orig = {0, 0, 0};
    float Pm[3][4]; float R[9]; float sx = 0, sy = 0; const float *rot = NULL;
    if (Ts_precomp != NULL) {
        memcpy(T, Ts_precomp + 9 * (size_t)idx, sizeof(T));
    } else {
        p_orig.x = means3D[3 * idx]; p_orig.y = means3D[3 * idx + 1]; p_orig.z = means3D[3 * idx + 2];
        rot = rots + 4 * idx; sx = scales[2 * idx]; sy = scales[2 * idx + 1];
        quat_to_rotmat(rot, R);
        /* S = scale_to_mat(scale, 1.0f): scale_modifier is ignored in backward (quirk) */
        float s0 = 1.0f * sx, s1 = 1.0f * sy;
        float M[3][4] = {{R[0] * s0, R[1] * s0, R[2] * s0, 0.0f}, {R[3] * s1, R[4] * s1, R[5] * s1, 0.0f},
                         {p_orig.x, p_orig.y, p_orig.z, 1.0f}};
        float hw = (float)((float)W / 2.0), cw = (float)((float)(W - 1) / 2.0);
        float hh = (float)((float)H / 2.0), ch = (float)((float)(H - 1) / 2.0);
        for (int i = 0; i < 4; i++) { /* P = world2ndc * ndc2pix */
            Pm[0][i] = proj[4 * i + 0] * hw + proj[4 * i + 3] * cw;
            Pm[1][i] = proj[4 * i + 1] * hh + proj[4 * i + 3] * ch;
            Pm[2][i] = proj[4 * i + 3];
        }
        for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) /* T = transpose(M) * P */
            T[c * 3 + r] = M[r][0] * Pm[c][0] + M[r][1] * Pm[c][1] + M[r][2] * Pm[c][2] + M[r][3] * Pm[c][3];
        f3 l2 = {R[6], R[7], R[8]};
        normal = transformVec4x3(l2, view);
    }
    float dL_dT[9];
    memcpy(dL_dT, dL_dTs + 9 * (size_t)idx, sizeof(dL_dT));
    float mx = dL_dmean2Ds[3 * idx], my = dL_dmean2Ds[3 * idx + 1];
    if (mx != 0 || my != 0) {
        const float *T0 = T, *T1 = T + 3, *T2 = T + 6;
        const float distance = T2[0] * T2[0] + T2[1] * T2[1] - T2[2] * T2[2];
        const float f = 1 / (distance);
        const float dpx_dT00 = f * T2[0], dpx_dT01 = f * T2[1], dpx_dT02 = -f * T2[2];
        const float dpy_dT10 = f * T2[0], dpy_dT11 = f * T2[1], dpy_dT12 = -f * T2[2];
        const float dpx_dT30 = T0[0] * (f - 2 * f * f * T2[0] * T2[0]);
        const float dpx_dT31 = T0[1] * (f - 2 * f * f * T2[1] * T2[1]);
        const float dpx_dT32 = -T0[2] * (f + 2 * f * f * T2[2] * T2[2]);
        const float dpy_dT30 = T1[0] * (f - 2 * f * f * T2[0] * T2[0]);
        const float dpy_dT31 = T1[1] * (f - 2 * f * f * T2[1] * T2[1]);
        const float dpy_dT32 = -T1[2] * (f + 2 * f * f * T2[2] * T2[2]);
        dL_dT[0] += mx * dpx_dT00; dL_dT[1] += mx * dpx_dT01; dL_dT[2] += mx * dpx_dT02;
        dL_dT[3] += my * dpy_dT10; dL_dT[4] += my * dpy_dT11; dL_dT[5] += my * dpy_dT12;
        dL_dT[6] += mx * dpx_dT30 + my * dpy_dT30;
        dL_dT[7] += mx * dpx_dT31 + my * dpy_dT31;
        dL_dT[8] += mx * dpx_dT32 + my * dpy_dT32;
        if (Ts_precomp != NULL) { memcpy(dL_dTs + 9 * (size_t)idx, dL_dT, sizeof(dL_dT)); return; }
    }
    if (Ts_precomp != NULL) return;
    /* dL_dM = P * transpose(dL_dT): dL_dM[c][i] = sum_k P[k][i] * dL_dT[k][c] */
    float dL_dM[3][4];
    for (int c = 0; c < 3; c++) for (int i = 0; i < 4; i++)
        dL_dM[c][i] = Pm[0][i] * dL_dT[0 * 3 + c] + Pm[1][i] * dL_dT[1 * 3 + c] + Pm[2][i] * dL_dT[2 * 3 + c];
    f3 dn = {dL_dnormals[3 * idx], dL_dnormals[3 * idx + 1], dL_dnormals[3 * idx + 2]};
    f3 dL_dtn = transformVec4x3Transpose(dn, view);
    f3 p_view = transformPoint4x3(p_orig, view);
    float cosv = -((p_view.x * normal.x + p_view.y * normal.y) + p_view.z * normal.z);
    float multiplier = cosv > 0 ? 1.0f : -1.0f;
    dL_dtn.x = multiplier * dL_dtn.x; dL_dtn.y = multiplier * dL_dtn.y; dL_dtn.z = multiplier * dL_dtn.z;
    float dL_dRS[9] = {dL_dM[0][0], dL_dM[0][1], dL_dM[0][2], dL_dM[1][0], dL_dM[1][1], dL_dM[1][2],
                       dL_dtn.x, dL_dtn.y, dL_dtn.z};
    float dL_dR[9] = {dL_dRS[0] * sx, dL_dRS[1] * sx, dL_dRS[2] * sx, dL_dRS[3] * sy, dL_dRS[4] * sy,
                      dL_dRS[5] * sy, dL_dRS[6], dL_dRS[7], dL_dRS[8]};
    quat_to_rotmat_vjp(rot, dL_dR, dL_drots + 4 * (size_t)idx);
    dL_dscales[2 * idx] = dL_dRS[0] * R[0] + dL_dRS[1] * R[1] + dL_dRS[2] * R[2];
    dL_dscales[2 * idx + 1] = dL_dRS[3] * R[3] + dL_dRS[4] * R[4] + dL_dRS[5] * R[5];
    dL_dmeans[3 * idx] = dL_dM[2][0]; dL_dmeans[3 * idx + 1] = dL_dM[2][1]; dL_dmeans[3 * idx + 2] = dL_dM[2][2];
}

/* dsr/cuda_rasterizer/rasterizer_impl.cu:346-448 (Rasterizer::backward),
 * dsr/cuda_rasterizer/backward.cu:586-641 (preprocessCUDA backward) and
 * dsr/rasterize_points.cu:136-233 (zero-init of the nine outputs).
 * Uses the state left by oracle_forward.  All outputs are [P,*] float arrays. */
void oracle_backward(OracleState *s, int P, int D, int M, const float *background, int width, int height,
                     const float *means3D, const float *shs, const float *colors_precomp, const float *scales,
                     float scale_modifier, const float *rotations, const float *transMat_precomp,
                     const float *viewmatrix, const float *projmatrix, const float *campos, float tan_fovx,
                     float tan_fovy, const float *dL_dpix, const float *dL_depths, float *dL_dmean2D,
                     float *dL_dnormal, float *dL_dopacity, float *dL_dcolor, float *dL_dmean3D,
                     float *dL_dtransMat, float *dL_dsh, float *dL_dscale, float *dL_drot) {
    (void)scale_modifier;
    memset(dL_dmean2D, 0, 3 * (size_t)P * sizeof(float)); memset(dL_dnormal, 0, 3 * (size_t)P * sizeof(float));
    memset(dL_dopacity, 0, (size_t)P * sizeof(float)); memset(dL_dcolor, 0, 3 * (size_t)P * sizeof(float));
    memset(dL_dmean3D, 0, 3 * (size_t)P * sizeof(float)); memset(dL_dtransMat, 0, 9 * (size_t)P * sizeof(float));
    if (M > 0) memset(dL_dsh, 0, 3 * (size_t)P * M * sizeof(float));
    memset(dL_dscale, 0, 2 * (size_t)P * sizeof(float)); memset(dL_drot, 0, 4 * (size_t)P * sizeof(float));
    if (P == 0) return;
    const float focal_y = height / (2.0f * tan_fovy);
    const float focal_x = width / (2.0f * tan_fovx);
    const float *color_ptr = colors_precomp ? colors_precomp : s->rgb;
    const float *transMat_ptr = transMat_precomp ? transMat_precomp : s->transMat;

    double *aT = (double *)calloc(9 * (size_t)P, sizeof(double)), *aM = (double *)calloc(3 * (size_t)P, sizeof(double));
    double *aN = (double *)calloc(3 * (size_t)P, sizeof(double)), *aO = (double *)calloc(P, sizeof(double));
    double *aC = (double *)calloc(3 * (size_t)P, sizeof(double));
    render_bwd(s, background, color_ptr, transMat_ptr, dL_dpix, dL_depths, aT, aM, aN, aO, aC);
    for (size_t i = 0; i < 9 * (size_t)P; i++) dL_dtransMat[i] = (float)aT[i];
    for (size_t i = 0; i < 3 * (size_t)P; i++) { dL_dmean2D[i] = (float)aM[i]; dL_dnormal[i] = (float)aN[i]; dL_dcolor[i] = (float)aC[i]; }
    for (size_t i = 0; i < (size_t)P; i++) dL_dopacity[i] = (float)aO[i];
    free(aT); free(aM); free(aN); free(aO); free(aC);
    free(s->dL_dtransMat_raw); free(s->dL_dnormal3D); free(s->dL_dmean2D_raw);
    s->dL_dmean2D_raw = (float *)malloc(3 * (size_t)P * sizeof(float));
    memcpy(s->dL_dmean2D_raw, dL_dmean2D, 3 * (size_t)P * sizeof(float));
    s->dL_dtransMat_raw = (float *)malloc(9 * (size_t)P * sizeof(float));
    s->dL_dnormal3D = (float *)malloc(3 * (size_t)P * sizeof(float));
    memcpy(s->dL_dtransMat_raw, dL_dtransMat, 9 * (size_t)P * sizeof(float));
    memcpy(s->dL_dnormal3D, dL_dnormal, 3 * (size_t)P * sizeof(float));

    /* backward.cu:618-619 : W,H re-derived by float truncation (quirk: can be W-1 / H-1) */
    const int Wb = (int)(focal_x * tan_fovx * 2);
    const int Hb = (int)(focal_y * tan_fovy * 2);
    const float *Ts_precomp = scales ? NULL : transMat_ptr;
#pragma omp parallel for schedule(static)
    for (int idx = 0; idx < P; idx++) {
        if (!(s->radii[idx] > 0)) continue;
        transmat_backward(idx, Ts_precomp, means3D, scales, rotations, projmatrix, viewmatrix, Wb, Hb, dL_dnormal,
                          dL_dmean2D, dL_dtransMat, dL_dmean3D, dL_dscale, dL_drot);
        if (shs) sh_backward(idx, D, M, means3D, campos, shs, s->clamped, dL_dcolor, dL_dmean3D, dL_dsh);
        /* densification surrogate, backward.cu:637-640 */
        float depth = transMat_ptr[9 * (size_t)idx + 8];
        dL_dmean2D[3 * idx] = (float)(dL_dtransMat[9 * (size_t)idx + 2] * depth * 0.5 * (float)Wb);
        dL_dmean2D[3 * idx + 1] = (float)(dL_dtransMat[9 * (size_t)idx + 5] * depth * 0.5 * (float)Hb);
    }
}

/* ------------------------------------------------------------------------------------ */
/* simple-knn                                                                             */

/* knn/simple_knn.cu:131-183 defines the result: for every point, the mean of the three
 * smallest squared distances to points with a different index (duplicates give 0, fewer
 * than three neighbours leave FLT_MAX terms).  The Morton ordering and box pruning of the
 * reference (simple_knn.cu:54-117,147-181) are an exact search, so the result is
 * algorithm independent; this restatement is the brute-force definition with the
 * reference's distance expression and updateKBest<3> insertion (simple_knn.cu:131-145). */
void oracle_knn(int P, const float *points, float *meanDists) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < P; i++) {
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        float rx = points[3 * i], ry = points[3 * i + 1], rz = points[3 * i + 2];
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            float dx = points[3 * j] - rx, dy = points[3 * j + 1] - ry, dz = points[3 * j + 2] - rz;
            float dist = dx * dx + dy * dy + dz * dz;
            for (int k = 0; k < 3; k++) if (best[k] > dist) { float t = best[k]; best[k] = dist; dist = t; }
        }
        meanDists[i] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}

/* ------------------------------------------------------------------------------------ */
/* Workload statistics (not part of the restatement): how many (pixel, splat) pairs of the
 * tile lists are evaluated / pass the alpha test / are blended, at pixel, 8x8-quadrant and
 * tile granularity.  Used to size the wave-level skipping of the HIP kernels. */
void oracle_pair_stats(const OracleState *s, double *out /*8*/) {
    const int W = s->W, H = s->H, gx = s->tiles_x, gy = s->tiles_y;
    const size_t N = (size_t)W * H;
    double pairs = 0, pass = 0, blended = 0, quad_any_pass = 0, quad_total = 0, tile_any_pass = 0, tile_total = 0, quad_any_blend = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+:pairs,pass,blended,quad_any_pass,quad_total,tile_any_pass,tile_total,quad_any_blend)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (uint32_t i = r0; i < r1; i++) {
            uint32_t id = s->point_list[i];
            const float *no = s->normal_opacity + 4 * (size_t)id;
            int tile_any = 0;
            for (int q = 0; q < 4; q++) {
                int qa = 0, qb = 0;
                for (int l = 0; l < 64; l++) {
                    int px = tx * 16 + (q & 1) * 8 + (l & 7), py = ty * 16 + (q >> 1) * 8 + (l >> 3);
                    if (px >= W || py >= H) continue;
                    size_t pix = (size_t)W * py + px;
                    pairs += 1;
                    PairEval e;
                    if (!eval_pair((float)px, (float)py, s->means2D + 2 * (size_t)id, s->transMat + 9 * (size_t)id, no[3], &e)) continue;
                    pass += 1; qa = 1;
                    if ((i - r0) < s->n_contrib[pix]) { blended += 1; qb = 1; }
                }
                quad_total += 1; quad_any_pass += qa; quad_any_blend += qb; tile_any |= qb;
            }
            tile_total += 1; tile_any_pass += tile_any;
        }
    }
    (void)N;
    out[0] = pairs; out[1] = pass; out[2] = blended; out[3] = quad_total; out[4] = quad_any_pass;
    out[5] = quad_any_blend; out[6] = tile_total; out[7] = tile_any_pass;
}

/* Per-instance flag (test infrastructure): 1 if at least one pixel of the instance's tile passes
 * the per-pixel tests of eval_pair for that splat (p.z != 0, depth >= near, alpha >= 1/255),
 * i.e. the instance is NOT provably dead.  The HIP path may drop exactly the instances whose flag
 * is 0 (alpha-cutoff tile culling); tests assert it never drops a flagged one. */
void oracle_instance_flags(const OracleState *s, uint8_t *flags /*R*/) {
    const int W = s->W, H = s->H, gx = s->tiles_x, gy = s->tiles_y;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (uint32_t i = r0; i < r1; i++) {
            uint32_t id = s->point_list[i];
            const float *no = s->normal_opacity + 4 * (size_t)id;
            uint8_t f = 0;
            for (int l = 0; l < 256 && !f; l++) {
                int px = tx * 16 + (l & 15), py = ty * 16 + (l >> 4);
                if (px >= W || py >= H) continue;
                PairEval e;
                if (eval_pair((float)px, (float)py, s->means2D + 2 * (size_t)id, s->transMat + 9 * (size_t)id, no[3], &e)) f = 1;
            }
            flags[i] = f;
        }
    }
}

/* Decision margins per pixel (test infrastructure, not part of the restatement).
 *
 * The per-pixel loop of forward.cu:346-433 takes three kinds of discrete decisions on computed floats:
 *   (0) skip the splat:  alpha < 1/255 (forward.cu:389) or depth < near (forward.cu:379)
 *   (1) stop the pixel:  T (1 - alpha) < 1e-4 (forward.cu:394-399)
 *   (2) median depth:    T > 0.5 (forward.cu:412-417)
 * Any two correct single-precision evaluations of the same formulas (this file with libm's expf and IEEE
 * division; the HIP kernels with v_exp_f32 / v_rcp_f32; the CUDA reference with its own expf) differ by ~1e-6
 * relative in alpha and T, so a pixel whose value sits within that distance of a threshold may legitimately
 * fall on either side.  This function re-walks every pixel's list exactly like render_fwd and records, per
 * decision kind, the smallest relative distance |value - threshold| / threshold met on the way:
 * out[k * N + pix], k = 0..2 (FLT_MAX if the decision never came up).  Parity tests use it to PROVE that every
 * pixel whose outputs differ from the HIP path beyond the tolerance sits on such a threshold
 * (tests/common.py::explain_mismatches). */
void oracle_pixel_margins(const OracleState *s, const float *transMats, float *out /*3N*/) {
    const int W = s->W, H = s->H, gx = s->tiles_x, gy = s->tiles_y;
    const size_t N = (size_t)W * H;
    if (!transMats) transMats = s->transMat;
    for (size_t i = 0; i < 3 * N; i++) out[i] = FLT_MAX;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
            int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (px >= W || py >= H) continue;
            size_t pix_id = (size_t)W * py + px;
            float pxf = (float)px, pyf = (float)py;
            float T = 1.0f;
            float m0 = FLT_MAX, m1 = FLT_MAX, m2 = FLT_MAX;
            for (uint32_t i = r0; i < r1; i++) {
                uint32_t id = s->point_list[i];
                const float *no = s->normal_opacity + 4 * (size_t)id;
                PairEval e;
                memset(&e, 0, sizeof e);
                int ok = eval_pair(pxf, pyf, s->means2D + 2 * (size_t)id, transMats + 9 * (size_t)id, no[3], &e);
                if (e.pz != 0.0f) {  /* p.z == 0 is exact on every side (same fma sequence) */
                    float md = fabsf(e.depth - near_n) / near_n;
                    if (md < m0) m0 = md;
                    if (e.depth >= near_n) {
                        float a = no[3] * expf(-0.5f * fminf(e.rho3d, e.rho2d));
                        float ma = fabsf(a - 1.0f / 255.0f) * 255.0f;
                        if (ma < m0) m0 = ma;
                    }
                }
                if (!ok) continue;
                float test_T = T * (1 - e.alpha);
                float mt = fabsf(test_T - 0.0001f) / 0.0001f;
                if (mt < m1) m1 = mt;
                if (test_T < 0.0001f) break;
                float mh = fabsf(T - 0.5f) / 0.5f;
                if (mh < m2) m2 = mh;
                T = test_T;
            }
            out[pix_id] = m0; out[pix_id + N] = m1; out[pix_id + 2 * N] = m2;
        }
    }
}

/* The (pixel, Gaussian) pairs whose SKIP decision (forward.cu:378 depth >= near, :391 alpha >= 1/255) sits within
 * `margin` (relative, as in oracle_pixel_margins) of its threshold and is reached before the pixel stops.  Taken the other
 * way such a decision adds or removes one blended splat of weight ~T/255: invisible in the outputs below a few list
 * positions, but it is the WHOLE contribution of that pixel to that Gaussian's gradient rows (tests/common.py).
 * Returns the number of pairs; the first min(count, max_n) are stored (order unspecified). */
long oracle_skip_suspects(const OracleState *s, const float *transMats, float margin, long max_n, int64_t *out_pix,
                          int32_t *out_gid, float *out_margins /* 3N as oracle_pixel_margins writes them, or NULL */) {
    const int W = s->W, H = s->H, gx = s->tiles_x, gy = s->tiles_y;
    const size_t N = (size_t)W * H;
    if (!transMats) transMats = s->transMat;
    long count = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        for (int ly = 0; ly < BLOCK_Y; ly++) for (int lx = 0; lx < BLOCK_X; lx++) {
            int px = tx * BLOCK_X + lx, py = ty * BLOCK_Y + ly;
            if (px >= W || py >= H) continue;
            float pxf = (float)px, pyf = (float)py;
            float T = 1.0f;
            float m0 = FLT_MAX, m1 = FLT_MAX, m2 = FLT_MAX;  /* the three rows of oracle_pixel_margins */
            for (uint32_t i = r0; i < r1; i++) {
                uint32_t id = s->point_list[i];
                const float *no = s->normal_opacity + 4 * (size_t)id;
                PairEval e;
                memset(&e, 0, sizeof e);
                int ok = eval_pair(pxf, pyf, s->means2D + 2 * (size_t)id, transMats + 9 * (size_t)id, no[3], &e);
                if (e.pz != 0.0f) {
                    float m = fabsf(e.depth - near_n) / near_n;
                    if (e.depth >= near_n) {
                        float a = no[3] * expf(-0.5f * fminf(e.rho3d, e.rho2d));
                        m = fminf(m, fabsf(a - 1.0f / 255.0f) * 255.0f);
                    }
                    if (m < m0) m0 = m;
                    if (m < margin) {
                        long k;
#pragma omp atomic capture
                        k = count++;
                        if (k < max_n) { out_pix[k] = (int64_t)W * py + px; out_gid[k] = (int32_t)id; }
                    }
                }
                if (!ok) continue;
                float test_T = T * (1 - e.alpha);
                float mt = fabsf(test_T - 0.0001f) / 0.0001f;
                if (mt < m1) m1 = mt;
                if (test_T < 0.0001f) break;
                float mh = fabsf(T - 0.5f) / 0.5f;
                if (mh < m2) m2 = mh;
                T = test_T;
            }
            if (out_margins) {
                size_t pix_id = (size_t)W * py + px;
                out_margins[pix_id] = m0; out_margins[pix_id + N] = m1; out_margins[pix_id + 2 * N] = m2;
            }
        }
    }
    return count;
}

/* What a pixel looks like when its near-threshold decisions fall the other way (tests/common.py: the check on flipped
 * pixels).  The forward loop of render_fwd is walked again for each of the n pixels in pix_ids, 2^PIXEL_ALT_BITS times:
 * in walk b, the d-th decision met that sits within `margin` (relative) of its threshold -- the same three kinds and the
 * same distances as oracle_pixel_margins -- is taken the OTHER way if bit d of b is set (d < PIXEL_ALT_BITS; later ones
 * are taken as computed).  Walk 0 is the frame as rendered.  out[(k * ALT + b) * 12 + m]: m = 0..2 colour, 3..9 the
 * seven `others` maps, 10 = last contributor's Gaussian id (-1: none), 11 = median contributor's Gaussian id (-1). */
#define PIXEL_ALT_BITS 4
int oracle_pixel_alt_count(void) { return 1 << PIXEL_ALT_BITS; }
void oracle_pixel_alternatives(const OracleState *s, const float *features, const float *transMats, const float *bg,
                               float margin, int n, const int64_t *pix_ids, float *out) {
    const int W = s->W, gx = s->tiles_x;
    const float mscale = far_n / (far_n - near_n);
    const int ALT = 1 << PIXEL_ALT_BITS;
    if (!transMats) transMats = s->transMat;
    if (!features) features = s->rgb;
#pragma omp parallel for schedule(dynamic, 1)
    for (int k = 0; k < n; k++) {
        const int px = (int)(pix_ids[k] % W), py = (int)(pix_ids[k] / W);
        const int tile = (py / BLOCK_Y) * gx + px / BLOCK_X;
        const uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        const float pxf = (float)px, pyf = (float)py;
        for (int b = 0; b < ALT; b++) {
            int d = 0;  /* near-threshold decisions met so far in this walk */
            float T = 1.0f;
            float C[3] = {0, 0, 0}, Nn[3] = {0, 0, 0};
            float Dd = 0, M1 = 0, M2 = 0, distortion = 0, median_depth = 0;
            long last_id = -1, median_id = -1;
#define FLIP_IF_CLOSE(decision, dist) do { if ((dist) < margin) { if (d < PIXEL_ALT_BITS && ((b >> d) & 1)) (decision) = !(decision); d++; } } while (0)
            for (uint32_t i = r0; i < r1; i++) {
                uint32_t id = s->point_list[i];
                const float *no = s->normal_opacity + 4 * (size_t)id;
                PairEval e;
                memset(&e, 0, sizeof e);
                eval_pair(pxf, pyf, s->means2D + 2 * (size_t)id, transMats + 9 * (size_t)id, no[3], &e);
                if (e.pz == 0.0f) continue;  /* exact on every side */
                /* forward.cu:378 depth >= near, :391 alpha >= 1/255 (eval_pair leaves depth, rho3d, rho2d behind even
                 * where it rejected the pair) */
                int near_ok = !(e.depth < near_n);
                FLIP_IF_CLOSE(near_ok, fabsf(e.depth - near_n) / near_n);
                if (!near_ok) continue;
                float G = expf(-0.5f * fminf(e.rho3d, e.rho2d));
                float alpha = fminf(0.99f, no[3] * G);
                int alpha_ok = !(alpha < 1.0f / 255.0f);
                FLIP_IF_CLOSE(alpha_ok, fabsf(no[3] * G - 1.0f / 255.0f) * 255.0f);
                if (!alpha_ok) continue;
                float depth = e.depth;
                float test_T = T * (1 - alpha);
                int go_on = !(test_T < 0.0001f);
                FLIP_IF_CLOSE(go_on, fabsf(test_T - 0.0001f) / 0.0001f);
                if (!go_on) break;
                float w = alpha * T;
                float A = 1 - T;
                float m = mscale * (1 - near_n / depth);
                distortion += (m * m * A + M2 - 2 * m * M1) * w;
                Dd += depth * w;
                M1 += m * w;
                M2 += m * m * w;
                int front = T > 0.5f;
                FLIP_IF_CLOSE(front, fabsf(T - 0.5f) / 0.5f);
                if (front) { median_depth = depth; median_id = (long)id; }
                for (int ch = 0; ch < 3; ch++) Nn[ch] += no[ch] * w;
                for (int ch = 0; ch < 3; ch++) C[ch] += features[3 * (size_t)id + ch] * w;
                T = test_T;
                last_id = (long)id;
            }
#undef FLIP_IF_CLOSE
            float *o = out + ((size_t)k * ALT + b) * 12;
            for (int ch = 0; ch < 3; ch++) o[ch] = C[ch] + T * bg[ch];
            o[3 + DEPTH_OFFSET] = Dd;
            o[3 + ALPHA_OFFSET] = 1 - T;
            for (int ch = 0; ch < 3; ch++) o[3 + NORMAL_OFFSET + ch] = Nn[ch];
            o[3 + MIDDEPTH_OFFSET] = median_depth;
            o[3 + DISTORTION_OFFSET] = distortion;
            o[10] = (float)last_id;
            o[11] = (float)median_id;
        }
    }
}

/* knn/simple_knn.cu:131-183 for a SUBSET of query points (all P points are candidates): lets the parity
 * test check distCUDA2 at 3e5 .. 1.5e6 points against the brute-force definition on a few thousand queries. */
void oracle_knn_queries(int P, const float *points, int nq, const int *queries, float *meanDists /*nq*/) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int qi = 0; qi < nq; qi++) {
        int i = queries[qi];
        float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
        float rx = points[3 * i], ry = points[3 * i + 1], rz = points[3 * i + 2];
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            float dx = points[3 * j] - rx, dy = points[3 * j + 1] - ry, dz = points[3 * j + 2] - rz;
            float dist = dx * dx + dy * dy + dz * dz;
            for (int k = 0; k < 3; k++) if (best[k] > dist) { float t = best[k]; best[k] = dist; dist = t; }
        }
        meanDists[qi] = (best[0] + best[1] + best[2]) / 3.0f;
    }
}

/* Lane-utilisation model of the blend backward (tools/lane_util_model.py, profiles/r04_lane_util_model.txt).
 * For every (list entry, tile) that some pixel blends, the 256-bit mask of the pixels that blend it -- exactly the
 * backward's `act` predicate: list position below the pixel's last contributor and the alpha test passed -- is formed and
 * the issue slots different static work units would need are counted.  A "visit" is one pass of the wave (64 lanes) over
 * the per-pixel arithmetic; useful lanes = blending pixels / (64 x visits).
 *   out[0]  entries (with at least one blending pixel)          out[1]  blending (entry, pixel) pairs
 *   out[2]  visits, 8x8 quadrants (the shipped kernel: lane = the same position in each of the four quadrants)
 *   out[3]  visits, 16x4 row strips            out[4]  visits, 4x16 column strips
 *   out[5]  visits if the two half-waves could choose their 8x4 half-quadrant independently (lanes 0..31 own the top
 *           halves of the four quadrants, lanes 32..63 the bottom halves: max(#top halves hit, #bottom halves hit))
 *   out[6]  visits if the four 16-lane rows could choose their 4x4 cell independently (row r owns cell r of each
 *           quadrant: max over r of the quadrants whose cell r is hit)
 *   out[7]  lower bound: ceil(blending pixels / 64) per entry (perfect packing)
 *   out[8]  quadrant visits saved if two CONSECUTIVE blending entries of a tile whose masks are disjoint inside a
 *           quadrant shared one visit there (greedy pairing, back to front)
 *   out[9]  quadrant visits in which at most 32 lanes blend and they all sit in one half (top or bottom 8x4)
 *   out[10] quadrant visits in which the blending lanes sit in a single 16-lane row (8x2 pixels)
 *   out[11] quadrant visits with at most 16 blending lanes, out[12] with at most 8 (gfx950 executes a VALU instruction
 *           with <= 16 enabled lanes in a slow mode: profiles/r04_exec_lane_threshold.txt), out[13] / out[14] the
 *           blending pixels in those visits */
void oracle_lane_model(const OracleState *s, double *out /*15*/) {
    const int W = s->W, H = s->H, gx = s->tiles_x, gy = s->tiles_y;
    double acc[15];
    memset(acc, 0, sizeof acc);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0, a9 = 0, a10 = 0, a11 = 0, a12 = 0, a13 = 0, a14 = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+:a0,a1,a2,a3,a4,a5,a6,a7,a8,a9,a10,a11,a12,a13,a14)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        unsigned char prev[256];
        int have_prev = 0;
        for (uint32_t ii = r1; ii > r0; ii--) {  /* back to front, as the backward walks the list */
            uint32_t i = ii - 1;
            uint32_t id = s->point_list[i];
            const float *no = s->normal_opacity + 4 * (size_t)id;
            unsigned char m[256];
            int n = 0;
            for (int l = 0; l < 256; l++) {
                int lx = l & 15, ly = l >> 4;
                int px = tx * 16 + lx, py = ty * 16 + ly;
                m[l] = 0;
                if (px >= W || py >= H) continue;
                size_t pix = (size_t)W * py + px;
                if ((i - r0) >= s->n_contrib[pix]) continue;
                PairEval e;
                if (!eval_pair((float)px, (float)py, s->means2D + 2 * (size_t)id, s->transMat + 9 * (size_t)id, no[3], &e)) continue;
                m[l] = 1;
                n++;
            }
            if (n == 0) continue;
            a0 += 1; a1 += n; a7 += (n + 63) / 64;
            int quad[4] = {0, 0, 0, 0}, rowstrip[4] = {0, 0, 0, 0}, colstrip[4] = {0, 0, 0, 0};
            int top[4] = {0, 0, 0, 0}, bot[4] = {0, 0, 0, 0}, cell[4][4], rows8x2[4][4];
            memset(cell, 0, sizeof cell);
            memset(rows8x2, 0, sizeof rows8x2);
            for (int l = 0; l < 256; l++) {
                if (!m[l]) continue;
                int lx = l & 15, ly = l >> 4, q = (lx >> 3) + 2 * (ly >> 3);
                quad[q]++;
                rowstrip[ly >> 2] = 1; colstrip[lx >> 2] = 1;
                if ((ly & 7) < 4) top[q]++; else bot[q]++;
                cell[q][((lx >> 2) & 1) + 2 * ((ly >> 2) & 1)] = 1;
                rows8x2[q][(ly & 7) >> 1] = 1;  /* lane = (lx & 7) + 8 (ly & 7): 16-lane row = (ly & 7) >> 1 */
            }
            int nq = 0, ntop = 0, nbot = 0;
            for (int q = 0; q < 4; q++) {
                nq += quad[q] != 0; ntop += top[q] != 0; nbot += bot[q] != 0;
                a3 += rowstrip[q]; a4 += colstrip[q];
                if (quad[q] && quad[q] <= 32 && (top[q] == 0 || bot[q] == 0)) a9 += 1;
                if (quad[q] && rows8x2[q][0] + rows8x2[q][1] + rows8x2[q][2] + rows8x2[q][3] == 1) a10 += 1;
                if (quad[q] && quad[q] <= 16) { a11 += 1; a13 += quad[q]; }
                if (quad[q] && quad[q] <= 8) { a12 += 1; a14 += quad[q]; }
            }
            a2 += nq;
            a5 += ntop > nbot ? ntop : nbot;
            int best = 0;
            for (int r = 0; r < 4; r++) {
                int c = (cell[0][r] != 0) + (cell[1][r] != 0) + (cell[2][r] != 0) + (cell[3][r] != 0);
                if (c > best) best = c;
            }
            a6 += best;
            if (have_prev) {  /* could this entry share its quadrant visits with the previous (deeper) blending entry? */
                int saved = 0;
                for (int q = 0; q < 4; q++) {
                    int both = 0, pq = 0, disjoint = 1;
                    for (int l = 0; l < 256; l++) {
                        int lx = l & 15, ly = l >> 4;
                        if ((lx >> 3) + 2 * (ly >> 3) != q) continue;
                        pq |= prev[l]; both |= m[l];
                        if (prev[l] && m[l]) disjoint = 0;
                    }
                    if (pq && both && disjoint) saved++;
                }
                a8 += saved;
                have_prev = saved ? 0 : 1;  /* greedy: a merged pair is used up */
                if (have_prev) memcpy(prev, m, 256);
            } else {
                memcpy(prev, m, 256);
                have_prev = 1;
            }
        }
    }
    acc[0] = a0; acc[1] = a1; acc[2] = a2; acc[3] = a3; acc[4] = a4; acc[5] = a5; acc[6] = a6; acc[7] = a7; acc[8] = a8;
    acc[9] = a9; acc[10] = a10; acc[11] = a11; acc[12] = a12; acc[13] = a13; acc[14] = a14;
    memcpy(out, acc, sizeof acc);
}

/* Workload statistics at 4x4-cell granularity (design study for the blend forward, not part of the restatement):
 * out[0] = (entry, quadrant) pairs with a pixel that passes the alpha test before the pixel's last contributor,
 * out[1] = (entry, 4x4 cell) pairs with such a pixel, out[2] = such (entry, pixel) pairs,
 * out[3] = sum over quadrants and 256-entry batches of max over the quadrant's four cells of their pair counts
 *          (the steps a wave needs if its four 16-lane rows walk their own cell lists and meet at batch ends). */
void oracle_cell_stats(const OracleState *s, double *out /*4*/) {
    const int W = s->W, H = s->H, gx = s->tiles_x, gy = s->tiles_y;
    double quad_pairs = 0, cell_pairs = 0, pix_pairs = 0, row_steps = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+:quad_pairs,cell_pairs,pix_pairs,row_steps)
    for (int tile = 0; tile < gx * gy; tile++) {
        int tx = tile % gx, ty = tile / gx;
        uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
        int batch_cells[16];
        memset(batch_cells, 0, sizeof batch_cells);
        for (uint32_t i = r0; i < r1; i++) {
            uint32_t id = s->point_list[i];
            const float *no = s->normal_opacity + 4 * (size_t)id;
            int cellhit[16];
            memset(cellhit, 0, sizeof cellhit);
            for (int l = 0; l < 256; l++) {
                int lx = l & 15, ly = l >> 4;
                int px = tx * 16 + lx, py = ty * 16 + ly;
                if (px >= W || py >= H) continue;
                size_t pix = (size_t)W * py + px;
                if ((i - r0) >= s->n_contrib[pix]) continue;
                PairEval e;
                if (!eval_pair((float)px, (float)py, s->means2D + 2 * (size_t)id, s->transMat + 9 * (size_t)id, no[3], &e)) continue;
                pix_pairs += 1;
                /* quadrant q = (lx >> 3) + 2 (ly >> 3); cell inside it = ((lx >> 2) & 1) + 2 ((ly >> 2) & 1) */
                cellhit[((lx >> 3) + 2 * (ly >> 3)) * 4 + ((lx >> 2) & 1) + 2 * ((ly >> 2) & 1)] = 1;
            }
            for (int q = 0; q < 4; q++) {
                int any = 0;
                for (int c = 0; c < 4; c++) { any |= cellhit[4 * q + c]; cell_pairs += cellhit[4 * q + c]; batch_cells[4 * q + c] += cellhit[4 * q + c]; }
                quad_pairs += any;
            }
            if (((i - r0) & 255) == 255 || i + 1 == r1) {
                for (int q = 0; q < 4; q++) {
                    int m = 0;
                    for (int c = 0; c < 4; c++) if (batch_cells[4 * q + c] > m) m = batch_cells[4 * q + c];
                    row_steps += m;
                }
                memset(batch_cells, 0, sizeof batch_cells);
            }
        }
    }
    out[0] = quad_pairs; out[1] = cell_pairs; out[2] = pix_pairs; out[3] = row_steps;
}
