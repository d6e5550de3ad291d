// Per-Gaussian kernels: forward preprocess (K1), near-plane visibility (K9) and the
// per-Gaussian backward (K8).  One thread per Gaussian, streaming; HBM-bound.
//
// Reference semantics: dsr/cuda_rasterizer/forward.cu:20-253, backward.cu:20-139,443-641,
// auxiliary.h.  The arithmetic order is the reference's (glm products expanded in glm's
// summation order), compiled with -ffp-contract=off.
#include "g4s_internal.h"
#include "g4s_device.h"

namespace g4s {

__constant__ float c_SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                 -1.0925484305920792f, 0.5462742152960396f};
__constant__ float c_SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                 0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                 -0.5900435899266435f};
#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f

// forward.cu:75-115 -- T = transpose(splat2world) * world2ndc * ndc2pix, rows Tu,Tv,Tw.
__device__ __forceinline__ void compute_transmat(F3 p, float sx, float sy, float mod, const float* R,
                                                 const float* proj, int W, int H, float* T) {
    const float s0 = mod * sx, s1 = mod * sy;
    const float Mrow[3][4] = {{R[0] * s0, R[1] * s0, R[2] * s0, 0.0f},
                              {R[3] * s1, R[4] * s1, R[5] * s1, 0.0f},
                              {p.x, p.y, p.z, 1.0f}};
    const float hw = (float)((float)W / 2.0), cw = (float)((float)(W - 1) / 2.0);
    const float hh = (float)((float)H / 2.0), ch = (float)((float)(H - 1) / 2.0);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        float c[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
            c[j] = Mrow[r][0] * proj[j] + Mrow[r][1] * proj[4 + j] + Mrow[r][2] * proj[8 + j] +
                   Mrow[r][3] * proj[12 + j];
        T[0 * 3 + r] = c[0] * hw + c[3] * cw;
        T[1 * 3 + r] = c[1] * hh + c[3] * ch;
        T[2 * 3 + r] = c[3];
    }
}

// forward.cu:119-147
__device__ __forceinline__ bool compute_aabb(const float* T, float cutoff, float& px, float& py, float& ex,
                                             float& ey) {
    const float* T0 = T;
    const float* T1 = T + 3;
    const float* T3 = T + 6;
    const float t0 = cutoff * cutoff, t1 = cutoff * cutoff, t2 = -1.0f;
    const float distance = (T3[0] * T3[0] * t0 + T3[1] * T3[1] * t1) + T3[2] * T3[2] * t2;
    const float inv = 1 / distance;
    const float f0 = inv * t0, f1 = inv * t1, f2 = inv * t2;
    if (distance == 0.0f) return false;
    px = (f0 * T0[0] * T3[0] + f1 * T0[1] * T3[1]) + f2 * T0[2] * T3[2];
    py = (f0 * T1[0] * T3[0] + f1 * T1[1] * T3[1]) + f2 * T1[2] * T3[2];
    const float tmp0 = (f0 * T0[0] * T0[0] + f1 * T0[1] * T0[1]) + f2 * T0[2] * T0[2];
    const float tmp1 = (f0 * T1[0] * T1[0] + f1 * T1[1] * T1[1]) + f2 * T1[2] * T1[2];
    const float h0 = px * px - tmp0, h1 = py * py - tmp1;
    ex = sqrtf(fmaxf(1e-4f, h0));
    ey = sqrtf(fmaxf(1e-4f, h1));
    return true;
}

// Conservative pixel-space bounding box of the region where a splat can reach alpha >= 1/255:
//   alpha = min(0.99, opa * exp(-min(rho3d, rho2d) / 2)) >= 1/255  <=>  min(rho3d, rho2d) <= 2 ln(255 opa) =: thr
// i.e. the union of the projected ellipse {rho3d <= thr} (compute_aabb's formula, forward.cu:119-147,
// evaluated at cutoff^2 = thr instead of 9) and the low-pass disk {rho2d <= thr} around the 3-sigma
// centre.  thr is inflated by 0.1 % + 1e-3 and the box by one pixel + 0.1 %, so no float rounding in
// the per-pixel evaluation can put a passing pixel outside it; where the cutoff conic is not an
// ellipse (the disk crosses the camera plane) the box is unbounded.  The kernels only use the box
// to SKIP work (tiles at emit, 8x8 quadrants in the blend loops); it never changes a result.
__device__ __forceinline__ void alpha_cutoff_box(const float* T, float cx, float cy, float opa, float4& box) {
    box = make_float4(1.0f, 1.0f, 0.0f, 0.0f);  // empty
    const float thr = 2.0f * logf(255.0f * opa);
    if (!(thr > 0.0f)) return;  // can never pass the alpha test (also catches NaN)
    const float t = thr * 1.001f + 1e-3f;
    const float BIG = 3.0e38f;
    float lo_x = -BIG, lo_y = -BIG, hi_x = BIG, hi_y = BIG;
    const float* Tu = T;
    const float* Tv = T + 3;
    const float* Tw = T + 6;
    const float d = t * (Tw[0] * Tw[0] + Tw[1] * Tw[1]) - Tw[2] * Tw[2];
    if (d < 0.0f) {
        const float f0 = t / d, f2 = -1.0f / d;
        const float ccx = f0 * (Tu[0] * Tw[0] + Tu[1] * Tw[1]) + f2 * Tu[2] * Tw[2];
        const float ccy = f0 * (Tv[0] * Tw[0] + Tv[1] * Tw[1]) + f2 * Tv[2] * Tw[2];
        const float hx = ccx * ccx - (f0 * (Tu[0] * Tu[0] + Tu[1] * Tu[1]) + f2 * Tu[2] * Tu[2]);
        const float hy = ccy * ccy - (f0 * (Tv[0] * Tv[0] + Tv[1] * Tv[1]) + f2 * Tv[2] * Tv[2]);
        const float ex = sqrtf(fmaxf(hx, 0.0f)), ey = sqrtf(fmaxf(hy, 0.0f));
        const float r2 = sqrtf(0.5f * t);  // rho2d = 2 |d|^2 <= t
        const float ax0 = fminf(ccx - ex, cx - r2), ax1 = fmaxf(ccx + ex, cx + r2);
        const float ay0 = fminf(ccy - ey, cy - r2), ay1 = fmaxf(ccy + ey, cy + r2);
        // any NaN / inf in the algebra above => keep the unbounded box
        if (fabsf(ax0) < BIG && fabsf(ax1) < BIG && fabsf(ay0) < BIG && fabsf(ay1) < BIG) {
            lo_x = ax0 - (1.0f + 1e-3f * fabsf(ax0));
            hi_x = ax1 + (1.0f + 1e-3f * fabsf(ax1));
            lo_y = ay0 - (1.0f + 1e-3f * fabsf(ay0));
            hi_y = ay1 + (1.0f + 1e-3f * fabsf(ay1));
        }
    }
    box = make_float4(lo_x, lo_y, hi_x, hi_y);
}

// The same region, exactly: rho3d <= t is the set of pixels (x, y) with  p.x^2 + p.y^2 - t p.z^2 <= 0  where
// p = k x l = Tu x Tv - x (Tw x Tv) - y (Tu x Tw) is LINEAR in the pixel, i.e. a conic.  When it is an ellipse it is
// returned in its eigenframe -- centre e, unit major axis u, inverse squared semi-axes 1/a^2 <= 1/b^2 -- so that
// the blend kernels can evaluate  ((d.u)/a)^2 + ((d.v)/b)^2 <= 1  without cancellation even for needles with an
// aspect ratio of 10^4 (the expanded form m11 dx^2 + 2 m12 dx dy + m22 dy^2 loses all digits there).  Both
// semi-axes are enlarged: r -> 1.002 sqrt(r^2 + 0.75^2) (0.2 % + three quarters of a pixel in quadrature; that
// absorbs the float storage of e and the per-pixel rounding of the blend loops and keeps the minor axis above
// 0.75 px).  Also the squared radius of the low-pass disk (rho2d = 2 |pixel - centre|^2 <= t).  Evaluated in
// double, once per visible Gaussian.  out[0..6] = ex, ey, ux, uy, 1/a^2, 1/b^2, r2; 1/a^2 = 0: no ellipse.
struct CutoffConic {
    double ex, ey, ux, uy, a2, b2;  // centre, unit major axis, squared semi-axes
};
// the conic  p.x^2 + p.y^2 - t p.z^2 <= 0  as an ellipse; false if it is not one (or is empty)
__device__ bool cutoff_conic(const double* A, const double* B, const double* D, double t, CutoffConic& c, bool bound_error) {
    auto dotg = [t](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] - t * a[2] * b[2]; };
    const double aa = dotg(A, A), ab = dotg(A, B), bb = dotg(B, B), a0 = dotg(A, D), b0 = dotg(B, D), c0 = dotg(D, D);
    const double det = aa * bb - ab * ab;
    if (!(aa > 0.0) || !(bb > 0.0) || !(det > 0.0)) return false;
    // (double-precision divisions are the expensive part of this function -- ~35 instructions each -- so every quotient
    // by the same denominator shares one reciprocal: 5 divisions per conic instead of 11.  The last-bit difference in
    // double is far below the float rounding of the stored ellipse and its 0.2 % + 0.75 px margin.)
    const double inv_det = 1.0 / det;
    const double nx = a0 * bb - b0 * ab, ny = b0 * aa - a0 * ab;
    c.ex = nx * inv_det;
    c.ey = ny * inv_det;
    const double kappa = a0 * c.ex + b0 * c.ey - c0;  // -Q(e):  (x - e)^T H (x - e) <= kappa
    if (!(kappa > 0.0)) return false;
    // kappa is a small difference of large terms and scales both semi-axes.  For needles with an aspect ratio in the
    // thousands its rounding error in double precision reaches percents (found by tools/fuzz_sweep.py: a hair-thin
    // splat whose long axis came out 1.3 % short and lost one contributing pixel).  A first-order error bound decides:
    // the ellipse is only used where kappa is good to 1e-3, i.e. the axes to 5e-4, a quarter of their 0.2 % margin.
    if (bound_error) {
        const double u = 4.5e-16;  // two roundings per product / sum
        const double rel_det = u * (fabs(aa * bb) + fabs(ab * ab)) * inv_det;
        const double ex_err = fabs(c.ex) * (u * (fabs(a0 * bb) + fabs(b0 * ab)) / fmax(fabs(nx), 1e-300) + rel_det);
        const double ey_err = fabs(c.ey) * (u * (fabs(b0 * aa) + fabs(a0 * ab)) / fmax(fabs(ny), 1e-300) + rel_det);
        const double kappa_err = fabs(a0) * ex_err + fabs(b0) * ey_err + u * (fabs(a0 * c.ex) + fabs(b0 * c.ey) + fabs(c0));
        if (!(kappa_err < 1e-3 * kappa)) return false;
    }
    // eigen-decomposition of H = [aa ab; ab bb]: small eigenvalue <-> major axis
    const double tr = aa + bb, disc = sqrt(fmax((aa - bb) * (aa - bb) + 4.0 * ab * ab, 0.0));
    const double lmax = 0.5 * (tr + disc);
    const double inv_lmax = 1.0 / lmax;
    const double lmin = det * inv_lmax;  // (tr - disc) / 2 without the cancellation
    if (!(lmin > 0.0)) return false;
    double ux, uy;  // eigenvector of lmin
    if (fabs(ab) > 1e-300) {
        ux = lmin - bb; uy = ab;
        if (fabs(aa - lmin) > fabs(bb - lmin)) { ux = ab; uy = lmin - aa; }
    } else {
        ux = aa <= bb ? 1.0 : 0.0; uy = aa <= bb ? 0.0 : 1.0;
    }
    const double un2 = ux * ux + uy * uy;
    if (!(un2 > 0.0)) return false;
    const double inv_un = 1.0 / sqrt(un2);
    c.ux = ux * inv_un; c.uy = uy * inv_un;
    c.a2 = kappa * (lmax * inv_det);  // kappa / lmin with 1 / lmin = lmax / det
    c.b2 = kappa * inv_lmax;
    return true;
}

__device__ __forceinline__ void alpha_cutoff_ellipse(const float* T, float opa, float* out) {
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = 0.0f;
    const float thr = 2.0f * logf(255.0f * opa);
    if (!(thr > 0.0f)) return;
    const double t = (double)thr * 1.001 + 1e-3;
    out[6] = (float)(0.5 * t * 1.001 + 0.01);
    const double Tu[3] = {T[0], T[1], T[2]}, Tv[3] = {T[3], T[4], T[5]}, Tw[3] = {T[6], T[7], T[8]};
    auto cross = [](const double* a, const double* b, double* c) {
        c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
    };
    double A[3], B[3], D[3];
    cross(Tw, Tv, A);
    cross(Tu, Tw, B);
    cross(Tu, Tv, D);
    // The conic of a needle seen obliquely is nearly degenerate: its long axis then depends on the 7th digit of t
    // and of T, i.e. on rounding the blend loops do not share.  Accept the ellipse only where it is well
    // conditioned: growing the cutoff by 1e-4 (ten times any rounding in the per-pixel evaluation) must move the
    // centre by less than a tenth of a pixel and the axes by less than 0.1 %; the larger of the two is used.
    CutoffConic c0, c1;
    if (!cutoff_conic(A, B, D, t, c0, false) || !cutoff_conic(A, B, D, t * (1.0 + 1e-4), c1, true)) return;
    const double shift2 = (c1.ex - c0.ex) * (c1.ex - c0.ex) + (c1.ey - c0.ey) * (c1.ey - c0.ey);
    if (!(shift2 < 0.01) || !(c1.a2 < c0.a2 * 1.002) || !(c1.b2 < c0.b2 * 1.002 + 1e-6) || !(c1.a2 >= c0.a2 * 0.999)) return;
    const double a2 = c1.a2 + 0.5625, b2 = c1.b2 + 0.5625;  // semi-axes^2 + 0.75^2
    const double s2 = 1.002 * 1.002;
    const double vals[6] = {c1.ex, c1.ey, c1.ux, c1.uy, 1.0 / (a2 * s2), 1.0 / (b2 * s2)};
    for (int i = 0; i < 6; i++)
        if (!(fabs(vals[i]) < 1e30)) return;  // NaN / overflow: no ellipse
    for (int i = 0; i < 6; i++) out[i] = (float)vals[i];
    // (no validity flag: 1 / a^2 = out[4] > 0 says there is an ellipse; out[7] belongs to the caller)
}

// Parts (b) and (c) of the REC_AFFINE certificate (g4s_device.h): true if the interpolated depth cannot fall below the
// near plane where the splat passes the alpha test and the low-pass exponent rho2d can never be the smaller one where
// it matters -- at every integer pixel whose rho2d could still pass the alpha test (a disk of <= 2.4 px around the
// centre: a box of at most 5 x 5 pixels) the 3-D exponent is smaller and outside the tie band -- evaluated with the
// blend loops' own affine arithmetic (eval_rho_affine).
//
// Wave-cooperative: EVERY lane of the wave calls it (converged), `want` = this lane has a splat that passed part (a).
// A quarter of the lanes carry such a splat (the rest are culled), each with up to 25 pixels to look at; walked by the
// owning lane alone that is a 25-trip loop for the whole wave.  Here the splats of the wave are parked in LDS, 25 slots
// each, and all 64 lanes take (splat, pixel) pairs: ~7 trips instead of 25 (preprocess_fwd 0.100 -> see DESIGN section 7).
// s_par: 64 x 16 words of this wave, s_fail: 64 words of this wave.  The pixels visited and the arithmetic per pixel
// are those of the serial walk, so the decision is the same bit.
constexpr int CERT_SLOT_WORDS = 16;
__device__ __forceinline__ bool lowpass_never_matters_wave(bool want, const float* T, const SplatAffine& af, float cx, float cy,
                                                           float opa, float* s_par, uint32_t* s_fail) {
    float tt = 0.0f;
    if (want) {
        const float thr = 2.0f * logf(255.0f * opa);
        want = thr > 0.0f && fabsf(cx) < 1e7f && fabsf(cy) < 1e7f;
        tt = thr * 1.01f + 0.1f;
        // depth = s.x Tw.x + s.y Tw.y + Tw.z with |s|^2 <= t wherever alpha passes: never below the near plane?
        if (want) want = T[8] - sqrtf(tt) * sqrtf(T[6] * T[6] + T[7] * T[7]) > NEAR_N * 1.05f;
    }
    const float r = sqrtf(0.5f * tt) + 0.01f;
    const int xlo = (int)ceilf(cx - r), ylo = (int)ceilf(cy - r), xhi = (int)floorf(cx + r), yhi = (int)floorf(cy + r);
    // opacity <= 1 gives r <= 2.39, i.e. at most 5 x 5 pixels; a caller's opacity above 1 simply does not qualify
    if (want) want = xhi - xlo <= 4 && yhi - ylo <= 4;
    // Most splats are much larger than the low-pass disk, and for those no pixel needs to be looked at:  with
    // N(d) = (p'.x, p'.y) = N0 + M d and p'.z(d) = Dc'.z + (A'.z, B'.z) . d,  |N| <= |N0| + |M|_F |d| and
    // |p'.z| >= zmin := |Dc'.z| - |(A'.z, B'.z)| r on the disk, so  rho3d = |N|^2 / p'.z^2 <= rho2d / 2 = (F / 2) |d|^2
    // wherever |N0| + |M|_F |d| <= sqrt(F / 2) zmin |d|, i.e. for every pixel at |d| >= dmin (the nearest integer pixel)
    // once  sqrt(F / 2) zmin > |M|_F  and  |N0| <= (sqrt(F / 2) zmin - |M|_F) dmin.  A factor of two between the
    // exponents is far outside the tie band and outside anything the float evaluation can move (part (a): 4e-5), so
    // the walk below would pass: `sure`.  The walk itself is left to the few splats this bound cannot decide.
    bool sure = false;
    if (want) {
        const float kf = sqrtf(0.5f * FILTER_INV_SQUARE);
        const float zmin = fabsf(af.Dc[2]) - sqrtf(fmaf(af.A[2], af.A[2], af.B[2] * af.B[2])) * r;
        const float mf = sqrtf(fmaf(af.A[0], af.A[0], af.B[0] * af.B[0]) + fmaf(af.A[1], af.A[1], af.B[1] * af.B[1]));
        const float n0 = sqrtf(fmaf(af.Dc[0], af.Dc[0], af.Dc[1] * af.Dc[1]));
        const float fx = cx - floorf(cx), fy = cy - floorf(cy);
        const float dxm = fminf(fx, 1.0f - fx), dym = fminf(fy, 1.0f - fy);
        const float dmin = sqrtf(fmaf(dxm, dxm, dym * dym));
        const float slack = kf * zmin - mf;  // (NaN anywhere fails the comparisons)
        sure = slack > 0.0f && dmin >= 1e-3f && n0 <= 0.99f * slack * dmin;
    }
    const bool walk = want && !sure;
    const uint64_t m = __ballot(walk);
    if (m == 0) return want;  // (uniform)
    const int lane = lane_id();
    const int rank = (int)__popcll(m & lanes_below_mask());
    if (walk) {
        float4* q = reinterpret_cast<float4*>(s_par + rank * CERT_SLOT_WORDS);
        q[0] = make_float4(af.A[0], af.A[1], af.A[2], af.B[0]);
        q[1] = make_float4(af.B[1], af.B[2], af.Dc[0], af.Dc[1]);
        q[2] = make_float4(af.Dc[2], cx, cy, tt);
        q[3] = make_float4(__int_as_float(xlo), __int_as_float(ylo), __int_as_float(xhi), __int_as_float(yhi));
        s_fail[rank] = 0u;
    }
    __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations complete in order; this only pins the compiler's)
    const int total = 25 * (int)__popcll(m);
    for (int base = 0; base < total; base += 64) {  // (uniform)
        const int i = base + lane;
        if (i < total) {
            const int rk = i / 25, pix = i - 25 * rk, iy = pix / 5, ix = pix - 5 * iy;
            const float4* q = reinterpret_cast<const float4*>(s_par + rk * CERT_SLOT_WORDS);
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
            const int xx = __float_as_int(q3.x) + ix, yy = __float_as_int(q3.y) + iy;
            if (xx <= __float_as_int(q3.z) && yy <= __float_as_int(q3.w)) {
                SplatAffine f;
                f.A[0] = q0.x; f.A[1] = q0.y; f.A[2] = q0.z;
                f.B[0] = q0.w; f.B[1] = q1.x; f.B[2] = q1.y;
                f.Dc[0] = q1.z; f.Dc[1] = q1.w; f.Dc[2] = q2.x;
                const float scx = q2.y, scy = q2.z, stt = q2.w;
                const float ddx = (float)xx - scx, ddy = (float)yy - scy;
                if (!(FILTER_INV_SQUARE * fmaf(ddx, ddx, ddy * ddy) > stt)) {  // eval_rho_affine's rho2d, before the costly part
                    PairEval e;
                    bool tie;
                    if (eval_rho_affine((float)xx, (float)yy, scx, scy, f, e, tie) && !(e.rho2d > stt) &&
                        (tie || !(e.rho3d <= e.rho2d)))
                        s_fail[rk] = 1u;
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return want && (sure || s_fail[rank] == 0u);
}

// Loads the 3*(deg+1)^2 active SH floats of Gaussian idx into registers.  vec16: the records are
// 192 B ([16][3] floats) on a 16-byte aligned base, so they are fetched as 16-byte quads.  rest != NULL: split
// layout, coefficient 0 lives in shs[P][3] and coefficients 1..M-1 in rest[P][M-1][3] (the two parameter tensors
// of the reference's GaussianModel, 2dgs/scene/gaussian_model.py:  get_features = cat(_features_dc, _features_rest)).
__device__ __forceinline__ void load_sh(const float* __restrict__ shs, const float* __restrict__ rest, size_t idx, int M,
                                        int deg, bool vec16, float* sh /*48*/) {
    const int n = 3 * (deg + 1) * (deg + 1);  // active floats (vec16 path)
    if (rest != nullptr) {
        const float* p0 = shs + idx * 3;
        const float* p1 = rest + idx * (size_t)(M - 1) * 3;
        sh[0] = p0[0]; sh[1] = p0[1]; sh[2] = p0[2];
        // one unconditional run per SH band, so that the loads of a band merge into wide accesses
        if (deg > 0) {
#pragma unroll
            for (int i = 3; i < 12; i++) sh[i] = p1[i - 3];
            if (deg > 1) {
#pragma unroll
                for (int i = 12; i < 27; i++) sh[i] = p1[i - 3];
                if (deg > 2) {
#pragma unroll
                    for (int i = 27; i < 48; i++) sh[i] = p1[i - 3];
                }
            }
        }
    } else if (vec16) {
        const float4* p = reinterpret_cast<const float4*>(shs + idx * 48);
#pragma unroll
        for (int q = 0; q < 12; q++) {
            if (4 * q < n) {
                const float4 v = p[q];
                sh[4 * q] = v.x; sh[4 * q + 1] = v.y; sh[4 * q + 2] = v.z; sh[4 * q + 3] = v.w;
            }
        }
    } else {
        const float* p = shs + idx * (size_t)M * 3;
        sh[0] = p[0]; sh[1] = p[1]; sh[2] = p[2];
        if (deg > 0) {
#pragma unroll
            for (int i = 3; i < 12; i++) sh[i] = p[i];
            if (deg > 1) {
#pragma unroll
                for (int i = 12; i < 27; i++) sh[i] = p[i];
                if (deg > 2) {
#pragma unroll
                    for (int i = 27; i < 48; i++) sh[i] = p[i];
                }
            }
        }
    }
}

// forward.cu:20-71.  sh = this Gaussian's active coefficients in registers (load_sh).
__device__ __forceinline__ void sh_to_rgb(int deg, const float* sh, F3 pos, F3 campos, float* rgb,
                                          uint32_t& clamp_bits) {
    const float dx = pos.x - campos.x, dy = pos.y - campos.y, dz = pos.z - campos.z;
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    float r[3];
#pragma unroll
    for (int c = 0; c < 3; c++) r[c] = SH_C0 * sh[c];
    if (deg > 0) {
#pragma unroll
        for (int c = 0; c < 3; c++)
            r[c] = r[c] - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
#pragma unroll
            for (int c = 0; c < 3; c++)
                r[c] = r[c] + c_SH_C2[0] * xy * sh[12 + c] + c_SH_C2[1] * yz * sh[15 + c] +
                       c_SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] + c_SH_C2[3] * xz * sh[21 + c] +
                       c_SH_C2[4] * (xx - yy) * sh[24 + c];
            if (deg > 2) {
#pragma unroll
                for (int c = 0; c < 3; c++)
                    r[c] = r[c] + c_SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + c] + c_SH_C3[1] * xy * z * sh[30 + c] +
                           c_SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] +
                           c_SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                           c_SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] +
                           c_SH_C3[5] * z * (xx - yy) * sh[42 + c] + c_SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
            }
        }
    }
    clamp_bits = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        r[c] += 0.5f;
        if (r[c] < 0) clamp_bits |= (1u << c);
        rgb[c] = fmaxf(r[c], 0.0f);
    }
}

// K1: forward.cu:150-253.  Also seeds the depth sort (key = depth bits, CULLED_KEY if the
// Gaussian emits nothing; payload = index), computes the conservative alpha-cutoff bounding
// box (record quad 5) and accumulates the reference's instance count (num_rendered).
// SH_MODE: 0 = packed [P,16,3] on a 16-byte aligned base (quad loads), 1 = packed generic, 2 = split (dc / rest).
// One instantiation per layout: a runtime switch inside one kernel costs the common layout ~20 % (register copies at the joins).
// 5 waves per SIMD (96 VGPRs, three 4-byte spills) instead of the 4 that the compiler's own 100 registers allow: the kernel
// is bound by the latency of its per-workgroup chain (loads -> cull -> the double-precision tail of the 28 % of the lanes
// that bin something), and a fifth resident wave hides some of it: 0.107 -> 0.102 ms (profiles/r05_ab_k1_occupancy.txt).
// (6 waves = 80 VGPRs spill 80 registers: 0.125 ms, LAB_NOTES section 3.)
template <int SH_MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) preprocess_fwd_kernel(PreprocessArgs a) {
    __shared__ __attribute__((aligned(16))) float s_cert[4][64 * CERT_SLOT_WORDS];
    __shared__ uint32_t s_cert_fail[4][64];
    const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
    const bool in_range = idx < a.P;
    int radius_out = 0;
    uint32_t touched_ref = 0, touched = 0, key = CULLED_KEY, clamp_bits = 0;
    // What a Gaussian that reaches the innermost block below hands to the two steps behind the joins (the wave-wide
    // certificate and the record store).  Deliberately NOT initialised: they are read only where radius_out > 0 says they
    // were written, and a zero on the culled paths would cost a register copy per word at each of the five joins.
    float T[9], cx, cy, opa, rgb[3];
    F3 normal;
    float4 box;
    int tx0, ty0, tx1, ty1;
    SplatAffine af;
    bool want = false;  // part (a) of the REC_AFFINE certificate holds (g4s_device.h)

    if (in_range) {
        const F3 p = mk3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
        const F3 p_view = xform_point_4x3(p, a.viewmatrix);
        if (p_view.z > 0.2f) {  // in_frustum, auxiliary.h:184-209
            if (a.transMat_precomp == nullptr) {
                float R[9];
                const float4 q = reinterpret_cast<const float4*>(a.rotations)[idx];
                const float2 sc = reinterpret_cast<const float2*>(a.scales)[idx];
                quat_to_rotmat(q, R);
                compute_transmat(p, sc.x, sc.y, a.scale_modifier, R, a.projmatrix, a.W, a.H, T);
                normal = xform_vec_4x3(mk3(R[6], R[7], R[8]), a.viewmatrix);
            } else {
#pragma unroll
                for (int i = 0; i < 9; i++) T[i] = a.transMat_precomp[9 * (size_t)idx + i];
                normal = mk3(0.0f, 0.0f, 1.0f);
            }
            const float cosv = -((p_view.x * normal.x + p_view.y * normal.y) + p_view.z * normal.z);
            if (cosv != 0) {
                const float mult = cosv > 0 ? 1.0f : -1.0f;
                normal = mk3(mult * normal.x, mult * normal.y, mult * normal.z);
                float ex, ey;
                if (compute_aabb(T, 3.0f, cx, cy, ex, ey)) {
                    const float radius = ceilf(fmaxf(ex, ey));
                    int x0, y0, x1, y1;
                    get_rect(cx, cy, sat_int(radius), a.tiles_x, a.tiles_y, x0, y0, x1, y1);
                    const int area = (x1 - x0) * (y1 - y0);
                    if (area != 0) {
                        if (a.colors_precomp == nullptr) {
                            float sh[48];
                            load_sh(a.shs, SH_MODE == 2 ? a.shs_rest : nullptr, (size_t)idx, a.M, a.D, SH_MODE == 0, sh);
                            sh_to_rgb(a.D, sh, p, mk3(a.cam_pos[0], a.cam_pos[1], a.cam_pos[2]), rgb, clamp_bits);
                        } else {
                            rgb[0] = a.colors_precomp[3 * (size_t)idx];
                            rgb[1] = a.colors_precomp[3 * (size_t)idx + 1];
                            rgb[2] = a.colors_precomp[3 * (size_t)idx + 2];
                        }
                        opa = a.opacities[idx];
                        radius_out = sat_int(radius);
                        touched_ref = (uint32_t)area;
                        key = __float_as_uint(p_view.z);
                        alpha_cutoff_box(T, cx, cy, opa, box);
                        tight_tile_rect(box, x0, y0, x1, y1, tx0, ty0, tx1, ty1);
                        touched = (uint32_t)((tx1 - tx0) * (ty1 - ty0));
                        // the emit kernel expands exactly this rect: 8 bytes per Gaussian instead of two record quads + radius
                        if (touched != 0) a.tight_rect[idx] = make_uint2((uint32_t)tx0 | ((uint32_t)ty0 << 16), (uint32_t)(tx1 - tx0));
                        // Does the splat qualify for the affine ray-splat intersection (REC_AFFINE, g4s_device.h)?  Part (a)
                        // here; only splats that are binned can be asked for it.
                        if (touched != 0 && !a.no_fastpath) {
                            // how far from the centre the splat is evaluated with a chance to pass: its alpha-cutoff box, in the frame
                            const float bx0 = fmaxf(box.x, 0.0f), bx1 = fminf(box.z, (float)(a.W - 1));
                            const float by0 = fmaxf(box.y, 0.0f), by1 = fminf(box.w, (float)(a.H - 1));
                            const float ex = fmaxf(fabsf(bx0 - cx), fabsf(bx1 - cx)), ey = fmaxf(fabsf(by0 - cy), fabsf(by1 - cy));
                            const float smax = sqrtf(2.0f * logf(255.0f * opa) * 1.001f + 1e-3f);
                            splat_affine(T, cx, cy, ex, ey, smax, af);
                            want = af.ok;
                        }
                    }
                }
            }
        }
    }
    // parts (b) and (c), all lanes of the wave together
    const bool affine = lowpass_never_matters_wave(want, T, af, cx, cy, opa, s_cert[threadIdx.x >> 6], s_cert_fail[threadIdx.x >> 6]);
    if (in_range) {
        // The record of a Gaussian with radii == 0 is never read (emit / blend follow the tile lists, the backward looks
        // at radii first): writing only the visible ones saves 128 B x (P - V) of stores.
        if (radius_out > 0) {
            // q0.w = binned rect (width | height << 16) | REC_AFFINE; q2..q4 = T (forward.cu:197-200), or A', B', Dc' for
            // REC_AFFINE splats; q7.w = the rect's origin (x0 | y0 << 16)
            float c[9];
#pragma unroll
            for (int i = 0; i < 3; i++) {  // selects, not a branch: no copies of the nine words at a join
                c[i] = affine ? af.A[i] : T[i];
                c[3 + i] = affine ? af.B[i] : T[3 + i];
                c[6 + i] = affine ? af.Dc[i] : T[6 + i];
            }
            uint32_t bx0, bx1, by0, by1;
            box_quadrants(box.x, box.z, a.W, bx0, bx1);
            box_quadrants(box.y, box.w, a.H, by0, by1);
            float ell[8];
#pragma unroll
            for (int i = 0; i < 8; i++) ell[i] = 0.0f;
            if (touched != 0) alpha_cutoff_ellipse(T, opa, ell);  // ell[4] (1 / a^2) stays 0 when there is no ellipse
            float4* out = reinterpret_cast<float4*>(a.rec) + (size_t)idx * REC_QUADS;
            out[0] = make_float4(cx, cy, 0.0f /* inst_off: slots_and_compact_kernel */,
                                 __uint_as_float(rect_extent_word(tx1 - tx0, ty1 - ty0) | (affine ? REC_AFFINE : 0u)));
            out[1] = make_float4(normal.x, normal.y, normal.z, opa);
            out[2] = make_float4(c[0], c[1], c[2], c[3]);
            out[3] = make_float4(c[4], c[5], c[6], c[7]);
            out[4] = make_float4(c[8], rgb[0], rgb[1], rgb[2]);
            out[5] = make_float4(__uint_as_float(bx0), __uint_as_float(bx1), T[8], __uint_as_float(by0 | (by1 << 16)));
            out[6] = make_float4(ell[0], ell[1], ell[2], ell[3]);
            out[7] = make_float4(ell[4], ell[5], ell[6], __uint_as_float((uint32_t)tx0 | ((uint32_t)ty0 << 16)));
        }
        a.clamped[idx] = (uint8_t)clamp_bits;
        a.tiles_touched[idx] = touched;
        a.radii[idx] = radius_out;
        a.depth_keys[idx] = key;
    }
    // num_rendered of the reference = sum of its tiles_touched: per-block partial, summed by the
    // count-scan kernel (a same-address atomic per wave serialises at ~90 atomics/us on this part)
    __shared__ uint32_t s_ref[4], s_tight[4], s_vis[4], s_kmin[4], s_kmax[4];
    const uint32_t wsum = wave_sum_u32(touched_ref);
    const uint32_t tsum = wave_sum_u32(touched);
    const uint32_t vsum = (uint32_t)__popcll(__ballot(touched > 0));
    // range of the depth keys that take part in the sort (binning.hip: three passes over key - smallest key)
    const uint32_t kmax = wave_max_u32_full_wave(touched > 0 ? key : 0u);
    const uint32_t kmin = ~wave_max_u32_full_wave(touched > 0 ? ~key : 0u);
    if (lane_id() == 0) {
        s_ref[threadIdx.x >> 6] = wsum;
        s_tight[threadIdx.x >> 6] = tsum;
        s_vis[threadIdx.x >> 6] = vsum;
        s_kmin[threadIdx.x >> 6] = kmin;
        s_kmax[threadIdx.x >> 6] = kmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        a.ref_block_sums[blockIdx.x] = (s_ref[0] + s_ref[1]) + (s_ref[2] + s_ref[3]);
        // binned-instance count of this block in INDEX order: base of the gradient-record slots
        a.idx_block_sums[blockIdx.x] = (s_tight[0] + s_tight[1]) + (s_tight[2] + s_tight[3]);
        // Gaussians of this block that emit instances: base of their slots in the compacted depth-sort input
        a.vis_block_sums[blockIdx.x] = (s_vis[0] + s_vis[1]) + (s_vis[2] + s_vis[3]);
        a.key_min_blocks[blockIdx.x] = min(min(s_kmin[0], s_kmin[1]), min(s_kmin[2], s_kmin[3]));
        a.key_max_blocks[blockIdx.x] = max(max(s_kmax[0], s_kmax[1]), max(s_kmax[2], s_kmax[3]));
    }
}

void launch_preprocess_fwd(const PreprocessArgs& a, hipStream_t s) {
    if (a.P <= 0) return;
    const dim3 grid((a.P + 255) / 256), block(256);
    if (a.shs_rest != nullptr) hipLaunchKernelGGL(preprocess_fwd_kernel<2>, grid, block, 0, s, a);
    else if (a.sh_vec16) hipLaunchKernelGGL(preprocess_fwd_kernel<0>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(preprocess_fwd_kernel<1>, grid, block, 0, s, a);
}

// K9: rasterizer_impl.cu:54-66
__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ view,
                                    uint8_t* __restrict__ present) {
    const int idx = (int)(blockIdx.x * 256 + threadIdx.x);
    if (idx >= P) return;
    const F3 p = mk3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    present[idx] = xform_point_4x3(p, view).z > 0.2f ? 1 : 0;
}
void launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}

// ---- backward ----------------------------------------------------------------------------

// auxiliary.h:237-281, v_R column-major
__device__ __forceinline__ float4 quat_to_rotmat_vjp(const float4 q, const float* v_R) {
    const float s = 1.0f / sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    const float w = q.x * s, x = q.y * s, y = q.z * s, z = q.w * s;
#define VR(c, r) v_R[(c) * 3 + (r)]
    float4 v;
    v.x = 2.f * (x * (VR(1, 2) - VR(2, 1)) + y * (VR(2, 0) - VR(0, 2)) + z * (VR(0, 1) - VR(1, 0)));
    v.y = 2.f * (-2.f * x * (VR(1, 1) + VR(2, 2)) + y * (VR(0, 1) + VR(1, 0)) + z * (VR(0, 2) + VR(2, 0)) +
                 w * (VR(1, 2) - VR(2, 1)));
    v.z = 2.f * (x * (VR(0, 1) + VR(1, 0)) - 2.f * y * (VR(0, 0) + VR(2, 2)) + z * (VR(1, 2) + VR(2, 1)) +
                 w * (VR(2, 0) - VR(0, 2)));
    v.w = 2.f * (x * (VR(0, 2) + VR(2, 0)) + y * (VR(1, 2) + VR(2, 1)) - 2.f * z * (VR(0, 0) + VR(1, 1)) +
                 w * (VR(0, 1) - VR(1, 0)));
#undef VR
    return v;
}

// auxiliary.h:130-140
__device__ __forceinline__ F3 dnormvdv(F3 v, F3 dv) {
    const float sum2 = v.x * v.x + v.y * v.y + v.z * v.z;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    return mk3(((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32,
               (-v.x * v.y * dv.x + (sum2 - v.y * v.y) * dv.y - v.z * v.y * dv.z) * invsum32,
               (-v.x * v.z * dv.x - v.y * v.z * dv.y + (sum2 - v.z * v.z) * dv.z) * invsum32);
}

// The SH basis values sh_backward multiplies the colour gradient with (coef[i] = d rgb / d sh_i; zero above the active
// degree): the same expressions in the same order, so the products are the bits sh_backward stores.
__device__ __forceinline__ void sh_coefficients(int deg, F3 pos, F3 campos, float* coef /*16*/) {
    const F3 d = mk3(pos.x - campos.x, pos.y - campos.y, pos.z - campos.z);
    const float len = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
    const float x = d.x / len, y = d.y / len, z = d.z / len;
#pragma unroll
    for (int i = 0; i < 16; i++) coef[i] = 0.0f;
    coef[0] = SH_C0;
    if (deg > 0) {
        coef[1] = -SH_C1 * y;
        coef[2] = SH_C1 * z;
        coef[3] = -SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            coef[4] = c_SH_C2[0] * xy;
            coef[5] = c_SH_C2[1] * yz;
            coef[6] = c_SH_C2[2] * (2.f * zz - xx - yy);
            coef[7] = c_SH_C2[3] * xz;
            coef[8] = c_SH_C2[4] * (xx - yy);
            if (deg > 2) {
                coef[9] = c_SH_C3[0] * y * (3.f * xx - yy);
                coef[10] = c_SH_C3[1] * xy * z;
                coef[11] = c_SH_C3[2] * y * (4.f * zz - xx - yy);
                coef[12] = c_SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                coef[13] = c_SH_C3[4] * x * (4.f * zz - xx - yy);
                coef[14] = c_SH_C3[5] * z * (xx - yy);
                coef[15] = c_SH_C3[6] * x * (xx - 3.f * yy);
            }
        }
    }
}

// backward.cu:20-139.  Writes all M coefficients of dL_dsh (zeros above the active degree)
// and returns the view-direction term to add to dL_dmean.
// ACC: the coefficients are ADDED to what dsh holds (gradient accumulation over views) and nothing is cleared.
template <bool ACC>
__device__ __forceinline__ F3 sh_backward(int deg, int M, const float* sh, F3 pos, F3 campos, uint32_t clamp_bits,
                                          const float* dL_dcolor, float* __restrict__ dsh, float* __restrict__ dsh_rest,
                                          bool vec16) {
    const F3 dir_orig = mk3(pos.x - campos.x, pos.y - campos.y, pos.z - campos.z);
    const float len = sqrtf(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
    const float x = dir_orig.x / len, y = dir_orig.y / len, z = dir_orig.z / len;
    float dRGB[3];
#pragma unroll
    for (int c = 0; c < 3; c++) dRGB[c] = dL_dcolor[c] * ((clamp_bits >> c) & 1u ? 0.0f : 1.0f);
    float dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
    float coef[16];
#pragma unroll
    for (int i = 0; i < 16; i++) coef[i] = 0.0f;
    coef[0] = SH_C0;
    if (deg > 0) {
        coef[1] = -SH_C1 * y;
        coef[2] = SH_C1 * z;
        coef[3] = -SH_C1 * x;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            dx[c] = -SH_C1 * sh[9 + c];
            dy[c] = -SH_C1 * sh[3 + c];
            dz[c] = SH_C1 * sh[6 + c];
        }
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            coef[4] = c_SH_C2[0] * xy;
            coef[5] = c_SH_C2[1] * yz;
            coef[6] = c_SH_C2[2] * (2.f * zz - xx - yy);
            coef[7] = c_SH_C2[3] * xz;
            coef[8] = c_SH_C2[4] * (xx - yy);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                dx[c] += c_SH_C2[0] * y * sh[12 + c] + c_SH_C2[2] * 2.f * -x * sh[18 + c] + c_SH_C2[3] * z * sh[21 + c] +
                         c_SH_C2[4] * 2.f * x * sh[24 + c];
                dy[c] += c_SH_C2[0] * x * sh[12 + c] + c_SH_C2[1] * z * sh[15 + c] + c_SH_C2[2] * 2.f * -y * sh[18 + c] +
                         c_SH_C2[4] * 2.f * -y * sh[24 + c];
                dz[c] += c_SH_C2[1] * y * sh[15 + c] + c_SH_C2[2] * 2.f * 2.f * z * sh[18 + c] + c_SH_C2[3] * x * sh[21 + c];
            }
            if (deg > 2) {
                coef[9] = c_SH_C3[0] * y * (3.f * xx - yy);
                coef[10] = c_SH_C3[1] * xy * z;
                coef[11] = c_SH_C3[2] * y * (4.f * zz - xx - yy);
                coef[12] = c_SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy);
                coef[13] = c_SH_C3[4] * x * (4.f * zz - xx - yy);
                coef[14] = c_SH_C3[5] * z * (xx - yy);
                coef[15] = c_SH_C3[6] * x * (xx - 3.f * yy);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    dx[c] += (c_SH_C3[0] * sh[27 + c] * 3.f * 2.f * xy + c_SH_C3[1] * sh[30 + c] * yz +
                              c_SH_C3[2] * sh[33 + c] * -2.f * xy + c_SH_C3[3] * sh[36 + c] * -3.f * 2.f * xz +
                              c_SH_C3[4] * sh[39 + c] * (-3.f * xx + 4.f * zz - yy) + c_SH_C3[5] * sh[42 + c] * 2.f * xz +
                              c_SH_C3[6] * sh[45 + c] * 3.f * (xx - yy));
                    dy[c] += (c_SH_C3[0] * sh[27 + c] * 3.f * (xx - yy) + c_SH_C3[1] * sh[30 + c] * xz +
                              c_SH_C3[2] * sh[33 + c] * (-3.f * yy + 4.f * zz - xx) +
                              c_SH_C3[3] * sh[36 + c] * -3.f * 2.f * yz + c_SH_C3[4] * sh[39 + c] * -2.f * xy +
                              c_SH_C3[5] * sh[42 + c] * -2.f * yz + c_SH_C3[6] * sh[45 + c] * -3.f * 2.f * xy);
                    dz[c] += (c_SH_C3[1] * sh[30 + c] * xy + c_SH_C3[2] * sh[33 + c] * 4.f * 2.f * yz +
                              c_SH_C3[3] * sh[36 + c] * 3.f * (2.f * zz - xx - yy) + c_SH_C3[4] * sh[39 + c] * 4.f * 2.f * xz +
                              c_SH_C3[5] * sh[42 + c] * (xx - yy));
                }
            }
        }
    }
    const int nact = (deg + 1) * (deg + 1);
    auto put = [](float* p, float v) { if (ACC) *p += v; else *p = v; };
    if (dsh_rest != nullptr) {  // split layout: dsh = this Gaussian's [3], dsh_rest = its [M-1][3]
        put(dsh + 0, coef[0] * dRGB[0]); put(dsh + 1, coef[0] * dRGB[1]); put(dsh + 2, coef[0] * dRGB[2]);
#pragma unroll
        for (int i = 1; i < 16; i++) {
            if (i < M && (!ACC || i < nact)) {
                const float cf = (i < nact) ? coef[i] : 0.0f;
                put(dsh_rest + 3 * (i - 1) + 0, cf * dRGB[0]); put(dsh_rest + 3 * (i - 1) + 1, cf * dRGB[1]);
                put(dsh_rest + 3 * (i - 1) + 2, cf * dRGB[2]);
            }
        }
        if (!ACC) for (int i = 45; i < (M - 1) * 3; i++) dsh_rest[i] = 0.0f;
    } else if (vec16) {  // M == 16, 16-byte aligned 192-byte record: twelve 16-byte stores
        float o[48];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float cf = (i < nact) ? coef[i] : 0.0f;
            o[3 * i] = cf * dRGB[0]; o[3 * i + 1] = cf * dRGB[1]; o[3 * i + 2] = cf * dRGB[2];
        }
        float4* o4 = reinterpret_cast<float4*>(dsh);
        if (ACC) {  // read-modify-write of the quads that hold active coefficients (all twelve loads first)
            const int nq = (3 * nact + 3) >> 2;
            float4 old[12];
#pragma unroll
            for (int q = 0; q < 12; q++) if (q < nq) old[q] = o4[q];
#pragma unroll
            for (int q = 0; q < 12; q++)
                if (q < nq) o4[q] = make_float4(old[q].x + o[4 * q], old[q].y + o[4 * q + 1], old[q].z + o[4 * q + 2], old[q].w + o[4 * q + 3]);
        } else {
#pragma unroll
            for (int q = 0; q < 12; q++) o4[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (i < M && (!ACC || i < nact)) {
                const float cf = (i < nact) ? coef[i] : 0.0f;
                put(dsh + 3 * i + 0, cf * dRGB[0]); put(dsh + 3 * i + 1, cf * dRGB[1]); put(dsh + 3 * i + 2, cf * dRGB[2]);
            }
        }
        if (!ACC) for (int i = 48; i < M * 3; i++) dsh[i] = 0.0f;
    }
    const F3 dL_ddir = mk3(dx[0] * dRGB[0] + dx[1] * dRGB[1] + dx[2] * dRGB[2],
                           dy[0] * dRGB[0] + dy[1] * dRGB[1] + dy[2] * dRGB[2],
                           dz[0] * dRGB[0] + dz[1] * dRGB[1] + dz[2] * dRGB[2]);
    return dnormvdv(dir_orig, dL_ddir);
}

// K8: backward.cu:586-641 + compute_transmat_aabb :443-584.
//
// Phase 1 folds every Gaussian's per-instance gradient records (written by the blend backward,
// one per (tile, Gaussian) instance, contiguous at [inst_off, inst_off + count)) in a fixed order
// -- the deterministic replacement of the reference's float atomics.  The block's 256 x 18
// (Gaussian, term) sums are spread over the threads so that consecutive lanes read consecutive
// floats of a record (coalesced 72-byte runs) and land in LDS.  Phase 2 is one thread per
// Gaussian.  Every output element is written (zeros for invisible Gaussians).
constexpr int K8_SUM_STRIDE = GRAD_FLOATS + 1;  // LDS row stride, odd => conflict-free column reads
// LDS staging of the block's outputs (floats per Gaussian: mean2D 3, normal 3, opacity 1, colour 3, mean3D 3, T 9, scale 2,
// rotation 4 = 28), one row-major [256, w] region per tensor
constexpr int K8_OUT_MEAN2D = 0, K8_OUT_NORMAL = 768, K8_OUT_OPACITY = 1536, K8_OUT_COLOR = 1792, K8_OUT_MEAN3D = 2560,
              K8_OUT_TRANSMAT = 3328, K8_OUT_SCALE = 5632, K8_OUT_ROT = 6144, K8_OUT_STATS = 7168, K8_OUT_FLOATS = 30;
// (K8_OUT_STATS: the optional per-view densification statistics [256, 2] = (||dL_dmean2D.xy||, visible) of
// g4s_rasterizer_backward_accumulate)

// K8, phase 1: the fold.
//
// A ROW of 16 lanes folds one Gaussian, four Gaussians per wave at a time, two such groups in flight: lane (kk, c)
// of a row reads quad c of record 4 s + kk, s = 0, 1, ... -- a typical run of ~10 records is three steps, all
// of whose loads are issued before any is consumed -- and the four kk partials are combined with two DPP row
// rotates.  The first design spent a whole wave (and two ds_bpermute exchanges) on every Gaussian: ~100 wave
// instructions per visible Gaussian, 40 M per launch at S3 -- it was bound by instruction issue, not by memory
// (0.169 ms for 0.44 GB).  Summation order: fixed (per lane ascending record index, then the DPP tree), so the
// gradients stay bit-reproducible.
//
// Unless the blend backward has cleared dL_dsh on the side (PreprocessBwdArgs::sh_prezeroed, frames that run the
// one-wave kernel), the fold phase also clears the dL_dsh rows of the Gaussians phase 2 will not write (radii == 0) -- 72 %
// of a 288 MB tensor at S3 -- with coalesced stores that overlap its own latency-bound gather.
struct FoldShZero {
    float* base;      // dL_dsh (or its [P,M-1,3] rest part); NULL = nothing to clear
    int row_floats;   // floats per Gaussian
};
__device__ __forceinline__ void fold_zero_rows(const FoldShZero z, int P, const uint8_t* s_vis) {
    if (z.base == nullptr) return;
    const int t = (int)threadIdx.x;
    const int rows = imin_(256, P - (int)blockIdx.x * 256);
    const size_t first = (size_t)blockIdx.x * 256 * z.row_floats;
    const int total = rows * z.row_floats;
    float* dst = z.base + first;
    if ((z.row_floats & 3) == 0 && (((size_t)dst) & 15) == 0) {  // rows are whole 16-byte quads
        const int rq = z.row_floats >> 2;
        for (int q = t; q < (total >> 2); q += 256)
            if (!s_vis[q / rq]) reinterpret_cast<float4*>(dst)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (int i = t; i < total; i += 256)
            if (!s_vis[i / z.row_floats]) dst[i] = 0.0f;
    }
}

// Phase 1 of K8 (device function: all 256 threads of the block call it).  On return -- after the trailing barrier --
// s_sum[g * K8_SUM_STRIDE + i] holds folded term i of local Gaussian g.
__device__ __forceinline__ void fold_block(const PreprocessBwdArgs& a, const FoldShZero z0, const FoldShZero z1,
                                           float* s_sum, uint32_t* s_off, uint32_t* s_cnt, uint8_t* s_vis) {
    const int t = (int)threadIdx.x;
    const int idx = (int)(blockIdx.x * 256 + t);
    const bool in_range = idx < a.P;
    const bool visible = in_range && a.radii[idx] > 0;
    const float4* rq = reinterpret_cast<const float4*>(a.rec) + (size_t)(in_range ? idx : 0) * REC_QUADS;
    uint32_t my_off = 0, my_cnt = 0;
    if (visible) {
        const float4 q0 = rq[0];
        my_off = __float_as_uint(q0.z);
        my_cnt = rect_tiles(__float_as_uint(q0.w));
        // (only a frame that overflowed its presized capacity has slots beyond the buffers: never read them)
        if (my_off >= a.n_slots) my_cnt = 0;
        else if (my_cnt > a.n_slots - my_off) my_cnt = a.n_slots - my_off;
    }
    s_off[t] = my_off;
    s_cnt[t] = my_cnt;
    s_vis[t] = visible ? 1 : 0;
    __syncthreads();
    fold_zero_rows(z0, a.P, s_vis);
    fold_zero_rows(z1, a.P, s_vis);
    {
        const int lane = lane_id();
        const int wbase = (t >> 6) * 64;  // first local Gaussian of this wave
        const int row = lane >> 4, kk = (lane >> 2) & 3, c = lane & 3;
        // Runs longer than FOLD_ROW_MAX records (near splats that cover hundreds of tiles) are folded by the whole
        // wave, 128 records per trip; in a 16-lane row they would take cnt/16 dependent round trips while the rest of
        // the wave idles (one 2 000-instance splat: 125 trips)
        constexpr uint32_t FOLD_ROW_MAX = 48;
        uint64_t big = __ballot(my_cnt > FOLD_ROW_MAX);
        while (big) {
            const int j = (int)__builtin_ctzll(big);
            big &= big - 1;
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)my_cnt, j);
            const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)my_off, j);
            const float4* base4 = reinterpret_cast<const float4*>(a.grad_inst + (size_t)off * GRAD_STRIDE);
            const uint32_t k16 = (uint32_t)(lane >> 2);  // record within a group of 16
            float4 sA = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 sB = make_float2(0.f, 0.f);
            for (uint32_t k0 = 0; k0 < cnt; k0 += 128) {
                float4 x[8];
                uint8_t f[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t k = k0 + 16 * i + k16;
                    const uint32_t kc = k < cnt ? k : 0u;
                    x[i] = base4[(size_t)kc * (GRAD_STRIDE / 4) + c];
                    f[i] = a.rec_flag[off + kc];
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t k = k0 + 16 * i + k16;
                    if (k < cnt && (f[i] & 1)) { sA.x += x[i].x; sA.y += x[i].y; sA.z += x[i].z; sA.w += x[i].w; }
                    if (k < cnt && (f[i] & 2) && c == 0) {
                        const float2 y = *reinterpret_cast<const float2*>(a.grad_inst + (size_t)(off + k) * GRAD_STRIDE + 16);
                        sB.x += y.x; sB.y += y.y;
                    }
                }
            }
            // over the 16 record lanes: two rotates inside each row of 16, then the four rows
            sA.x += dpp_f32<0x124>(sA.x); sA.y += dpp_f32<0x124>(sA.y); sA.z += dpp_f32<0x124>(sA.z); sA.w += dpp_f32<0x124>(sA.w);
            sA.x += dpp_f32<0x128>(sA.x); sA.y += dpp_f32<0x128>(sA.y); sA.z += dpp_f32<0x128>(sA.z); sA.w += dpp_f32<0x128>(sA.w);
            sA.x += __shfl_xor(sA.x, 16, 64); sA.y += __shfl_xor(sA.y, 16, 64); sA.z += __shfl_xor(sA.z, 16, 64); sA.w += __shfl_xor(sA.w, 16, 64);
            sA.x += __shfl_xor(sA.x, 32, 64); sA.y += __shfl_xor(sA.y, 32, 64); sA.z += __shfl_xor(sA.z, 32, 64); sA.w += __shfl_xor(sA.w, 32, 64);
            sB.x += dpp_f32<0x124>(sB.x); sB.y += dpp_f32<0x124>(sB.y);
            sB.x += dpp_f32<0x128>(sB.x); sB.y += dpp_f32<0x128>(sB.y);
            sB.x += __shfl_xor(sB.x, 16, 64); sB.y += __shfl_xor(sB.y, 16, 64);
            sB.x += __shfl_xor(sB.x, 32, 64); sB.y += __shfl_xor(sB.y, 32, 64);
            float* dst = s_sum + (wbase + j) * K8_SUM_STRIDE;
            if (lane < 4) { dst[4 * lane] = sA.x; dst[4 * lane + 1] = sA.y; dst[4 * lane + 2] = sA.z; dst[4 * lane + 3] = sA.w; }
            if (lane == 0) { dst[16] = sB.x; dst[17] = sB.y; }
        }
        uint64_t vis = __ballot(my_cnt != 0 && my_cnt <= FOLD_ROW_MAX);
        constexpr int U = 3;  // groups of four Gaussians in flight
        while (vis) {
            // group u, row r folds the (4 u + r)-th remaining member of the wave that has records
            int jj[U];
            uint32_t cn[U], of[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                uint64_t m = vis;
                for (int i = 0; i < row; i++) m &= m - 1;  // drop the members the lower rows take
                jj[u] = m ? (int)__builtin_ctzll(m) : -1;
#pragma unroll
                for (int i = 0; i < 4; i++) vis &= vis - 1;  // (scalar) the four members of this group are taken
                cn[u] = jj[u] >= 0 ? s_cnt[wbase + jj[u]] : 0u;
                of[u] = jj[u] >= 0 ? s_off[wbase + jj[u]] : 0u;
            }
            float4 acc[U];
            float2 accB[U];
#pragma unroll
            for (int u = 0; u < U; u++) { acc[u] = make_float4(0.f, 0.f, 0.f, 0.f); accB[u] = make_float2(0.f, 0.f); }
            // the longest run of the eight decides the trip count (uniform); four steps = 16 records per trip
            uint32_t cmax = 0;  // scalar: cn[u] is uniform inside a row
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int r = 0; r < 4; r++) cmax = max(cmax, (uint32_t)__builtin_amdgcn_readlane((int)cn[u], 16 * r));
            for (uint32_t k0 = 0; k0 < cmax; k0 += 16) {
                float4 x[U][4];
                uint8_t f[U][4];
                // the validity bytes first, then only the records the blend backward wrote (about half of the slots
                // belong to instances that no pixel blended): the kernel is bound by HBM traffic, and the 80-byte
                // records are most of what it reads
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t k = k0 + 4 * i + kk;
                        f[u][i] = k < cn[u] ? a.rec_flag[of[u] + k] : (uint8_t)0;
                    }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const float4* base4 = reinterpret_cast<const float4*>(a.grad_inst + (size_t)of[u] * GRAD_STRIDE);
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t k = k0 + 4 * i + kk;
                        x[u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (f[u][i] & 1) x[u][i] = base4[(size_t)k * (GRAD_STRIDE / 4) + c];
                    }
                }
                // records the blend backward never wrote hold garbage (possibly NaN): select, do not multiply
#pragma unroll
                for (int u = 0; u < U; u++) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t k = k0 + 4 * i + kk;
                        if (f[u][i] & 1) {  // (f is 0 beyond the run)
                            acc[u].x += x[u][i].x; acc[u].y += x[u][i].y; acc[u].z += x[u][i].z; acc[u].w += x[u][i].w;
                        }
                        if ((f[u][i] & 2) && c == 0) {  // the rare low-pass centre terms 16..17
                            const float2 y = *reinterpret_cast<const float2*>(a.grad_inst + (size_t)(of[u] + k) * GRAD_STRIDE + 16);
                            accB[u].x += y.x; accB[u].y += y.y;
                        }
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                float4 sA = acc[u];
                float2 sB = accB[u];
                // over kk: rotate by 4 and 8 lanes inside the row of 16
                sA.x += dpp_f32<0x124>(sA.x); sA.y += dpp_f32<0x124>(sA.y); sA.z += dpp_f32<0x124>(sA.z); sA.w += dpp_f32<0x124>(sA.w);
                sA.x += dpp_f32<0x128>(sA.x); sA.y += dpp_f32<0x128>(sA.y); sA.z += dpp_f32<0x128>(sA.z); sA.w += dpp_f32<0x128>(sA.w);
                sB.x += dpp_f32<0x124>(sB.x); sB.y += dpp_f32<0x124>(sB.y);
                sB.x += dpp_f32<0x128>(sB.x); sB.y += dpp_f32<0x128>(sB.y);
                if (jj[u] >= 0 && kk == 0) {
                    float* dst = s_sum + (wbase + jj[u]) * K8_SUM_STRIDE;
                    dst[4 * c] = sA.x; dst[4 * c + 1] = sA.y; dst[4 * c + 2] = sA.z; dst[4 * c + 3] = sA.w;  // terms 4c..4c+3
                    if (c == 0) { dst[16] = sB.x; dst[17] = sB.y; }
                }
            }
        }
        // members without records fold to zero
        if (my_cnt == 0) {
#pragma unroll
            for (int i = 0; i < GRAD_FLOATS; i++) s_sum[t * K8_SUM_STRIDE + i] = 0.0f;
        }
    }
    __syncthreads();
}

// K8: fold (phase 1, above) + one thread per Gaussian (phase 2) in ONE kernel: the 256 x 18 folded terms of the
// block stay in LDS -- the first design wrote them to HBM from a lean fold kernel and read them back here
// (2 x 108 MB at S3 and a launch), to give the gather more resident waves; once the fold stopped being issue-bound
// (16-lane rows) that round trip was the larger cost.
// ACC (PreprocessBwdArgs::accumulate, gradient accumulation over views): the parameter gradients -- dL_dmean3D,
// dL_dopacity, dL_dscale, dL_drot, dL_dsh -- are ADDED to what their tensors hold; blocks without a visible Gaussian
// touch none of them and nothing is zero-filled.  The per-view outputs (dL_dmean2D, dL_dcolor, dL_dnormal,
// dL_dtransMat) are written as always.
template <int SH_MODE, bool ACC>
__global__ void __launch_bounds__(256) preprocess_bwd_kernel(PreprocessBwdArgs a, FoldShZero z0, FoldShZero z1) {
    // one LDS buffer, used twice: the folded terms (256 x 19 floats) of phase 1, then the block's 256 x 28 output floats
    // on their way to coalesced 16-byte stores (K8_OUT_* below)
    __shared__ float s_buf[256 * K8_OUT_FLOATS];
    float* s_sum = s_buf;
    __shared__ uint32_t s_off[256], s_cnt[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_vis[256];  // (read back sixteen bytes at a time for the packed rows)
    const int t = (int)threadIdx.x;
    const int idx = (int)(blockIdx.x * 256 + t);
    const bool in_range = idx < a.P;
    fold_block(a, z0, z1, s_sum, s_off, s_cnt, s_vis);
    const bool visible = s_vis[t] != 0;
    // the packed gradient row of this Gaussian, if the caller asked for them (visible Gaussians in index order)
    // (the rows of a block are consecutive in the packed buffer: they are assembled in LDS and leave as one coalesced run)
    uint32_t prank = 0, pcount = 0, pbase = 0;  // this Gaussian's rank among the block's visible ones, their number, the block's first row
    if (a.packed_rows != nullptr) {
#pragma unroll
        for (int w = 0; w < 4; w++) {  // (s_vis is one byte per thread, 0 / 1: sixty-four bytes per wave)
            uint32_t c = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint4 v = reinterpret_cast<const uint4*>(s_vis)[4 * w + i];
                c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
            }
            if (w < (t >> 6)) prank += c;
            pcount += c;
        }
        prank += (uint32_t)__popcll(__ballot(visible) & lanes_below_mask());
        pbase = a.packed_block_offs[blockIdx.x];
    }
    const float4* rq = reinterpret_cast<const float4*>(a.rec) + (size_t)(in_range ? idx : 0) * REC_QUADS;

    // record order (blend backward): 0..14 = colour, normal, T; 15 = opacity; 16..17 = low-pass centre terms
    float g[GRAD_FLOATS];
#pragma unroll
    for (int i = 0; i < 15; i++) g[i] = s_sum[t * K8_SUM_STRIDE + i];
    g[17] = s_sum[t * K8_SUM_STRIDE + 15];
    g[15] = s_sum[t * K8_SUM_STRIDE + 16];
    g[16] = s_sum[t * K8_SUM_STRIDE + 17];
    // g: [0..2] colour, [3..5] normal, [6..14] T (Tu,Tv,Tw), [15..16] mean2D, [17] opacity
    float dmean3[3] = {0, 0, 0}, dscale[2] = {0, 0};
    float4 drot = make_float4(0, 0, 0, 0);
    float dmean2[3] = {0, 0, 0};
    float dT_out[9];
#pragma unroll
    for (int i = 0; i < 9; i++) dT_out[i] = g[6 + i];
    constexpr bool split_sh = SH_MODE == 2;
    float* dsh = a.M > 0 ? a.dL_dsh + (size_t)idx * (split_sh ? 1 : a.M) * 3 : nullptr;
    float* dsh_rest = split_sh ? a.dL_dsh_rest + (size_t)idx * (a.M - 1) * 3 : nullptr;

    if (visible) {
        const bool precomp = (a.scales == nullptr);
        const float4 q0 = rq[0], q2 = rq[2], q3 = rq[3], q4 = rq[4], q5 = rq[5];
        const bool affine = (__float_as_uint(q0.w) & REC_AFFINE) != 0;
        const float depth_T8 = q5.z;  // transMats[idx*9+8]
        float T[9];   // the backward's own T (scale_modifier ignored, truncated W / H: backward.cu:481, 618-619)
        float Tf[9];  // the forward's T: what the moments of a REC_AFFINE splat refer to
        float Pm[3][4];
        float R[9];
        F3 normal = mk3(0, 0, 0), p_orig = mk3(0, 0, 0);
        float sx = 0, sy = 0;
        float4 rot = make_float4(1, 0, 0, 0);
        if (precomp) {
#pragma unroll
            for (int i = 0; i < 9; i++) Tf[i] = T[i] = a.transMat_precomp[9 * (size_t)idx + i];
        } else {
            p_orig = mk3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
            rot = reinterpret_cast<const float4*>(a.rotations)[idx];
            const float2 sc = reinterpret_cast<const float2*>(a.scales)[idx];
            sx = sc.x;
            sy = sc.y;
            quat_to_rotmat(rot, R);
            const float s0 = 1.0f * sx, s1 = 1.0f * sy;  // scale_modifier ignored (backward.cu:481)
            const float Mrow[3][4] = {{R[0] * s0, R[1] * s0, R[2] * s0, 0.0f},
                                      {R[3] * s1, R[4] * s1, R[5] * s1, 0.0f},
                                      {p_orig.x, p_orig.y, p_orig.z, 1.0f}};
            const float hw = (float)((float)a.W / 2.0), cw = (float)((float)(a.W - 1) / 2.0);
            const float hh = (float)((float)a.H / 2.0), ch = (float)((float)(a.H - 1) / 2.0);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Pm[0][i] = a.projmatrix[4 * i + 0] * hw + a.projmatrix[4 * i + 3] * cw;
                Pm[1][i] = a.projmatrix[4 * i + 1] * hh + a.projmatrix[4 * i + 3] * ch;
                Pm[2][i] = a.projmatrix[4 * i + 3];
            }
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int r = 0; r < 3; r++)
                    T[c * 3 + r] = Mrow[r][0] * Pm[c][0] + Mrow[r][1] * Pm[c][1] + Mrow[r][2] * Pm[c][2] +
                                   Mrow[r][3] * Pm[c][3];
            normal = xform_vec_4x3(mk3(R[6], R[7], R[8]), a.viewmatrix);
            if (affine) compute_transmat(p_orig, sx, sy, a.scale_modifier, R, a.projmatrix, a.frame_W, a.frame_H, Tf);  // as K1 did
        }
        float dL_dT[9];
#pragma unroll
        for (int i = 0; i < 9; i++) dL_dT[i] = g[6 + i];  // general path: the blend backward summed dL/dTu, dL/dTv, dL/dTw
        // REC_AFFINE: dL/dT from the moments S, X, Y of dL/dp' (backward.cu:396-426, taken once per Gaussian instead of
        // once per pixel).  With Q(x, y) = x A + y B + D = k x l, A = Tv x Tw, B = Tw x Tu, D = Tu x Tv, det = det(T) and
        // p' = gam Q, gam = 1 / det:
        //   dL = sum_pixels dL/dp' . (gam dQ + Q dgam),   sum dL/dp' . p' = -sum depth dL/ddepth =: -Zd   (s is homogeneous in p')
        //   dL/dTu = gam (Y x Tw + S x lc) + gdet A         kc = cx Tw - Tu, lc = cy Tw - Tv (k, l at the splat centre:
        //   dL/dTv = gam (Tw x X + kc x S) + gdet B          the moments X, Y are taken about it, so the frame-sized
        //   dL/dTw = -cx (.)u - cy (.)v + gam (lc x X + Y x kc) + gdet D      terms cancel here, once, not in the sums)
        // with gdet = Zd / det = dL/d det (depth = det / Q.z), (.)u / (.)v the gam parts of the first two lines.
        if (affine) {
            const float cx = q0.x, cy = q0.y;
            const F3 Ap = mk3(q2.x, q2.y, q2.z), Bp = mk3(q2.w, q3.x, q3.y), Dp = mk3(q3.z, q3.w, q4.x);
            const F3 S = mk3(g[6], g[7], g[8]), X = mk3(g[9], g[10], g[11]), Y = mk3(g[12], g[13], g[14]);
            const F3 Tu = mk3(Tf[0], Tf[1], Tf[2]), Tv = mk3(Tf[3], Tf[4], Tf[5]), Tw = mk3(Tf[6], Tf[7], Tf[8]);
            auto crs = [](F3 u, F3 v) { return mk3(fmaf(u.y, v.z, -(u.z * v.y)), fmaf(u.z, v.x, -(u.x * v.z)), fmaf(u.x, v.y, -(u.y * v.x))); };
            auto dot3 = [](F3 u, F3 v) { return fmaf(u.x, v.x, fmaf(u.y, v.y, u.z * v.z)); };
            const float Zd = -(dot3(Ap, X) + dot3(Bp, Y) + dot3(Dp, S));
            const F3 kc = mk3(fmaf(cx, Tw.x, -Tu.x), fmaf(cx, Tw.y, -Tu.y), fmaf(cx, Tw.z, -Tu.z));
            const F3 lc = mk3(fmaf(cy, Tw.x, -Tv.x), fmaf(cy, Tw.y, -Tv.y), fmaf(cy, Tw.z, -Tv.z));
            // det(T) = (kc x lc) . Tw in double: near edge-on splats lose digits in the 2 x 2 minors
            const double kd[3] = {(double)cx * Tw.x - Tu.x, (double)cx * Tw.y - Tu.y, (double)cx * Tw.z - Tu.z};
            const double ld[3] = {(double)cy * Tw.x - Tv.x, (double)cy * Tw.y - Tv.y, (double)cy * Tw.z - Tv.z};
            const double det = (kd[1] * ld[2] - kd[2] * ld[1]) * Tw.x + (kd[2] * ld[0] - kd[0] * ld[2]) * Tw.y +
                               (kd[0] * ld[1] - kd[1] * ld[0]) * Tw.z;
            const float gam = 1.0f / (float)det;  // (REC_AFFINE: det(T) is invertible; one single-precision division, not a double one)
            const float gdet = Zd * gam;
            const F3 u1 = crs(Y, Tw), u2 = crs(S, lc), v1 = crs(Tw, X), v2 = crs(kc, S), w1 = crs(lc, X), w2 = crs(Y, kc);
            const F3 U = mk3(gam * (u1.x + u2.x), gam * (u1.y + u2.y), gam * (u1.z + u2.z));
            const F3 V = mk3(gam * (v1.x + v2.x), gam * (v1.y + v2.y), gam * (v1.z + v2.z));
            const F3 Wg = mk3(gam * (w1.x + w2.x), gam * (w1.y + w2.y), gam * (w1.z + w2.z));
            const F3 Aa = crs(Tv, Tw), Ba = crs(Tw, Tu), Da = crs(Tu, Tv);
            dL_dT[0] = fmaf(gdet, Aa.x, U.x); dL_dT[1] = fmaf(gdet, Aa.y, U.y); dL_dT[2] = fmaf(gdet, Aa.z, U.z);
            dL_dT[3] = fmaf(gdet, Ba.x, V.x); dL_dT[4] = fmaf(gdet, Ba.y, V.y); dL_dT[5] = fmaf(gdet, Ba.z, V.z);
            dL_dT[6] = fmaf(gdet, Da.x, Wg.x - fmaf(cx, U.x, cy * V.x));
            dL_dT[7] = fmaf(gdet, Da.y, Wg.y - fmaf(cx, U.y, cy * V.y));
            dL_dT[8] = fmaf(gdet, Da.z, Wg.z - fmaf(cx, U.z, cy * V.z));
#pragma unroll
            for (int i = 0; i < 9; i++) dT_out[i] = dL_dT[i];
        }
        const float mx = g[15], my = g[16];
        bool early_out = false;
        if (mx != 0 || my != 0) {  // backward.cu:513-556, low-pass centre path (cutoff-1 formula)
            const float* T0 = T;
            const float* T1 = T + 3;
            const float* T2 = T + 6;
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
            if (precomp) {
#pragma unroll
                for (int i = 0; i < 9; i++) dT_out[i] = dL_dT[i];
                early_out = true;
            }
        }
        if (!precomp && !early_out) {
            float dL_dM[3][4];
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int i = 0; i < 4; i++)
                    dL_dM[c][i] = Pm[0][i] * dL_dT[0 * 3 + c] + Pm[1][i] * dL_dT[1 * 3 + c] + Pm[2][i] * dL_dT[2 * 3 + c];
            F3 dL_dtn = xform_vec_4x3_T(mk3(g[3], g[4], g[5]), a.viewmatrix);
            const F3 p_view = xform_point_4x3(p_orig, a.viewmatrix);
            const float cosv = -((p_view.x * normal.x + p_view.y * normal.y) + p_view.z * normal.z);
            const float mult = cosv > 0 ? 1.0f : -1.0f;
            dL_dtn = mk3(mult * dL_dtn.x, mult * dL_dtn.y, mult * dL_dtn.z);
            const float dL_dRS[9] = {dL_dM[0][0], dL_dM[0][1], dL_dM[0][2], dL_dM[1][0], dL_dM[1][1],
                                     dL_dM[1][2], dL_dtn.x,    dL_dtn.y,    dL_dtn.z};
            const float dL_dR[9] = {dL_dRS[0] * sx, dL_dRS[1] * sx, dL_dRS[2] * sx, dL_dRS[3] * sy, dL_dRS[4] * sy,
                                    dL_dRS[5] * sy, dL_dRS[6],      dL_dRS[7],      dL_dRS[8]};
            drot = quat_to_rotmat_vjp(rot, dL_dR);
            dscale[0] = dL_dRS[0] * R[0] + dL_dRS[1] * R[1] + dL_dRS[2] * R[2];
            dscale[1] = dL_dRS[3] * R[3] + dL_dRS[4] * R[4] + dL_dRS[5] * R[5];
            dmean3[0] = dL_dM[2][0];
            dmean3[1] = dL_dM[2][1];
            dmean3[2] = dL_dM[2][2];
        }
        if (a.shs != nullptr) {
            float sh[48];
            load_sh(a.shs, SH_MODE == 2 ? a.shs_rest : nullptr, (size_t)idx, a.M, a.D, SH_MODE == 0, sh);
            const F3 dm = sh_backward<ACC>(a.D, a.M, sh, mk3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]),
                                      mk3(a.campos[0], a.campos[1], a.campos[2]), a.clamped[idx], g, dsh, dsh_rest, SH_MODE == 0);
            dmean3[0] += dm.x;
            dmean3[1] += dm.y;
            dmean3[2] += dm.z;
        }
        // densification surrogate (backward.cu:637-640): uses the raw blend-accumulated dL_dT
        // (or, on the precomputed-T path with a centre gradient, the folded one the reference wrote back)
        dmean2[0] = (float)(dT_out[2] * depth_T8 * 0.5 * (float)a.W);
        dmean2[1] = (float)(dT_out[5] * depth_T8 * 0.5 * (float)a.H);
    }
    // (rows of dL_dsh that belong to invisible Gaussians: cleared by the fold phase above or beside the blend backward)
    // Outputs: every thread parks its 28 floats in LDS, then the block writes each tensor's 256-row region with
    // coalesced 16-byte stores -- 7 store instructions per thread instead of 28 strided dword stores, three quarters of
    // which only carried the zeros of invisible Gaussians.
    // (the barrier: every thread has taken its folded terms out of s_buf)
    const bool any_visible = __syncthreads_or(visible ? 1 : 0) != 0;
    {
        float* o = s_buf;
        o[K8_OUT_MEAN2D + 3 * t] = dmean2[0]; o[K8_OUT_MEAN2D + 3 * t + 1] = dmean2[1]; o[K8_OUT_MEAN2D + 3 * t + 2] = dmean2[2];
        o[K8_OUT_NORMAL + 3 * t] = g[3]; o[K8_OUT_NORMAL + 3 * t + 1] = g[4]; o[K8_OUT_NORMAL + 3 * t + 2] = g[5];
        o[K8_OUT_OPACITY + t] = g[17];
        o[K8_OUT_COLOR + 3 * t] = g[0]; o[K8_OUT_COLOR + 3 * t + 1] = g[1]; o[K8_OUT_COLOR + 3 * t + 2] = g[2];
        o[K8_OUT_MEAN3D + 3 * t] = dmean3[0]; o[K8_OUT_MEAN3D + 3 * t + 1] = dmean3[1]; o[K8_OUT_MEAN3D + 3 * t + 2] = dmean3[2];
#pragma unroll
        for (int i = 0; i < 9; i++) o[K8_OUT_TRANSMAT + 9 * t + i] = dT_out[i];
        o[K8_OUT_SCALE + 2 * t] = dscale[0]; o[K8_OUT_SCALE + 2 * t + 1] = dscale[1];
        *reinterpret_cast<float4*>(o + K8_OUT_ROT + 4 * t) = drot;
        // gaussian_model.py:649-651: the norm of THIS view's screen-space gradient (its z is 0) and the visibility count
        o[K8_OUT_STATS + 2 * t] = visible ? sqrtf(dmean2[0] * dmean2[0] + dmean2[1] * dmean2[1]) : 0.0f;
        o[K8_OUT_STATS + 2 * t + 1] = visible ? 1.0f : 0.0f;
    }
    __syncthreads();
    {
        const int rows = imin_(256, a.P - (int)blockIdx.x * 256);
        auto flush = [&](float* base, int w, int lds_off, bool acc) {
            if (base == nullptr) return;
            if (acc && !any_visible) return;  // adding a block of zeros: leave the sums alone
            float* dst = base + (size_t)blockIdx.x * 256 * w;
            const float* src = s_buf + lds_off;
            const int n = rows * w;
            if ((((size_t)dst) & 15) == 0) {
                for (int i = t; i < (n >> 2); i += 256) {
                    float4 v = reinterpret_cast<const float4*>(src)[i];
                    if (acc) {
                        const float4 u = reinterpret_cast<const float4*>(dst)[i];
                        v = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
                    }
                    reinterpret_cast<float4*>(dst)[i] = v;
                }
                for (int i = (n & ~3) + t; i < n; i += 256) dst[i] = acc ? dst[i] + src[i] : src[i];
            } else {
                for (int i = t; i < n; i += 256) dst[i] = acc ? dst[i] + src[i] : src[i];
            }
        };
        flush(a.dL_dmean2D, 3, K8_OUT_MEAN2D, false);
        flush(a.dL_dnormal, 3, K8_OUT_NORMAL, false);
        flush(a.dL_dopacity, 1, K8_OUT_OPACITY, ACC);
        flush(a.dL_dcolor, 3, K8_OUT_COLOR, false);
        flush(a.dL_dmean3D, 3, K8_OUT_MEAN3D, ACC);
        flush(a.dL_dtransMat, 9, K8_OUT_TRANSMAT, false);
        flush(a.dL_dscale, 2, K8_OUT_SCALE, ACC);
        flush(a.dL_drot, 4, K8_OUT_ROT, ACC);
        flush(a.view_stats, 2, K8_OUT_STATS, ACC);
    }
    // The packed copy of the visible Gaussians' parameter gradients (PreprocessBwdArgs::packed_rows): the block's rows are
    // consecutive in the buffer, so they are assembled in LDS -- the SH block is formed again from the colour gradient and
    // the direction, 48 multiplies, instead of keeping 48 registers alive across the flush above -- and leave as one
    // coalesced run of dwords.  (Written straight from the threads, 61 dword stores at a 244-byte lane stride, the same
    // rows cost the kernel 0.064 ms at S3; tools/micro/prepack_cost.py.)  K8_PACK_ROWS rows fit the buffer; a block with
    // more visible Gaussians (half of its 256) writes the surplus directly.
    if (a.packed_rows != nullptr && pcount != 0) {
        const int RW = 3 * a.M + 13;
        const int lds_rows = imin_((int)pcount, (256 * K8_OUT_FLOATS) / RW);
        __syncthreads();  // the flush has read s_buf
        if (visible && pbase + prank < a.packed_capacity) {
            float* row = (int)prank < lds_rows ? s_buf + (size_t)prank * RW : a.packed_rows + (size_t)(pbase + prank) * RW;
            row[0] = dmean3[0]; row[1] = dmean3[1]; row[2] = dmean3[2];
            // dL_dsh = coef(direction) x dL_dcolor, clamped channels zeroed (sh_backward, backward.cu:20-139)
            const F3 pos = mk3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
            float coef[16];
            sh_coefficients(a.D, pos, mk3(a.campos[0], a.campos[1], a.campos[2]), coef);
            const uint32_t cb = a.clamped[idx];
            // (times 0 / 1 like sh_backward, not a select: a clamped channel's gradient keeps the sign of its zero)
            const float d0 = g[0] * ((cb & 1u) ? 0.0f : 1.0f), d1 = g[1] * ((cb & 2u) ? 0.0f : 1.0f), d2 = g[2] * ((cb & 4u) ? 0.0f : 1.0f);
            for (int i = 0; i < a.M; i++) {
                const float cf = i < 16 ? coef[i] : 0.0f;
                row[3 + 3 * i] = cf * d0; row[3 + 3 * i + 1] = cf * d1; row[3 + 3 * i + 2] = cf * d2;
            }
            float* q = row + 3 + 3 * a.M;
            q[0] = g[17];
            q[1] = dscale[0]; q[2] = dscale[1];
            q[3] = drot.x; q[4] = drot.y; q[5] = drot.z; q[6] = drot.w;
            q[7] = sqrtf(dmean2[0] * dmean2[0] + dmean2[1] * dmean2[1]);  // (the view_stats columns)
            q[8] = 1.0f;
            q[9] = __int_as_float(idx);
        }
        __syncthreads();
        const uint32_t room = a.packed_capacity > pbase ? a.packed_capacity - pbase : 0u;
        const int n = imin_(lds_rows, (int)(room < 256u ? room : 256u)) * RW;
        float* dst = a.packed_rows + (size_t)pbase * RW;
        for (int i = t; i < n; i += 256) dst[i] = s_buf[i];
    }
}

void launch_preprocess_bwd(const PreprocessBwdArgs& a, hipStream_t s) {
    if (a.P <= 0) return;
    // dL_dsh rows of the Gaussians phase 2 does not write: cleared by the fold phase unless the blend backward did it
    FoldShZero z0{nullptr, 0}, z1{nullptr, 0};
    if (a.M > 0 && a.dL_dsh != nullptr && !a.sh_prezeroed && !a.accumulate) {
        if (a.shs_rest != nullptr) {
            z0 = FoldShZero{a.dL_dsh, 3};
            if (a.M > 1) z1 = FoldShZero{a.dL_dsh_rest, (a.M - 1) * 3};
        } else {
            z0 = FoldShZero{a.dL_dsh, a.M * 3};
        }
    }
    const dim3 grid((a.P + 255) / 256), block(256);
    if (a.accumulate) {
        if (a.shs_rest != nullptr) hipLaunchKernelGGL((preprocess_bwd_kernel<2, true>), grid, block, 0, s, a, z0, z1);
        else if (a.sh_vec16) hipLaunchKernelGGL((preprocess_bwd_kernel<0, true>), grid, block, 0, s, a, z0, z1);
        else hipLaunchKernelGGL((preprocess_bwd_kernel<1, true>), grid, block, 0, s, a, z0, z1);
    } else {
        if (a.shs_rest != nullptr) hipLaunchKernelGGL((preprocess_bwd_kernel<2, false>), grid, block, 0, s, a, z0, z1);
        else if (a.sh_vec16) hipLaunchKernelGGL((preprocess_bwd_kernel<0, false>), grid, block, 0, s, a, z0, z1);
        else hipLaunchKernelGGL((preprocess_bwd_kernel<1, false>), grid, block, 0, s, a, z0, z1);
    }
}

}  // namespace g4s
