// Fused map post-processing of render() (include/g4s_render_maps.h).
//
// Reference semantics: 2dgs/gaussian_renderer/__init__.py:117-164 and 2dgs/utils/point_utils.py:9-37
// (about fifteen element-wise torch kernels + two 3x3 matmuls over [H,W,3] + a cross product).
//
// MI355X design: pure HBM streaming.  One thread per pixel, 64x4-pixel blocks (a wave reads 256
// contiguous bytes of every plane); the 4-neighbour stencil of the depth-to-normal step re-reads
// neighbouring pixels through L1/L2 instead of staging tiles (the planes are read once from HBM:
// 28 B in + 64 B out per pixel forward).  The backward is the gather form of the stencil's adjoint
// -- every pixel recomputes the normals of its four neighbours from the saved surf_depth -- so there
// are no atomics and the result is bit-reproducible.  The camera algebra (two 4x4 / 3x3 inverses)
// runs once per call in one thread, in double, into a 160-byte device scratch.
#include "g4s_internal.h"
#include "g4s_device.h"

namespace g4s {

// ---- camera algebra (point_utils.py:10-21) ------------------------------------------------------
__device__ bool invert4(const double* m, double* inv) {
    // adjugate / determinant of a row-major 4x4
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0.0) return false;
    const double r = 1.0 / det;
    for (int i = 0; i < 16; i++) inv[i] *= r;
    return true;
}

// cam: [0..8] Rv = world_view_transform[:3,:3] (row-major), [9..17] Rd (rays_d = (x, y, 1) @ Rd),
//      [18..20] rays_o.  A singular camera gives NaNs, as torch.inverse would raise.
__global__ void maps_camera_kernel(const float* __restrict__ wvt, const float* __restrict__ fpt, int W, int H,
                                   float* __restrict__ cam) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double Wm[16], Winv[16], FP[16];
    for (int i = 0; i < 16; i++) { Wm[i] = wvt[i]; FP[i] = fpt[i]; }
    const bool ok = invert4(Wm, Winv);  // c2w = (wvt^T)^-1 = Winv^T
    // projection_matrix = c2w^T @ full_proj_transform = Winv @ FP
    double M[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += Winv[4 * r + k] * FP[4 * k + c];
            M[4 * r + c] = s;
        }
    // ndc2pix (4x3) = [[W/2,0,0,W/2],[0,H/2,0,H/2],[0,0,0,1]]^T ; intrins = (M @ ndc2pix)[:3,:3]^T
    const double N[12] = {W / 2.0, 0, 0, 0, H / 2.0, 0, 0, 0, 0, W / 2.0, H / 2.0, 1.0};
    double K[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 4; k++) s += M[4 * r + k] * N[3 * k + c];
            K[3 * c + r] = s;  // transposed
        }
    const double det = K[0] * (K[4] * K[8] - K[5] * K[7]) - K[1] * (K[3] * K[8] - K[5] * K[6]) + K[2] * (K[3] * K[7] - K[4] * K[6]);
    const double rdet = 1.0 / det;
    double Ki[9];
    Ki[0] = (K[4] * K[8] - K[5] * K[7]) * rdet; Ki[1] = (K[2] * K[7] - K[1] * K[8]) * rdet; Ki[2] = (K[1] * K[5] - K[2] * K[4]) * rdet;
    Ki[3] = (K[5] * K[6] - K[3] * K[8]) * rdet; Ki[4] = (K[0] * K[8] - K[2] * K[6]) * rdet; Ki[5] = (K[2] * K[3] - K[0] * K[5]) * rdet;
    Ki[6] = (K[3] * K[7] - K[4] * K[6]) * rdet; Ki[7] = (K[1] * K[6] - K[0] * K[7]) * rdet; Ki[8] = (K[0] * K[4] - K[1] * K[3]) * rdet;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            cam[3 * r + c] = wvt[4 * r + c];
            // rays_d = pix @ Ki^T @ c2w[:3,:3]^T = pix @ (Ki^T @ Winv[:3,:3])
            double s = 0;
            for (int k = 0; k < 3; k++) s += Ki[3 * k + r] * Winv[4 * k + c];
            cam[9 + 3 * r + c] = (float)(ok ? s : nan);
        }
    for (int c = 0; c < 3; c++) cam[18 + c] = (float)(ok ? Winv[12 + c] : nan);  // rays_o = c2w[:3,3]
}

struct MapsCam {
    float Rv[9], Rd[9], o[3];
};
__device__ __forceinline__ MapsCam load_cam(const float* __restrict__ cam) {
    MapsCam c;
#pragma unroll
    for (int i = 0; i < 9; i++) { c.Rv[i] = cam[i]; c.Rd[i] = cam[9 + i]; }
#pragma unroll
    for (int i = 0; i < 3; i++) c.o[i] = cam[18 + i];
    return c;
}

// torch.nan_to_num(x, 0, 0): nan -> 0, +inf -> 0, -inf -> lowest finite (neginf keeps its default)
__device__ __forceinline__ float nan_to_num00(float v) {
    if (v != v) return 0.0f;
    if (v == __builtin_huge_valf()) return 0.0f;
    if (v == -__builtin_huge_valf()) return -3.4028234663852886e38f;
    return v;
}
__device__ __forceinline__ bool passes_grad(float v) { return v == v && fabsf(v) != __builtin_huge_valf(); }

__device__ __forceinline__ float surf_depth_of(float D, float a, float med, float omr, float r, float* expected) {
    const float e = nan_to_num00(D / a);
    if (expected) *expected = e;
    return e * omr + r * nan_to_num00(med);
}

__device__ __forceinline__ F3 ray_dir(const MapsCam& c, int x, int y) {
    const float xf = (float)x, yf = (float)y;
    return mk3(fmaf(xf, c.Rd[0], fmaf(yf, c.Rd[3], c.Rd[6])), fmaf(xf, c.Rd[1], fmaf(yf, c.Rd[4], c.Rd[7])),
               fmaf(xf, c.Rd[2], fmaf(yf, c.Rd[5], c.Rd[8])));
}
__device__ __forceinline__ F3 back_project(const MapsCam& c, int x, int y, float depth) {
    const F3 d = ray_dir(c, x, y);
    return mk3(depth * d.x + c.o[0], depth * d.y + c.o[1], depth * d.z + c.o[2]);  // point_utils.py:23
}
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

struct MapsArgs {
    int W, H;
    float r, omr;  // depth_ratio, 1 - depth_ratio
    const float* allmap;
    const float* cam;
    // forward outputs
    float *rend_alpha, *rend_normal, *rend_normal_cam, *rend_depth, *rend_dist, *surf_depth, *surf_normal, *surf_normal_cam;
    // backward inputs / output
    const float* surf_depth_in;
    const float *g_alpha, *g_normal, *g_normal_cam, *g_depth, *g_dist, *g_surf_depth, *g_surf_normal, *g_surf_normal_cam;
    float* g_allmap;
};

constexpr int MAPS_BX = 64, MAPS_BY = 4;

__global__ void __launch_bounds__(MAPS_BX * MAPS_BY) maps_fwd_kernel(MapsArgs a) {
    const int x = (int)(blockIdx.x * MAPS_BX + threadIdx.x % MAPS_BX), y = (int)(blockIdx.y * MAPS_BY + threadIdx.x / MAPS_BX);
    if (x >= a.W || y >= a.H) return;
    const MapsCam c = load_cam(a.cam);
    const size_t N = (size_t)a.W * a.H, p = (size_t)a.W * y + x;
    const float* m = a.allmap;
    const float D = m[p], al = m[p + N], n0 = m[p + 2 * N], n1 = m[p + 3 * N], n2 = m[p + 4 * N], med = m[p + 5 * N];
    a.rend_alpha[p] = al;
    a.rend_dist[p] = m[p + 6 * N];
    // (n @ Rv^T)[i] = sum_j n[j] Rv[i][j]   (__init__.py:123)
    a.rend_normal[p] = n0 * c.Rv[0] + n1 * c.Rv[1] + n2 * c.Rv[2];
    a.rend_normal[p + N] = n0 * c.Rv[3] + n1 * c.Rv[4] + n2 * c.Rv[5];
    a.rend_normal[p + 2 * N] = n0 * c.Rv[6] + n1 * c.Rv[7] + n2 * c.Rv[8];
    a.rend_normal_cam[p] = n0;
    a.rend_normal_cam[p + N] = n1;
    a.rend_normal_cam[p + 2 * N] = n2;
    float e;
    const float sd = surf_depth_of(D, al, med, a.omr, a.r, &e);
    a.rend_depth[p] = e;
    a.surf_depth[p] = sd;
    F3 sn = mk3(0, 0, 0);
    if (x >= 1 && x < a.W - 1 && y >= 1 && y < a.H - 1) {  // point_utils.py:33-36: 1-pixel border stays zero
        auto P = [&](int xx, int yy) {
            const size_t q = (size_t)a.W * yy + xx;
            return back_project(c, xx, yy, surf_depth_of(m[q], m[q + N], m[q + 5 * N], a.omr, a.r, nullptr));
        };
        const F3 up = P(x, y - 1), dn = P(x, y + 1), lf = P(x - 1, y), rt = P(x + 1, y);
        const F3 dx = mk3(dn.x - up.x, dn.y - up.y, dn.z - up.z);  // along rows (the reference calls it dx)
        const F3 dy = mk3(rt.x - lf.x, rt.y - lf.y, rt.z - lf.z);
        const F3 cr = cross3(dx, dy);
        const float nrm = sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
        const float inv = 1.0f / fmaxf(nrm, 1e-12f);  // F.normalize eps
        sn = mk3(cr.x * inv * al, cr.y * inv * al, cr.z * inv * al);  // * rend_alpha.detach()  (:146)
    }
    a.surf_normal[p] = sn.x;
    a.surf_normal[p + N] = sn.y;
    a.surf_normal[p + 2 * N] = sn.z;
    // (sn @ Rv)[i] = sum_j sn[j] Rv[j][i]   (__init__.py:149)
    a.surf_normal_cam[p] = sn.x * c.Rv[0] + sn.y * c.Rv[3] + sn.z * c.Rv[6];
    a.surf_normal_cam[p + N] = sn.x * c.Rv[1] + sn.y * c.Rv[4] + sn.z * c.Rv[7];
    a.surf_normal_cam[p + 2 * N] = sn.x * c.Rv[2] + sn.y * c.Rv[5] + sn.z * c.Rv[8];
}

// Adjoint of the normal of interior pixel (x, y) with respect to its two difference vectors:
// returns dL/d(dx) in *gdx and dL/d(dy) in *gdy.
__device__ __forceinline__ void normal_adjoint(const MapsArgs& a, const MapsCam& c, int x, int y, F3* gdx, F3* gdy) {
    const size_t N = (size_t)a.W * a.H, q = (size_t)a.W * y + x;
    const float* sdm = a.surf_depth_in;
    const F3 up = back_project(c, x, y - 1, sdm[q - a.W]), dn = back_project(c, x, y + 1, sdm[q + a.W]);
    const F3 lf = back_project(c, x - 1, y, sdm[q - 1]), rt = back_project(c, x + 1, y, sdm[q + 1]);
    const F3 dx = mk3(dn.x - up.x, dn.y - up.y, dn.z - up.z), dy = mk3(rt.x - lf.x, rt.y - lf.y, rt.z - lf.z);
    const F3 cr = cross3(dx, dy);
    const float nrm = sqrtf(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
    // total gradient of the world-space surf_normal: direct + through surf_normal_cam = sn @ Rv
    F3 g = mk3(0, 0, 0);
    if (a.g_surf_normal) g = mk3(a.g_surf_normal[q], a.g_surf_normal[q + N], a.g_surf_normal[q + 2 * N]);
    if (a.g_surf_normal_cam) {
        const float h0 = a.g_surf_normal_cam[q], h1 = a.g_surf_normal_cam[q + N], h2 = a.g_surf_normal_cam[q + 2 * N];
        g.x += h0 * c.Rv[0] + h1 * c.Rv[1] + h2 * c.Rv[2];
        g.y += h0 * c.Rv[3] + h1 * c.Rv[4] + h2 * c.Rv[5];
        g.z += h0 * c.Rv[6] + h1 * c.Rv[7] + h2 * c.Rv[8];
    }
    const float al = a.allmap[q + N];  // detached weight
    g = mk3(g.x * al, g.y * al, g.z * al);
    F3 gc;
    if (nrm > 1e-12f) {  // v / |v|: (g - u (u.g)) / |v|
        const float inv = 1.0f / nrm;
        const F3 u = mk3(cr.x * inv, cr.y * inv, cr.z * inv);
        const float ug = u.x * g.x + u.y * g.y + u.z * g.z;
        gc = mk3((g.x - u.x * ug) * inv, (g.y - u.y * ug) * inv, (g.z - u.z * ug) * inv);
    } else {  // clamped denominator: v / eps
        gc = mk3(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f);
    }
    *gdx = cross3(dy, gc);  // c = dx x dy  =>  dL/ddx = dy x gc, dL/ddy = gc x dx
    *gdy = cross3(gc, dx);
}

__global__ void __launch_bounds__(MAPS_BX * MAPS_BY) maps_bwd_kernel(MapsArgs a) {
    const int x = (int)(blockIdx.x * MAPS_BX + threadIdx.x % MAPS_BX), y = (int)(blockIdx.y * MAPS_BY + threadIdx.x / MAPS_BX);
    if (x >= a.W || y >= a.H) return;
    const MapsCam c = load_cam(a.cam);
    const size_t N = (size_t)a.W * a.H, p = (size_t)a.W * y + x;
    const float D = a.allmap[p], al = a.allmap[p + N], med = a.allmap[p + 5 * N];

    // gradient reaching this pixel's back-projected point from the normals of its four neighbours
    F3 gP = mk3(0, 0, 0);
    if (a.g_surf_normal || a.g_surf_normal_cam) {
        auto interior = [&](int xx, int yy) { return xx >= 1 && xx < a.W - 1 && yy >= 1 && yy < a.H - 1; };
        F3 gdx, gdy;
        if (interior(x, y - 1)) { normal_adjoint(a, c, x, y - 1, &gdx, &gdy); gP = mk3(gP.x + gdx.x, gP.y + gdx.y, gP.z + gdx.z); }
        if (interior(x, y + 1)) { normal_adjoint(a, c, x, y + 1, &gdx, &gdy); gP = mk3(gP.x - gdx.x, gP.y - gdx.y, gP.z - gdx.z); }
        if (interior(x - 1, y)) { normal_adjoint(a, c, x - 1, y, &gdx, &gdy); gP = mk3(gP.x + gdy.x, gP.y + gdy.y, gP.z + gdy.z); }
        if (interior(x + 1, y)) { normal_adjoint(a, c, x + 1, y, &gdx, &gdy); gP = mk3(gP.x - gdy.x, gP.y - gdy.y, gP.z - gdy.z); }
    }
    const F3 d = ray_dir(c, x, y);
    float g_sd = gP.x * d.x + gP.y * d.y + gP.z * d.z;
    if (a.g_surf_depth) g_sd += a.g_surf_depth[p];
    float g_e = g_sd * a.omr;
    if (a.g_depth) g_e += a.g_depth[p];
    // nan_to_num backward passes no gradient through replaced values; the division backward then is torch's
    // grad / other and (-grad * self) / (other * other) -- which is NaN at alpha == 0 (0/0), exactly like the
    // reference.  Such pixels have no contributor, so the rasterizer's backward never reads the value.
    const float e = D / al;
    const float g_em = passes_grad(e) ? g_e : 0.0f;
    const float gD = g_em / al;
    float gA = (-g_em * D) / (al * al);
    if (a.g_alpha) gA += a.g_alpha[p];
    const float gM = passes_grad(med) ? g_sd * a.r : 0.0f;
    float gn0 = 0, gn1 = 0, gn2 = 0;
    if (a.g_normal_cam) { gn0 = a.g_normal_cam[p]; gn1 = a.g_normal_cam[p + N]; gn2 = a.g_normal_cam[p + 2 * N]; }
    if (a.g_normal) {
        const float h0 = a.g_normal[p], h1 = a.g_normal[p + N], h2 = a.g_normal[p + 2 * N];
        gn0 += h0 * c.Rv[0] + h1 * c.Rv[3] + h2 * c.Rv[6];
        gn1 += h0 * c.Rv[1] + h1 * c.Rv[4] + h2 * c.Rv[7];
        gn2 += h0 * c.Rv[2] + h1 * c.Rv[5] + h2 * c.Rv[8];
    }
    float* o = a.g_allmap;
    o[p] = gD;
    o[p + N] = gA;
    o[p + 2 * N] = gn0;
    o[p + 3 * N] = gn1;
    o[p + 4 * N] = gn2;
    o[p + 5 * N] = gM;
    o[p + 6 * N] = a.g_dist ? a.g_dist[p] : 0.0f;
}

}  // namespace g4s

using namespace g4s;

extern "C" void g4s_maps_launch_internal(int fwd, int W, int H, float depth_ratio, const float* allmap, const float* wvt,
                                         const float* fpt, float* cam, float* const* outs, const float* surf_depth_in,
                                         const float* const* grads, float* g_allmap, hipStream_t s) {
    hipLaunchKernelGGL(maps_camera_kernel, dim3(1), dim3(64), 0, s, wvt, fpt, W, H, cam);
    MapsArgs a{};
    a.W = W; a.H = H; a.r = depth_ratio; a.omr = (float)(1.0 - (double)depth_ratio);
    a.allmap = allmap; a.cam = cam;
    const dim3 grid((W + MAPS_BX - 1) / MAPS_BX, (H + MAPS_BY - 1) / MAPS_BY), block(MAPS_BX * MAPS_BY);
    if (fwd) {
        a.rend_alpha = outs[0]; a.rend_normal = outs[1]; a.rend_normal_cam = outs[2]; a.rend_depth = outs[3];
        a.rend_dist = outs[4]; a.surf_depth = outs[5]; a.surf_normal = outs[6]; a.surf_normal_cam = outs[7];
        hipLaunchKernelGGL(maps_fwd_kernel, grid, block, 0, s, a);
    } else {
        a.surf_depth_in = surf_depth_in;
        a.g_alpha = grads[0]; a.g_normal = grads[1]; a.g_normal_cam = grads[2]; a.g_depth = grads[3]; a.g_dist = grads[4];
        a.g_surf_depth = grads[5]; a.g_surf_normal = grads[6]; a.g_surf_normal_cam = grads[7];
        a.g_allmap = g_allmap;
        hipLaunchKernelGGL(maps_bwd_kernel, grid, block, 0, s, a);
    }
}

// ---- packed rows for the visible-rows gradient exchange (g4splat_amd/parallel.py) -----------------
// mode bit 0: direction (0 = pack: rows -> buffer, 1 = unpack: buffer -> rows)
// mode bit 1: buffer layout (0 = segment after segment, packed[seg_off(s) * n + j * w_s + c]; 1 = row-major [n, sum w],
//             packed[j * sum_w + seg_off(s) + c] -- what an all_to_all with per-destination row ranges needs)
// mode bit 2: unpack ADDS to the rows instead of overwriting them (the owner's accumulation of one source's rows; a
//             source holds a row at most once, so there are no duplicate indices inside one launch)
// mode bit 3: (row-major only) buffer rows are sum w + 1 floats: the last one is the row's index as int32 bits --
//             pack writes it, unpack reads it instead of idx[] (idx may be NULL then)
// A row's floats over all segments (58 + 2 for the gradient bucket) are spread over the lanes of a wave --
// lane -> (segment, column) is fixed for the whole kernel, so there is no per-element division -- and each wave
// walks rows j, j + #waves, ...: both sides move contiguous w_s-float runs.  Rows wider than 64 floats take
// several lane passes.
namespace g4s {
struct RowSegs {
    float* ptr[8];
    int width[8];
    int nseg;
};
__global__ void __launch_bounds__(256) pack_rows_kernel(RowSegs segs, const long long* __restrict__ idx, int n,
                                                        float* __restrict__ packed, int mode) {
    const int lane = (int)(threadIdx.x & 63);
    const int wave = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), nwaves = (int)(gridDim.x * 4);
    const bool unpack = (mode & 1) != 0, row_major = (mode & 2) != 0, add = (mode & 4) != 0, carry = (mode & 8) != 0;
    int row_floats = 0;
    for (int s = 0; s < segs.nseg; s++) row_floats += segs.width[s];
    // carry (row-major only): every buffer row ends with its row index as an int32 column -- written by pack, and
    // read by unpack INSTEAD of idx[] (the rows and their indices then travel in one all_to_all)
    const int buf_floats = row_floats + (carry ? 1 : 0);
    for (int f0 = 0; f0 < buf_floats; f0 += 64) {
        // this lane's (segment, column) for float f0 + lane of a row
        const int f = f0 + lane;
        int seg = -1, col = 0, w = 1;
        size_t seg_off = 0;  // floats of a row before this segment
        {
            int base = 0;
            size_t off = 0;
            for (int s = 0; s < segs.nseg; s++) {
                if (f >= base && f < base + segs.width[s]) { seg = s; col = f - base; w = segs.width[s]; seg_off = off; }
                base += segs.width[s];
                off += (size_t)segs.width[s];
            }
        }
        const bool index_lane = carry && f == row_floats;
        if (seg < 0 && !index_lane) continue;
        if (index_lane) {
            if (!unpack)
                for (int j = wave; j < n; j += nwaves) packed[(size_t)j * buf_floats + row_floats] = __int_as_float((int)idx[j]);
            continue;
        }
        float* sp = segs.ptr[seg];
        float* pp = row_major ? packed + f : packed + seg_off * (size_t)n + col;
        const size_t pstride = row_major ? (size_t)buf_floats : (size_t)w;
        for (int j = wave; j < n; j += nwaves) {
            const long long row = (carry && unpack) ? (long long)__float_as_int(packed[(size_t)j * buf_floats + row_floats]) : idx[j];
            float* src = sp + (size_t)row * w + col;
            float* dst = pp + (size_t)j * pstride;
            if (!unpack) *dst = *src;
            else if (add) *src += *dst;
            else *src = *dst;
        }
    }
}

// The owner's accumulation of ALL sources in one launch (g4s_accumulate_rows, include/g4s_rasterizer.h).  The per-source
// form -- one pack_rows_kernel(mode 15) launch per source -- reads and writes a destination row once per source that
// holds it and pays a launch per source; here a workgroup owns ACC_CHUNK consecutive destination rows: it loads them into
// LDS, finds each source's rows of the chunk (the sources' rows ascend by index: a 32-ary search by 32 lanes per source,
// all sources at once), adds them source after source -- the same order of additions per element as the per-source
// launches, so the same bits -- and writes the chunk back: 7 x (launch + read + write) become 1 x.
constexpr int ACC_CHUNK = 64;  // (measured at the metric size, eight ranks: 32 rows 0.088 ms, 48 0.075, 64 0.076, 128 0.096)
struct AccSources {
    int off[8], cnt[8];  // rows [off, off + cnt) of the buffer came from source i (ascending row indices)
    int nsrc;
    // the owner's own contribution (already in the segments' rows) takes position `own_pos` in the order of additions:
    // 0 = first (own + s0 + s1 + ...), k = behind the first k sources (((0 + s0 + ... + s_{k-1}) + own) + s_k + ...): with the
    // sources in rank order and own_pos = the owner's rank, every row is summed in RANK ORDER whoever owns it
    // (g4s_accumulate_rows_ordered)
    int own_pos;
};
__global__ void __launch_bounds__(256) accumulate_rows_kernel(RowSegs segs, AccSources src, const float* __restrict__ packed,
                                                              int row_lo, int row_hi) {
    extern __shared__ float s_tile[];  // [ACC_CHUNK][row_floats]
    __shared__ int s_b0[8], s_b1[8];
    const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
    int row_floats = 0;
    for (int s = 0; s < segs.nseg; s++) row_floats += segs.width[s];
    const int buf_floats = row_floats + 1;
    const int chunk_lo = row_lo + (int)blockIdx.x * ACC_CHUNK;
    const int chunk_hi = min(chunk_lo + ACC_CHUNK, row_hi);
    const int rows = chunk_hi - chunk_lo;
    // The chunk's rows into LDS.  This lane's (segment, column) for float f0 + lane of a row is fixed (as in
    // pack_rows_kernel); a wave takes rows wave, wave + 4, ...: every load of the loop is in flight at once.  (Other
    // resident workgroups -- up to ten per CU -- cover the latency of this copy and of the search below.)
    for (int f0 = 0; f0 < row_floats; f0 += 64) {
        const int f = f0 + lane;
        int seg = -1, col = 0, w = 1;
        {
            int base = 0;
            for (int s = 0; s < segs.nseg; s++) {
                if (f >= base && f < base + segs.width[s]) { seg = s; col = f - base; w = segs.width[s]; }
                base += segs.width[s];
            }
        }
        if (seg >= 0) {
            const float* sp = segs.ptr[seg] + (size_t)chunk_lo * w + col;
            if (src.own_pos == 0) {
#pragma unroll 4
                for (int r = wave; r < rows; r += 4) s_tile[r * row_floats + f] = sp[(size_t)r * w];
            } else {  // the owner's rows join the sum behind the first own_pos sources (below): start from zero
#pragma unroll 4
                for (int r = wave; r < rows; r += 4) s_tile[r * row_floats + f] = 0.0f;
            }
        }
    }
    // each source's rows of this chunk: group g = 32 lanes searches source g for both ends at once
    {
        const int g = t >> 5, l32 = t & 31;
        const bool live = g < src.nsrc;
        const int off = live ? src.off[g] : 0, cnt = live ? src.cnt[g] : 0;
        int lo[2] = {0, 0}, hi[2] = {cnt, cnt};
        const int target[2] = {chunk_lo, chunk_hi};
        for (int it = 0; it < 7; it++) {  // 32^7 > 2^31 rows (uniform trip count: the ballots need every lane)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int span = hi[e] - lo[e];
                const int step = (span + 31) >> 5;
                const int pos = lo[e] + l32 * step;
                const bool valid = span > 0 && pos < hi[e];
                int v = 0x7fffffff;
                if (valid) v = __float_as_int(packed[(size_t)(off + pos) * buf_floats + row_floats]);
                const uint64_t b = __ballot(valid && v < target[e]);
                const uint64_t bv = __ballot(valid);
                const int c = __popc((uint32_t)(b >> (32 * (lane >> 5))));    // probes below the target: a prefix
                const int nv = __popc((uint32_t)(bv >> (32 * (lane >> 5))));  // valid probes
                if (span > 0) {
                    const int nlo = c == 0 ? lo[e] : lo[e] + (c - 1) * step + 1;
                    const int nhi = c == 0 ? lo[e] : (c < nv ? lo[e] + c * step : hi[e]);
                    lo[e] = nlo; hi[e] = nhi;
                }
            }
            if (__syncthreads_or((hi[0] - lo[0]) | (hi[1] - lo[1])) == 0) break;  // every group has both ends
        }
        if (live && l32 == 0) { s_b0[g] = lo[0]; s_b1[g] = lo[1]; }
    }
    __syncthreads();
    // Sources in order; a wave takes rows b0 + wave, + 4, ... of the source's sub-range, four of them in flight (a source
    // holds a row once: the waves never meet on a tile row).
    for (int s = 0; s <= src.nsrc; s++) {
        if (s == src.own_pos && s != 0) {  // (uniform) the owner's own rows, in their place in the order
            for (int f0 = 0; f0 < row_floats; f0 += 64) {
                const int f = f0 + lane;
                int seg = -1, col = 0, w = 1;
                {
                    int base = 0;
                    for (int g = 0; g < segs.nseg; g++) {
                        if (f >= base && f < base + segs.width[g]) { seg = g; col = f - base; w = segs.width[g]; }
                        base += segs.width[g];
                    }
                }
                if (seg >= 0) {
                    const float* sp = segs.ptr[seg] + (size_t)chunk_lo * w + col;
#pragma unroll 4
                    for (int r = wave; r < rows; r += 4) s_tile[r * row_floats + f] += sp[(size_t)r * w];
                }
            }
            __syncthreads();
        }
        if (s == src.nsrc) break;
        const int b0 = s_b0[s], b1 = s_b1[s];
        const float* base = packed + (size_t)src.off[s] * buf_floats;
        for (int j0 = b0 + wave; j0 < b1; j0 += 16) {
            int r[4];
            float v[4][4];  // up to 4 x 64 floats per row (launch check: rows of at most 240 floats)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = j0 + 4 * u;
                const bool ok = j < b1;
                const float* rowp = base + (size_t)(ok ? j : b0) * buf_floats;
                r[u] = ok ? __float_as_int(rowp[row_floats]) - chunk_lo : -1;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int f = lane + 64 * q;
                    v[u][q] = f < row_floats ? rowp[f] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                // (unsigned: a received row whose index column lies outside this chunk -- a source that is not sorted, or
                // that holds a row of another shard -- is dropped instead of being added outside the LDS tile)
                if ((unsigned)r[u] < (unsigned)rows) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int f = lane + 64 * q;
                        if (f < row_floats) s_tile[r[u] * row_floats + f] += v[u][q];
                    }
                }
            }
        }
        __syncthreads();  // (the next source may hold the same rows)
    }
    for (int f0 = 0; f0 < row_floats; f0 += 64) {
        const int f = f0 + lane;
        int seg = -1, col = 0, w = 1;
        {
            int base = 0;
            for (int s = 0; s < segs.nseg; s++) {
                if (f >= base && f < base + segs.width[s]) { seg = s; col = f - base; w = segs.width[s]; }
                base += segs.width[s];
            }
        }
        if (seg >= 0) {
            float* sp = segs.ptr[seg] + (size_t)chunk_lo * w + col;
#pragma unroll 4
            for (int r = wave; r < rows; r += 4) sp[(size_t)r * w] = s_tile[r * row_floats + f];
        }
    }
}
}  // namespace g4s

extern "C" int g4s_accumulate_rows_launch_internal(int nseg, float* const* ptrs, const int* widths, int nsrc, const int* src_off,
                                                   const int* src_cnt, const float* packed, int row_lo, int row_hi,
                                                   hipStream_t s, int own_pos) {
    g4s::RowSegs segs{};
    segs.nseg = nseg;
    int row_floats = 0;
    for (int i = 0; i < nseg; i++) { segs.ptr[i] = ptrs[i]; segs.width[i] = widths[i]; row_floats += widths[i]; }
    if (row_hi <= row_lo) return 0;
    const size_t lds = (size_t)g4s::ACC_CHUNK * row_floats * sizeof(float);
    if (lds > 60 * 1024) return -1;
    const int blocks = (row_hi - row_lo + g4s::ACC_CHUNK - 1) / g4s::ACC_CHUNK;
    if (own_pos != 0 && nsrc > 8) return -2;  // (the ordered form needs all sources in one launch)
    for (int s0 = 0; s0 < nsrc; s0 += 8) {  // (more than eight sources: eight per launch, in order)
        g4s::AccSources src{};
        src.nsrc = nsrc - s0 < 8 ? nsrc - s0 : 8;
        src.own_pos = own_pos;
        for (int i = 0; i < src.nsrc; i++) { src.off[i] = src_off[s0 + i]; src.cnt[i] = src_cnt[s0 + i]; }
        hipLaunchKernelGGL(g4s::accumulate_rows_kernel, dim3(blocks), dim3(256), lds, s, segs, src, packed, row_lo, row_hi);
    }
    return 0;
}

extern "C" void g4s_pack_rows_launch_internal(int nseg, float* const* ptrs, const int* widths, const long long* idx, int n,
                                              float* packed, int mode, hipStream_t s) {
    RowSegs segs{};
    segs.nseg = nseg;
    for (int i = 0; i < nseg; i++) { segs.ptr[i] = ptrs[i]; segs.width[i] = widths[i]; }
    if (n <= 0) return;
    const int blocks = (n + 31) / 32 < 8192 ? (n + 31) / 32 : 8192;  // >= 8 rows per wave once n is large
    hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, s, segs, idx, n, packed, mode);
}
