// Fused L1 + SSIM photometric loss with its gradient (include/g4s_losses.h).
//
// Reference semantics: 2dgs/utils/loss_utils.py:17-18 (l1_loss), :31-33 (gaussian), :46-79 (ssim / _ssim:
// five depthwise 11x11 convolutions with zero padding), combined as train_with_refine_depth.py:382-383.
//
// MI355X design: 16x16-pixel tiles per (channel) block; the 26x26 halo tile of both images is staged in LDS
// once and the 11x11 Gaussian window is applied separably (11 + 11 taps instead of 121) to the five moments
// at the same time.  Kernel 1 turns the moments into the SSIM value and its three partial-derivative maps
// (d/dmu1 total, d/dE[x^2], d/dE[xy]) and reduces the SSIM / L1 sums per block; kernel 2 convolves the three
// maps (the adjoint of a zero-padded symmetric convolution is the same convolution) and adds the L1 sign
// term; a one-block kernel folds the block partials in double, in a fixed order.  HBM traffic ~ 130 B/pixel
// per channel-pixel against ~25 full-image passes of the eager formulation.
#include "g4s_internal.h"
#include "g4s_device.h"

namespace g4s {

constexpr int SS_T = 16, SS_R = 5, SS_IN = SS_T + 2 * SS_R;  // 26
constexpr int SS_LD = SS_IN + 1;                             // LDS row stride (odd: conflict-free columns)
// gaussian(11, 1.5) / sum, as float32 exactly like loss_utils.py:31-33 builds it
__device__ constexpr float kGauss[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                                         0.10936068743467331f,  0.21300552785396576f,   0.26601171493530273f,
                                         0.21300552785396576f,  0.10936068743467331f,   0.036000773310661316f,
                                         0.0075987582094967365f, 0.001028380123898387f};
constexpr float SS_C1 = 0.01f * 0.01f, SS_C2 = 0.03f * 0.03f;

struct PhotoArgs {
    int W, H;
    float lambda;
    const float* image;
    const float* gt;
    float* maps;      // [3 maps][3 channels][H][W]
    float* partials;  // [blocks][2]: ssim sum, l1 sum
    float* out3;
    float* dL_dimage;
    int nblocks;
};

// sum over a 256-thread block in a fixed order; result valid in thread 0
__device__ __forceinline__ float block_sum256(float v, float* s4) {
    v = wave_sum_to_lane63(v);
    if ((threadIdx.x & 63) == 63) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

__global__ void __launch_bounds__(256) ssim_fwd_kernel(PhotoArgs a) {
    __shared__ float s_x[SS_IN][SS_LD], s_y[SS_IN][SS_LD];
    __shared__ float s_h[5][SS_IN][SS_T + 1];
    __shared__ float s_red[2][4];
    const int ch = (int)blockIdx.z, tx0 = (int)blockIdx.x * SS_T, ty0 = (int)blockIdx.y * SS_T;
    const size_t N = (size_t)a.W * a.H;
    const float* img = a.image + ch * N;
    const float* gt = a.gt + ch * N;
    for (int i = (int)threadIdx.x; i < SS_IN * SS_IN; i += 256) {
        const int r = i / SS_IN, c = i % SS_IN, gx = tx0 - SS_R + c, gy = ty0 - SS_R + r;
        const bool in = gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;  // zero padding (padding=window_size//2)
        s_x[r][c] = in ? img[(size_t)gy * a.W + gx] : 0.0f;
        s_y[r][c] = in ? gt[(size_t)gy * a.W + gx] : 0.0f;
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < SS_IN * SS_T; i += 256) {  // rows of the halo tile, filtered along x
        const int r = i / SS_T, c = i % SS_T;
        float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = kGauss[k], x = s_x[r][c + k], y = s_y[r][c + k];
            m1 = fmaf(g, x, m1); m2 = fmaf(g, y, m2);
            e11 = fmaf(g, x * x, e11); e22 = fmaf(g, y * y, e22); e12 = fmaf(g, x * y, e12);
        }
        s_h[0][r][c] = m1; s_h[1][r][c] = m2; s_h[2][r][c] = e11; s_h[3][r][c] = e22; s_h[4][r][c] = e12;
    }
    __syncthreads();
    const int tx = (int)threadIdx.x % SS_T, ty = (int)threadIdx.x / SS_T, px = tx0 + tx, py = ty0 + ty;
    float m1 = 0, m2 = 0, e11 = 0, e22 = 0, e12 = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float g = kGauss[k];
        m1 = fmaf(g, s_h[0][ty + k][tx], m1); m2 = fmaf(g, s_h[1][ty + k][tx], m2);
        e11 = fmaf(g, s_h[2][ty + k][tx], e11); e22 = fmaf(g, s_h[3][ty + k][tx], e22);
        e12 = fmaf(g, s_h[4][ty + k][tx], e12);
    }
    float ssim = 0, l1 = 0;
    if (px < a.W && py < a.H) {
        const float mu1sq = m1 * m1, mu2sq = m2 * m2, mu12 = m1 * m2;
        const float s1 = e11 - mu1sq, s2 = e22 - mu2sq, s12 = e12 - mu12;
        const float A1 = 2 * mu12 + SS_C1, A2 = 2 * s12 + SS_C2, B1 = mu1sq + mu2sq + SS_C1, B2 = s1 + s2 + SS_C2;
        const float rB1 = 1.0f / B1, rB2 = 1.0f / B2;
        ssim = (A1 * A2) * (rB1 * rB2);
        // partial derivatives of the map value with respect to E[x^2], E[xy] and (in total) mu1
        const float dS_de11 = -ssim * rB2;
        const float dS_de12 = 2 * A1 * (rB1 * rB2);
        const float dS_dm1 = 2 * m2 * A2 * (rB1 * rB2) - 2 * m1 * ssim * rB1 - 2 * m1 * dS_de11 - m2 * dS_de12;
        const size_t p = (size_t)py * a.W + px;
        a.maps[(0 * 3 + ch) * N + p] = dS_dm1;
        a.maps[(1 * 3 + ch) * N + p] = dS_de11;
        a.maps[(2 * 3 + ch) * N + p] = dS_de12;
        l1 = fabsf(s_x[ty + SS_R][tx + SS_R] - s_y[ty + SS_R][tx + SS_R]);
    }
    const float bs = block_sum256(ssim, s_red[0]);
    const float bl = block_sum256(l1, s_red[1]);
    if (threadIdx.x == 0) {
        const int b = (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
        a.partials[2 * b] = bs;
        a.partials[2 * b + 1] = bl;
    }
}

// one block of 1024 threads; four independent loads in flight per thread (a serial loop of dependent round trips over
// the 22 500 block partials of a 1600x1200 frame took 24 us); fixed association => bit-reproducible
__global__ void __launch_bounds__(1024) photo_reduce_kernel(PhotoArgs a) {
    __shared__ double s_s[1024], s_l[1024];
    double s = 0, l = 0;
    const float2* part = reinterpret_cast<const float2*>(a.partials);
    for (int i0 = (int)threadIdx.x; i0 < a.nblocks; i0 += 4096) {
        float2 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + 1024 * k;
            v[k] = part[i < a.nblocks ? i : 0];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i0 + 1024 * k < a.nblocks) { s += v[k].x; l += v[k].y; }
    }
    s_s[threadIdx.x] = s; s_l[threadIdx.x] = l;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s_s[threadIdx.x] += s_s[threadIdx.x + o]; s_l[threadIdx.x] += s_l[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = 3.0 * (double)a.W * (double)a.H;
        const double ssim = s_s[0] / n, l1 = s_l[0] / n;
        a.out3[0] = (float)((1.0 - (double)a.lambda) * l1 + (double)a.lambda * (1.0 - ssim));
        a.out3[1] = (float)l1;
        a.out3[2] = (float)ssim;
    }
}

__global__ void __launch_bounds__(256) ssim_bwd_kernel(PhotoArgs a) {
    __shared__ float s_m[3][SS_IN][SS_LD];
    __shared__ float s_h[3][SS_IN][SS_T + 1];
    const int ch = (int)blockIdx.z, tx0 = (int)blockIdx.x * SS_T, ty0 = (int)blockIdx.y * SS_T;
    const size_t N = (size_t)a.W * a.H;
    for (int i = (int)threadIdx.x; i < SS_IN * SS_IN; i += 256) {
        const int r = i / SS_IN, c = i % SS_IN, gx = tx0 - SS_R + c, gy = ty0 - SS_R + r;
        const bool in = gx >= 0 && gx < a.W && gy >= 0 && gy < a.H;
        const size_t p = (size_t)gy * a.W + gx;
#pragma unroll
        for (int m = 0; m < 3; m++) s_m[m][r][c] = in ? a.maps[(m * 3 + ch) * N + p] : 0.0f;
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < SS_IN * SS_T; i += 256) {
        const int r = i / SS_T, c = i % SS_T;
        float v0 = 0, v1 = 0, v2 = 0;
#pragma unroll
        for (int k = 0; k < 11; k++) {
            const float g = kGauss[k];
            v0 = fmaf(g, s_m[0][r][c + k], v0); v1 = fmaf(g, s_m[1][r][c + k], v1); v2 = fmaf(g, s_m[2][r][c + k], v2);
        }
        s_h[0][r][c] = v0; s_h[1][r][c] = v1; s_h[2][r][c] = v2;
    }
    __syncthreads();
    const int tx = (int)threadIdx.x % SS_T, ty = (int)threadIdx.x / SS_T, px = tx0 + tx, py = ty0 + ty;
    if (px >= a.W || py >= a.H) return;
    float cm = 0, c11 = 0, c12 = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) {
        const float g = kGauss[k];
        cm = fmaf(g, s_h[0][ty + k][tx], cm); c11 = fmaf(g, s_h[1][ty + k][tx], c11); c12 = fmaf(g, s_h[2][ty + k][tx], c12);
    }
    const size_t p = (size_t)py * a.W + px;
    const float x = a.image[ch * N + p], y = a.gt[ch * N + p];
    const float inv_n = 1.0f / (3.0f * (float)a.W * (float)a.H);
    const float d = x - y;
    const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);  // torch.abs backward: sign(0) = 0
    const float dssim = cm + 2.0f * x * c11 + y * c12;               // d(sum of the SSIM map)/dx
    a.dL_dimage[ch * N + p] = ((1.0f - a.lambda) * sgn - a.lambda * dssim) * inv_n;
}

}  // namespace g4s

using namespace g4s;

extern "C" size_t g4s_photometric_workspace(int width, int height) {
    if (width <= 0 || height <= 0) return 0;
    const size_t N = (size_t)width * height;
    const size_t blocks = (size_t)((width + SS_T - 1) / SS_T) * ((height + SS_T - 1) / SS_T) * 3;
    return align_up(9 * N * 4) + align_up(blocks * 8) + 256;
}

extern "C" void g4s_photometric_launch_internal(int W, int H, const float* image, const float* gt, float lambda, float* out3,
                                                float* dL_dimage, char* workspace, hipStream_t s) {
    PhotoArgs a{};
    a.W = W; a.H = H; a.lambda = lambda; a.image = image; a.gt = gt; a.out3 = out3; a.dL_dimage = dL_dimage;
    const size_t N = (size_t)W * H;
    char* w = align_ptr(workspace);
    a.maps = (float*)w;
    a.partials = (float*)(w + align_up(9 * N * 4));
    const dim3 grid((W + SS_T - 1) / SS_T, (H + SS_T - 1) / SS_T, 3);
    a.nblocks = (int)(grid.x * grid.y * grid.z);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL(photo_reduce_kernel, dim3(1), dim3(1024), 0, s, a);
    if (dL_dimage) hipLaunchKernelGGL(ssim_bwd_kernel, grid, dim3(256), 0, s, a);
}

// ---- geometry regularisers of the training step (include/g4s_losses.h) ------------------------------
// normal_error.mean() = mean(1 - sum_c rend_normal_c surf_normal_c) and rend_dist.mean()
// (train_with_refine_depth.py:391-396): one pass over the seven planes + a fixed-order reduction forward, one pass
// backward, instead of ~8 element-wise / reduction launches each way.
namespace g4s {
struct GeoRegArgs {
    long long N;  // pixels
    int vec;      // N % 4 == 0 and every base 16-byte aligned
    const float *rn, *sn, *dist;
    float* partials;  // [blocks][2]
    int nblocks;
    float* out2;
    const float* g2;
    float *d_rn, *d_sn, *d_dist;
};

__global__ void __launch_bounds__(256) georeg_fwd_kernel(GeoRegArgs a) {
    __shared__ float s_red[2][4];
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    float e = 0.0f, d = 0.0f;
    if (p0 < a.N) {
        float r[3][4], n[3][4], t[4];
        const int cnt = (int)((a.N - p0) < 4 ? (a.N - p0) : 4);
        if (a.vec) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                *reinterpret_cast<float4*>(r[c]) = *reinterpret_cast<const float4*>(a.rn + c * a.N + p0);
                *reinterpret_cast<float4*>(n[c]) = *reinterpret_cast<const float4*>(a.sn + c * a.N + p0);
            }
            *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(a.dist + p0);
        } else {
            for (int i = 0; i < 4; i++) {
                const long long p = p0 + (i < cnt ? i : 0);
                for (int c = 0; c < 3; c++) { r[c][i] = a.rn[c * a.N + p]; n[c][i] = a.sn[c * a.N + p]; }
                t[i] = a.dist[p];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i < cnt) {
                e += 1.0f - ((r[0][i] * n[0][i] + r[1][i] * n[1][i]) + r[2][i] * n[2][i]);
                d += t[i];
            }
        }
    }
    const float be = block_sum256(e, s_red[0]);
    const float bd = block_sum256(d, s_red[1]);
    if (threadIdx.x == 0) { a.partials[2 * blockIdx.x] = be; a.partials[2 * blockIdx.x + 1] = bd; }
}

__global__ void __launch_bounds__(1024) georeg_reduce_kernel(GeoRegArgs a) {
    __shared__ double s_e[1024], s_d[1024];
    double e = 0, d = 0;
    const float2* part = reinterpret_cast<const float2*>(a.partials);
    for (int i0 = (int)threadIdx.x; i0 < a.nblocks; i0 += 4096) {
        float2 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + 1024 * k;
            v[k] = part[i < a.nblocks ? i : 0];
        }
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (i0 + 1024 * k < a.nblocks) { e += v[k].x; d += v[k].y; }
    }
    s_e[threadIdx.x] = e; s_d[threadIdx.x] = d;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s_e[threadIdx.x] += s_e[threadIdx.x + o]; s_d[threadIdx.x] += s_d[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.out2[0] = (float)(s_e[0] / (double)a.N);
        a.out2[1] = (float)(s_d[0] / (double)a.N);
    }
}

__global__ void __launch_bounds__(256) georeg_bwd_kernel(GeoRegArgs a) {
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= a.N) return;
    const float inv = 1.0f / (float)a.N;
    const float gn = -(a.g2[0] * inv), gd = a.g2[1] * inv;  // d mean / d element = 1/N, as autograd's mean backward
    const int cnt = (int)((a.N - p0) < 4 ? (a.N - p0) : 4);
    if (a.vec) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float4 r = *reinterpret_cast<const float4*>(a.rn + c * a.N + p0);
            const float4 n = *reinterpret_cast<const float4*>(a.sn + c * a.N + p0);
            *reinterpret_cast<float4*>(a.d_rn + c * a.N + p0) = make_float4(gn * n.x, gn * n.y, gn * n.z, gn * n.w);
            *reinterpret_cast<float4*>(a.d_sn + c * a.N + p0) = make_float4(gn * r.x, gn * r.y, gn * r.z, gn * r.w);
        }
        *reinterpret_cast<float4*>(a.d_dist + p0) = make_float4(gd, gd, gd, gd);
    } else {
        for (int i = 0; i < cnt; i++) {
            for (int c = 0; c < 3; c++) {
                a.d_rn[c * a.N + p0 + i] = gn * a.sn[c * a.N + p0 + i];
                a.d_sn[c * a.N + p0 + i] = gn * a.rn[c * a.N + p0 + i];
            }
            a.d_dist[p0 + i] = gd;
        }
    }
}
}  // namespace g4s

extern "C" size_t g4s_geometry_regularizers_workspace(int width, int height) {
    const size_t blocks = ((size_t)width * height + 1023) / 1024;
    return g4s::align_up(blocks * 8) + 256;
}

static inline bool georeg_al16(const void* p) { return ((size_t)p & 15) == 0; }

extern "C" void g4s_georeg_launch_internal(int fwd, int W, int H, const float* rn, const float* sn, const float* dist, float* out2,
                                           const float* g2, float* d_rn, float* d_sn, float* d_dist, char* workspace,
                                           hipStream_t s) {
    using namespace g4s;
    GeoRegArgs a{};
    a.N = (long long)W * H;
    a.rn = rn; a.sn = sn; a.dist = dist; a.out2 = out2; a.g2 = g2; a.d_rn = d_rn; a.d_sn = d_sn; a.d_dist = d_dist;
    a.nblocks = (int)((a.N + 1023) / 1024);
    a.vec = (a.N % 4 == 0) && georeg_al16(rn) && georeg_al16(sn) && (fwd ? georeg_al16(dist) : (georeg_al16(d_rn) && georeg_al16(d_sn) && georeg_al16(d_dist)));
    if (fwd) {
        a.partials = (float*)align_ptr(workspace);
        hipLaunchKernelGGL(georeg_fwd_kernel, dim3(a.nblocks), dim3(256), 0, s, a);
        hipLaunchKernelGGL(georeg_reduce_kernel, dim3(1), dim3(1024), 0, s, a);
    } else {
        hipLaunchKernelGGL(georeg_bwd_kernel, dim3(a.nblocks), dim3(256), 0, s, a);
    }
}

// ---- fused Adam over up to eight parameter segments (include/g4s_optim.h) ------------------------------
namespace g4s {
struct AdamSegs {
    float* p[8];
    const float* g[8];
    float* m[8];
    float* v[8];
    long long n[8];        // elements
    long long first[9];    // first float4-block of each segment in the launch's block space (prefix sums)
    float step_size[8];    // lr / (1 - beta1^t)
    float inv_sqrt_bc2[8]; // 1 / sqrt(1 - beta2^t)
    int nseg;
    float w1, w2, beta2, eps;  // 1 - beta1, 1 - beta2 (formed in double on the host), beta2, eps
    const float* coef;     // g4s_adam_step_device: step_size[s] = coef[s], inv_sqrt_bc2[s] = coef[8 + s] (device memory,
                           // written by adam_prep_kernel in front of this launch); NULL: the two arrays above
};

// g4s_adam_step_device: the step counts and learning rates live on the device, so that a captured launch (hipGraph) does
// the right update at every replay.  One thread per segment: t <- t + 1, then the two bias-correction factors in double,
// exactly as the host does for g4s_adam_step.
struct AdamPrep {
    float* step[8];   // per segment: torch's capturable state["step"] (a float32 scalar on the device), incremented here
    const double* lr; // [nseg] on the device (double, like the Python floats the host path divides)
    float* coef;      // [16] scratch on the device
    int nseg;
    double beta1, beta2;
};
__global__ void adam_prep_kernel(AdamPrep a) {
    const int i = (int)threadIdx.x;
    if (i >= a.nseg) return;
    const float t = *a.step[i] + 1.0f;
    *a.step[i] = t;
    const double bc1 = 1.0 - pow(a.beta1, (double)t), bc2 = 1.0 - pow(a.beta2, (double)t);
    a.coef[i] = (float)(a.lr[i] / bc1);
    a.coef[8 + i] = (float)(1.0 / sqrt(bc2));
}

// One thread per 4 consecutive floats (16-byte accesses when the segment base is 16-byte aligned, which torch
// allocations are; the tail and misaligned bases fall back to scalar accesses).
__global__ void __launch_bounds__(256) adam_kernel(AdamSegs a) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;  // float4-block index over all segments
    int s = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) s += (i < a.nseg && q >= a.first[i]) ? 1 : 0;
    const long long e0 = (q - a.first[s]) * 4;
    if (q >= a.first[a.nseg] || e0 >= a.n[s]) return;
    float* p = a.p[s] + e0;
    const float* g = a.g[s] + e0;
    float* m = a.m[s] + e0;
    float* v = a.v[s] + e0;
    const float w1 = a.w1, w2 = a.w2;
    const float ss = a.coef ? a.coef[s] : a.step_size[s], ib = a.coef ? a.coef[8 + s] : a.inv_sqrt_bc2[s];
    const bool vec = e0 + 4 <= a.n[s] && (((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0;
    float pv[4], gv[4], mv[4], vv[4];
    const int cnt = vec ? 4 : (int)((a.n[s] - e0) < 4 ? (a.n[s] - e0) : 4);
    if (vec) {
        *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(p);
        *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(g);
        *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(m);
        *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(v);
    } else {
        for (int i = 0; i < cnt; i++) { pv[i] = p[i]; gv[i] = g[i]; mv[i] = m[i]; vv[i] = v[i]; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i >= cnt) break;
        mv[i] = mv[i] + w1 * (gv[i] - mv[i]);              // exp_avg.lerp_(grad, 1 - beta1)
        vv[i] = a.beta2 * vv[i] + w2 * gv[i] * gv[i];      // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vv[i]) * ib + a.eps;     // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
        pv[i] = pv[i] - ss * (mv[i] / denom);              // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
    if (vec) {
        *reinterpret_cast<float4*>(p) = *reinterpret_cast<const float4*>(pv);
        *reinterpret_cast<float4*>(m) = *reinterpret_cast<const float4*>(mv);
        *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(vv);
    } else {
        for (int i = 0; i < cnt; i++) { p[i] = pv[i]; m[i] = mv[i]; v[i] = vv[i]; }
    }
}

// Densification statistics of one view (2dgs/scene/gaussian_model.py:649-651 and the max_radii2D update of the
// training loop, train_with_refine_depth.py): for the Gaussians selected by `filter`
//   xyz_gradient_accum += |dL/dmean2D|_2,  denom += 1,  max_radii2D = max(max_radii2D, radii)
// in one pass (28 B per Gaussian) instead of boolean-mask gathers, a norm and scatters (about twenty launches).
__global__ void __launch_bounds__(256) densify_stats_kernel(int P, const float* __restrict__ grad, const uint8_t* __restrict__ filter,
                                                            const int* __restrict__ radii, float* __restrict__ accum,
                                                            float* __restrict__ denom, float* __restrict__ max_radii) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= P || !filter[i]) return;
    const float gx = grad[3 * i], gy = grad[3 * i + 1], gz = grad[3 * i + 2];
    accum[i] += sqrtf((gx * gx + gy * gy) + gz * gz);
    denom[i] += 1.0f;
    if (max_radii != nullptr) max_radii[i] = fmaxf(max_radii[i], (float)radii[i]);
}
}  // namespace g4s

// Activations of the Gaussian parameters as render() reads them (2dgs/scene/gaussian_model.py:157-192, without the
// optional mip filter): scales = exp(_scaling), rotations = _rotation / max(|_rotation|, 1e-12), opacity =
// sigmoid(_opacity) -- one pass instead of ~5 element-wise / reduction launches, and one pass for their backward
// instead of ~9.
namespace g4s {
__global__ void __launch_bounds__(256) activations_fwd_kernel(int P, const float2* __restrict__ scaling,
                                                              const float4* __restrict__ rotation,
                                                              const float* __restrict__ opacity, float2* __restrict__ scales,
                                                              float4* __restrict__ rots, float* __restrict__ opac) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= P) return;
    const float2 s = scaling[i];
    scales[i] = make_float2(expf(s.x), expf(s.y));
    const float4 q = rotation[i];
    const float n = fmaxf(sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w), 1e-12f);
    rots[i] = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
    opac[i] = 1.0f / (1.0f + expf(-opacity[i]));
}

__global__ void __launch_bounds__(256) activations_bwd_kernel(int P, const float2* __restrict__ scales,
                                                              const float4* __restrict__ rotation,
                                                              const float* __restrict__ opac, const float2* __restrict__ g_scales,
                                                              const float4* __restrict__ g_rots, const float* __restrict__ g_opac,
                                                              float2* __restrict__ d_scaling, float4* __restrict__ d_rotation,
                                                              float* __restrict__ d_opacity) {
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= P) return;
    const float2 y = scales[i], gs = g_scales[i];
    d_scaling[i] = make_float2(gs.x * y.x, gs.y * y.y);  // exp' = exp
    const float4 q = rotation[i], g = g_rots[i];
    const float norm = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    if (norm > 1e-12f) {  // y = q / |q|:  dq = (g - y <y, g>) / |q|
        const float inv = 1.0f / norm;
        const float4 u = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        const float d = ((u.x * g.x + u.y * g.y) + u.z * g.z) + u.w * g.w;
        d_rotation[i] = make_float4((g.x - u.x * d) * inv, (g.y - u.y * d) * inv, (g.z - u.z * d) * inv, (g.w - u.w * d) * inv);
    } else {              // clamped denominator: y = q / 1e-12
        d_rotation[i] = make_float4(g.x * 1e12f, g.y * 1e12f, g.z * 1e12f, g.w * 1e12f);
    }
    const float o = opac[i];
    d_opacity[i] = g_opac[i] * o * (1.0f - o);  // sigmoid' = y (1 - y)
}
}  // namespace g4s

extern "C" void g4s_activations_launch_internal(int fwd, int P, const float* scaling_or_scales, const float* rotation,
                                                const float* opacity_or_opac, const float* g_scales, const float* g_rots,
                                                const float* g_opac, float* out_s, float* out_r, float* out_o, hipStream_t s) {
    if (P <= 0) return;
    const dim3 grid((unsigned)((P + 255) / 256)), block(256);
    if (fwd)
        hipLaunchKernelGGL(g4s::activations_fwd_kernel, grid, block, 0, s, P, (const float2*)scaling_or_scales, (const float4*)rotation,
                           opacity_or_opac, (float2*)out_s, (float4*)out_r, out_o);
    else
        hipLaunchKernelGGL(g4s::activations_bwd_kernel, grid, block, 0, s, P, (const float2*)scaling_or_scales, (const float4*)rotation,
                           opacity_or_opac, (const float2*)g_scales, (const float4*)g_rots, g_opac, (float2*)out_s, (float4*)out_r,
                           out_o);
}

extern "C" void g4s_densify_stats_launch_internal(int P, const float* grad, const unsigned char* filter, const int* radii,
                                                  float* accum, float* denom, float* max_radii, hipStream_t s) {
    if (P <= 0) return;
    hipLaunchKernelGGL(g4s::densify_stats_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, P, grad, filter, radii,
                       accum, denom, max_radii);
}

extern "C" void g4s_adam_device_launch_internal(int nseg, float* const* params, const float* const* grads,
                                                float* const* exp_avg, float* const* exp_avg_sq, const long long* numel,
                                                const double* lr_dev, float* const* step_dev, float* coef_dev, double beta1,
                                                double beta2, double eps, hipStream_t s) {
    g4s::AdamPrep pr{};
    pr.nseg = nseg; pr.lr = lr_dev; pr.coef = coef_dev; pr.beta1 = beta1; pr.beta2 = beta2;
    g4s::AdamSegs a{};
    a.nseg = nseg; a.w1 = (float)(1.0 - beta1); a.w2 = (float)(1.0 - beta2); a.beta2 = (float)beta2; a.eps = (float)eps;
    a.coef = coef_dev;
    long long blocks4 = 0;
    for (int i = 0; i < nseg; i++) {
        pr.step[i] = step_dev[i];
        a.p[i] = params[i]; a.g[i] = grads[i]; a.m[i] = exp_avg[i]; a.v[i] = exp_avg_sq[i]; a.n[i] = numel[i];
        a.first[i] = blocks4;
        blocks4 += (numel[i] + 3) / 4;
    }
    for (int i = nseg; i <= 8; i++) a.first[i] = blocks4;
    hipLaunchKernelGGL(g4s::adam_prep_kernel, dim3(1), dim3(8), 0, s, pr);  // (the counts advance even when all segments are empty)
    if (blocks4 == 0) return;
    hipLaunchKernelGGL(g4s::adam_kernel, dim3((unsigned)((blocks4 + 255) / 256)), dim3(256), 0, s, a);
}

extern "C" void g4s_adam_launch_internal(int nseg, float* const* params, const float* const* grads, float* const* exp_avg,
                                         float* const* exp_avg_sq, const long long* numel, const double* lr, const int* step,
                                         double beta1, double beta2, double eps, hipStream_t s) {
    AdamSegs a{};
    a.nseg = nseg; a.w1 = (float)(1.0 - beta1); a.w2 = (float)(1.0 - beta2); a.beta2 = (float)beta2; a.eps = (float)eps;
    long long blocks4 = 0;
    for (int i = 0; i < nseg; i++) {
        a.p[i] = params[i]; a.g[i] = grads[i]; a.m[i] = exp_avg[i]; a.v[i] = exp_avg_sq[i]; a.n[i] = numel[i];
        a.first[i] = blocks4;
        blocks4 += (numel[i] + 3) / 4;
        const double bc1 = 1.0 - pow(beta1, (double)step[i]), bc2 = 1.0 - pow(beta2, (double)step[i]);
        a.step_size[i] = (float)(lr[i] / bc1);
        a.inv_sqrt_bc2[i] = (float)(1.0 / sqrt(bc2));
    }
    for (int i = nseg; i <= 8; i++) a.first[i] = blocks4;
    if (blocks4 == 0) return;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((blocks4 + 255) / 256)), dim3(256), 0, s, a);
}
