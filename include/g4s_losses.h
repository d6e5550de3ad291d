/*
 * g4s_losses.h -- C ABI of the fused photometric loss of G4Splat's training step
 * (SURVEY.md 8(f) f2), libg4s_hip.so.
 *
 * Replaces, for one [3,H,W] render and its ground-truth image,
 *   Ll1  = l1_loss(image, gt)                                   2d-gaussian-splatting/utils/loss_utils.py:17-18
 *   ssim = ssim(image, gt)      (11x11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2)
 *                                                               utils/loss_utils.py:31-33, 46-79
 *   loss = (1 - lambda_dssim) Ll1 + lambda_dssim (1 - ssim)     train_with_refine_depth.py:382-383
 * and their autograd backward (five grouped 11x11 convolutions forward, their transposes backward, ~25
 * element-wise kernels) by two tiled kernels + one reduction.  The value AND dloss/dimage are produced by the
 * same call -- the training step always needs both -- so the autograd binding
 * (g4splat_amd/losses.py) only scales the stored gradient in its backward.
 */
#ifndef G4S_LOSSES_H_INCLUDED
#define G4S_LOSSES_H_INCLUDED

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Device scratch needed for a width x height image (three derivative maps per channel + block partials). */
size_t g4s_photometric_workspace(int width, int height);

/*
 *   image, gt   [3,H,W] float32 (device)
 *   out3        device float[3]: loss, Ll1, ssim
 *   dL_dimage   [3,H,W]: d loss / d image (every element written); may be NULL (value only)
 * Sums are reduced in a fixed order (no atomics): bit-reproducible.
 */
int g4s_photometric_loss(int width, int height, const float* image, const float* gt, float lambda_dssim, float* out3,
                         float* dL_dimage, char* workspace, size_t workspace_bytes, void* stream);

/*
 * Geometry regularisers of the training step, fused (train_with_refine_depth.py:391-396):
 *   out2[0] = mean over pixels of  1 - sum_c rend_normal[c] * surf_normal[c]     ("normal_error.mean()")
 *   out2[1] = mean over pixels of  rend_dist                                      ("rend_dist.mean()")
 * rend_normal, surf_normal [3,H,W], rend_dist [1,H,W] float32 (device), out2 device float[2]; the caller applies
 * lambda_normal / lambda_dist.  Fixed-order reduction (bit-reproducible).  The backward takes the cotangents of the
 * two means (device float[2]) and writes every element of the three gradients:
 *   dL_drend_normal = -(g0/N) surf_normal,  dL_dsurf_normal = -(g0/N) rend_normal,  dL_drend_dist = g1/N.
 */
size_t g4s_geometry_regularizers_workspace(int width, int height);
int g4s_geometry_regularizers_forward(int width, int height, const float* rend_normal, const float* surf_normal,
                                      const float* rend_dist, float* out2, char* workspace, size_t workspace_bytes,
                                      void* stream);
int g4s_geometry_regularizers_backward(int width, int height, const float* rend_normal, const float* surf_normal,
                                       const float* grad_out2, float* dL_drend_normal, float* dL_dsurf_normal,
                                       float* dL_drend_dist, void* stream);

#ifdef __cplusplus
}
#endif
#endif
