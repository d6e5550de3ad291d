/*
 * g4s_optim.h -- C ABI of the fused Adam step over the Gaussian parameter groups,
 * libg4s_hip.so.
 *
 * Replaces torch.optim.Adam.step() as the reference configures it
 * (2d-gaussian-splatting/scene/gaussian_model.py:248-266: six parameter groups xyz / f_dc / f_rest / opacity /
 * scaling / rotation with their own learning rates, betas (0.9, 0.999), eps = 1e-15, no weight decay, no amsgrad)
 * -- ~10 full passes over 4 x 232 B per Gaussian in the foreach implementation -- by ONE kernel that reads
 * param / grad / exp_avg / exp_avg_sq once and writes param / exp_avg / exp_avg_sq once.  Same update rule:
 *
 *   m <- m + (1 - beta1) (g - m)              v <- beta2 v + (1 - beta2) g g
 *   p <- p - (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 */
#ifndef G4S_OPTIM_H_INCLUDED
#define G4S_OPTIM_H_INCLUDED

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * nseg (1..8) segments; HOST arrays of device pointers params/grads/exp_avg/exp_avg_sq, element counts
 * numel, learning rates lr and step counts step (t >= 1, AFTER the increment, per segment like torch's
 * per-parameter state["step"]).  Tensors are float32, updated in place; the hyper-parameters are doubles like
 * the Python floats torch works from (1 - beta2 must be formed in double: in float32 it is off by 1.3e-5).
 */
int g4s_adam_step(int nseg, float* const* params, const float* const* grads, float* const* exp_avg,
                  float* const* exp_avg_sq, const long long* numel, const double* lr, const int* step, double beta1,
                  double beta2, double eps, void* stream);

/*
 * The same update with the step counts and learning rates read from DEVICE memory, for launches captured in a hipGraph
 * (a replay must advance t and may see a new learning rate): lr_dev [nseg] float64 on the device (doubles, like `lr` above); step_dev = HOST array of
 * nseg device pointers to float32 scalars holding t BEFORE the call (torch's capturable state["step"]); each is
 * incremented by one on the device, then used.  coef_dev: 16 floats of device scratch owned by the caller for the
 * lifetime of the launch (the bias-correction factors, computed in double on the device like g4s_adam_step does on the
 * host).  Two launches.
 */
int g4s_adam_step_device(int nseg, float* const* params, const float* const* grads, float* const* exp_avg,
                         float* const* exp_avg_sq, const long long* numel, const double* lr_dev, float* const* step_dev,
                         float* coef_dev, double beta1, double beta2, double eps, void* stream);

/*
 * Densification statistics of one rendered view, fused.  Replaces GaussianModel.add_densification_stats
 * (2d-gaussian-splatting/scene/gaussian_model.py:649-651) and the max_radii2D update of the training loop
 * (train_with_refine_depth.py, "max_radii2D[visibility_filter] = max(max_radii2D[visibility_filter], radii[...])"):
 * for every i with update_filter[i] != 0
 *     xyz_gradient_accum[i] += || grad_mean2D[i, 0:3] ||_2      denom[i] += 1
 *     max_radii2D[i] = max(max_radii2D[i], radii[i])            (skipped when max_radii2D is NULL)
 * grad_mean2D [P,3] is the rasterizer's dL_dmean2D (the gradient of `viewspace_points`), update_filter one byte
 * per Gaussian (a torch bool tensor), radii int32 [P]; the three statistics are float32 [P], updated in place.
 */
int g4s_densify_stats(int P, const float* grad_mean2D, const unsigned char* update_filter, const int* radii,
                      float* xyz_gradient_accum, float* denom, float* max_radii2D, void* stream);

/*
 * Parameter activations as render() reads them (2d-gaussian-splatting/scene/gaussian_model.py:157-192, mip filter
 * off): scales = exp(_scaling) [P,2], rotations = normalize(_rotation) [P,4] (x / max(|x|, 1e-12), as
 * torch.nn.functional.normalize), opacities = sigmoid(_opacity) [P].  One launch each way instead of ~5 / ~9.
 * The backward takes the activated scales / opacities the forward produced and the raw rotations.
 * Scale tensors 8-byte, rotation tensors 16-byte aligned.
 */
int g4s_activations_forward(int P, const float* scaling_raw, const float* rotation_raw, const float* opacity_raw,
                            float* scales, float* rotations, float* opacities, void* stream);
int g4s_activations_backward(int P, const float* scales, const float* rotation_raw, const float* opacities,
                             const float* dL_dscales, const float* dL_drotations, const float* dL_dopacities,
                             float* dL_dscaling_raw, float* dL_drotation_raw, float* dL_dopacity_raw, void* stream);

/*
 * Stream compaction of Gaussian rows: the device side of prune_points / densify_and_clone / densify_and_split
 * (2d-gaussian-splatting/scene/gaussian_model.py:510-541, 583-626), which the reference expresses as one boolean-mask
 * indexing per tensor (six parameters, twelve Adam moments, three statistics).
 *
 *   g4s_compact_scan    scans `keep` (one byte per row, a torch bool tensor) ONCE: wave ballot + popcount per 256 rows,
 *                       a single-block scan of the block counts; the number of kept rows goes to *out_count (device int).
 *                       `workspace` (>= g4s_compact_workspace(P) bytes) holds the scan for the gathers that follow.
 *   g4s_compact_gather  copies the kept rows of `nseg` row-major float tensors src[s] = [P, widths[s]] to
 *                       dst[s] + dst_row0 * widths[s], in index order (stable, like mask indexing), eight tensors per
 *                       launch.  src / dst / widths are HOST arrays.  dst[s] must hold dst_row0 + count rows.
 */
size_t g4s_compact_workspace(int P);
int g4s_compact_scan(int P, const unsigned char* keep, int* out_count, char* workspace, size_t workspace_bytes, void* stream);
int g4s_compact_gather(int P, const unsigned char* keep, const char* workspace, int nseg, const float* const* src,
                       float* const* dst, const int* widths, long long dst_row0, void* stream);

#ifdef __cplusplus
}
#endif
#endif
