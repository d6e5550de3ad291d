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

#ifdef __cplusplus
}
#endif
#endif
