/*
 * g4s_render_maps.h -- C ABI of the fused map post-processing that follows the rasterizer in
 * G4Splat's `render()`, libg4s_hip.so.
 *
 * Replaces the ~15 element-wise torch kernels of
 *   2d-gaussian-splatting/gaussian_renderer/__init__.py:117-164   (alpha / normal / depth maps)
 *   2d-gaussian-splatting/utils/point_utils.py:9-37               (depths_to_points, depth_to_normal)
 * by ONE forward and ONE backward kernel (plus a one-thread camera-algebra kernel).  The reference has
 * no C++ interface for this step (it is Python); the entry points below are what a maintainer would
 * call from `render()` through a `torch.autograd.Function` -- g4splat_amd/render_maps.py is that binding.
 *
 * Conventions as in g4s_rasterizer.h: device pointers, planar [C,H,W] float32 maps, the caller's
 * hipStream_t as void*, int status + g4s_last_error().
 */
#ifndef G4S_RENDER_MAPS_H_INCLUDED
#define G4S_RENDER_MAPS_H_INCLUDED

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes of device scratch both calls need (the derived camera: rotation, ray matrix, origin). */
size_t g4s_render_maps_workspace(void);

/*
 * Forward.  gaussian_renderer/__init__.py:117-164 with `allmap` = the rasterizer's [7,H,W] output
 * (0 = sum w*depth, 1 = alpha, 2..4 = view-space normal, 5 = median depth, 6 = distortion):
 *
 *   rend_alpha      [1,H,W] = allmap[1]                                            (:118)
 *   rend_normal     [3,H,W] = allmap[2:5] rotated to world space                   (:121-123)
 *   rend_normal_cam [3,H,W] = allmap[2:5]                                          (:122)
 *   rend_depth      [1,H,W] = nan_to_num(allmap[0] / allmap[1], 0, 0)              (:130-131)
 *   rend_dist       [1,H,W] = allmap[6]                                            (:134)
 *   surf_depth      [1,H,W] = rend_depth (1 - depth_ratio) + depth_ratio nan_to_num(allmap[5], 0, 0)   (:139)
 *   surf_normal     [3,H,W] = normalize(cross(dP/drow, dP/dcol)) * alpha, P = back-projected surf_depth,
 *                             central differences, zero on the 1-pixel border   (:142-146, point_utils.py:26-37)
 *   surf_normal_cam [3,H,W] = surf_normal rotated to view space                    (:149)
 *
 * world_view_transform / full_proj_transform: the camera's 4x4 matrices exactly as the reference
 * stores them (row-major torch tensors, row-vector convention, scene/cameras.py:55-57).
 */
int g4s_render_maps_forward(int width, int height, const float* allmap, const float* world_view_transform,
                            const float* full_proj_transform, float depth_ratio, float* rend_alpha, float* rend_normal,
                            float* rend_normal_cam, float* rend_depth, float* rend_dist, float* surf_depth,
                            float* surf_normal, float* surf_normal_cam, char* workspace, size_t workspace_bytes,
                            void* stream);

/*
 * Backward: dL/dallmap [7,H,W] (every element written) from the gradients of the eight maps; any
 * dL_* pointer may be NULL (= zero).  surf_depth is the forward's output.  Alpha is detached inside
 * surf_normal exactly as in the reference (:146).  Gather form, no atomics: bit-reproducible.
 */
int g4s_render_maps_backward(int width, int height, const float* allmap, const float* surf_depth,
                             const float* world_view_transform, const float* full_proj_transform, float depth_ratio,
                             const float* dL_rend_alpha, const float* dL_rend_normal, const float* dL_rend_normal_cam,
                             const float* dL_rend_depth, const float* dL_rend_dist, const float* dL_surf_depth,
                             const float* dL_surf_normal, const float* dL_surf_normal_cam, float* dL_dallmap,
                             char* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
