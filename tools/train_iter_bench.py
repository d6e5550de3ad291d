"""Whole training iteration on the MI355X at the metric's size (S3: 1.5 M surfels, 1600x1200, SH degree 3), product path
only: render() [HIP rasterizer + fused maps] -> fused L1+SSIM loss + normal-consistency and distortion regularisers
(train_with_refine_depth.py:378-399) -> backward -> FusedAdam step + densification statistics.

    python tools/train_iter_bench.py [--iters 40] [--torch-adam] [--graph]

--graph: the same iteration captured once in a HIP graph and replayed (g4splat_amd.graphed.TrainStepGraph; presized
rasterizer state, FusedAdam with step counts and learning rates on the device): one launch from the host per iteration.

Prints wall time per iteration and the per-kernel-group milliseconds from the library's profiling hooks."""
import argparse
import ctypes
import json
import math
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g4splat_amd import _lib, synthetic  # noqa: E402
from g4splat_amd.gaussian_model import GaussianModel  # noqa: E402
from g4splat_amd.gaussian_renderer import render  # noqa: E402
from g4splat_amd.losses import geometry_regularizers, photometric_loss  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--P", type=int, default=1_500_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the fused kernel")
    ap.add_argument("--graph", action="store_true", help="replay the iteration from a HIP graph")
    ap.add_argument("--no-kernel-timing", action="store_true", help="leave the library's per-kernel event pairs off (they cost ~5 us each)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    scene = synthetic.scene_room(a.P, seed=0)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    model = GaussianModel(sh_degree=3)
    model.create_from_parameters(t(scene.means3D), t(scene.scales), t(scene.rotations), torch.rand((a.P, 3), device=dev))
    with torch.no_grad():
        model._opacity.copy_(torch.logit(t(scene.opacities).clamp(1e-4, 1 - 1e-4)))
        model._features_rest.copy_(t(scene.shs[:, 1:, :]))
    model.active_sh_degree = 3
    model.training_setup(fused=not a.torch_adam, capturable=a.graph)
    cams = []
    for c in synthetic.room_cameras(8, a.width, a.height, fovx_deg=90.0):
        cams.append(SimpleNamespace(image_width=a.width, image_height=a.height, FoVx=2 * math.atan(c.tanfovx),
                                    FoVy=2 * math.atan(c.tanfovy), world_view_transform=t(c.world_view_transform),
                                    full_proj_transform=t(c.full_proj_transform), camera_center=t(c.camera_center),
                                    znear=0.01, zfar=100.0))
    gts = [torch.rand((3, a.height, a.width), device=dev) for _ in cams]
    pipe = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)

    def iteration(i):
        out = render(cams[i % 8], model, pipe, bg)
        loss, _l1, _s = photometric_loss(out["render"], gts[i % 8], 0.2)
        normal_mean, dist_mean = geometry_regularizers(out["rend_normal"], out["surf_normal"], out["rend_dist"])
        total = loss + 0.05 * normal_mean + 100.0 * dist_mean
        total.backward()
        with torch.no_grad():
            model.add_densification_stats(out["viewspace_points"], out["visibility_filter"], out["radii"])
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)

    if a.graph:
        from g4splat_amd.diff_surfel_rasterization import _C
        from g4splat_amd.graphed import TrainStepGraph
        R, empty = 0, torch.empty(0, device=dev)
        with torch.no_grad():
            for cam in cams:
                fw = _C.rasterize_gaussians(bg, model.get_xyz, empty, model.get_opacity, model.get_scaling, model.get_rotation,
                                            1.0, empty, cam.world_view_transform, cam.full_proj_transform,
                                            math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), a.height, a.width,
                                            model.get_features, 3, cam.camera_center, False, False)
                R = max(R, int(fw[0]))
        del fw

        def body(out, gt):
            loss, _l1, _s = photometric_loss(out["render"], gt, 0.2)
            normal_mean, dist_mean = geometry_regularizers(out["rend_normal"], out["surf_normal"], out["rend_dist"])
            return loss + 0.05 * normal_mean + 100.0 * dist_mean
        step = TrainStepGraph(model, body, cams[0], (3, a.height, a.width), instance_capacity=int(R * 1.3), pipe=pipe, bg=bg)

        def iteration(i):  # noqa: F811
            model.update_learning_rate(i + 1)
            step(cams[i % 8], gts[i % 8])
    for i in range(8):
        iteration(i)
    torch.cuda.synchronize()
    lib.g4s_profile_reset()
    lib.g4s_profile_enable(0 if (a.graph or a.no_kernel_timing) else 1)
    t0 = time.perf_counter()
    for i in range(a.iters):
        iteration(i)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters * 1e3
    lib.g4s_profile_enable(0)
    ker = {}
    for k in range(lib.g4s_profile_kernels()):
        ms, cnt = ctypes.c_double(), ctypes.c_int()
        lib.g4s_profile_read(k, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            ker[lib.g4s_profile_name(k).decode()] = round(ms.value / a.iters, 4)
    print(json.dumps({"P": a.P, "resolution": [a.width, a.height], "iters": a.iters, "ms_per_iteration": round(wall, 3),
                      "optimizer": "torch.optim.Adam" if a.torch_adam else "FusedAdam", "hip_graph": bool(a.graph),
                      "library_kernels_ms_per_iteration": ker, "library_kernels_sum_ms": round(sum(ker.values()), 3)}))


if __name__ == "__main__":
    main()
