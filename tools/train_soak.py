"""Soak run of the product training path on one MI355X: a room scene is re-learnt from its own renders for several
hundred iterations with the reference's schedule in miniature -- SH degree raised every 100 iterations, densify +
prune every 100 from iteration 100 (gaussian_model.py:628-647), opacity reset once -- and the run reports loss,
PSNR against the targets, the Gaussian count and whether anything became non-finite.

    python tools/train_soak.py [--iters 600] [--P 120000] [--width 640] [--height 480] [--graph]

--graph: the iterations between two schedule events (every 100th iteration: SH degree, densification, opacity reset, the
distortion weight) are replayed from a HIP graph (g4splat_amd.graphed.TrainStepGraph), which is captured again after every
event; the event iterations themselves run eagerly.  Same kernels in the same order: the log must equal the eager run's."""
import argparse
import json
import math
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from g4splat_amd import synthetic  # noqa: E402
from g4splat_amd.gaussian_model import GaussianModel  # noqa: E402
from g4splat_amd.gaussian_renderer import render  # noqa: E402
from g4splat_amd.losses import geometry_regularizers, photometric_loss  # noqa: E402


def psnr(img1, img2):  # 2dgs/utils/image_utils.py:19-21
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=600)
    ap.add_argument("--P", type=int, default=120_000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)  # (densify_and_split draws its children from torch's generator: two runs are comparable)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    scene = synthetic.scene_room(a.P, seed=0, scale_mean=0.04)
    cams = []
    for c in synthetic.room_cameras(8, a.width, a.height, fovx_deg=90.0):
        cams.append(SimpleNamespace(image_width=a.width, image_height=a.height, FoVx=2 * math.atan(c.tanfovx),
                                    FoVy=2 * math.atan(c.tanfovy), world_view_transform=t(c.world_view_transform),
                                    full_proj_transform=t(c.full_proj_transform), camera_center=t(c.camera_center),
                                    znear=0.01, zfar=100.0))
    pipe = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)

    def build(means, scales, rots, opac, shs, degree):
        m = GaussianModel(sh_degree=3)
        m.create_from_parameters(t(means), t(scales), t(rots), torch.rand((means.shape[0], 3), device=dev))
        with torch.no_grad():
            m._opacity.copy_(torch.logit(t(opac).clamp(1e-4, 1 - 1e-4)))
            m._features_dc.copy_(t(shs[:, :1, :]))
            m._features_rest.copy_(t(shs[:, 1:, :]))
        m.active_sh_degree = degree
        return m

    truth = build(scene.means3D, scene.scales, scene.rotations, scene.opacities, scene.shs, 3)
    with torch.no_grad():
        targets = [render(c, truth, pipe, bg)["render"].clone() for c in cams]
    # the trainee: a third of the surfels, displaced, grey, twice as large
    rng = np.random.default_rng(1)
    keep = rng.choice(a.P, a.P // 3, replace=False)
    shs0 = np.zeros_like(scene.shs[keep])
    model = build(scene.means3D[keep] + rng.normal(0, 0.02, (keep.size, 3)).astype(np.float32), scene.scales[keep] * 2.0,
                  scene.rotations[keep], np.full_like(scene.opacities[keep], 0.3), shs0, 0)
    model.training_setup(capturable=a.graph)
    extent = 5.0

    def evaluate():
        with torch.no_grad():
            return float(torch.stack([psnr(render(c, model, pipe, bg)["render"][None], g[None]).mean()
                                      for c, g in zip(cams, targets)]).mean())

    log = {"psnr_start": evaluate(), "P_start": int(model.get_xyz.shape[0])}
    finite = True
    densify_ms = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step, captures = None, 0

    def capture(first_it):
        """A graph for the iterations from `first_it` up to the next schedule event."""
        from g4splat_amd.diff_surfel_rasterization import _C
        from g4splat_amd.graphed import TrainStepGraph
        R, empty = 0, torch.empty(0, device=dev)
        with torch.no_grad():
            for cam in cams:
                fw = _C.rasterize_gaussians(bg, model.get_xyz, empty, model.get_opacity, model.get_scaling, model.get_rotation, 1.0,
                                            empty, cam.world_view_transform, cam.full_proj_transform, math.tan(cam.FoVx * 0.5),
                                            math.tan(cam.FoVy * 0.5), a.height, a.width, model.get_features, model.active_sh_degree,
                                            cam.camera_center, False, False)
                R = max(R, int(fw[0]))
        w_dist = 100.0 if first_it > a.iters // 2 else 0.0

        def body(out, gt):
            loss, _l1, _s = photometric_loss(out["render"], gt, 0.2)
            nm, dm = geometry_regularizers(out["rend_normal"], out["surf_normal"], out["rend_dist"])
            return loss + 0.05 * nm + (w_dist * dm if w_dist else 0.0)
        return TrainStepGraph(model, body, cams[0], (3, a.height, a.width), instance_capacity=int(R * 1.5) + 1024, pipe=pipe, bg=bg)

    for it in range(1, a.iters + 1):
        if it % 100 == 0 and model.active_sh_degree < 3:
            model.active_sh_degree += 1
        i = (it * 5) % 8
        if a.graph and it % 100 != 0:
            if step is None:
                step = capture(it)
                captures += 1
            step(cams[i], targets[i])
            continue
        if step is not None:
            assert not step.overflowed(), it
            step = None
        out = render(cams[i], model, pipe, bg)
        loss, _l1, _s = photometric_loss(out["render"], targets[i], 0.2)
        nm, dm = geometry_regularizers(out["rend_normal"], out["surf_normal"], out["rend_dist"])
        total = loss + 0.05 * nm + (100.0 * dm if it > a.iters // 2 else 0.0)
        total.backward()
        with torch.no_grad():
            model.add_densification_stats(out["viewspace_points"], out["visibility_filter"], out["radii"])
            if it >= 100 and it % 100 == 0 and it < a.iters - 50:
                torch.cuda.synchronize()
                td = time.perf_counter()
                model.densify_and_prune(0.0002, 0.05, extent, 20 if it > 300 else None)
                torch.cuda.synchronize()
                densify_ms.append((time.perf_counter() - td) * 1e3)
            if it == a.iters // 2:
                model.reset_opacity()
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
        if it % 100 == 0:
            lv = float(total.detach())
            finite = finite and math.isfinite(lv)
            log[f"it{it}"] = {"loss": round(lv, 5), "P": int(model.get_xyz.shape[0])}
    torch.cuda.synchronize()
    log["ms_per_iteration"] = round((time.perf_counter() - t0) / a.iters * 1e3, 3)
    log["densify_and_prune_ms"] = {"calls": len(densify_ms), "mean": round(float(np.mean(densify_ms)), 2) if densify_ms else None,
                                   "max": round(float(np.max(densify_ms)), 2) if densify_ms else None}
    log["psnr_end"] = evaluate()
    for p in (model._xyz, model._scaling, model._rotation, model._opacity, model._features_dc, model._features_rest):
        finite = finite and bool(torch.isfinite(p).all())
    log["finite"] = finite
    log["hip_graph"] = {"captures": captures} if a.graph else False
    log["P_end"] = int(model.get_xyz.shape[0])
    log["max_memory_GB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 3)
    print(json.dumps(log))
    assert finite and log["psnr_end"] > log["psnr_start"] + 3.0


if __name__ == "__main__":
    main()
