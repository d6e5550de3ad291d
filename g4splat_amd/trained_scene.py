"""A scene with a TRAINED distribution, as a seeded recipe (bench.py --workload s3t; tests/test_gpu_parity.py).

Every other synthetic scene of this repository is drawn i.i.d.: surfels on box faces with a log-normal size, a normal
jitter of a few degrees, unimodal opacities (synthetic.scene_room).  What the reference's loop produces after thousands
of iterations (train_with_refine_depth.py:583-593: densify / split / prune every 100 iterations, SH degree raised every
1000, opacity reset every 3000; arguments/__init__.py:90-94) looks different -- elongated splats from clones that
drifted apart, split children at 1/1.6 of their parent's size, an opacity histogram with one mode near the reset value
and one near 1, deep tiles where many translucent surfels overlap -- and that is the distribution the blend kernels
meet in production.  No dataset exists in the build container or on the GPU box, so the stand-in is made here: the
room scene is RE-LEARNT from its own renders through the product's training path (render() -> L1 + SSIM + normal /
distortion regularisers -> backward -> fused Adam, densify_and_prune on the device) with the reference's schedule in
miniature, and the trainee -- not the ground truth -- is the scene.  Seeded (numpy for the perturbation, torch for
densify_and_split's normal draws): the same build on the same hardware gives the same bits.

Needs the GPU: the recipe IS the product path.  bench.py trains it before the warm-up (a few seconds at 1.5 M surfels)."""
import math
from types import SimpleNamespace

import numpy as np

from . import synthetic


def scene_trained(seed=0, iters=1000, P=1_500_000, width=1600, height=1200, nviews=8, device="cuda:0", log=None,
                  start_factor=1.6):
    """-> (synthetic.Scene on the host, info dict).  `P` surfels of synthetic.scene_room(P, seed) are the ground truth;
    the trainee starts as `start_factor` x P of them (every one once, the rest drawn again) displaced by 2 cm, 1.5 x as
    large, grey, opacity 0.3, SH degree 0 -- the pruning after the opacity reset takes about 40 % away again, which
    leaves a scene of about P surfels."""
    import torch

    from .gaussian_model import GaussianModel
    from .gaussian_renderer import render
    from .losses import geometry_regularizers, photometric_loss

    dev = torch.device(device)
    torch.manual_seed(seed)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    truth_scene = synthetic.scene_room(P, seed=seed)
    cams = []
    for c in synthetic.room_cameras(nviews, width, height, fovx_deg=90.0):
        cams.append(SimpleNamespace(image_width=width, image_height=height, FoVx=2 * math.atan(c.tanfovx),
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

    truth = build(truth_scene.means3D, truth_scene.scales, truth_scene.rotations, truth_scene.opacities, truth_scene.shs, 3)
    with torch.no_grad():
        targets = [render(c, truth, pipe, bg)["render"].clone() for c in cams]
    del truth
    rng = np.random.default_rng(seed + 1)
    pick = np.concatenate([np.arange(P), rng.integers(0, P, max(int(P * (start_factor - 1.0)), 0))])
    model = build(truth_scene.means3D[pick] + rng.normal(0, 0.02, (pick.size, 3)).astype(np.float32),
                  truth_scene.scales[pick] * 1.5, truth_scene.rotations[pick], np.full_like(truth_scene.opacities[pick], 0.3),
                  np.zeros_like(truth_scene.shs[pick]), 0)
    model.training_setup()
    extent = 5.0
    every = max(iters // 10, 1)  # the reference's schedule, compressed: ten densification rounds, three SH raises, one reset
    history = []
    for it in range(1, iters + 1):
        if it % every == 0 and model.active_sh_degree < 3:
            model.active_sh_degree += 1
        i = (it * 5) % len(cams)
        out = render(cams[i], model, pipe, bg)
        loss, _l1, _s = photometric_loss(out["render"], targets[i], 0.2)
        nm, dm = geometry_regularizers(out["rend_normal"], out["surf_normal"], out["rend_dist"])
        total = loss + 0.05 * nm + (100.0 * dm if it > iters // 2 else 0.0)
        total.backward()
        with torch.no_grad():
            model.add_densification_stats(out["viewspace_points"], out["visibility_filter"], out["radii"])
            if it >= every and it % every == 0 and it < iters - every // 2:
                model.densify_and_prune(0.0002, 0.05, extent, 20 if it > 3 * every else None)
                history.append((it, int(model.get_xyz.shape[0])))
            if it == iters // 2:
                model.reset_opacity()
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
        if log is not None and it % every == 0:
            log(f"scene_trained: iteration {it}, {int(model.get_xyz.shape[0])} surfels, loss {float(total.detach()):.4f}")
    with torch.no_grad():
        scene = synthetic.Scene(means3D=model.get_xyz.detach().cpu().numpy().astype(np.float32),
                                scales=model.get_scaling.detach().cpu().numpy().astype(np.float32),
                                rotations=model.get_rotation.detach().cpu().numpy().astype(np.float32),
                                opacities=model.get_opacity.detach().cpu().numpy().astype(np.float32),
                                shs=model.get_features.detach().cpu().numpy().astype(np.float32))
    sc = scene.scales
    info = {"seed": seed, "iterations": iters, "P_truth": int(P), "P_start": int(pick.size), "P": int(scene.means3D.shape[0]), "surfels_after_each_densification": history,
            "aspect_ratio_median": float(np.median(sc.max(1) / np.maximum(sc.min(1), 1e-12))),
            "aspect_ratio_p99": float(np.quantile(sc.max(1) / np.maximum(sc.min(1), 1e-12), 0.99)),
            "opacity_below_0.1": float((scene.opacities < 0.1).mean()), "opacity_above_0.9": float((scene.opacities > 0.9).mean())}
    del model, targets
    torch.cuda.empty_cache()
    return scene, info
