"""Golden vectors for the step right behind the rasterizer, produced by RUNNING THE
REFERENCE'S OWN CODE on the CPU of the build container:

    render_tail.npz   2d-gaussian-splatting/gaussian_renderer/__init__.py:19-166 `render()` end to end, with the
                      compiled rasterizer replaced by a stand-in that returns seeded (colour, radii, allmap) tensors
                      -- so everything behind the rasterizer call is the reference's: alpha / normal / depth maps,
                      nan_to_num, the depth_ratio blend, utils/point_utils.py:9-37 (depths_to_points,
                      depth_to_normal), the detach on alpha, the view<->world rotations -- and, by back-propagating
                      seeded cotangents of all eight maps through it, d(loss)/d(allmap) of the reference's autograd.

The fixtures pin oracle/render_maps_ref.py (CPU test) and the HIP kernels of csrc/maps.hip (GPU test).
Run in the build container only (needs /root/reference):  python tests/golden/make_golden_maps.py"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import reference_modules  # noqa: E402

MAPS = ("rend_alpha", "rend_normal", "rend_normal_cam", "rend_depth", "rend_dist", "surf_depth", "surf_normal", "surf_normal_cam")


def main():
    from g4splat_amd import synthetic
    rng = np.random.default_rng(77)
    out = {}
    cases = [(37, 26, 70.0, 0.0), (48, 32, 95.0, 1.0), (29, 41, 50.0, 0.35)]
    fake = types.ModuleType("diff_surfel_rasterization")

    class Settings(types.SimpleNamespace):
        pass

    state = {}

    class Rasterizer:
        def __init__(self, raster_settings):
            self.raster_settings = raster_settings

        def __call__(self, **kw):
            state["kwargs"] = kw
            return state["color"], state["radii"], state["allmap"]

    fake.GaussianRasterizationSettings = lambda **k: Settings(**k)
    fake.GaussianRasterizer = Rasterizer
    with reference_modules({"diff_surfel_rasterization": fake}):
        import gaussian_renderer as ref_gr
        for ci, (W, H, fov, ratio) in enumerate(cases):
            # a camera that is neither axis-aligned nor at the origin
            eye = rng.uniform(-1, 1, 3)
            cam = synthetic.look_at_camera(eye, eye + rng.normal(size=3), (0.1, 1.0, 0.05), math.radians(fov), W, H)
            view = types.SimpleNamespace(image_width=W, image_height=H, FoVx=cam.FoVx, FoVy=cam.FoVy, znear=0.01, zfar=100.0,
                                         world_view_transform=torch.tensor(cam.world_view_transform),
                                         full_proj_transform=torch.tensor(cam.full_proj_transform),
                                         camera_center=torch.tensor(cam.camera_center))
            # allmap as the rasterizer produces it: [0] sum w*depth, [1] alpha in [0,1] with exact zeros and ones,
            # [2:5] un-normalised view-space normals, [5] median depth (0 where nothing was hit), [6] distortion
            alpha = rng.uniform(0, 1, (H, W)).astype(np.float32)
            alpha[rng.uniform(size=(H, W)) < 0.08] = 0.0
            alpha[rng.uniform(size=(H, W)) < 0.05] = 1.0
            depth = rng.uniform(0.5, 6.0, (H, W)).astype(np.float32)
            n = rng.normal(size=(3, H, W)).astype(np.float32)
            allmap = np.concatenate([(depth * alpha)[None], alpha[None], n * alpha[None],
                                     (depth * rng.uniform(0.9, 1.1, (H, W)).astype(np.float32) * (alpha > 0))[None],
                                     (rng.uniform(0, 0.05, (H, W)).astype(np.float32) * alpha)[None]], 0).astype(np.float32)
            P = 5
            pc = types.SimpleNamespace(get_xyz=torch.zeros(P, 3), get_opacity=torch.ones(P, 1), get_scaling=torch.ones(P, 2),
                                       get_rotation=torch.ones(P, 4), get_features=torch.zeros(P, 16, 3), active_sh_degree=3,
                                       max_sh_degree=3)
            pipe = types.SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, depth_ratio=ratio, debug=False)
            state["color"] = torch.tensor(rng.uniform(0, 1, (3, H, W)).astype(np.float32))
            state["radii"] = torch.tensor([3, 0, 7, 0, 1], dtype=torch.int32)
            state["allmap"] = torch.tensor(allmap, requires_grad=True)
            rets = ref_gr.render(view, pc, pipe, torch.zeros(3))
            assert tuple(rets.keys()) == ("render", "viewspace_points", "visibility_filter", "radii", "rend_alpha", "rend_normal",
                                          "rend_normal_cam", "rend_dist", "surf_depth", "surf_normal", "surf_normal_cam",
                                          "rend_depth"), tuple(rets.keys())
            cot = {m: torch.tensor(rng.normal(size=tuple(rets[m].shape)).astype(np.float32)) for m in MAPS}
            loss = sum((rets[m] * cot[m]).sum() for m in MAPS)
            loss.backward()
            out[f"c{ci}_meta"] = np.array([W, H, ratio], np.float64)
            out[f"c{ci}_wvt"] = cam.world_view_transform
            out[f"c{ci}_fpt"] = cam.full_proj_transform
            out[f"c{ci}_allmap"] = allmap
            out[f"c{ci}_visibility_filter"] = rets["visibility_filter"].numpy()
            for m in MAPS:
                out[f"c{ci}_{m}"] = rets[m].detach().numpy()
                out[f"c{ci}_cot_{m}"] = cot[m].numpy()
            out[f"c{ci}_dL_dallmap"] = state["allmap"].grad.numpy()
            # what render() handed to the rasterizer (argument names / identity of the tensors)
            assert set(state["kwargs"]) == {"means3D", "means2D", "shs", "colors_precomp", "opacities", "scales", "rotations",
                                            "cov3D_precomp"}
    np.savez_compressed(os.path.join(HERE, "render_tail.npz"), **out)
    print("wrote render_tail.npz", os.path.getsize(os.path.join(HERE, "render_tail.npz")))


if __name__ == "__main__":
    main()
