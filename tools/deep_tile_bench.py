"""Backward time on a frame whose work sits in a few very deep tiles (the one-wave-per-tile kernel's worst case),
with and without the four-wave deep-tile kernel (option "bwd_hot_threshold"), and on the metric workload.

    python tools/deep_tile_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import cotangents, scene_inputs  # noqa: E402
from g4splat_amd.diff_surfel_rasterization import _C  # noqa: E402
from g4splat_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")


def run(inp, label):
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    a = {k: t(v) for k, v in inp.items() if isinstance(v, np.ndarray)}
    gc, go = (t(x) for x in cotangents(inp["H"], inp["W"], seed=1))
    fw = _C.rasterize_gaussians(a["bg"], a["means3D"], a["colors"], a["opacity"], a["scales"], a["rotations"], 1.0,
                                a["transMat"], a["view"], a["proj"], inp["tanfovx"], inp["tanfovy"], inp["H"], inp["W"],
                                a["sh"], inp["D"], a["campos"], False, False)
    R, _c, _o, radii, geom, binning, img = fw
    for thr in ("default", "1000000000", "0"):
        _lib.set_option("bwd_hot_threshold", int(thr) if thr != "default" else _lib.OPTION_UNSET)
        ts = []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _C.rasterize_gaussians_backward(a["bg"], a["means3D"], radii, a["colors"], a["scales"], a["rotations"], 1.0,
                                            a["transMat"], a["view"], a["proj"], inp["tanfovx"], inp["tanfovy"], gc, go,
                                            a["sh"], inp["D"], a["campos"], geom, R, binning, img, False)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"{label}: R={R} threshold={thr:>10s}: backward {1e3 * min(ts):8.3f} ms", flush=True)
    _lib.set_option("bwd_hot_threshold", _lib.OPTION_UNSET)


run(scene_inputs(P=60000, W=48, H=32, seed=77, D=1, opacity_max=0.03, scale_mul=4.0, fov_deg=110.0), "6 deep tiles (10^4 each)")
run(scene_inputs(P=400000, W=256, H=256, seed=78, D=1, opacity_max=0.02, scale_mul=2.0, fov_deg=60.0), "256 tiles, translucent fog")
import bench  # noqa: E402
scene, cams, d, dcams, (P, W, H, D) = bench.build_scene("s3", dev)
cam = dcams[0]
inp = dict(bg=np.zeros(3, np.float32), means3D=scene.means3D, colors=np.zeros((0,), np.float32), opacity=scene.opacities,
           scales=scene.scales, rotations=scene.rotations, transMat=np.zeros((0,), np.float32),
           view=cams[0].world_view_transform, proj=cams[0].full_proj_transform, tanfovx=cams[0].tanfovx,
           tanfovy=cams[0].tanfovy, H=H, W=W, sh=scene.shs, D=D, campos=cams[0].camera_center)
run(inp, "metric workload s3")
