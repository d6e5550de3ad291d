"""sha1 of every output and gradient of a few frames (S3 view 0, a 10 k scene at every SH degree, precomputed colours):
run with two library builds to see whether a change is bit-identical.      python tools/grad_digest.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import cotangents, run_hip, scene_inputs  # noqa: E402
from g4splat_amd import _lib, synthetic  # noqa: E402

E0 = np.zeros(0, np.float32)
print(_lib.load().g4s_version().decode())


def digest(tag, inp):
    h = run_hip(inp, cotangents(inp["H"], inp["W"], seed=3))
    parts = [("color", h["color"]), ("others", h["others"]), ("radii", h["radii"])] + sorted(h["grads"].items())
    print(tag, " ".join(f"{k}:{hashlib.sha1(np.ascontiguousarray(v).tobytes()).hexdigest()[:10]}" for k, v in parts))


for D in range(4):
    digest(f"10k D={D}", scene_inputs(P=10000, W=256, H=256, seed=D, D=D))
digest("sub-pixel", scene_inputs(P=5000, W=177, H=130, seed=9, D=1, scale_mul=0.02))
P, W, H = 1_500_000, 1600, 1200
scene = synthetic.scene_room(P, seed=0)
cam = synthetic.room_cameras(8, W, H, fovx_deg=90.0)[0]
digest("S3 view 0", dict(bg=np.zeros(3, np.float32), means3D=scene.means3D, colors=E0, opacity=scene.opacities,
                         scales=scene.scales, rotations=scene.rotations, scale_modifier=1.0, transMat=E0,
                         view=cam.world_view_transform, proj=cam.full_proj_transform, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                         H=H, W=W, sh=scene.shs, D=3, campos=cam.camera_center))
