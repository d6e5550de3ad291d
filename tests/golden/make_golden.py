"""Generates the committed golden fixtures under tests/golden/ (run in the BUILD container only).

Part A imports the reference's own Python helpers from /root/reference (they never travel to
the GPU box) and records their outputs on seeded inputs:
  camera_*.npz   getWorld2View2 / getProjectionMatrix + the Camera.__init__ matrix algebra
                 (2dgs/utils/graphics_utils.py:38-71, 2dgs/scene/cameras.py:49-58)
  sh_eval.npz    eval_sh (2dgs/utils/sh_utils.py:57-112) for degrees 0..3
  transmat.npz   the python formulation of the splat->pixel matrix T
                 (2dgs/gaussian_renderer/__init__.py:64-75 with build_scaling_rotation,
                 2dgs/utils/general_utils.py:79-111, 2dgs/scene/gaussian_model.py:30-36)
  losses.npz     l1_loss / ssim / psnr (2dgs/utils/loss_utils.py, image_utils.py) on seeded images
Part B records the CPU oracle's outputs on a small seeded scene (oracle_small.npz): a regression
pin of the restatement itself (NOT of the reference, which has no runnable path here).

    python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/2d-gaussian-splatting"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def reference_part():
    sys.path.insert(0, REF)
    from utils import graphics_utils as gu
    from utils import sh_utils as shu
    from utils import general_utils as genu
    from utils import loss_utils as lu
    from utils import image_utils as iu

    rng = np.random.default_rng(20240607)
    # ---- cameras -------------------------------------------------------------------------
    cams = []
    for i in range(4):
        A = rng.normal(size=(3, 3))
        Q, _ = np.linalg.qr(A)
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        t = rng.normal(size=3)
        fovx, W, H = math.radians(rng.uniform(40, 100)), int(rng.integers(64, 400)), int(rng.integers(64, 300))
        fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
        # Camera.__init__, scene/cameras.py:55-58 (CPU instead of .cuda())
        wvt = torch.tensor(gu.getWorld2View2(Q, t, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = gu.getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        cams.append(dict(R=Q, t=t, fovx=fovx, fovy=fovy, W=W, H=H, world_view_transform=wvt.numpy(),
                         full_proj_transform=full.numpy(), camera_center=center.numpy()))
    np.savez(os.path.join(HERE, "camera.npz"), **{f"{k}_{i}": np.asarray(v) for i, c in enumerate(cams) for k, v in c.items()})

    # ---- SH ---------------------------------------------------------------------------------
    P = 257
    sh = rng.normal(0, 0.5, (P, 16, 3)).astype(np.float32)
    dirs = rng.normal(size=(P, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out = {"sh": sh, "dirs": dirs}
    for deg in range(4):
        # render(): shs_view = features.transpose(1,2) -> [P,3,16]; colours = clamp_min(eval_sh + 0.5, 0)
        res = shu.eval_sh(deg, torch.tensor(sh).transpose(1, 2), torch.tensor(dirs))
        out[f"rgb_deg{deg}"] = torch.clamp_min(res + 0.5, 0.0).numpy()
        out[f"raw_deg{deg}"] = res.numpy()
    np.savez(os.path.join(HERE, "sh_eval.npz"), **out)

    # ---- T (python compute_cov3D path) -------------------------------------------------------
    orig_zeros = torch.zeros
    genu.torch.zeros = lambda *a, **k: orig_zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"})
    try:
        P = 200
        xyz = rng.normal(0, 1.0, (P, 3)).astype(np.float32) + np.array([0, 0, 4], np.float32)
        scaling = np.exp(rng.uniform(math.log(0.01), math.log(0.3), (P, 2))).astype(np.float32)
        rot = rng.normal(size=(P, 4)).astype(np.float32)  # raw (un-normalised) as stored in _rotation
        mod = 0.8
        c = cams[0]
        W, H = c["W"], c["H"]
        s3 = torch.cat([torch.tensor(scaling) * mod, torch.ones(P, 1)], dim=-1)
        RS = genu.build_scaling_rotation(s3, torch.tensor(rot)).permute(0, 2, 1)  # gaussian_model.py:31
        trans = orig_zeros((P, 4, 4))
        trans[:, :3, :3] = RS
        trans[:, 3, :3] = torch.tensor(xyz)
        trans[:, 3, 3] = 1
        near, far = 0.01, 100.0
        ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, far - near, near],
                                [0, 0, 0, 1]]).float().T
        world2pix = torch.tensor(c["full_proj_transform"]) @ ndc2pix
        T = (trans[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)
    finally:
        genu.torch.zeros = orig_zeros
    np.savez(os.path.join(HERE, "transmat.npz"), xyz=xyz, scaling=scaling, rot=rot, mod=np.float32(mod), cam=0,
             T=T.numpy())

    # ---- losses / metric -----------------------------------------------------------------------
    a = torch.tensor(rng.uniform(0, 1, (3, 48, 64)).astype(np.float32))
    b = torch.clamp(a + torch.tensor(rng.normal(0, 0.1, (3, 48, 64)).astype(np.float32)), 0, 1)
    np.savez(os.path.join(HERE, "losses.npz"), a=a.numpy(), b=b.numpy(), l1=lu.l1_loss(a, b).numpy(),
             ssim=lu.ssim(a, b).numpy(), psnr=iu.psnr(a[None], b[None]).numpy())
    sys.path.remove(REF)


def oracle_part():
    from common import cotangents, run_oracle, scene_inputs
    from oracle import oracle as om
    inp = scene_inputs(P=600, W=96, H=64, seed=42, D=3, bg=(0.25, 0.5, 0.75), scale_mul=2.5)
    g = cotangents(64, 96, seed=5)
    o = run_oracle(om, inp, g)
    keep = {k: v for k, v in inp.items() if isinstance(v, np.ndarray)}
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), scalars=np.array([inp["scale_modifier"], inp["tanfovx"], inp["tanfovy"], inp["H"], inp["W"], inp["D"]], np.float64),
                        dL_dcolor=g[0], dL_dothers=g[1], R=np.int64(o["R"]), color=o["color"], others=o["others"],
                        radii=o["radii"], **{"in_" + k: v for k, v in keep.items()},
                        **{"grad_" + k: v for k, v in o["grads"].items()})


if __name__ == "__main__":
    reference_part()
    oracle_part()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
