"""Backward time of a workload as a function of the list depth above which a tile goes to the four-wave kernel
(option "bwd_hot_threshold"; default: max(4 x average list, 2048)).  One wave per tile leaves a tail when a few tiles are
much deeper than the rest: this sweep says whether handing more of them to the four-wave kernel pays.
    python tools/hot_threshold_sweep.py [s3|s2|s5] [view ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from common import hip_state  # noqa: E402
from g4splat_amd import _lib  # noqa: E402
from g4splat_amd.diff_surfel_rasterization import _C  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "s3"
views = [int(v) for v in sys.argv[2:]] or [0, 2]
dev = torch.device("cuda", 0)
scene, cams, d, dcams, (P, W, H, D) = bench.build_scene(wl, dev)
e, bg = torch.empty(0, device=dev), torch.zeros(3, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
gc, go = torch.randn((3, H, W), device=dev, generator=g), torch.randn((7, H, W), device=dev, generator=g)
for v in views:
    c = dcams[v]
    f = _C.rasterize_gaussians(bg, d["means3D"], e, d["opacity"], d["scales"], d["rotations"], 1.0, e, c["view"], c["proj"],
                               c["tanfovx"], c["tanfovy"], H, W, d["sh"], D, c["campos"], False, False)
    st = hip_state(dict(R=f[0], geom=f[4], binning=f[5], img=f[6]), dict(means3D=scene.means3D, W=W, H=H))
    last = st["n_contrib"][0].reshape(H, W)
    th, tw = (H + 15) // 16, (W + 15) // 16
    pad = np.zeros((th * 16, tw * 16), last.dtype)
    pad[:H, :W] = last
    n_live = pad.reshape(th, 16, tw, 16).max(axis=(1, 3)).reshape(-1)
    print(f"{wl} view {v}: {len(n_live)} tiles, n_live mean {n_live.mean():.0f} p50 {np.median(n_live):.0f} p90 {np.percentile(n_live, 90):.0f} "
          f"p99 {np.percentile(n_live, 99):.0f} max {n_live.max()}")

    def bwd():
        return _C.rasterize_gaussians_backward(bg, d["means3D"], f[3], e, d["scales"], d["rotations"], 1.0, e, c["view"], c["proj"],
                                               c["tanfovx"], c["tanfovy"], gc, go, d["sh"], D, c["campos"], f[4], f[0], f[5], f[6], False)

    for thr in (None, 3000, 2000, 1500, 1200, 1000, 800, 600, 400, -1):
        ctx = _lib.option("bwd_hot_threshold", thr) if thr is not None else None
        if ctx:
            ctx.__enter__()
        for _ in range(3):
            bwd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            bwd()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        if ctx:
            ctx.__exit__(None, None, None)
        hot = len(n_live) if thr == -1 else int((n_live > (thr if thr is not None else max(4 * int(f[0]) // len(n_live), 2048))).sum())
        print(f"   threshold {str(thr):>5s}: backward {ms:.3f} ms  ({hot} tiles to the four-wave kernel)")
