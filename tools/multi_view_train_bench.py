"""A multi-view training step on one MI355X: V views per optimiser step (gradient accumulation; SURVEY.md 8(e)'s 8 views over
fewer than 8 GPUs), every view through the reference's own call sequence -- gaussian_renderer.render(), the fused L1+SSIM
loss and the regularisers, loss.backward() -- then ONE FusedAdam step.  Timed twice: the views one after the other on the
default stream, and each view inside a ViewPipeline slot (k streams; the autograd node picks the slot's PresizedState up by
itself).  Prints ms per view for both and checks that the parameters after the timed steps agree.

    python tools/multi_view_train_bench.py [--views 4] [--k 2] [--steps 10] [--P 1500000 --width 1600 --height 1200]"""
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
from g4splat_amd.diff_surfel_rasterization import _C  # noqa: E402
from g4splat_amd.gaussian_model import GaussianModel  # noqa: E402
from g4splat_amd.gaussian_renderer import render  # noqa: E402
from g4splat_amd.losses import geometry_regularizers, photometric_loss  # noqa: E402
from g4splat_amd.pipeline import ViewPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--P", type=int, default=1_500_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)  # intended: one leaf, several forward streams
    scene = synthetic.scene_room(a.P, seed=0)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    cams = []
    for c in synthetic.room_cameras(8, a.width, a.height, fovx_deg=90.0):
        cams.append(SimpleNamespace(image_width=a.width, image_height=a.height, FoVx=2 * math.atan(c.tanfovx),
                                    FoVy=2 * math.atan(c.tanfovy), world_view_transform=t(c.world_view_transform),
                                    full_proj_transform=t(c.full_proj_transform), camera_center=t(c.camera_center),
                                    znear=0.01, zfar=100.0))
    gts = [torch.rand((3, a.height, a.width), device=dev) for _ in cams]
    cfg = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)

    def fresh_model():
        m = GaussianModel(sh_degree=3)
        g = torch.Generator(device=dev).manual_seed(0)
        m.create_from_parameters(t(scene.means3D), t(scene.scales), t(scene.rotations), torch.rand((a.P, 3), device=dev, generator=g))
        with torch.no_grad():
            m._opacity.copy_(torch.logit(t(scene.opacities).clamp(1e-4, 1 - 1e-4)))
            m._features_rest.copy_(t(scene.shs[:, 1:, :]))
        m.active_sh_degree = 3
        m.training_setup(fused=True)
        for grp in m.optimizer.param_groups:
            for p in grp["params"]:
                p.grad = torch.zeros_like(p)  # persistent gradient buffers (zero_grad(set_to_none=False) below)
        return m

    def one_view(m, v):
        out = render(cams[v % 8], m, cfg, bg)
        loss, _l1, _s = photometric_loss(out["render"], gts[v % 8], 0.2)
        normal_mean, dist_mean = geometry_regularizers(out["rend_normal"], out["surf_normal"], out["rend_dist"])
        (loss + 0.05 * normal_mean + 100.0 * dist_mean).backward()
        if pipe_ref[0] is not None:
            pipe_ref[0].after_previous_view()  # the statistics below are shared by the views
        with torch.no_grad():
            m.add_densification_stats(out["viewspace_points"], out["visibility_filter"], out["radii"])

    pipe_ref = [None]

    def step(m, s, pipe):
        pipe_ref[0] = pipe
        for j in range(a.views):
            v = s * a.views + j
            if pipe is None:
                one_view(m, v)
            else:
                with pipe.slot(j):
                    one_view(m, v)
        if pipe is not None:
            pipe.join()
        with torch.no_grad():
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=False)

    def timed(pipe):
        m = fresh_model()
        if pipe is not None:
            pipe.release_hooks()
            pipe.order_accumulation(m.parameters())
        for s in range(2):
            step(m, s, pipe)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(2, 2 + a.steps):
            step(m, s, pipe)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (a.steps * a.views) * 1e3
        return ms, [p.detach().clone() for p in (m._xyz, m._features_dc, m._scaling, m._rotation, m._opacity,
                                                  m.xyz_gradient_accum, m.denom)]

    # instance capacity of a slot: the largest count over the views, with headroom
    m0 = fresh_model()
    R, empty = 0, torch.empty(0, device=dev)
    with torch.no_grad():
        for cam in cams:
            fw = _C.rasterize_gaussians(bg, m0.get_xyz, empty, m0.get_opacity, m0.get_scaling, m0.get_rotation, 1.0, empty,
                                        cam.world_view_transform, cam.full_proj_transform, math.tan(cam.FoVx * 0.5),
                                        math.tan(cam.FoVy * 0.5), a.height, a.width, m0.get_features, 3, cam.camera_center,
                                        False, False)
            R = max(R, int(fw[0]))
    del m0, fw
    ms_seq, p_seq = timed(None)
    _ms, p_seq2 = timed(None)  # the same run again: how far two identical runs drift apart (Adam amplifies last-bit differences)
    pipe = ViewPipeline(a.P, a.width, a.height, int(R * 1.3), dev, k=a.k)
    ms_pipe, p_pipe = timed(pipe)
    assert not pipe.overflowed()
    for _ in range(3):  # the ordering must hold every time, not most of the time
        _ms2, p_again = timed(pipe)
        assert all(torch.equal(x, y) for x, y in zip(p_pipe, p_again)), "two pipelined runs differ"
    rel = lambda A, B: max(float((x - y).abs().max() / (x.abs().max() + 1e-30)) for x, y in zip(A, B))
    drift, drift_same = rel(p_seq, p_pipe), rel(p_seq, p_seq2)
    print(json.dumps({"P": a.P, "resolution": [a.width, a.height], "views_per_step": a.views, "steps": a.steps,
                      "ms_per_view_sequential": round(ms_seq, 3), "ms_per_view_in_slots": round(ms_pipe, 3), "streams": a.k,
                      "speedup": round(ms_seq / ms_pipe, 3),
                      "parameters_after_the_steps_max_rel_diff": drift,
                      "same_between_two_sequential_runs": drift_same}))


if __name__ == "__main__":
    main()
