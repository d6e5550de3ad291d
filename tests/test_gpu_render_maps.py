"""Fused render() map post-processing (HIP, include/g4s_render_maps.h) vs the plain-torch restatement of
2dgs/gaussian_renderer/__init__.py:117-164 + utils/point_utils.py:9-37 (oracle/render_maps_ref.py).

Tolerances: maps <= 1e-4 abs (BASELINE north_star's output bar), gradients <= 1e-3 of the largest
reference gradient per tensor."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from g4splat_amd import synthetic
from g4splat_amd.render_maps import render_maps
from oracle.render_maps_ref import render_maps as maps_ref

pytestmark = pytest.mark.gpu

NAMES = ("rend_alpha", "rend_normal", "rend_normal_cam", "rend_depth", "rend_dist", "surf_depth", "surf_normal",
         "surf_normal_cam")


def _camera(W, H, eye=(0.3, -0.2, -3.0), dev="cpu"):
    cam = synthetic.look_at_camera(eye, (0, 0, 0), (0, 1, 0), 1.0, W, H)
    return SimpleNamespace(image_width=W, image_height=H,
                           world_view_transform=torch.tensor(cam.world_view_transform, device=dev),
                           full_proj_transform=torch.tensor(cam.full_proj_transform, device=dev))


def _allmap(W, H, seed, holes=True):
    """A plausible allmap: smooth depth field with steps, alpha in (0,1], unnormalised normals."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    alpha = np.clip(rng.uniform(0.2, 1.0, (H, W)) + 0.1 * np.sin(xx / 7), 0.05, 1.0).astype(np.float32)
    depth = (2.5 + 0.4 * np.sin(xx / 9.0) * np.cos(yy / 5.0) + 0.05 * rng.normal(size=(H, W))).astype(np.float32)
    depth[yy > 0.7 * H] += 1.0  # a depth discontinuity
    n = rng.normal(size=(3, H, W)).astype(np.float32) * alpha
    med = (depth + 0.02 * rng.normal(size=(H, W))).astype(np.float32)
    dist = rng.uniform(0, 0.1, (H, W)).astype(np.float32)
    am = np.stack([depth * alpha, alpha, n[0], n[1], n[2], med, dist]).astype(np.float32)
    if holes and H > 4 and W > 4:
        am[1, 2, 3] = 0.0; am[0, 2, 3] = 0.0          # 0/0 -> nan -> 0
        am[1, 3, 1] = 0.0; am[0, 3, 1] = 0.5          # x/0 -> +inf -> 0
        am[5, 1, 2] = np.nan                           # nan median
        am[5, 4, 4] = np.inf
    return am


def _run_both(W, H, seed, ratio, grad_seed=1, holes=True):
    am = _allmap(W, H, seed, holes)
    ref_in = torch.tensor(am, requires_grad=True)
    ref = maps_ref(ref_in, _camera(W, H), ratio)
    hip_in = torch.tensor(am, device="cuda:0", requires_grad=True)
    out = render_maps(hip_in, _camera(W, H, dev="cuda:0"), ratio)
    g = torch.Generator().manual_seed(grad_seed)
    cots = {k: torch.randn(ref[k].shape, generator=g) for k in NAMES}
    sum((ref[k] * cots[k]).sum() for k in NAMES).backward()
    sum((out[k] * cots[k].to("cuda:0")).sum() for k in NAMES).backward()
    torch.cuda.synchronize()
    return ref, out, ref_in.grad, hip_in.grad.cpu()


@pytest.mark.parametrize("W,H,ratio", [(96, 64, 0.0), (96, 64, 1.0), (133, 77, 0.37), (64, 4, 0.5), (5, 9, 0.25)])
def test_maps_forward_backward_match_reference(hip_lib, W, H, ratio):
    ref, out, g_ref, g_hip = _run_both(W, H, seed=W + H, ratio=ratio)
    for k in NAMES:
        a, b = out[k].detach().cpu(), ref[k].detach()
        assert a.shape == b.shape, k
        assert (a - b).abs().max() <= 1e-4, (k, float((a - b).abs().max()))
    # alpha == 0 pixels: torch's division backward yields NaN (0/0) for channels 0 and 1; the HIP op reproduces it
    assert torch.equal(torch.isnan(g_hip), torch.isnan(g_ref))
    assert int(torch.isnan(g_ref).sum()) == (4 if min(W, H) > 4 else 0)
    g_hip, g_ref = torch.nan_to_num(g_hip, 0.0), torch.nan_to_num(g_ref, 0.0)
    for ch in range(7):
        d = (g_hip[ch] - g_ref[ch]).abs().max()
        assert d <= 1e-3 * g_ref[ch].abs().max() + 1e-6, (ch, float(d), float(g_ref[ch].abs().max()))


@pytest.mark.parametrize("case", [0, 1, 2])
def test_maps_match_the_reference_render_tail(hip_lib, case):
    """The HIP kernels against tests/golden/render_tail.npz: outputs and d/d(allmap) of the REFERENCE's own render()
    tail + utils/point_utils.py (run on the CPU of the build container with a stand-in rasterizer,
    tests/golden/make_golden_maps.py) -- tilted, off-origin cameras, alpha with exact zeros and ones, depth_ratio 0 /
    1 / 0.35.  Bars: 1e-5 abs on the maps (contract 1e-4), 1e-4 of the largest gradient per plane (contract 1e-3)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "render_tail.npz"))
    W, H, ratio = z[f"c{case}_meta"]
    cam = SimpleNamespace(image_width=int(W), image_height=int(H),
                          world_view_transform=torch.tensor(z[f"c{case}_wvt"], device="cuda:0"),
                          full_proj_transform=torch.tensor(z[f"c{case}_fpt"], device="cuda:0"))
    am = torch.tensor(z[f"c{case}_allmap"], device="cuda:0", requires_grad=True)
    out = render_maps(am, cam, float(ratio))
    for m in NAMES:
        d = np.abs(out[m].detach().cpu().numpy() - z[f"c{case}_{m}"]).max()
        assert d <= 1e-5, (m, float(d))
    sum((out[m] * torch.tensor(z[f"c{case}_cot_{m}"], device="cuda:0")).sum() for m in NAMES).backward()
    got, want = am.grad.cpu().numpy(), z[f"c{case}_dL_dallmap"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    got, want = np.nan_to_num(got), np.nan_to_num(want)
    for ch in range(7):
        assert np.abs(got[ch] - want[ch]).max() <= 1e-4 * np.abs(want[ch]).max() + 1e-7, ch


@pytest.mark.parametrize("W,H", [(1, 1), (2, 2), (3, 3), (2, 7)])
def test_maps_degenerate_sizes(hip_lib, W, H):
    """No interior pixel (or exactly one): surf_normal is all-zero border / a single normal."""
    ref, out, g_ref, g_hip = _run_both(W, H, seed=5, ratio=0.5, holes=False)
    for k in NAMES:
        assert (out[k].detach().cpu() - ref[k].detach()).abs().max() <= 1e-4, k
    assert (g_hip - g_ref).abs().max() <= 1e-3 * g_ref.abs().max() + 1e-6


def test_maps_partial_gradients_and_determinism(hip_lib):
    """Only some outputs used (the other cotangents arrive as None); same inputs -> bitwise same gradients."""
    W, H = 80, 56
    am = _allmap(W, H, 3)
    cam_c, cam_g = _camera(W, H), _camera(W, H, dev="cuda:0")
    ref_in = torch.tensor(am, requires_grad=True)
    ref = maps_ref(ref_in, cam_c, 0.5)
    (1 - (ref["rend_normal"] * ref["surf_normal"]).sum(0)).mean().backward()
    grads = []
    for _ in range(2):
        hip_in = torch.tensor(am, device="cuda:0", requires_grad=True)
        out = render_maps(hip_in, cam_g, 0.5)
        (1 - (out["rend_normal"] * out["surf_normal"]).sum(0)).mean().backward()
        grads.append(hip_in.grad.cpu())
    assert torch.equal(torch.nan_to_num(grads[0], 7.0), torch.nan_to_num(grads[1], 7.0))
    assert torch.equal(torch.isnan(grads[0]), torch.isnan(ref_in.grad))
    grads[0], ref_in.grad.data = torch.nan_to_num(grads[0], 0.0), torch.nan_to_num(ref_in.grad, 0.0)
    assert (grads[0] - ref_in.grad).abs().max() <= 1e-3 * ref_in.grad.abs().max() + 1e-9


def test_maps_errors(hip_lib):
    cam = _camera(8, 8, dev="cuda:0")
    with pytest.raises(RuntimeError):
        render_maps(torch.zeros((7, 8, 8)), cam, 0.5)           # host tensor: no CPU path
    with pytest.raises(RuntimeError):
        render_maps(torch.zeros((6, 8, 8), device="cuda:0"), cam, 0.5)


def test_maps_speed_vs_eager_torch(hip_lib, capsys):
    """1600x1200 (the metric's resolution): the fused op must beat the reference's eager torch formulation on the
    same GPU; prints both and the HBM fraction of the fused kernels (92 B/pixel forward, 108 B/pixel backward)."""
    import ctypes
    import json
    import time
    W, H = 1600, 1200
    am = torch.tensor(_allmap(W, H, 11), device="cuda:0")
    cam = _camera(W, H, dev="cuda:0")
    cots = {k: torch.randn((c, H, W), device="cuda:0") for k, c in zip(NAMES, (1, 3, 3, 1, 1, 1, 3, 3))}

    def step(fn):
        x = am.clone().requires_grad_(True)
        out = fn(x, cam, 0.5)
        torch.autograd.backward([out[k] for k in NAMES], [cots[k] for k in NAMES])
        return x.grad

    def wall(fn, n=20):
        for _ in range(5):
            step(fn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(fn)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_ref = wall(maps_ref)
    hip_lib.g4s_profile_reset()
    hip_lib.g4s_profile_enable(1)
    t_hip = wall(render_maps)
    hip_lib.g4s_profile_enable(0)
    ker = {}
    for k in range(hip_lib.g4s_profile_kernels()):
        ms, cnt = ctypes.c_double(), ctypes.c_int()
        hip_lib.g4s_profile_read(k, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            ker[hip_lib.g4s_profile_name(k).decode()] = ms.value / cnt.value
    N = W * H
    rep = {"resolution": [W, H], "eager_torch_fwd_bwd_ms": round(t_ref, 3), "fused_fwd_bwd_ms": round(t_hip, 3),
           "maps_fwd_kernel_ms": round(ker["maps_fwd"], 4), "maps_bwd_kernel_ms": round(ker["maps_bwd"], 4),
           "maps_fwd_GBps": round(N * 92 / ker["maps_fwd"] / 1e6, 1), "maps_bwd_GBps": round(N * 108 / ker["maps_bwd"] / 1e6, 1)}
    with capsys.disabled():
        print("\nrender_maps timing:", json.dumps(rep))
    assert t_hip < t_ref
