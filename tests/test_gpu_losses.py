"""Fused photometric loss (HIP, include/g4s_losses.h) against golden vectors produced by the reference's own
loss_utils.py (tests/golden/photometric.npz -- pinned parity) and against the torch restatement at the
metric's resolution.  Tolerances: scalars 1e-5 abs, gradient 1e-3 of its largest entry (measured ~1e-6)."""
import os

import numpy as np
import pytest
import torch

from g4splat_amd.losses import photometric_loss
from oracle import losses_ref

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["ragged", "tile", "tiny", "zeros"])
def test_golden_vectors_of_the_reference(hip_lib, name):
    g = np.load(os.path.join(G, "photometric.npz"))
    x = torch.tensor(g[f"{name}_image"], device="cuda:0", requires_grad=True)
    y = torch.tensor(g[f"{name}_gt"], device="cuda:0")
    loss, l1, ss = photometric_loss(x, y, float(g[f"{name}_lambda"]))
    loss.backward()
    assert abs(float(l1) - float(g[f"{name}_l1"])) <= 1e-6
    assert abs(float(ss) - float(g[f"{name}_ssim"])) <= 1e-5
    assert abs(float(loss.detach()) - float(g[f"{name}_loss"])) <= 1e-5
    want = g[f"{name}_grad"]
    got = x.grad.cpu().numpy()
    assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-9  # what it actually achieves
    assert not l1.requires_grad and not ss.requires_grad


def test_metrics_ssim_is_the_fused_kernel_and_matches_the_reference(hip_lib):
    """g4splat_amd.metrics.ssim (the reference's name, loss_utils.py:46-79) == the reference's value (golden), with a
    gradient; psnr / l1_loss follow their inputs to the device."""
    from g4splat_amd import metrics
    g = np.load(os.path.join(G, "losses.npz"))
    a = torch.tensor(g["a"], device="cuda:0", requires_grad=True)
    b = torch.tensor(g["b"], device="cuda:0")
    s = metrics.ssim(a, b)
    assert abs(float(s.detach()) - float(g["ssim"])) <= 1e-5
    s.backward()
    assert a.grad is not None and float(a.grad.abs().max()) > 0
    assert abs(float(metrics.ssim(a.detach()[None], b[None])) - float(g["ssim"])) <= 1e-5
    np.testing.assert_allclose(metrics.psnr(a.detach()[None], b[None]).cpu().numpy(), g["psnr"], rtol=1e-5)
    np.testing.assert_allclose(metrics.l1_loss(a.detach(), b).cpu().numpy(), g["l1"], rtol=1e-5)


def test_metric_resolution_vs_restatement_and_speed(hip_lib, capsys):
    import ctypes
    import json
    import time
    H, W = 1200, 1600
    g = torch.Generator(device="cuda:0").manual_seed(1)
    gt = torch.rand((3, H, W), device="cuda:0", generator=g)
    img = (gt + 0.1 * torch.randn((3, H, W), device="cuda:0", generator=g)).clamp(0, 1)
    x1 = img.clone().requires_grad_(True)
    loss, l1, ss = photometric_loss(x1, gt, 0.2)
    (3.0 * loss).backward()  # upstream factor reaches the gradient
    x2 = img.clone().requires_grad_(True)
    rl, r1, rs = losses_ref.photometric_loss(x2, gt, 0.2)  # eager torch on the same GPU
    (3.0 * rl).backward()
    assert abs(float(loss.detach()) - float(rl.detach())) <= 1e-5 and abs(float(l1) - float(r1.detach())) <= 1e-6 and abs(float(ss) - float(rs.detach())) <= 1e-5
    d = (x1.grad - x2.grad).abs().max()
    assert float(d) <= 1e-3 * float(x2.grad.abs().max())
    # bit-reproducible
    x3 = img.clone().requires_grad_(True)
    l3, _, _ = photometric_loss(x3, gt, 0.2)
    (3.0 * l3).backward()
    assert torch.equal(l3, loss) and torch.equal(x3.grad, x1.grad)

    def wall(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def hip_step():
        x = img.clone().requires_grad_(True)
        photometric_loss(x, gt, 0.2)[0].backward()

    def ref_step():
        x = img.clone().requires_grad_(True)
        losses_ref.photometric_loss(x, gt, 0.2)[0].backward()

    t_ref = wall(ref_step)
    hip_lib.g4s_profile_reset()
    hip_lib.g4s_profile_enable(1)
    t_hip = wall(hip_step)
    hip_lib.g4s_profile_enable(0)
    ker = {}
    for k in range(hip_lib.g4s_profile_kernels()):
        ms, cnt = ctypes.c_double(), ctypes.c_int()
        hip_lib.g4s_profile_read(k, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            ker[hip_lib.g4s_profile_name(k).decode()] = ms.value / cnt.value
    with capsys.disabled():
        print("\nphotometric loss timing:", json.dumps({"resolution": [W, H], "eager_torch_fwd_bwd_ms": round(t_ref, 3),
                                                         "fused_fwd_bwd_ms": round(t_hip, 3),
                                                         "fused_kernels_ms": round(ker["photometric_loss"], 4)}))
    assert t_hip < t_ref


def test_errors(hip_lib):
    a = torch.zeros((3, 8, 8), device="cuda:0")
    with pytest.raises(RuntimeError):
        photometric_loss(a.cpu(), a.cpu(), 0.2)
    with pytest.raises(RuntimeError):
        photometric_loss(a, torch.zeros((3, 8, 9), device="cuda:0"), 0.2)
    loss, l1, ss = photometric_loss(a, a, 0.2)  # identical images, no gradient requested
    assert float(l1) == 0.0 and abs(float(ss) - 1.0) <= 1e-6 and abs(float(loss)) <= 1e-6


@pytest.mark.parametrize("H,W", [(1200, 1600), (37, 53)])
def test_geometry_regularizers_match_torch(hip_lib, H, W):
    """g4s_geometry_regularizers_* against the reference's formulation (train_with_refine_depth.py:391-396)."""
    from g4splat_amd.losses import geometry_regularizers
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    mk = lambda *s: torch.randn(s, generator=g).to(dev)
    rn, sn, rd = mk(3, H, W), mk(3, H, W), mk(1, H, W).abs()
    leaves_a = [t.clone().requires_grad_(True) for t in (rn, sn, rd)]
    leaves_b = [t.clone().requires_grad_(True) for t in (rn, sn, rd)]
    ne, dm = geometry_regularizers(*leaves_a)
    (0.05 * ne + 100.0 * dm).backward()
    normal_error = (1 - (leaves_b[0] * leaves_b[1]).sum(dim=0))[None]
    ne_t, dm_t = normal_error.mean(), leaves_b[2].mean()
    (0.05 * ne_t + 100.0 * dm_t).backward()
    torch.cuda.synchronize()
    # the fused means are accumulated in double; torch's float32 tree sum is the looser of the two
    ref_ne = normal_error.double().mean().item()
    ref_dm = leaves_b[2].double().mean().item()
    assert abs(ne.item() - ref_ne) <= 2e-7 * max(1.0, abs(ref_ne)) and abs(ne.item() - ne_t.item()) <= 2e-5
    assert abs(dm.item() - ref_dm) <= 2e-7 * max(1.0, abs(ref_dm)) and abs(dm.item() - dm_t.item()) <= 2e-5
    for a, b in zip(leaves_a, leaves_b):
        assert a.grad.shape == b.grad.shape
        assert (a.grad - b.grad).abs().max() <= 3e-7 * b.grad.abs().max()
    # bit-reproducible
    ne2, dm2 = geometry_regularizers(rn, sn, rd)
    assert ne2.item() == ne.item() and dm2.item() == dm.item()
