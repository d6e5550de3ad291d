"""Fused render maps, photometric loss and geometry regularisers against their torch restatements over many odd
image sizes (down to 1 x 1, sizes below the SSIM window, non-multiples of every block size).
    python tools/maps_loss_sweep.py [count]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_render_maps import NAMES, _run_both  # noqa: E402
from g4splat_amd.losses import geometry_regularizers, photometric_loss  # noqa: E402
from oracle import losses_ref  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
bad = 0
for seed in range(count):
    rng = np.random.default_rng(seed)
    W, H = int(rng.choice([1, 2, 3, 5, 8, 11, 16, 17, 31, 63, 64, 65, 100, 257])), int(rng.choice([1, 2, 4, 7, 10, 12, 16, 33, 64, 90, 129]))
    ratio = float(rng.choice([0.0, 0.3, 1.0]))
    tag = f"seed {seed} {W}x{H}"
    try:
        ref, out, g_ref, g_hip = _run_both(W, H, seed, ratio, grad_seed=seed + 1)
        for k in NAMES:
            a, b = out[k].detach().cpu(), ref[k].detach()
            assert a.shape == b.shape, ("maps shape", k)
            assert torch.equal(torch.isnan(a), torch.isnan(b)), ("maps nan", k)
            assert (torch.nan_to_num(a - b)).abs().max() <= 1e-4, ("maps", k, float((torch.nan_to_num(a - b)).abs().max()))
        assert torch.equal(torch.isnan(g_hip), torch.isnan(g_ref)), "maps grad nan"
        scale = float(torch.nan_to_num(g_ref).abs().max()) + 1e-12
        assert float(torch.nan_to_num(g_hip - g_ref).abs().max()) <= 1e-3 * scale, "maps grad"
        # photometric loss
        g = torch.Generator().manual_seed(seed)
        img, gt = torch.rand((3, H, W), generator=g), torch.rand((3, H, W), generator=g)
        a = img.clone().to(dev).requires_grad_(True)
        b = img.clone().requires_grad_(True)
        la, l1a, sa = photometric_loss(a, gt.to(dev), 0.2)
        lb, l1b, sb = losses_ref.photometric_loss(b, gt, 0.2)
        la.backward(); lb.backward()
        assert abs(float(la.detach()) - float(lb.detach())) <= 2e-5 * max(1.0, abs(float(lb.detach()))), ("loss", float(la.detach()), float(lb.detach()))
        assert float((a.grad.cpu() - b.grad).abs().max()) <= 2e-5 * float(b.grad.abs().max()) + 1e-9, "loss grad"
        # regularisers
        rn, sn, rd = torch.randn((3, H, W), generator=g), torch.randn((3, H, W), generator=g), torch.rand((1, H, W), generator=g)
        ne, dm = geometry_regularizers(rn.to(dev), sn.to(dev), rd.to(dev))
        ne_t = float((1 - (rn.double() * sn.double()).sum(0)).mean()); dm_t = float(rd.double().mean())
        assert abs(float(ne) - ne_t) <= 1e-6 * max(1.0, abs(ne_t)) and abs(float(dm) - dm_t) <= 1e-6, "regularisers"
    except AssertionError as ex:
        bad += 1
        print("MISMATCH", tag, ex, flush=True)
print(f"done: {count} sizes, {bad} mismatches")
