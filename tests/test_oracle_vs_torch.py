"""Cross-validation of the C oracle against the independent pure-PyTorch autograd renderer
(oracle/torch_ref.py): forward outputs, and every gradient of the explicit backward that is a
true gradient.  Gaussians whose low-pass (2-D filter) branch fired are excluded
from the position/shape comparison: there the reference deliberately uses a different centre
derivative (backward.cu:513-541), asserted separately in test_oracle_quirks.py."""
import numpy as np
import pytest
import torch

from common import EMPTY, cotangents, rel_err
from g4splat_amd import synthetic
from oracle import torch_ref


def _scene(P, W, H, seed, faceon):
    scene, cam = synthetic.scene_random(P, seed=seed, width=W, height=H, opacity_max=0.95)
    rng = np.random.default_rng(seed + 100)
    if faceon:
        q = np.concatenate([np.ones((P, 1)), rng.normal(0, 0.15, (P, 3))], 1)
        rot = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
        scales = rng.uniform(0.08, 0.3, (P, 2)).astype(np.float32)
    else:
        rot, scales = scene.rotations, (scene.scales * 3).astype(np.float32)
    return scene, cam, rot, scales


@pytest.mark.parametrize("faceon,seed", [(True, 5), (False, 3)])
def test_forward_and_true_gradients(oracle_mod, faceon, seed):
    P, W, H, D = 200, 64, 48, 3
    scene, cam, rot, scales = _scene(P, W, H, seed, faceon)
    bg = np.array([0.2, 0.5, 0.7], np.float32)
    o = oracle_mod.Oracle()
    R, color, others, radii = o.rasterize_gaussians(bg, scene.means3D, EMPTY, scene.opacities, scales, rot, 1.0, EMPTY,
                                                    cam.world_view_transform, cam.full_proj_transform, cam.tanfovx,
                                                    cam.tanfovy, H, W, scene.shs, D, cam.camera_center)
    dt = torch.float64
    t = lambda a: torch.tensor(a, dtype=dt)
    m, s, q = t(scene.means3D).requires_grad_(), t(scales).requires_grad_(), t(rot).requires_grad_()
    op, sh = t(scene.opacities).requires_grad_(), t(scene.shs).requires_grad_()
    cols = torch_ref.sh_to_rgb(D, sh, m, t(cam.camera_center))
    c2, o2, r2, aux = torch_ref.render(m, s, q, op, cols, t(cam.world_view_transform), t(cam.full_proj_transform), W, H,
                                       t(bg))
    np.testing.assert_array_equal(r2.numpy(), radii)
    assert np.abs(c2.detach().numpy() - color).max() < 2e-5
    assert np.abs(o2.detach().numpy() - others).max() < 1e-4
    gc, go = cotangents(H, W, seed=seed)
    g = o.rasterize_gaussians_backward(gc, go)
    ((c2 * t(gc)).sum() + (o2 * t(go)).sum()).backward()
    lowpass = np.abs(o.state("dL_dmean2D_raw")).sum(1) > 0
    keep = ~lowpass
    assert keep.sum() > P // 3
    assert rel_err(g["opacity"], op.grad.numpy()) < 2e-5
    assert rel_err(g["sh"], sh.grad.numpy()) < 2e-5
    assert rel_err(g["means3D"][keep], m.grad.numpy()[keep]) < 5e-5
    assert rel_err(g["scales"][keep], s.grad.numpy()[keep]) < 5e-5
    assert rel_err(g["rotations"][keep], q.grad.numpy()[keep]) < 5e-5


def test_precomputed_colour_gradient(oracle_mod):
    P, W, H = 150, 48, 48
    scene, cam, rot, scales = _scene(P, W, H, 9, True)
    rng = np.random.default_rng(1)
    cols_np = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    bg = np.zeros(3, np.float32)
    o = oracle_mod.Oracle()
    o.rasterize_gaussians(bg, scene.means3D, cols_np, scene.opacities, scales, rot, 1.0, EMPTY,
                          cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, EMPTY, 0,
                          cam.camera_center)
    gc, go = cotangents(H, W, seed=2)
    g = o.rasterize_gaussians_backward(gc, go * 0)
    dt = torch.float64
    t = lambda a: torch.tensor(a, dtype=dt)
    cols = t(cols_np).requires_grad_()
    c2, o2, _, _ = torch_ref.render(t(scene.means3D), t(scales), t(rot), t(scene.opacities), cols,
                                    t(cam.world_view_transform), t(cam.full_proj_transform), W, H, t(bg))
    (c2 * t(gc)).sum().backward()
    assert rel_err(g["colors"], cols.grad.numpy()) < 2e-5
