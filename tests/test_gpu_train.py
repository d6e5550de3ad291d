"""Training-level parity (last row): the same small synthetic scene is optimised for 40
iterations twice from the same initialisation --

  * on the MI355X with the product path (HIP rasterizer, fused render maps, fused photometric loss), and
  * on the CPU with the checkers (oracle rasterizer, torch restatements of the maps and the loss),

with the reference's loss (train_with_refine_depth.py:382-399: photometric + normal-consistency + distortion)
and the reference's Adam groups.  The two runs must stay together: loss curves within 2 % of each other at
every iteration, final renders > 45 dB PSNR apart from each other, and the loss must actually go down."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import contextlib

from cpu_rasterizer import oracle_backend
from g4splat_amd import synthetic
from g4splat_amd.gaussian_model import GaussianModel
from g4splat_amd.gaussian_renderer import render
from g4splat_amd.losses import photometric_loss
from oracle import losses_ref

pytestmark = pytest.mark.gpu
W, H, P, ITERS = 96, 64, 400, 40


def _cams(dev):
    out = []
    for i in range(4):
        a = 0.6 * i - 0.9
        cam = synthetic.look_at_camera((2.6 * np.sin(a), 0.3 * (i - 1.5), -2.6 * np.cos(a)), (0, 0, 0), (0, 1, 0), 1.0, W, H)
        out.append(SimpleNamespace(image_width=W, image_height=H, FoVx=cam.FoVx, FoVy=cam.FoVy,
                                   world_view_transform=torch.tensor(cam.world_view_transform, device=dev),
                                   full_proj_transform=torch.tensor(cam.full_proj_transform, device=dev),
                                   camera_center=torch.tensor(cam.camera_center, device=dev), znear=0.01, zfar=100.0))
    return out


def _model(seed, dev, jitter):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-0.9, 0.9, (P, 3)).astype(np.float32)
    cols = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    sc = rng.uniform(0.06, 0.2, (P, 2)).astype(np.float32)
    rot = rng.normal(size=(P, 4)).astype(np.float32)
    if jitter:  # the trainee starts away from the scene that produced the targets
        j = np.random.default_rng(seed + 1)
        pts = pts + j.normal(0, 0.03, pts.shape).astype(np.float32)
        cols = np.clip(cols + j.normal(0, 0.2, cols.shape), 0, 1).astype(np.float32)
        sc = sc * j.uniform(0.7, 1.3, sc.shape).astype(np.float32)
    m = GaussianModel(sh_degree=3)
    t = lambda a: torch.tensor(a, device=dev)
    m.create_from_parameters(t(pts), t(sc), t(rot), t(cols))
    with torch.no_grad():
        m._opacity += 2.5
    m.active_sh_degree = 1
    return m


def _train(dev, targets):
    hip = dev.type == "cuda"
    cams = _cams(dev)
    model = _model(0, dev, jitter=True)
    model.training_setup()
    pipe = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    bg = torch.tensor([0.0, 0.0, 0.0], device=dev)
    backend = contextlib.nullcontext if hip else oracle_backend  # CPU run: oracle rasterizer + torch maps
    losses = []
    for it in range(ITERS):
        cam, gt = cams[it % 4], targets[it % 4].to(dev)
        with backend():
            out = render(cam, model, pipe, bg)
        if hip:
            loss, _l1, _s = photometric_loss(out["render"], gt, 0.2)
        else:
            loss, _l1, _s = losses_ref.photometric_loss(out["render"], gt, 0.2)
        normal_error = (1 - (out["rend_normal"] * out["surf_normal"]).sum(dim=0))[None]
        total = loss + 0.05 * normal_error.mean() + 100.0 * out["rend_dist"].mean()  # lambda_normal, lambda_dist
        total.backward()
        with torch.no_grad():
            model.add_densification_stats(out["viewspace_points"], out["visibility_filter"])
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
        losses.append(float(total.detach()))
    with torch.no_grad():
        with backend():
            final = [render(c, model, pipe, bg)["render"].cpu() for c in cams]
    return np.array(losses), final, model


def test_hip_training_tracks_cpu_checker_training(hip_lib):
    cpu, gpu = torch.device("cpu"), torch.device("cuda:0")
    # targets: renders of the un-jittered scene by the CPU checker
    pipe = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    with torch.no_grad():
        truth = _model(0, cpu, jitter=False)
        with oracle_backend():
            targets = [render(c, truth, pipe, torch.zeros(3))["render"] for c in _cams(cpu)]
    l_cpu, f_cpu, m_cpu = _train(cpu, targets)
    l_gpu, f_gpu, m_gpu = _train(gpu, targets)
    assert l_cpu[-4:].mean() < 0.8 * l_cpu[:4].mean(), "the optimisation must make progress"
    assert np.abs(l_gpu - l_cpu).max() <= 0.02 * np.abs(l_cpu).max(), np.abs(l_gpu - l_cpu).max()
    for a, b in zip(f_gpu, f_cpu):
        mse = float(((a - b) ** 2).mean())
        assert 10 * np.log10(1.0 / max(mse, 1e-12)) > 45.0
    # parameters drifted apart by far less than they moved (on average: Adam with eps = 1e-15 turns a noise-level
    # gradient into a full +-lr step, so individual barely-constrained coordinates may take opposite signs)
    start = _model(0, cpu, True)._xyz
    moved = (m_cpu._xyz.detach() - start).abs().mean()
    drift = (m_gpu._xyz.detach().cpu() - m_cpu._xyz.detach()).abs().mean()
    assert drift <= 0.1 * moved, (float(drift), float(moved))
    assert torch.equal(m_gpu.denom.cpu(), m_cpu.denom)


def test_densify_on_gpu_then_keep_training(hip_lib):
    """prune / clone / split on HIP tensors with FusedAdam's state, followed by more iterations of the product path."""
    dev = torch.device("cuda:0")
    cams = _cams(dev)
    model = _model(0, dev, jitter=True)
    model.training_setup()
    from g4splat_amd.optim import FusedAdam
    assert isinstance(model.optimizer, FusedAdam)
    pipe = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    gt = torch.rand((3, H, W), device=dev)

    def iterate(n):
        last = None
        for it in range(n):
            out = render(cams[it % 4], model, pipe, bg)
            loss, _, _ = photometric_loss(out["render"], gt, 0.2)
            loss.backward()
            with torch.no_grad():
                model.max_radii2D = torch.maximum(model.max_radii2D, out["radii"].float())
                model.add_densification_stats(out["viewspace_points"], out["visibility_filter"])
                model.optimizer.step()
                model.optimizer.zero_grad(set_to_none=True)
            last = float(loss.detach())
        return last

    iterate(6)
    n0 = model.get_xyz.shape[0]
    torch.manual_seed(0)
    thr = float((model.xyz_gradient_accum / model.denom.clamp(min=1)).median())  # about half of the set qualifies
    model.densify_and_prune(thr, 0.005, extent=2.0, max_screen_size=None)
    n1 = model.get_xyz.shape[0]
    assert n1 > n0
    for p in model.parameters():
        assert p.shape[0] == n1 and p.is_cuda
        st = model.optimizer.state[p]
        assert st["exp_avg"].shape == p.shape and float(st["step"]) == 6
    model.reset_opacity()
    assert float(model.get_opacity.detach().max()) <= 0.0100001
    assert np.isfinite(iterate(6))
    assert float(model.optimizer.state[model._xyz]["step"]) == 12


def test_config3_densify_and_prune_schedule_at_metric_size(hip_lib, capsys):
    """BASELINE config 3: 1.5 M surfels, 1600x1200, the densify + prune loop -- the schedule of
    arguments/__init__.py:90-94 (densify every 100 iterations between 500 and 15 000 of 30 000, opacity reset every
    3 000, SH degree up every 1 000) compressed 100:1 to 300 iterations: densify_and_prune at 50 / 100 / 150, one
    opacity reset at 125, SH degree raised every 10 iterations.  Product path only (render(), fused loss and
    regularisers, FusedAdam, fused statistics, HIP compaction in prune / clone / split): a displaced, recoloured copy
    of the room scene is re-fitted to renders of the original from eight views.  Checks: the loss falls, every
    parameter and moment stays finite, the set really is edited (and the optimiser state follows it), PSNR improves."""
    import time
    from g4splat_amd.losses import geometry_regularizers
    dev = torch.device("cuda:0")
    P, W_, H_ = 1_500_000, 1600, 1200
    truth_scene = synthetic.scene_room(P, seed=0)
    cams_np = synthetic.room_cameras(8, W_, H_, fovx_deg=90.0)
    cams = [SimpleNamespace(image_width=W_, image_height=H_, FoVx=c.FoVx, FoVy=c.FoVy, znear=0.01, zfar=100.0,
                            world_view_transform=torch.tensor(c.world_view_transform, device=dev),
                            full_proj_transform=torch.tensor(c.full_proj_transform, device=dev),
                            camera_center=torch.tensor(c.camera_center, device=dev)) for c in cams_np]

    def model_from(scene, jitter):
        rng = np.random.default_rng(5)
        m = GaussianModel(3)
        xyz = scene.means3D + (rng.normal(0, 0.004, scene.means3D.shape).astype(np.float32) if jitter else 0)
        cols = np.clip(scene.shs[:, 0, :] * 0.28209479177387814 + 0.5, 0, 1)
        if jitter:
            cols = 0.5 * cols + 0.25
        m.create_from_parameters(torch.tensor(xyz, device=dev), torch.tensor(scene.scales, device=dev),
                                 torch.tensor(scene.rotations, device=dev), torch.tensor(cols.astype(np.float32), device=dev), 1.0)
        with torch.no_grad():
            op = np.clip(scene.opacities, 1e-4, 1 - 1e-4)
            m._opacity.copy_(torch.tensor(np.log(op / (1 - op)), device=dev))
            if not jitter:
                m._features_rest.copy_(torch.tensor(scene.shs[:, 1:, :], device=dev))
        return m

    pipe = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    truth = model_from(truth_scene, False)
    truth.active_sh_degree = 3
    with torch.no_grad():
        targets = [render(c, truth, pipe, bg)["render"].clone() for c in cams]
    del truth
    model = model_from(truth_scene, True)
    model.training_setup()
    psnr = lambda a, b: float(10 * torch.log10(1.0 / ((a - b) ** 2).mean()))
    with torch.no_grad():
        psnr0 = np.mean([psnr(render(c, model, pipe, bg)["render"], t) for c, t in zip(cams, targets)])
    losses, sizes = [], [int(model.get_xyz.shape[0])]
    torch.manual_seed(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(1, 301):
        model.update_learning_rate(it * 100)
        if it % 10 == 0:
            model.oneupSHdegree()
        cam, gt = cams[it % 8], targets[it % 8]
        out = render(cam, model, pipe, bg)
        loss, _l1, _s = photometric_loss(out["render"], gt, 0.2)
        nrm, dist_l = geometry_regularizers(out["rend_normal"], out["surf_normal"], out["rend_dist"])
        total = loss + (0.05 * nrm if it > 70 else 0.0) + (100.0 * dist_l if it > 30 else 0.0)  # lambda_normal / lambda_dist
        total.backward()
        with torch.no_grad():
            if it < 150:
                model.add_densification_stats(out["viewspace_points"], out["visibility_filter"], out["radii"])
                if it in (50, 100):
                    model.densify_and_prune(0.0002, 0.05, 6.0, 20 if it > 30 else None)
                    sizes.append(int(model.get_xyz.shape[0]))
                if it == 125:
                    model.reset_opacity()
            model.optimizer.step()
            model.optimizer.zero_grad(set_to_none=True)
        if it % 10 == 0:
            losses.append(float(total.detach()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = int(model.get_xyz.shape[0])
    for p in model.parameters():
        assert p.shape[0] == n and bool(torch.isfinite(p).all())
        st = model.optimizer.state[p]
        assert st["exp_avg"].shape == p.shape and bool(torch.isfinite(st["exp_avg"]).all()) and bool(torch.isfinite(st["exp_avg_sq"]).all())
    assert model.xyz_gradient_accum.shape[0] == n and model.max_radii2D.shape[0] == n
    assert len(set(sizes)) > 1, sizes  # the set was edited
    with torch.no_grad():
        psnr1 = np.mean([psnr(render(c, model, pipe, bg)["render"], t) for c, t in zip(cams, targets)])
    with capsys.disabled():
        print(f"\nC3 schedule at 1.5 M: sizes {sizes} -> {n}, loss {losses[0]:.4f} -> {losses[-1]:.4f}, PSNR {psnr0:.2f} -> {psnr1:.2f} dB, "
              f"{dt / 300 * 1e3:.2f} ms per iteration incl. densification")
    assert psnr1 > psnr0 + 1.0, (psnr0, psnr1)
