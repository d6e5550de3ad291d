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


def _densify_state(P, dev, seed=11):
    """A room-scene model at size P with optimiser moments and densification statistics arranged so that ONE
    densify_and_prune clones, splits and prunes a material share of the rows (>= 5 % each)."""
    import types
    scene = synthetic.scene_room(P, seed=0)
    rng = np.random.default_rng(seed)
    m = GaussianModel(3)
    cols = np.clip(scene.shs[:, 0, :] * 0.28209479177387814 + 0.5, 0, 1).astype(np.float32)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    m.create_from_parameters(t(scene.means3D), t(scene.scales), t(scene.rotations), t(cols), 1.0)
    with torch.no_grad():
        op = np.clip(scene.opacities, 1e-4, 1 - 1e-4).astype(np.float32)
        op[rng.random(P) < 0.08] = 0.02            # 8 % below min_opacity = 0.05: pruned
        m._opacity.copy_(t(np.log(op / (1 - op))))
        sc = scene.scales.copy()
        big = rng.random(P) < 0.5                   # half the rows above percent_dense * extent = 0.06: split candidates
        sc[big] = rng.uniform(0.07, 0.09, (int(big.sum()), 2)).astype(np.float32)
        sc[~big] = np.minimum(sc[~big], 0.05)
        m._scaling.copy_(t(np.log(sc)))
        m._features_rest.copy_(t(scene.shs[:, 1:, :]))
    m.training_setup()
    # moments as after a few Adam steps, statistics as after a few views: ~14 % of the rows above the threshold
    for p_ in m.parameters():
        st = m.optimizer.state[p_]
        st["step"] = torch.tensor(3.0)
        st["exp_avg"] = t(rng.normal(0, 1e-3, tuple(p_.shape)).astype(np.float32))
        st["exp_avg_sq"] = t(rng.uniform(0, 1e-6, tuple(p_.shape)).astype(np.float32))
    m.denom = t(rng.integers(0, 4, (P, 1)).astype(np.float32))            # zeros included: NaN statistics (0 / 0)
    m.xyz_gradient_accum = t((rng.random((P, 1)) < 0.14).astype(np.float32) * 1e-3) * torch.clamp(m.denom, min=1.0)
    m.max_radii2D = t(rng.uniform(0, 30, P).astype(np.float32))           # some beyond the screen-size limit of 20
    return m


def _snapshot(m):
    out = {}
    for name, p_ in (("xyz", m._xyz), ("f_dc", m._features_dc), ("f_rest", m._features_rest), ("opacity", m._opacity),
                     ("scaling", m._scaling), ("rotation", m._rotation)):
        out[name] = p_.detach().cpu().numpy()
        st = m.optimizer.state[p_]
        out[name + "_m"] = st["exp_avg"].cpu().numpy()
        out[name + "_v"] = st["exp_avg_sq"].cpu().numpy()
    out["accum"], out["denom"], out["radii"] = m.xyz_gradient_accum.cpu().numpy(), m.denom.cpu().numpy(), m.max_radii2D.cpu().numpy()
    return out


def test_config3_densification_at_size_edits_a_material_share_of_the_rows(hip_lib, capsys, monkeypatch):
    """Verdict r2 item 8 (gaussian_model.py:583-647 at BASELINE config 3's size).  One densify_and_prune on the 1.5 M
    room scene in a state where clone, split and prune EACH take >= 5 % of the rows, on the GPU (FusedAdam state, HIP
    stream compaction):
      * the selections equal the reference's rules evaluated with plain torch mask arithmetic, the row count is
        P + clones + splits - pruned exactly, kept rows (parameters AND Adam moments) equal mask indexing bit for bit;
      * on a 200 000-row subset the same call runs on the CPU path of densify.py -- the one that replays the
        reference's own GaussianModel bit for bit (tests/test_densify.py) -- from the same state with the same normal
        draws: identical row counts and selections, tensors equal to float round-off."""
    real_normal = torch.normal
    draws = {}

    def normal_from_cpu(mean, std, generator=None):  # the GPU's generator differs from the CPU's: both sides take the CPU stream
        key = tuple(std.shape)
        if key not in draws:
            g = torch.Generator().manual_seed(1234)
            draws[key] = real_normal(mean=torch.zeros(std.shape), std=torch.ones(std.shape), generator=g)
        return draws[key].to(std.device) * std + mean

    monkeypatch.setattr(torch, "normal", normal_from_cpu)
    dev = torch.device("cuda:0")
    report = []
    for P in (1_500_000, 200_000):
        m = _densify_state(P, dev)
        before = _snapshot(m)
        with torch.no_grad():
            grads = m.xyz_gradient_accum / m.denom
            grads[grads.isnan()] = 0.0
            big = m.get_scaling.max(dim=1).values > 0.01 * 6.0
            hot = grads.squeeze(1) >= 0.0002
            n_clone = int((hot & ~big).sum())
            clone_idx = torch.nonzero(hot & ~big).squeeze(1).cpu().numpy()
            # after the clone pass the set is [P originals | clones]; the split looks at the first P statistics only
            n_split = int((hot & big).sum())
            m.densify_and_prune(0.0002, 0.05, 6.0, 20)
        after = _snapshot(m)
        n_after = after["xyz"].shape[0]
        grown = P + n_clone + n_split  # + clones, + 2 children - 1 parent per split
        pruned = grown - n_after
        report.append((P, n_clone, n_split, pruned, n_after))
        assert n_clone >= 0.05 * P and n_split >= 0.05 * P and pruned >= 0.05 * P, report[-1]
        for k in after:
            assert after[k].shape[0] == n_after and np.isfinite(after[k]).all(), k
        # the surviving ORIGINAL rows keep their order: they are the rows of `before` that are neither split nor pruned
        keep0 = ~(hot & big).cpu().numpy()
        op_b = 1 / (1 + np.exp(-before["opacity"][:, 0].astype(np.float64)))
        sc_b = np.exp(before["scaling"].astype(np.float64)).max(axis=1)
        # (max_radii2D was reset by the clone / split passes' densification_postfix, gaussian_model.py:579-581, so the
        # screen-size term of the prune rule never fires here -- in the reference neither)
        keep0 &= ~((op_b < 0.05) | (sc_b > 0.6))
        n0 = int(keep0.sum())
        # rows whose opacity sits within float round-off of the threshold may fall either way between exp() variants
        edge = np.abs(op_b - 0.05) < 1e-6
        if not edge.any():
            for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"):
                np.testing.assert_array_equal(after[k][:n0], before[k][keep0], err_msg=k)
                np.testing.assert_array_equal(after[k + "_m"][:n0], before[k + "_m"][keep0], err_msg=k)
                np.testing.assert_array_equal(after[k + "_v"][:n0], before[k + "_v"][keep0], err_msg=k)
        assert (after["accum"] == 0).all() and (after["denom"] == 0).all()  # statistics restart (gaussian_model.py:579-581)
        if P == 200_000:
            cpu = _densify_state(P, torch.device("cpu"))
            with torch.no_grad():
                cpu.densify_and_prune(0.0002, 0.05, 6.0, 20)
            want = _snapshot(cpu)
            for k in after:
                assert after[k].shape == want[k].shape, (k, after[k].shape, want[k].shape)
                scale = np.abs(want[k]).max() + 1e-12
                assert np.abs(after[k] - want[k]).max() <= 2e-5 * scale, (k, float(np.abs(after[k] - want[k]).max()))
        del m
    with capsys.disabled():
        for P, c, s_, pr, n in report:
            print(f"\nC3 densify_and_prune at P={P}: cloned {c} ({100 * c / P:.1f} %), split {s_} ({100 * s_ / P:.1f} %), "
                  f"pruned {pr} ({100 * pr / P:.1f} %) -> {n} rows")


def test_training_iteration_replayed_from_a_hip_graph_equals_the_eager_loop():
    """graphed.TrainStepGraph: render + fused loss + regularisers + backward + densification statistics + FusedAdam step
    (capturable: step counts and learning rates on the device) captured once and replayed, against the same iterations
    issued eagerly from Python -- three cameras in turn, the xyz learning rate changing every iteration.  The kernels and
    their order are the same, so parameters, Adam state and statistics must agree bit for bit; the capture's warm-up
    iterations must leave no trace."""
    from g4splat_amd.graphed import TrainStepGraph
    from g4splat_amd.losses import geometry_regularizers
    dev = torch.device("cuda", 0)
    cams = _cams(dev)[:3]
    g = torch.Generator(device=dev).manual_seed(3)
    gts = [torch.rand((3, H, W), device=dev, generator=g) for _ in cams]

    def body(out, gt):
        loss, _l1, _s = photometric_loss(out["render"], gt, 0.2)
        normal_mean, dist_mean = geometry_regularizers(out["rend_normal"], out["surf_normal"], out["rend_dist"])
        return loss + 0.05 * normal_mean + 100.0 * dist_mean

    def fresh():
        m = _model(0, dev, jitter=True)
        m.training_setup(capturable=True)
        return m

    pipe = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    iters = 9
    # eager
    a = fresh()
    losses_a = []
    for it in range(iters):
        a.update_learning_rate(it + 1)
        out = render(cams[it % 3], a, pipe, bg)
        loss = body(out, gts[it % 3])
        loss.backward()
        with torch.no_grad():
            a.add_densification_stats(out["viewspace_points"], out["visibility_filter"], out["radii"])
        a.optimizer.step()
        a.optimizer.zero_grad(set_to_none=True)
        losses_a.append(float(loss.detach()))
    # graph
    b = fresh()
    before = [p.detach().clone() for p in b.parameters()]
    step = TrainStepGraph(b, body, cams[0], (3, H, W), instance_capacity=200_000)
    for p, q in zip(b.parameters(), before):
        assert torch.equal(p, q)  # the warm-up left no trace
    assert all(float(st["step"]) == 0 for st in b.optimizer.state.values())
    losses_b = []
    for it in range(iters):
        b.update_learning_rate(it + 1)
        losses_b.append(float(step(cams[it % 3], gts[it % 3])))
    assert not step.overflowed() and step.replays == iters
    assert losses_a == losses_b, (losses_a, losses_b)
    for name, p, q in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), a.parameters(), b.parameters()):
        assert torch.equal(p, q), name
    for pa, pb in zip(a.parameters(), b.parameters()):
        sa, sb = a.optimizer.state[pa], b.optimizer.state[pb]
        assert float(sa["step"]) == float(sb["step"]) == iters
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum) and torch.equal(a.denom, b.denom)
    assert losses_a[-1] < losses_a[0]
    # densification changes P: the graph is captured again for the new set and carries on (the reference densifies every
    # 100 iterations, train_with_refine_depth.py:573-590); split children are drawn from torch's generator: seed both alike
    for m in (a, b):
        torch.manual_seed(7)
        with torch.no_grad():
            m.densify_and_prune(2e-6, 0.005, 3.0, 20)
    n_new = a.get_xyz.shape[0]
    assert n_new == b.get_xyz.shape[0] and n_new != P
    # the old graph holds the freed parameter tensors and the old P: a replay must be refused, not run
    with pytest.raises(RuntimeError, match="stale.*number of Gaussians"):
        step(cams[0], gts[0])
    step = TrainStepGraph(b, body, cams[0], (3, H, W), instance_capacity=300_000)
    deg = b.active_sh_degree
    b.active_sh_degree = deg - 1 if deg > 0 else deg + 1
    with pytest.raises(RuntimeError, match="stale.*active SH degree"):
        step(cams[0], gts[0])
    b.active_sh_degree = deg
    for it in range(iters, iters + 4):
        a.update_learning_rate(it + 1)
        out = render(cams[it % 3], a, pipe, bg)
        loss = body(out, gts[it % 3])
        loss.backward()
        with torch.no_grad():
            a.add_densification_stats(out["viewspace_points"], out["visibility_filter"], out["radii"])
        a.optimizer.step()
        a.optimizer.zero_grad(set_to_none=True)
        b.update_learning_rate(it + 1)
        lb = step(cams[it % 3], gts[it % 3])
        assert float(loss.detach()) == float(lb), it
    for name, p, q in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), a.parameters(), b.parameters()):
        assert torch.equal(p, q), name
    assert all(float(a.optimizer.state[p]["step"]) == float(b.optimizer.state[q]["step"]) == iters + 4
               for p, q in zip(a.parameters(), b.parameters()))
