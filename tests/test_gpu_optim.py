"""FusedAdam (HIP, include/g4s_optim.h) against torch.optim.Adam -- the optimiser the reference builds at
2dgs/scene/gaussian_model.py:248-266 (six groups, eps = 1e-15) -- on identical parameters and gradient sequences."""
import numpy as np
import pytest
import torch

from g4splat_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu

SHAPES = [(1003, 3), (1003, 1, 3), (1003, 15, 3), (1003, 1), (1003, 2), (1003, 4)]
LRS = [0.00016, 0.0025, 0.0025 / 20, 0.05, 0.005, 0.001]
NAMES = ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]


def _groups(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in SHAPES]
    return ps, [{"params": [p], "lr": lr, "name": n} for p, lr, n in zip(ps, LRS, NAMES)]


def test_matches_torch_adam_over_many_steps(hip_lib):
    pa, ga = _groups("cuda:0")
    pb, gb = _groups("cpu")
    a = FusedAdam(ga, lr=0.0, eps=1e-15)
    b = torch.optim.Adam(gb, lr=0.0, eps=1e-15)
    g = torch.Generator().manual_seed(9)
    for it in range(25):
        for x, y in zip(pa, pb):
            grad = torch.randn(y.shape, generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))
            if it % 5 == 3:
                grad[::3] = 0.0  # rows of invisible Gaussians: exact zero gradients still move through the moments
            y.grad = grad.clone()
            x.grad = grad.to("cuda:0")
        if it == 10:  # the reference's xyz learning-rate schedule edits param_groups in place (:268-274)
            a.param_groups[0]["lr"] = b.param_groups[0]["lr"] = 0.0001
        a.step()
        b.step()
    for x, y, n in zip(pa, pb, NAMES):
        d = (x.detach().cpu() - y.detach()).abs().max()
        moved = (y.detach() - _groups("cpu")[0][NAMES.index(n)].detach()).abs().max()
        # 2e-5 of the distance travelled, plus two float32 ulps of the parameter values themselves
        assert d <= 2e-5 * moved + 2.4e-7 * float(y.detach().abs().max()), (n, float(d), float(moved))
        sa, sb = a.state[x], b.state[y]
        assert float(sa["step"]) == float(sb["step"]) == 25
        # moments: relative to the tensor's scale (an element is a cancelling sum of gradients spanning 8 decades)
        for key in ("exp_avg", "exp_avg_sq"):
            want = sb[key]
            assert (sa[key].cpu() - want).abs().max() <= 1e-5 * want.abs().max(), (n, key)


def test_capturable_mode_equals_the_plain_mode(hip_lib):
    """capturable=True: step counts (float32 scalars, torch's convention) and learning rates live on the device
    (g4s_adam_step_device: the bias corrections are formed in double by a one-block kernel instead of by the host).  Same
    update, bit for bit, over 40 steps with a changing learning rate and with a parameter that sits out some steps -- its
    step count must not advance and nobody's slot in the learning-rate buffer may move."""
    dev = "cuda:0"
    pa, ga = _groups(dev)
    pb, gb = _groups(dev)
    a = FusedAdam(ga, lr=0.0, eps=1e-15)
    b = FusedAdam(gb, lr=0.0, eps=1e-15, capturable=True)
    g = torch.Generator().manual_seed(4)
    for it in range(40):
        for i, (x, y) in enumerate(zip(pa, pb)):
            if i == 2 and it % 4 == 1:
                x.grad = y.grad = None
                continue
            grad = (torch.randn(x.shape, generator=g) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=g)))).to(dev)
            x.grad, y.grad = grad.clone(), grad.clone()
        a.param_groups[0]["lr"] = b.param_groups[0]["lr"] = 0.00016 * (0.97 ** it)
        a.step()
        b.step()
    torch.cuda.synchronize()
    for x, y, n in zip(pa, pb, NAMES):
        assert torch.equal(x, y), n
        sa, sb = a.state[x], b.state[y]
        assert sb["step"].is_cuda and float(sa["step"]) == float(sb["step"]) == (30 if n == "f_rest" else 40)
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    # the state dict moves to torch.optim.Adam(capturable=True) and back
    c = torch.optim.Adam(_groups(dev)[1], lr=0.0, eps=1e-15, capturable=True)
    c.load_state_dict(b.state_dict())
    assert all(float(st["step"]) in (30.0, 40.0) for st in c.state.values())


def test_state_surgery_like_densification(hip_lib):
    """cat_tensors_to_optimizer (gaussian_model.py:528-549) replaces a parameter and extends its moments; the
    optimiser must carry on with the edited state.  Misaligned / odd-sized tensors take the scalar path."""
    dev = "cuda:0"
    p = torch.nn.Parameter(torch.randn(7, 3, device=dev))
    opt = FusedAdam([{"params": [p], "lr": 0.01, "name": "xyz"}], lr=0.0, eps=1e-15)
    p.grad = torch.ones_like(p)
    opt.step()
    st = opt.state.pop(p)
    ext = torch.zeros(5, 3, device=dev)
    st["exp_avg"] = torch.cat([st["exp_avg"], ext])
    st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"], ext])
    q = torch.nn.Parameter(torch.cat([p.detach(), torch.randn(5, 3, device=dev)]))
    opt.param_groups[0]["params"][0] = q
    opt.state[q] = st
    before = q.detach().clone()
    q.grad = torch.ones_like(q)
    opt.step()
    assert float(opt.state[q]["step"]) == 2 and (q.detach() != before).all()
    # a view with an odd element offset: 4-byte aligned only
    base = torch.randn(101, device=dev)
    r = torch.nn.Parameter(base[1:100])
    ref = torch.nn.Parameter(base[1:100].detach().cpu().clone())
    o1, o2 = FusedAdam([r], lr=0.1), torch.optim.Adam([ref], lr=0.1)
    for _ in range(3):
        r.grad = torch.full_like(r, 0.5)
        ref.grad = torch.full_like(ref, 0.5)
        o1.step()
        o2.step()
    assert torch.allclose(r.detach().cpu(), ref.detach(), rtol=1e-5, atol=1e-7)
    with pytest.raises(RuntimeError):
        h = torch.nn.Parameter(torch.zeros(3))
        h.grad = torch.zeros(3)
        FusedAdam([h]).step()


def test_speed_vs_torch_adam(hip_lib, capsys):
    """1.5 M Gaussians (58 floats each): one kernel vs torch.optim.Adam's foreach implementation on the same GPU."""
    import json
    import time
    dev = "cuda:0"
    P = 1_500_000
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 2), (P, 4)]

    def make(cls):
        ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
        for p in ps:
            p.grad = torch.randn_like(p)
        return cls([{"params": [p], "lr": lr} for p, lr in zip(ps, LRS)], lr=0.0, eps=1e-15)

    def wall(opt, n=20):
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_ref, t_hip = wall(make(torch.optim.Adam)), wall(make(FusedAdam))
    traffic = P * 58 * 4 * 7  # read p, g, m, v; write p, m, v
    with capsys.disabled():
        print("\nadam timing:", json.dumps({"P": P, "torch_foreach_ms": round(t_ref, 3), "fused_ms": round(t_hip, 3),
                                            "fused_GBps": round(traffic / t_hip / 1e6, 1)}))
    assert t_hip < t_ref


def test_fused_densification_stats_match_the_torch_formulation(hip_lib):
    """g4s_densify_stats against GaussianModel.add_densification_stats' host formulation
    (2dgs/scene/gaussian_model.py:649-651) + the training loop's max_radii2D update, over several views."""
    from g4splat_amd.optim import densify_stats
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    P = 100_003
    acc_h, den_h, mr_h = torch.zeros((P, 1)), torch.zeros((P, 1)), torch.zeros(P)
    acc_d, den_d, mr_d = acc_h.to(dev), den_h.to(dev), mr_h.to(dev)
    for view in range(4):
        grad = torch.randn((P, 3), generator=g) * 1e-3
        grad[:, 2] = 0
        radii = torch.randint(0, 40, (P,), generator=g, dtype=torch.int32)
        radii[torch.rand(P, generator=g) < 0.6] = 0
        filt = radii > 0
        acc_h[filt] += torch.norm(grad[filt], dim=-1, keepdim=True)
        den_h[filt] += 1
        mr_h[filt] = torch.max(mr_h[filt], radii[filt].float())
        densify_stats(grad.to(dev), filt.to(dev), acc_d, den_d, radii.to(dev), mr_d)
    torch.cuda.synchronize()
    assert torch.equal(den_d.cpu(), den_h) and torch.equal(mr_d.cpu(), mr_h)
    assert (acc_d.cpu() - acc_h).abs().max() <= 4e-7 * acc_h.abs().max()
    # without radii: max_radii2D untouched
    before = mr_d.clone()
    densify_stats(torch.ones((P, 3), device=dev), torch.ones(P, dtype=torch.bool, device=dev), acc_d, den_d)
    assert torch.equal(mr_d, before) and torch.equal(den_d.cpu(), den_h + 1)
    with pytest.raises(RuntimeError):
        densify_stats(torch.ones((P, 3), device=dev), torch.ones(P, dtype=torch.bool, device=dev), acc_d, den_d,
                      radii=torch.zeros(P, dtype=torch.int32, device=dev))


def test_fused_activations_match_torch(hip_lib):
    """g4s_activations_* against torch.exp / torch.nn.functional.normalize / torch.sigmoid and their autograd backward
    (the getters of 2dgs/scene/gaussian_model.py:157-192, mip filter off)."""
    from g4splat_amd.optim import fused_activations
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(12)
    P = 200_003
    raw = [torch.randn((P, 2), generator=g) * 1.5 - 3.0, torch.randn((P, 4), generator=g), torch.randn((P, 1), generator=g) * 3]
    raw[1][::1000] *= 1e-3  # nearly-degenerate quaternions
    raw[1][5] = 0.0         # an exactly-zero quaternion takes the clamped branch
    cot = [torch.randn(t.shape, generator=g) for t in raw]
    a = [t.clone().to(dev).requires_grad_(True) for t in raw]
    b = [t.clone().to(dev).requires_grad_(True) for t in raw]
    ya = fused_activations(*a)
    yb = (torch.exp(b[0]), torch.nn.functional.normalize(b[1]), torch.sigmoid(b[2]))
    sum((y * c.to(dev)).sum() for y, c in zip(ya, cot)).backward()
    sum((y * c.to(dev)).sum() for y, c in zip(yb, cot)).backward()
    torch.cuda.synchronize()
    for y, z in zip(ya, yb):
        assert y.shape == z.shape
        assert (y - z).detach().abs().max() <= 4e-7 * max(1.0, float(z.detach().abs().max()))
    for x, z, name in zip(a, b, ("scaling", "rotation", "opacity")):
        ref = z.grad
        err = (x.grad - ref).abs()
        tol = 2e-6 * ref.abs().clamp_min(1e-6) if name != "rotation" else 2e-5 * ref.abs().max(dim=1, keepdim=True).values.clamp_min(1e-6)
        mask = torch.ones(P, dtype=torch.bool, device=dev)
        mask[5] = False  # the zero quaternion: both give g / 1e-12, compare separately
        assert bool((err[mask] <= tol[mask]).all()), name
    assert torch.allclose(a[1].grad[5], b[1].grad[5], rtol=1e-5)


def test_compact_rows_equals_mask_indexing(hip_lib):
    """g4s_compact_scan / g4s_compact_gather (wave ballot + prefix sum) against torch's `t[keep]`: same rows, same
    order, for row widths 1 .. 48, sizes that are not multiples of 256, empty / full / single-row masks, and more than
    eight tensors per call (several gather launches against one scan)."""
    from g4splat_amd.optim import compact_rows
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    for P, frac in ((1, 1.0), (1, 0.0), (255, 0.5), (256, 0.5), (257, 0.5), (10_000, 0.03), (100_003, 0.7), (5000, 0.0),
                    (5000, 1.0), (1_500_000, 0.28)):
        keep = (torch.rand(P, generator=g) < frac).to(dev)
        shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 2), (P, 4), (P,), (P, 3), (P, 1, 3), (P, 15, 3), (P, 1)]
        ts = [torch.randn(sh, generator=g).to(dev) for sh in shapes]
        n, out = compact_rows(keep, ts, extra_rows=5)
        assert n == int(keep.sum())
        for t, o in zip(ts, out):
            assert o.shape == (n + 5,) + tuple(t.shape[1:])
            assert torch.equal(o[:n], t[keep])
    with pytest.raises(RuntimeError):
        compact_rows(torch.ones(4, dtype=torch.bool), [torch.ones(4, 3)])  # host tensors: no CPU path


def test_densification_on_the_gpu_replays_the_reference(hip_lib, monkeypatch):
    """The reference-generated densification fixture (tests/golden/densify.npz: the reference's GaussianModel run on the
    CPU through Adam steps, statistics, clone / split / prune with and without a screen-size limit, opacity reset)
    replayed on the GPU with FusedAdam and the HIP compaction kernels.  The split's torch.normal draw is taken from the
    CPU generator (the GPU's stream differs) so that the selections and the new points are comparable; values agree
    to float32 round-off (exp / log / Adam's arithmetic differ in the last bits between the two devices)."""
    import os
    import sys
    import types
    from g4splat_amd.gaussian_model import GaussianModel
    from g4splat_amd.optim import FusedAdam
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold_dir)
    try:
        import make_golden_densify as mg
    finally:
        sys.path.remove(gold_dir)
    z = np.load(os.path.join(gold_dir, "densify.npz"))
    dev = torch.device("cuda:0")
    real_normal = torch.normal
    monkeypatch.setattr(torch, "normal", lambda mean, std, generator=None: real_normal(mean=mean.cpu(), std=std.cpu()).to(mean.device))
    t = lambda a: torch.tensor(a, device=dev)
    gm = GaussianModel(3)
    gm.create_from_parameters(t(z["in_means"]), t(z["in_scales"]), t(z["in_quats"]), t(z["in_colors"]), 1.0)
    with torch.no_grad():
        gm._opacity.copy_(t(z["in_opacity_raw"]))
        gm._features_rest.copy_(t(z["in_f_rest"]))
    gm.training_setup(types.SimpleNamespace(**mg.ARGS))
    assert isinstance(gm.optimizer, FusedAdam)
    inp = dict(vs_grad=[z[f"in_vs_grad_{i}"] for i in range(4)], vs_filter=[z[f"in_vs_filter_{i}"] for i in range(4)],
               max_radii=z["in_max_radii"])

    # drive() as written for the host, with every host tensor it creates moved to the device
    real_tensor = torch.tensor
    monkeypatch.setattr(torch, "tensor", lambda *a, **k: real_tensor(*a, **k).to(dev) if "device" not in k else real_tensor(*a, **k))
    real_zeros = torch.zeros
    monkeypatch.setattr(torch, "zeros", lambda *a, **k: real_zeros(*a, **({"device": dev} | k)))
    out = {}

    def snapshot(m, tag, o):
        for f in mg.FIELDS:
            p = getattr(m, f)
            o[f"{tag}{f}"] = p.detach().cpu().numpy()
            st = m.optimizer.state.get(p, None)
            if st:
                o[f"{tag}{f}_exp_avg"] = st["exp_avg"].cpu().numpy()
                o[f"{tag}{f}_exp_avg_sq"] = st["exp_avg_sq"].cpu().numpy()
        o[f"{tag}_accum"] = m.xyz_gradient_accum.cpu().numpy()
        o[f"{tag}_denom"] = m.denom.cpu().numpy()
        o[f"{tag}_max_radii2D"] = m.max_radii2D.cpu().numpy()

    monkeypatch.setattr(mg, "snapshot", snapshot)
    mg.drive(gm, inp, out, lambda m: [g["params"][0] for g in m.optimizer.param_groups])
    for k in z.files:
        if k.startswith("in_") or k.startswith("initial"):
            continue
        assert out[k].shape == z[k].shape, (k, out[k].shape, z[k].shape)
        scale = np.abs(z[k]).max() + 1e-12
        assert np.abs(out[k] - z[k]).max() <= 2e-5 * scale, (k, float(np.abs(out[k] - z[k]).max()), float(scale))
