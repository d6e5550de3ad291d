"""Adaptive density control (g4splat_amd/densify.py) against the rules of 2dgs/scene/gaussian_model.py:436-651,
on CPU tensors with torch.optim.Adam (the same code runs on HIP tensors with FusedAdam)."""
import numpy as np
import torch

from g4splat_amd.densify import build_rotation
from g4splat_amd.gaussian_model import GaussianModel


def _model(P=40, seed=0):
    rng = np.random.default_rng(seed)
    m = GaussianModel(sh_degree=2)
    scales = np.where(np.arange(P)[:, None] % 2 == 0, 0.004, 0.3) * np.ones((P, 2))  # even rows small, odd rows large
    m.create_from_parameters(torch.tensor(rng.normal(size=(P, 3)).astype(np.float32)),
                             torch.tensor(scales.astype(np.float32)),
                             torch.tensor(rng.normal(size=(P, 4)).astype(np.float32)),
                             torch.tensor(rng.uniform(0, 1, (P, 3)).astype(np.float32)))
    with torch.no_grad():
        m._opacity[:] = 2.0  # sigmoid = 0.88
    m.training_setup(fused=False)
    # one optimiser step so that every group has moments
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    m.optimizer.step()
    m.optimizer.zero_grad(set_to_none=True)
    return m


def _moments(m, name):
    for g in m.optimizer.param_groups:
        if g["name"] == name:
            return m.optimizer.state[g["params"][0]]


def test_clone_split_prune_counts_and_state():
    m = _model()
    P = 40
    xyz0, rot0, sc0 = m._xyz.detach().clone(), m._rotation.detach().clone(), m.get_scaling.detach().clone()
    m.max_radii2D = torch.zeros(P)
    # gradient statistics: rows 0..9 above the threshold (5 small -> clone, 5 large -> split), row 39 NaN (0/0)
    m.xyz_gradient_accum = torch.zeros((P, 1))
    m.denom = torch.ones((P, 1))
    m.xyz_gradient_accum[:10] = 1.0
    m.denom[39] = 0.0
    with torch.no_grad():
        m._opacity[20] = -10.0  # transparent -> pruned
    torch.manual_seed(0)
    m.densify_and_prune(max_grad=0.5, min_opacity=0.005, extent=1.0, max_screen_size=None)
    # 40 + 5 clones + 2*5 children - 5 split parents - 1 transparent = 49
    assert m._xyz.shape[0] == 49
    for t in (m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation):
        assert t.shape[0] == 49 and t.requires_grad
    assert m.xyz_gradient_accum.shape == (49, 1) and not m.xyz_gradient_accum.any() and not m.denom.any()
    assert m.max_radii2D.shape == (49,)
    # optimiser: parameters rebound, moments follow the rows (kept rows keep theirs, new rows start at zero)
    for g in m.optimizer.param_groups:
        p = g["params"][0]
        assert p is getattr(m, {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                                "scaling": "_scaling", "rotation": "_rotation"}[g["name"]])
        st = m.optimizer.state[p]
        assert st["exp_avg"].shape == p.shape and st["exp_avg_sq"].shape == p.shape and float(st["step"]) == 1
    ea = _moments(m, "xyz")["exp_avg"]
    n_kept = 40 - 5 - 1
    assert (ea[:n_kept] != 0).all() and not ea[n_kept:].any()
    # survivors in original order: small rows 0,2,4,6,8 stay, large rows 1,3,5,7,9 are gone, row 20 is gone
    kept_idx = [i for i in range(40) if i not in (1, 3, 5, 7, 9, 20)]
    assert torch.allclose(m._xyz.detach()[:n_kept], xyz0[kept_idx])
    # clones are exact copies of rows 0,2,4,6,8
    assert torch.equal(m._xyz.detach()[n_kept:n_kept + 5], xyz0[[0, 2, 4, 6, 8]])
    # children: N=2 per parent, scale / 1.6, same rotation, displaced inside the parent's plane
    kids = m._xyz.detach()[n_kept + 5:]
    parents = [1, 3, 5, 7, 9] * 2
    assert torch.allclose(m.get_scaling.detach()[n_kept + 5:], sc0[parents] / 1.6, rtol=1e-5)
    assert torch.equal(m._rotation.detach()[n_kept + 5:], rot0[parents])
    normal = build_rotation(rot0[parents])[:, :, 2]
    off = kids - xyz0[parents]
    assert (off * normal).sum(1).abs().max() <= 1e-5 and off.norm(dim=1).max() <= 5 * 0.3 * 1.5


def test_prune_by_screen_and_world_size_and_mip_toggle():
    m = _model(P=12, seed=1)
    m.xyz_gradient_accum = torch.zeros((12, 1))
    m.denom = torch.ones((12, 1))
    m.max_radii2D = torch.zeros(12)
    m.max_radii2D[3] = 50.0
    m.use_mip_filter = True
    m.mip_filter = torch.full((12, 1), 10.0)  # would make every scale > 0.1 * extent if it were not switched off
    m.densify_and_prune(max_grad=1e9, min_opacity=0.005, extent=1.0, max_screen_size=20)
    # row 3 (screen size) and the six large rows (0.3 > 0.1 * extent) go; the mip filter is restored
    assert m._xyz.shape[0] == 12 - 1 - 6 + (1 if 3 % 2 == 1 else 0)
    assert m.use_mip_filter and m.mip_filter.shape[0] == m._xyz.shape[0]


def test_reset_opacity_zeroes_moments():
    m = _model(P=9)
    with torch.no_grad():
        m._opacity[0] = -8.0  # already below 0.01: unchanged
    before = m.get_opacity.detach().clone()
    old_param = m._opacity
    m.reset_opacity()
    assert m._opacity is not old_param
    o = m.get_opacity.detach()
    assert torch.allclose(o[1:], torch.full_like(o[1:], 0.01), atol=1e-6) and torch.allclose(o[0], before[0])
    st = _moments(m, "opacity")
    assert not st["exp_avg"].any() and not st["exp_avg_sq"].any() and float(st["step"]) == 1


def test_compute_mip_filter():
    from types import SimpleNamespace
    m = _model(P=6)
    with torch.no_grad():
        m._xyz[:] = torch.tensor([[0, 0, 2.0], [0, 0, 4.0], [0, 0, -1.0], [100.0, 0, 1.0], [0.1, 0.1, 1.0], [0, 0, 0.1]])
    cam = SimpleNamespace(R=np.eye(3, dtype=np.float32), T=np.zeros(3, np.float32), focal_x=100.0, focal_y=100.0,
                          image_width=200, image_height=100)
    cam2 = SimpleNamespace(R=np.eye(3, dtype=np.float32), T=np.array([0, 0, 1.0], np.float32), focal_x=50.0, focal_y=50.0,
                           image_width=200, image_height=100)
    m.compute_mip_filter([cam, cam2])
    f = m.mip_filter[:, 0]
    k = 0.2 ** 0.5 / 100.0
    # closest valid depth over both cameras; points nobody sees get the largest valid distance
    assert torch.allclose(f[0], torch.tensor(2.0 * k)) and torch.allclose(f[1], torch.tensor(4.0 * k))
    assert torch.allclose(f[4], torch.tensor(1.0 * k)) and torch.allclose(f[5], torch.tensor(1.1 * k))
    assert torch.allclose(f[2], f.max()) and torch.allclose(f[3], f.max())


def test_schedule_checkpoint_and_helpers(tmp_path):
    """update_learning_rate against the reference's get_expon_lr_func (golden), capture/restore round trip through
    torch.save, get_covariance against its definition, freeze_params, gs_scale_loss, the reference-named optimiser
    wrappers, training_setup from an OptimizationParams-like object."""
    import os
    from types import SimpleNamespace
    from g4splat_amd.gaussian_model import get_expon_lr_func
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr_schedule.npz"))
    for i in range(3):
        lr0, lr1, dsteps, dmult, msteps = g[f"args_{i}"]
        f = get_expon_lr_func(lr_init=lr0, lr_final=lr1, lr_delay_steps=int(dsteps), lr_delay_mult=dmult, max_steps=int(msteps))
        np.testing.assert_allclose([f(int(s)) for s in g["steps"]], g[f"lr_{i}"], rtol=1e-12, atol=0)
    m = _model(P=11)
    ta = SimpleNamespace(position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
                         position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005,
                         rotation_lr=0.001, percent_dense=0.02)
    m.spatial_lr_scale = 4.3
    m.training_setup(ta, fused=False)
    assert m.percent_dense == 0.02
    assert [grp["name"] for grp in m.optimizer.param_groups] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert abs(m.update_learning_rate(7000) - float(g["lr_0"][list(g["steps"]).index(7000)])) < 1e-15
    assert m.optimizer.param_groups[0]["lr"] == m.update_learning_rate(7000)
    # checkpoint tuple
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    m.optimizer.step()
    m.xyz_gradient_accum = torch.rand((11, 1))
    m.denom = torch.ones((11, 1))
    path = str(tmp_path / "chkpnt.pth")
    torch.save((m.capture(), 123), path)
    args, it = torch.load(path, weights_only=False)
    n = GaussianModel(sh_degree=2)
    n.restore(args, ta, fused=False)
    assert it == 123 and n.active_sh_degree == m.active_sh_degree and n.spatial_lr_scale == 4.3
    for a, b in zip(m.parameters(), n.parameters()):
        assert torch.equal(a, b)
    assert torch.equal(n.xyz_gradient_accum, m.xyz_gradient_accum)
    sa, sb = m.optimizer.state[m._xyz], n.optimizer.state[n._xyz]
    assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and float(sa["step"]) == float(sb["step"]) == 1
    # splat-to-world transform
    T = m.get_covariance(2.0)
    R = build_rotation(m._rotation.detach())
    s = m.get_scaling.detach() * 2.0
    assert torch.allclose(T[:, 0, :3], R[:, :, 0] * s[:, :1], atol=1e-6) and torch.allclose(T[:, 1, :3], R[:, :, 1] * s[:, 1:2], atol=1e-6)
    assert torch.allclose(T[:, 2, :3], R[:, :, 2], atol=1e-6) and torch.equal(T[:, 3, :3], m._xyz.detach()) and (T[:, 3, 3] == 1).all()
    assert float(m.gs_scale_loss(0.05).detach()) == float(torch.clamp(m.get_scaling.detach().max(1).values - 0.05, min=0).pow(2).sum())
    assert m.construct_list_of_attributes()[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    kept = m._prune_optimizer(torch.arange(11) % 2 == 0)
    assert kept["xyz"] is m._xyz and m._xyz.shape[0] == 6
    grown = m.cat_tensors_to_optimizer({"xyz": torch.zeros(2, 3), "f_dc": torch.zeros(2, 1, 3), "f_rest": torch.zeros(2, 8, 3),
                                        "opacity": torch.zeros(2, 1), "scaling": torch.zeros(2, 2), "rotation": torch.ones(2, 4)})
    assert grown["rotation"] is m._rotation and m._rotation.shape[0] == 8
    m.freeze_params()
    assert not any(p.requires_grad for p in m.parameters())
    m.set_mip_filter(True)
    assert m.use_mip_filter


def test_replay_of_the_reference_gaussian_model():
    """tests/golden/densify.npz was written by running the REFERENCE's GaussianModel (scene/gaussian_model.py) on the CPU
    of the build container through create_from_parameters -> training_setup -> 3 Adam steps -> 4 x
    add_densification_stats -> densify_and_prune (clone, split with its seeded torch.normal draw, prune, with and
    without a screen-size limit, NaN statistics) -> reset_opacity (tests/golden/make_golden_densify.py).  The same
    sequence on this repo's GaussianModel / densify.py must give the same parameters, the same Adam moments and the
    same statistics after every stage -- bit for bit: same torch ops on the same CPU generator."""
    import os
    import sys
    import types
    gold_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold_dir)
    try:
        import make_golden_densify as mg
    finally:
        sys.path.remove(gold_dir)
    z = np.load(os.path.join(gold_dir, "densify.npz"))
    inp = dict(vs_grad=[z[f"in_vs_grad_{i}"] for i in range(4)], vs_filter=[z[f"in_vs_filter_{i}"] for i in range(4)],
               max_radii=z["in_max_radii"])
    gm = GaussianModel(3)
    gm.create_from_parameters(torch.tensor(z["in_means"]), torch.tensor(z["in_scales"]), torch.tensor(z["in_quats"]),
                              torch.tensor(z["in_colors"]), 1.0)
    with torch.no_grad():
        gm._opacity.copy_(torch.tensor(z["in_opacity_raw"]))
        gm._features_rest.copy_(torch.tensor(z["in_f_rest"]))
    gm.training_setup(types.SimpleNamespace(**mg.ARGS), fused=False)
    out = {}
    mg.snapshot(gm, "initial", out)
    mg.drive(gm, inp, out, lambda m: [g["params"][0] for g in m.optimizer.param_groups])
    keys = [k for k in z.files if not k.startswith("in_")]
    assert len(keys) > 60
    for k in keys:
        assert k in out, k
        assert out[k].shape == z[k].shape, (k, out[k].shape, z[k].shape)
        np.testing.assert_array_equal(out[k], z[k], err_msg=k)
    assert z["after_densify_xyz"].shape[0] > z["initial_xyz"].shape[0]  # it did clone / split
