"""Drop-in surface of the Python packages (dsr/diff_surfel_rasterization/__init__.py:158-222,
knn/ext.cpp): names, field order, argument validation and error behaviour -- no GPU needed."""
import inspect

import pytest
import torch

from g4splat_amd import dropin
from g4splat_amd.diff_surfel_rasterization import (GaussianRasterizationSettings, GaussianRasterizer,
                                                   rasterize_gaussians)


def _settings():
    return GaussianRasterizationSettings(image_height=32, image_width=48, tanfovx=0.5, tanfovy=0.4,
                                         bg=torch.zeros(3), scale_modifier=1.0, viewmatrix=torch.eye(4),
                                         projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
                                         prefiltered=False, debug=False)


def test_settings_fields_in_reference_order():
    assert GaussianRasterizationSettings._fields == ("image_height", "image_width", "tanfovx", "tanfovy", "bg",
                                                     "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
                                                     "campos", "prefiltered", "debug")


def test_forward_signature():
    sig = inspect.signature(GaussianRasterizer.forward)
    assert list(sig.parameters) == ["self", "means3D", "means2D", "opacities", "shs", "colors_precomp", "scales",
                                    "rotations", "cov3D_precomp"]
    assert list(inspect.signature(rasterize_gaussians).parameters) == [
        "means3D", "means2D", "sh", "colors_precomp", "opacities", "scales", "rotations", "cov3Ds_precomp",
        "raster_settings"]


def test_argument_validation_messages():
    r = GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), shs=torch.zeros(4, 16, 3), colors_precomp=x)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), colors_precomp=x)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), colors_precomp=x, scales=torch.ones(4, 2),
          rotations=torch.ones(4, 4), cov3D_precomp=torch.ones(4, 9))


def test_cpu_tensors_are_rejected_like_the_reference():
    """CHECK_INPUT (rasterize_points.cu:27-28): no CPU fallback, ever."""
    r = GaussianRasterizer(_settings())
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        r(means3D=x, means2D=x, opacities=torch.ones(4, 1), colors_precomp=x, scales=torch.ones(4, 2),
          rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="means3D must have dimensions"):
        r(means3D=torch.zeros(4, 2), means2D=x, opacities=torch.ones(4, 1), colors_precomp=x,
          scales=torch.ones(4, 2), rotations=torch.ones(4, 4))
    from g4splat_amd.simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        distCUDA2(torch.zeros(5, 3))


def test_dropin_module_names():
    """`import diff_surfel_rasterization` / `from simple_knn._C import distCUDA2` resolve to this package."""
    dropin.install()
    import diff_surfel_rasterization as d
    from simple_knn._C import distCUDA2
    from diff_surfel_rasterization import GaussianRasterizationSettings as S2
    assert d.GaussianRasterizer is GaussianRasterizer and S2 is GaussianRasterizationSettings
    assert callable(distCUDA2) and callable(d._C.rasterize_gaussians) and callable(d._C.mark_visible)


def test_fused_adam_state_dict_is_interchangeable_with_torch_adam():
    """a checkpoint captured with FusedAdam must load into torch.optim.Adam (the reference's restore()) and
    step, and vice versa -- FusedAdam therefore carries torch.optim.Adam's full set of param-group keys.  (CPU: only
    construction / state_dict / load_state_dict are exercised on the fused side; its step() needs a GPU.)"""
    import torch
    from g4splat_amd.optim import FusedAdam
    mk = lambda: [{"params": [torch.nn.Parameter(torch.ones(5, 3))], "lr": 0.01, "name": "xyz"},
                  {"params": [torch.nn.Parameter(torch.ones(5, 1))], "lr": 0.05, "name": "opacity"}]
    fused = FusedAdam(mk(), lr=0.0, eps=1e-15)
    ref = torch.optim.Adam(mk(), lr=0.0, eps=1e-15)
    ref_keys = set(ref.state_dict()["param_groups"][0])
    assert ref_keys <= set(fused.state_dict()["param_groups"][0]), ref_keys - set(fused.state_dict()["param_groups"][0])
    # fused -> torch: populate a state, load, step
    for g in fused.param_groups:
        p = g["params"][0]
        fused.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.full_like(p, 0.1), "exp_avg_sq": torch.full_like(p, 0.2)}
    ref.load_state_dict(fused.state_dict())
    for g in ref.param_groups:
        g["params"][0].grad = torch.ones_like(g["params"][0])
    ref.step()  # raised KeyError('weight_decay') before
    assert float(ref.state[ref.param_groups[0]["params"][0]]["step"]) == 4.0
    # torch -> fused
    fused2 = FusedAdam(mk(), lr=0.0, eps=1e-15)
    fused2.load_state_dict(ref.state_dict())
    assert fused2.param_groups[0]["eps"] == 1e-15 and fused2.param_groups[1]["lr"] == 0.05
    with pytest.raises(ValueError):
        FusedAdam(mk(), weight_decay=0.1)
    bad = ref.state_dict()
    bad["param_groups"][0]["amsgrad"] = True
    fused2.load_state_dict(bad)
    fused2.param_groups[0]["params"][0].grad = torch.ones(5, 3)
    with pytest.raises(ValueError):
        fused2.step()  # non-default settings smuggled in through load_state_dict are refused at step()


def test_debug_snapshot_covers_the_split_sh_node(tmp_path, monkeypatch):
    """raster_settings.debug=True writes snapshot_fw.dump on a failing forward also on the split-SH path
    (the one render() takes), with both SH tensors in it."""
    import torch
    from g4splat_amd import diff_surfel_rasterization as dsr
    monkeypatch.chdir(tmp_path)

    def boom(*a):
        raise RuntimeError("injected failure")

    monkeypatch.setattr(dsr._C, "rasterize_gaussians", boom)
    P = 4
    rs = dsr.GaussianRasterizationSettings(image_height=8, image_width=8, tanfovx=1.0, tanfovy=1.0, bg=torch.zeros(3),
                                           scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=1,
                                           campos=torch.zeros(3), prefiltered=False, debug=True)
    with pytest.raises(RuntimeError, match="injected"):
        dsr.rasterize_gaussians(torch.zeros(P, 3), torch.zeros(P, 3), (torch.ones(P, 1, 3), torch.ones(P, 15, 3)), None,
                                torch.ones(P, 1), torch.ones(P, 2), torch.ones(P, 4), torch.empty(0), rs)
    dump = torch.load(tmp_path / "snapshot_fw.dump")
    pair = [a for a in dump if isinstance(a, tuple)]
    assert len(pair) == 1 and pair[0][0].shape == (P, 1, 3) and pair[0][1].shape == (P, 15, 3)


def test_presized_context_nests_and_is_per_thread():
    """diff_surfel_rasterization.presized: the active (state, workspace) is what the innermost `with` set, is restored on
    exit (also when the body raises) and is not seen by other threads (the autograd engine's workers take what the forward
    stored in the node instead)."""
    import threading
    import g4splat_amd.diff_surfel_rasterization as dsr
    a, b = object(), object()
    assert getattr(dsr._slot, "value", None) is None
    with dsr.presized(a, "wa"):
        assert dsr._slot.value == (a, "wa")
        seen = []
        th = threading.Thread(target=lambda: seen.append(getattr(dsr._slot, "value", None)))
        th.start(); th.join()
        assert seen == [None]
        with pytest.raises(ValueError):
            with dsr.presized(b):
                assert dsr._slot.value == (b, None)
                raise ValueError("x")
        assert dsr._slot.value == (a, "wa")
    assert dsr._slot.value is None


def test_product_metrics_match_the_reference_goldens():
    """g4splat_amd.metrics (the reference's utils/loss_utils.py / image_utils.py names a training loop logs with) against
    values computed by the reference's own functions (tests/golden/losses*.npz).  `ssim` runs the fused HIP kernel and is
    covered by tests/test_gpu_losses.py."""
    import os
    import numpy as np
    import torch
    from g4splat_amd import metrics
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    d = np.load(os.path.join(G, "losses.npz"))
    a, b = torch.tensor(d["a"]), torch.tensor(d["b"])
    np.testing.assert_allclose(metrics.l1_loss(a, b).numpy(), d["l1"], rtol=1e-6)
    np.testing.assert_allclose(metrics.psnr(a[None], b[None]).numpy(), d["psnr"], rtol=1e-6)
    m = np.load(os.path.join(G, "losses_misc.npz"))
    a, b, conf = (torch.tensor(m[k]) for k in ("a", "b", "conf"))
    assert abs(float(metrics.l1_loss_with_conf(a, b, conf)) - float(m["l1_conf"])) <= 1e-7
    assert abs(float(metrics.l2_loss(a, b)) - float(m["l2"])) <= 1e-7
    np.testing.assert_allclose(metrics.mse(a, b).numpy(), m["mse"], rtol=1e-6)
    with pytest.raises(RuntimeError, match="CUDA"):
        metrics.ssim(a, b)  # the SSIM is the HIP kernel: no CPU path
