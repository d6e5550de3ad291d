"""render() and the autograd operator on the MI355X vs the same host code driven by the CPU oracle."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from cpu_rasterizer import OracleRasterizer
from test_render_cpu import KEYS, _camera, _model
from g4splat_amd.gaussian_renderer import render
from oracle.render_maps_ref import render_maps as maps_ref

pytestmark = pytest.mark.gpu


def _to(cam, dev):
    d = dict(vars(cam))
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            d[k] = v.to(dev)
    return SimpleNamespace(**d)


def _loss(out):
    return (out["render"] ** 2).mean() + out["rend_dist"].mean() + (1 - (out["rend_normal"] * out["surf_normal"]).sum(0)).mean() \
        + 0.1 * out["surf_depth"].mean()


def test_render_matches_oracle_driven_render(hip_lib):
    cam, pipe = _camera(), SimpleNamespace(depth_ratio=0.5, compute_cov3D_python=False)
    ref_model, hip_model = _model(seed=2), _model(seed=2)
    ref = render(cam, ref_model, pipe, torch.tensor([0.1, 0.2, 0.3]), rasterizer_cls=OracleRasterizer, maps_fn=maps_ref)
    _loss(ref).backward()
    dev = torch.device("cuda:0")
    for name in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        setattr(hip_model, name, torch.nn.Parameter(getattr(hip_model, name).detach().to(dev)))
    out = render(_to(cam, dev), hip_model, pipe, torch.tensor([0.1, 0.2, 0.3], device=dev))
    assert set(out) == KEYS
    _loss(out).backward()
    torch.cuda.synchronize()
    for k in KEYS - {"viewspace_points"}:
        a, b = out[k].detach().cpu(), ref[k].detach()
        if a.dtype.is_floating_point:
            assert (a - b).abs().max() <= 2e-4, k
        else:
            assert torch.equal(a, b), k
    for p, q in zip(hip_model.parameters(), ref_model.parameters()):
        g, r = p.grad.cpu(), q.grad
        assert (g - r).abs().max() <= 1e-3 * r.abs().max() + 1e-9
    g2, r2 = out["viewspace_points"].grad.cpu(), ref["viewspace_points"].grad
    assert (g2 - r2).abs().max() <= 1e-3 * r2.abs().max() + 1e-9


def test_no_grad_render_and_large_scene_properties(hip_lib):
    """Size-independent properties on a larger scene: alpha in [0,1], colour = blend + T*bg (linearity in
    bg), idempotence (same inputs -> bitwise same outputs)."""
    from common import run_hip, scene_inputs
    inp = scene_inputs(P=200000, W=640, H=480, seed=13, D=2, bg=(0.0, 0.0, 0.0), scale_mul=0.6)
    a = run_hip(inp)
    b = run_hip(inp)
    np.testing.assert_array_equal(a["color"], b["color"])
    np.testing.assert_array_equal(a["others"], b["others"])
    alpha = a["others"][1]
    assert alpha.min() >= 0 and alpha.max() <= 1 + 1e-6
    inp2 = dict(inp)
    inp2["bg"] = np.array([1.0, 0.5, 0.25], np.float32)
    c = run_hip(inp2)
    T = 1 - alpha
    for ch in range(3):
        assert np.abs(c["color"][ch] - (a["color"][ch] + T * inp2["bg"][ch])).max() <= 1e-5
    np.testing.assert_array_equal(c["others"], a["others"])
    assert (a["radii"] > 0).sum() > 1000
