"""Pins the CPU oracle and the host-side helpers against golden vectors generated from the
reference's own Python helpers (tests/golden/make_golden.py) -- the only reference code that can be
run in this project; the CUDA rasterizer itself has no runnable path, so the
oracle's parity with it stays "unpinned" beyond these vectors and the cross-checks in
test_oracle_vs_torch.py."""
import os

import numpy as np

from common import EMPTY, run_oracle, scene_inputs
from g4splat_amd import synthetic

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_camera_matrices_match_reference_helpers():
    d = np.load(os.path.join(G, "camera.npz"))
    for i in range(4):
        cam = synthetic.make_camera(d[f"R_{i}"], d[f"t_{i}"], float(d[f"fovx_{i}"]), float(d[f"fovy_{i}"]),
                                    int(d[f"W_{i}"]), int(d[f"H_{i}"]))
        np.testing.assert_allclose(cam.world_view_transform, d[f"world_view_transform_{i}"], atol=1e-6)
        np.testing.assert_allclose(cam.full_proj_transform, d[f"full_proj_transform_{i}"], atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(cam.camera_center, d[f"camera_center_{i}"], atol=2e-5)


def _one_gaussian_scene(P):
    inp = scene_inputs(P=P, W=64, H=64, seed=0, D=0)
    inp["means3D"] = np.tile(np.array([[0.0, 0.0, 3.0]], np.float32), (P, 1))
    inp["scales"] = np.full((P, 2), 0.05, np.float32)
    inp["rotations"] = np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1))
    inp["opacity"] = np.full((P, 1), 0.5, np.float32)
    return inp


def test_sh_colours_match_reference_eval_sh(oracle_mod):
    d = np.load(os.path.join(G, "sh_eval.npz"))
    sh, dirs = d["sh"], d["dirs"]
    P = sh.shape[0]
    for deg in range(4):
        inp = _one_gaussian_scene(P)
        # place each Gaussian on its direction from the camera centre (origin), in front of the camera
        dd = dirs.copy()
        dd[:, 2] = np.abs(dd[:, 2]) + 2.5  # keep them inside the 60-degree frustum
        dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        inp["means3D"] = (dd * 3.0).astype(np.float32)
        inp["sh"] = sh
        inp["D"] = deg
        inp["campos"] = np.zeros(3, np.float32)
        o = run_oracle(oracle_mod, inp)
        vis = o["radii"] > 0
        assert vis.sum() > P // 2
        # reference: clamp_min(eval_sh(deg, sh^T, dir) + 0.5, 0) with dir = normalised (mean - campos)
        import torch
        ref = _eval_sh_numpy(deg, sh, inp["means3D"] / np.linalg.norm(inp["means3D"], axis=1, keepdims=True))
        np.testing.assert_allclose(o["oracle"].state("rgb")[vis], np.maximum(ref + 0.5, 0)[vis], atol=2e-6)
    # and the golden itself pins _eval_sh_numpy to the reference implementation
    for deg in range(4):
        np.testing.assert_allclose(_eval_sh_numpy(deg, sh, dirs), d[f"raw_deg{deg}"], atol=2e-6)


def _eval_sh_numpy(deg, sh, dirs):
    from oracle import torch_ref
    import torch
    # torch_ref.sh_to_rgb = clamp(eval + 0.5); recover the raw value through a shifted DC term
    big = sh.astype(np.float64)
    big[:, 0, :] += 100.0 / torch_ref.SH_C0
    v = torch_ref.sh_to_rgb(deg, torch.tensor(big, dtype=torch.float64), torch.tensor(dirs, dtype=torch.float64),
                            torch.zeros(3, dtype=torch.float64)).numpy()
    return (v - 0.5 - 100.0).astype(np.float64)


def test_transmat_matches_reference_python_formulation(oracle_mod):
    """T from the oracle's preprocess == the reference's compute_cov3D_python path (render(), :64-75)."""
    d = np.load(os.path.join(G, "transmat.npz"))
    c = np.load(os.path.join(G, "camera.npz"))
    i = int(d["cam"])
    P = d["xyz"].shape[0]
    rot = d["rot"] / np.linalg.norm(d["rot"], axis=1, keepdims=True)  # the model feeds normalised rotations
    o = oracle_mod.Oracle()
    W, H = int(c[f"W_{i}"]), int(c[f"H_{i}"])
    import math
    R, _, _, radii = o.rasterize_gaussians(
        np.zeros(3, np.float32), d["xyz"], np.full((P, 3), 0.5, np.float32), np.full((P, 1), 0.5, np.float32),
        d["scaling"], rot.astype(np.float32), float(d["mod"]), EMPTY, c[f"world_view_transform_{i}"],
        c[f"full_proj_transform_{i}"], math.tan(float(c[f"fovx_{i}"]) / 2), math.tan(float(c[f"fovy_{i}"]) / 2), H, W,
        EMPTY, 0, c[f"camera_center_{i}"])
    # T is written for every Gaussian in front of the near plane, even if culled later
    view = c[f"world_view_transform_{i}"]
    z = (np.concatenate([d["xyz"], np.ones((P, 1), np.float32)], 1) @ view)[:, 2]
    front = z > 0.2
    got, want = o.state("transMat")[front], d["T"][front]
    scale = np.abs(want).max(axis=1, keepdims=True)
    assert np.abs(got - want).max() <= 2e-5 * max(1.0, scale.max())
    np.testing.assert_allclose(got / scale, want / scale, atol=5e-6)


def test_oracle_regression_small_scene(oracle_mod):
    d = np.load(os.path.join(G, "oracle_small.npz"))
    sc = d["scalars"]
    inp = {k[3:]: d[k] for k in d.files if k.startswith("in_")}
    inp.update(scale_modifier=float(sc[0]), tanfovx=float(sc[1]), tanfovy=float(sc[2]), H=int(sc[3]), W=int(sc[4]),
               D=int(sc[5]))
    o = run_oracle(oracle_mod, inp, (d["dL_dcolor"], d["dL_dothers"]))
    assert o["R"] == int(d["R"])
    np.testing.assert_array_equal(o["radii"], d["radii"])
    np.testing.assert_allclose(o["color"], d["color"], atol=1e-6)
    np.testing.assert_allclose(o["others"], d["others"], atol=1e-5)
    for k, v in o["grads"].items():
        ref = d["grad_" + k]
        assert np.abs(v - ref).max() <= 1e-5 * (np.abs(ref).max() + 1e-30), k


def test_loss_helpers_match_reference():
    import torch
    from oracle import metrics_ref as metrics
    d = np.load(os.path.join(G, "losses.npz"))
    a, b = torch.tensor(d["a"]), torch.tensor(d["b"])
    np.testing.assert_allclose(metrics.l1_loss(a, b).numpy(), d["l1"], rtol=1e-6)
    np.testing.assert_allclose(metrics.ssim(a, b).numpy(), d["ssim"], rtol=1e-5)
    np.testing.assert_allclose(metrics.psnr(a[None], b[None]).numpy(), d["psnr"], rtol=1e-6)


def test_photometric_loss_restatement_matches_the_reference():
    """oracle/losses_ref.py against values AND gradients produced by the reference's own loss_utils.py
    (tests/golden/make_golden_losses.py imports it): this part of the oracle is pinned."""
    import torch
    from oracle import losses_ref
    g = np.load(os.path.join(G, "photometric.npz"))
    for name in ("ragged", "tile", "tiny", "zeros"):
        x = torch.tensor(g[f"{name}_image"], requires_grad=True)
        loss, l1, ss = losses_ref.photometric_loss(x, torch.tensor(g[f"{name}_gt"]), float(g[f"{name}_lambda"]))
        loss.backward()
        assert abs(float(l1) - float(g[f"{name}_l1"])) <= 1e-7
        assert abs(float(ss) - float(g[f"{name}_ssim"])) <= 1e-6
        assert abs(float(loss) - float(g[f"{name}_loss"])) <= 1e-6
        np.testing.assert_allclose(x.grad.numpy(), g[f"{name}_grad"], rtol=1e-5, atol=1e-9)


def test_remaining_losses_match_the_reference():
    """metrics.l1_loss_with_conf / l2_loss / smooth_loss / mse against values computed by the reference's own
    loss_utils.py (tests/golden/make_golden_misc.py)."""
    import torch
    from oracle import metrics_ref as metrics
    d = np.load(os.path.join(G, "losses_misc.npz"))
    a, b, conf, disp = (torch.tensor(d[k]) for k in ("a", "b", "conf", "disp"))
    assert abs(float(metrics.l1_loss_with_conf(a, b, conf)) - float(d["l1_conf"])) <= 1e-7
    assert abs(float(metrics.l2_loss(a, b)) - float(d["l2"])) <= 1e-7
    assert abs(float(metrics.smooth_loss(disp, a)) - float(d["smooth"])) <= 1e-6
    np.testing.assert_allclose(metrics.mse(a, b).numpy(), d["mse"], rtol=1e-6)


def test_render_maps_ref_matches_the_reference_render_tail():
    """oracle/render_maps_ref.py against the reference's own render() tail + point_utils, values and autograd
    gradients (fixture made by tests/golden/make_golden_maps.py by running the reference's code on the CPU)."""
    from types import SimpleNamespace
    import torch
    from oracle.render_maps_ref import render_maps
    z = np.load(os.path.join(G, "render_tail.npz"))
    names = ("rend_alpha", "rend_normal", "rend_normal_cam", "rend_depth", "rend_dist", "surf_depth", "surf_normal", "surf_normal_cam")
    for ci in range(3):
        W, H, ratio = z[f"c{ci}_meta"]
        cam = SimpleNamespace(image_width=int(W), image_height=int(H), world_view_transform=torch.tensor(z[f"c{ci}_wvt"]),
                              full_proj_transform=torch.tensor(z[f"c{ci}_fpt"]))
        am = torch.tensor(z[f"c{ci}_allmap"], requires_grad=True)
        out = render_maps(am, cam, float(ratio))
        for m in names:
            np.testing.assert_allclose(out[m].detach().numpy(), z[f"c{ci}_{m}"], rtol=0, atol=1e-6, err_msg=m)
        sum((out[m] * torch.tensor(z[f"c{ci}_cot_{m}"])).sum() for m in names).backward()
        want = z[f"c{ci}_dL_dallmap"]
        got = am.grad.numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.isnan(want).any()  # alpha == 0 pixels: torch's 0/0 backward, part of the reference's behaviour
        np.testing.assert_allclose(np.nan_to_num(got), np.nan_to_num(want), rtol=1e-5, atol=1e-6 * np.nanmax(np.abs(want)))
