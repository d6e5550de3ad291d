"""Golden vectors for adaptive density control, produced by RUNNING THE REFERENCE'S OWN
`GaussianModel` (2d-gaussian-splatting/scene/gaussian_model.py) on the CPU of the build container:

    densify.npz   a seeded 150-Gaussian model with populated Adam state and densification statistics, then
                  create_from_parameters (:225-246), training_setup (:248-266), three Adam steps,
                  add_densification_stats (:649-651), densify_and_prune (:628-647 -> densify_and_clone :612-626,
                  densify_and_split :583-610 incl. its torch.normal draw under torch.manual_seed, prune_points
                  :527-541), reset_opacity (:436-439) -- parameters, both Adam moments and the statistics after
                  every stage.

tests/test_densify.py replays the same sequence on g4splat_amd.gaussian_model.GaussianModel (CPU, same seed:
torch's CPU generator is deterministic) and requires identical tensors.
Run in the build container only (needs /root/reference):  python tests/golden/make_golden_densify.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_import import reference_modules  # noqa: E402

FIELDS = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
ARGS = dict(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01,
            position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05, scaling_lr=0.005, rotation_lr=0.001)


def make_inputs():
    rng = np.random.default_rng(4242)
    P = 150
    return dict(
        means=rng.normal(0, 1.5, (P, 3)).astype(np.float32),
        scales=np.exp(rng.uniform(np.log(0.004), np.log(0.2), (P, 2))).astype(np.float32),
        quats=rng.normal(size=(P, 4)).astype(np.float32),
        colors=rng.uniform(0, 1, (P, 3)).astype(np.float32),
        # three synthetic gradient sets for the Adam steps (one per parameter group and step)
        **{f"g{s}_{f}": None for s in range(3) for f in FIELDS},
        opacity_raw=rng.normal(0, 2.5, (P, 1)).astype(np.float32),
        f_rest=rng.normal(0, 0.1, (P, 15, 3)).astype(np.float32),
        vs_grad=[rng.normal(0, 3e-4, (P, 3)).astype(np.float32) for _ in range(4)],
        vs_filter=[rng.uniform(size=P) < 0.6 for _ in range(4)],
        max_radii=rng.uniform(0, 40, P).astype(np.float32),
        rng=rng)


def snapshot(gm, tag, out):
    for f in FIELDS:
        p = getattr(gm, f)
        out[f"{tag}{f}"] = p.detach().numpy().copy()
        st = gm.optimizer.state.get(p, None)
        if st:
            out[f"{tag}{f}_exp_avg"] = st["exp_avg"].numpy().copy()
            out[f"{tag}{f}_exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
    out[f"{tag}_accum"] = gm.xyz_gradient_accum.numpy().copy()
    out[f"{tag}_denom"] = gm.denom.numpy().copy()
    out[f"{tag}_max_radii2D"] = gm.max_radii2D.numpy().copy()


def drive(gm, inp, out, params_of):
    """The sequence both sides run.  `params_of(gm)` lists the six parameters in group order."""
    rng = np.random.default_rng(99)
    for step in range(3):
        for p in params_of(gm):
            g = rng.normal(0, 1e-3, tuple(p.shape)).astype(np.float32)
            out[f"adamgrad_{step}_{tuple(p.shape)}"] = g
            p.grad = torch.tensor(g)
        gm.optimizer.step()
        gm.optimizer.zero_grad(set_to_none=True)
    snapshot(gm, "after_adam", out)
    for i in range(4):
        vs = torch.zeros((gm.get_xyz.shape[0], 3), requires_grad=True)
        vs.grad = torch.tensor(inp["vs_grad"][i])
        gm.add_densification_stats(vs, torch.tensor(inp["vs_filter"][i]))
    gm.max_radii2D = torch.tensor(inp["max_radii"])
    snapshot(gm, "after_stats", out)
    torch.manual_seed(20240928)
    gm.densify_and_prune(0.0002, 0.05, 4.0, 20)
    snapshot(gm, "after_densify", out)
    gm.reset_opacity()
    snapshot(gm, "after_reset", out)
    # a second round right away, now without a screen-size limit (the `max_screen_size=None` branch of the trainer)
    n = gm.get_xyz.shape[0]
    r2 = np.random.default_rng(7)
    gm.xyz_gradient_accum = torch.tensor(r2.uniform(0, 6e-4, (n, 1)).astype(np.float32))
    gm.denom = torch.tensor(r2.integers(0, 3, (n, 1)).astype(np.float32))  # zeros -> 0/0 -> NaN -> 0
    torch.manual_seed(5)
    gm.densify_and_prune(0.0002, 0.005, 4.0, None)
    snapshot(gm, "after_densify2", out)


def main():
    inp = make_inputs()
    out = {}
    with reference_modules():
        from scene.gaussian_model import GaussianModel
        gm = GaussianModel(3)
        gm.create_from_parameters(torch.tensor(inp["means"]), torch.tensor(inp["scales"]), torch.tensor(inp["quats"]),
                                  torch.tensor(inp["colors"]), 1.0)
        with torch.no_grad():  # spread the opacities and give the higher SH bands content
            gm._opacity.copy_(torch.tensor(inp["opacity_raw"]))
            gm._features_rest.copy_(torch.tensor(inp["f_rest"]))
        gm.training_setup(types.SimpleNamespace(**ARGS))
        snapshot(gm, "initial", out)
        drive(gm, inp, out, lambda m: [g["params"][0] for g in m.optimizer.param_groups])
    for k in ("means", "scales", "quats", "colors", "opacity_raw", "f_rest", "max_radii"):
        out["in_" + k] = inp[k]
    for i in range(4):
        out[f"in_vs_grad_{i}"] = inp["vs_grad"][i]
        out[f"in_vs_filter_{i}"] = inp["vs_filter"][i]
    out = {k: v for k, v in out.items() if not k.startswith("adamgrad_")}
    np.savez_compressed(os.path.join(HERE, "densify.npz"), **out)
    print("wrote densify.npz", os.path.getsize(os.path.join(HERE, "densify.npz")),
          {k: out[k].shape[0] for k in out if k.endswith("_xyz")})


if __name__ == "__main__":
    main()
