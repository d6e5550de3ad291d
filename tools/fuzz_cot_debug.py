import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import cotangents, run_hip, run_oracle, scene_inputs, parity_report
import oracle.oracle as oracle_mod
for seed in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    W = int(rng.choice([1, 7, 16, 33, 100, 161, 250, 400])); H = int(rng.choice([1, 5, 16, 47, 96, 130, 300]))
    P = int(rng.choice([1, 2, 17, 300, 2000, 6000, 20000])); D = int(rng.integers(0, 4))
    inp = scene_inputs(P=P, W=W, H=H, seed=seed, D=D, bg=tuple(rng.uniform(0, 1, 3)),
                       scale_mul=float(rng.choice([0.02, 0.05, 0.5, 1.0, 4.0, 20.0])),
                       opacity_max=float(rng.choice([0.02, 0.3, 1.0])),
                       scale_modifier=float(rng.choice([1.0, 1.0, 0.7, 1.6])), fov_deg=float(rng.uniform(25, 115)))
    kind = seed % 4
    if kind == 1: inp["scales"] = (inp["scales"] * np.array([[8.0, 0.1]], np.float32)).astype(np.float32)
    elif kind == 2: inp["scales"] = (inp["scales"] * np.array([[0.01, 40.0]], np.float32)).astype(np.float32)
    g = cotangents(H, W, seed=seed)
    gc, go = g[0].copy(), g[1].copy()
    pick = seed % 5
    if pick == 0: go[:] = 0
    elif pick == 1: gc[:] = 0; go[1:] = 0
    elif pick == 2: gc[:] = 0; go[:5] = 0; go[6] = 0
    elif pick == 3: gc[:] = 0; go[:6] = 0
    else: gc[:] = 0; go[0:2] = 0; go[5:] = 0
    g = (gc, go)
    o = run_oracle(oracle_mod, inp, g)
    from g4splat_amd import _lib
    for opts in ({"no_fastpath": 1}, {"box_only": 1}, {"bwd_hot_threshold": 1 << 30}, {}):
        for k in ("no_fastpath", "box_only"): _lib.set_option(k, opts.get(k, 0))
        _lib.set_option("bwd_hot_threshold", opts.get("bwd_hot_threshold", _lib.OPTION_UNSET))
        h = run_hip(inp, g)
        d = np.abs(h["grads"]["transMat"].astype(np.float64) - o["grads"]["transMat"])
        print(" options", opts, "transMat max|d|", d.max(), "row", int(d.max(axis=1).argmax()))
    print(f"seed {seed}: P={P} {W}x{H} D={D} kind={kind} pick={pick} R={o['R']}")
    rep = parity_report(h, o, inp, oracle_mod)
    print(" suspects", rep["suspect_pixels"], "flipped", rep["flipped_pixels"], "worst", rep["flipped_worst"], "out_err_unexplained", rep["out_err_unexplained"])
    for n in o["grads"]:
        if n not in h["grads"]: continue
        a, b = h["grads"][n].astype(np.float64), o["grads"][n].astype(np.float64)
        if b.size == 0 or a.size == 0: continue
        d = np.abs(a - b); i = np.unravel_index(d.argmax(), d.shape) if d.size else None
        print(f"  {n:10s} max|o| {np.abs(b).max():.3e} max|d| {d.max():.3e} at {i} o={b[i]:.4e} h={a[i]:.4e}", rep["grads"].get(n, {}).get("rel_unexplained"), (rep.get("grads_masked") or {}).get(n, {}).get("rel"))

        if n == "means2D":
            r = int(i[0])
            print("   row", r, "radii h/o", h["radii"][r], o["radii"][r], "opacity", inp["opacity"][r], "scales", inp["scales"][r])
            for m in ("transMat", "means3D", "opacity", "scales", "rotations", "means2D"):
                if m in h["grads"] and h["grads"][m].size: print("    ", m, "h", h["grads"][m][r], "o", o["grads"][m][r])

    # linearity probe: the backward is linear in the cotangents, rounding noise is not.  g(3 c) / 3 - g(c) on the row
    # with the largest mismatch tells signal from noise.
    from common import hip_backward_again
    r = int(np.abs(h["grads"]["transMat"].astype(np.float64) - o["grads"]["transMat"]).max(axis=1).argmax())
    h3 = hip_backward_again(h, inp, (3.0 * g[0], 3.0 * g[1]))
    o3 = o["oracle"].rasterize_gaussians_backward(3.0 * g[0], 3.0 * g[1])
    print("  row", r, "HIP transMat g(c)     ", h["grads"]["transMat"][r])
    print("  row", r, "HIP transMat g(3c)/3  ", h3["transMat"][r] / 3.0)
    print("  row", r, "oracle transMat g(c)  ", o["grads"]["transMat"][r])
    print("  row", r, "oracle transMat g(3c)/3", o3["transMat"][r] / 3.0)
