"""Diagnose fuzz_sweep mismatches: python tools/fuzz_debug.py seed [seed ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import cotangents, hip_state, run_hip, run_oracle, scene_inputs  # noqa: E402
import oracle.oracle as oracle_mod  # noqa: E402
from g4splat_amd import _lib  # noqa: E402

for seed in [int(x) for x in sys.argv[1:]]:
    rng = np.random.default_rng(seed)
    W = int(rng.choice([1, 7, 16, 33, 100, 161, 250, 400]))
    H = int(rng.choice([1, 5, 16, 47, 96, 130, 300]))
    P = int(rng.choice([1, 2, 17, 300, 2000, 6000, 20000]))
    D = int(rng.integers(0, 4))
    inp = scene_inputs(P=P, W=W, H=H, seed=seed, D=D, bg=tuple(rng.uniform(0, 1, 3)),
                       scale_mul=float(rng.choice([0.02, 0.05, 0.5, 1.0, 4.0, 20.0])),
                       opacity_max=float(rng.choice([0.02, 0.3, 1.0])),
                       scale_modifier=float(rng.choice([1.0, 1.0, 0.7, 1.6])), fov_deg=float(rng.uniform(25, 115)))
    kind = seed % 4
    if kind == 1:
        inp["scales"] = (inp["scales"] * np.array([[8.0, 0.1]], np.float32)).astype(np.float32)
    elif kind == 2:
        inp["scales"] = (inp["scales"] * np.array([[0.01, 40.0]], np.float32)).astype(np.float32)
    scene_scale = float(os.environ.get("FUZZ_SCENE_SCALE", "1"))
    if scene_scale != 1.0:
        inp["means3D"] = (inp["means3D"] * scene_scale).astype(np.float32)
        inp["scales"] = (inp["scales"] * scene_scale).astype(np.float32)
    g = cotangents(H, W, seed=seed)
    o = run_oracle(oracle_mod, inp, g)
    print(f"seed {seed}: P={P} {W}x{H} D={D} kind={kind} R={o['R']} scale_modifier={inp['scale_modifier']}")
    for env in ({}, {"box_only": 1}, {"no_fastpath": 1}):
        for k in ("box_only", "no_fastpath"):
            _lib.set_option(k, env.get(k, 0))
        h = run_hip(inp, g)
        d = np.abs(h["color"] - o["color"]).max(axis=0)
        do = np.abs(h["others"] - o["others"]).max(axis=0)
        nbad = int((d > 1e-4).sum())
        y, x = np.unravel_index(np.argmax(d), d.shape)
        st = hip_state(h, inp)
        last_h = st["n_contrib"][0].reshape(H, W)[y, x]
        last_o = o["oracle"].state("n_contrib").reshape(2, H, W)[0][y, x]
        per_map = [float(np.abs(h["others"][c] - o["others"][c]).max()) for c in range(7)]
        print("   others per map (depth, alpha, n0, n1, n2, median, distortion):", ["%.2e" % v for v in per_map],
              "max |depth|", float(np.abs(o["others"][0]).max()), "max |median|", float(np.abs(o["others"][5]).max()))
        print(f"   {env or 'default'}: {nbad} pixels beyond 1e-4, worst {d.max():.3e} at ({x},{y}) others worst {do.max():.3e}; "
              f"alpha hip {h['others'][1][y, x]:.6f} oracle {o['others'][1][y, x]:.6f}; last contributor hip(list pos) {last_h} oracle {last_o}")
    for k in ("box_only", "no_fastpath"):
        _lib.set_option(k, 0)
