"""Extended fuzz of the HIP rasterizer against the CPU oracle (same generator as tests/test_gpu_parity.py::
test_fuzz_small_scenes, other seeds, plus needle / hair splats), both backward kernels.  Not part of the test suite:
a robustness sweep to run when the kernels change.

    python tools/fuzz_sweep.py [first_seed] [count] [kind]
FUZZ_BIG=1: frames of 640x360 .. 1601x1203 with 50 k .. 300 k Gaussians (the same random generator).
Criterion: tests/common.py::assert_parity (guard bars + threshold-margin proof); FUZZ_CONTRACT_BARS=1 or FUZZ_COT=1
(sparse cotangents) fall back to the north-star bars (1e-4 / 1e-3)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import cotangents, hip_state, run_hip, run_oracle, scene_inputs  # noqa: E402
from common import GRAD_RTOL, OUT_ATOL, assert_parity, rel_err  # noqa: E402
from test_gpu_parity import check_lists_against_oracle  # noqa: E402
import oracle.oracle as oracle_mod  # noqa: E402
from g4splat_amd import _lib  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    W = int(rng.choice([1, 7, 16, 33, 100, 161, 250, 400]))
    H = int(rng.choice([1, 5, 16, 47, 96, 130, 300]))
    P = int(rng.choice([1, 2, 17, 300, 2000, 6000, 20000]))
    if os.environ.get("FUZZ_BIG"):  # mid-size frames: thousands of tiles, lists hundreds deep (between the small fuzz and S2)
        W = int(rng.choice([640, 1000, 1333, 1601]))
        H = int(rng.choice([360, 480, 750, 1203]))
        P = int(rng.choice([50_000, 120_000, 300_000]))
    D = int(rng.integers(0, 4))
    inp = scene_inputs(P=P, W=W, H=H, seed=seed, D=D, bg=tuple(rng.uniform(0, 1, 3)),
                       # (FUZZ_BIG: splats of the small fuzz's upper sizes at 300 k Gaussians give tile lists 87 000 deep --
                       # 5e8 instances and minutes per scene; keep the lists in the hundreds)
                       scale_mul=float(rng.choice([0.02, 0.05, 0.2, 0.5] if os.environ.get("FUZZ_BIG") else
                                                  [0.02, 0.05, 0.5, 1.0, 4.0, 20.0])),
                       opacity_max=float(rng.choice([0.02, 0.3, 1.0])),
                       scale_modifier=float(rng.choice([1.0, 1.0, 0.7, 1.6])), fov_deg=float(rng.uniform(25, 115)))
    kind = seed % 4 if len(sys.argv) <= 3 else int(sys.argv[3])  # optional third argument: force the splat shape
    if kind == 1:
        inp["scales"] = (inp["scales"] * np.array([[8.0, 0.1]], np.float32)).astype(np.float32)
    elif kind == 2:
        inp["scales"] = (inp["scales"] * np.array([[0.01, 40.0]], np.float32)).astype(np.float32)
    scene_scale = float(os.environ.get("FUZZ_SCENE_SCALE", "1"))  # uniform scaling of the scene about the camera
    if scene_scale != 1.0:
        inp["means3D"] = (inp["means3D"] * scene_scale).astype(np.float32)
        inp["scales"] = (inp["scales"] * scene_scale).astype(np.float32)
    if os.environ.get("FUZZ_PRECOMP"):  # colours and / or the T matrices handed in precomputed (no SH, no scale + rotation)
        o0 = run_oracle(oracle_mod, inp)
        which = seed % 3
        if which in (0, 2):
            inp["colors"] = rng.uniform(0, 1, (P, 3)).astype(np.float32)
            inp["sh"] = np.zeros((0,), np.float32)
        if which in (1, 2):
            inp["transMat"] = o0["oracle"].state("transMat").copy()
            inp["scales"] = np.zeros((0,), np.float32)
            inp["rotations"] = np.zeros((0,), np.float32)
    g = cotangents(H, W, seed=seed)
    cot = os.environ.get("FUZZ_COT", "")  # sparse cotangents: only some of the ten output maps carry a gradient
    if cot:
        gc, go = g[0].copy(), g[1].copy()
        pick = seed % 5
        if pick == 0: go[:] = 0                      # colour loss only
        elif pick == 1: gc[:] = 0; go[1:] = 0        # depth only
        elif pick == 2: gc[:] = 0; go[:5] = 0; go[6] = 0   # median depth only
        elif pick == 3: gc[:] = 0; go[:6] = 0        # distortion only
        else: gc[:] = 0; go[0:2] = 0; go[5:] = 0     # normals only
        g = (gc, go)
    o = run_oracle(oracle_mod, inp, g)
    for mode in ("policy", "one-wave"):
        _lib.set_option("bwd_hot_threshold", (1 << 30) if mode == "one-wave" else _lib.OPTION_UNSET)
        h = run_hip(inp, g)
        tag = f"seed {seed} {mode}: P={P} {W}x{H} D={D} kind={kind}"
        try:
            if not cot and not os.environ.get("FUZZ_CONTRACT_BARS"):
                # default criterion: the test suite's -- guard bars (~10x the measured error) on everything that is not
                # explained by a decision threshold within 1e-5 of its value in the oracle (tests/common.py)
                assert_parity(h, o, inp, oracle_mod, tag=tag, scale_aware=scene_scale != 1.0)
                # (the list check walks every tile in Python: minutes per mid-size frame; the suite does it at S2 / S3 size)
                if o["R"] > 0 and mode == "policy" and not os.environ.get("FUZZ_BIG"):
                    check_lists_against_oracle(hip_state(h, inp), o["oracle"], oracle_mod)
                continue
            assert h["R"] == o["R"], "R"
            assert np.array_equal(h["radii"], o["radii"]), "radii"
            assert np.abs(h["color"] - o["color"]).max() <= OUT_ATOL, "color"
            assert np.abs(h["others"] - o["others"]).max() <= OUT_ATOL, "others"
            if o["R"] > 0 and mode == "policy":
                check_lists_against_oracle(hip_state(h, inp), o["oracle"], oracle_mod)
            for name in ("means3D", "scales", "rotations", "opacity", "sh", "colors", "transMat", "means2D"):
                if o["grads"][name].size == 0 or h["grads"][name].size == 0:
                    assert o["grads"][name].size == h["grads"][name].size or name in ("sh", "scales", "rotations", "colors"), "grad size " + name
                    continue
                if np.abs(o["grads"][name]).max() == 0:
                    assert np.abs(h["grads"][name]).max() == 0, "grad (expected zero) " + name
                    continue
                if cot:
                    # With a single map's cotangent some gradients are mathematically zero (the median depth of a
                    # ray-plane intersection does not depend on the in-plane scales, ...): both sides then return the
                    # rounding noise of cancelling terms.  Judge against the frame's overall gradient scale as well.
                    S = max(float(np.abs(o["grads"][n]).max()) for n in ("means3D", "transMat", "opacity") if o["grads"][n].size)
                    d = float(np.abs(h["grads"][name].astype(np.float64) - o["grads"][name]).max())
                    # (distortion of one or two splats, median depth w.r.t. scales: exactly zero in exact arithmetic --
                    # what is left is the random walk of ~H W roundings of O(|cotangent|) terms)
                    floor = 1e-7 * np.sqrt(H * W) * max(float(np.abs(g[0]).max()), float(np.abs(g[1]).max()))
                    # scale / rotation gradients are dL_dT pushed through the projection (entries ~ the focal length in
                    # pixels): for hair-thin splats (scale 1e-5) the true value is a difference of terms ~ S * focal
                    cond = 4e-6 * S * max(W, H) if name in ("scales", "rotations") else 0.0
                    assert d <= GRAD_RTOL * float(np.abs(o["grads"][name]).max()) + 1e-4 * S + floor + cond, "grad " + name
                    continue
                assert rel_err(h["grads"][name], o["grads"][name]) <= GRAD_RTOL, "grad " + name
        except AssertionError as ex:
            # a mismatch that survives box_only is a threshold flip of the per-pixel arithmetic (one contributor at
            # alpha ~ 1/255 or T ~ 1e-4 decided the other way: ~1e-7 of the pixels); one that disappears is a culling bug
            with _lib.option("box_only", 1):
                hb = run_hip(inp, g)
            cured = (np.abs(hb["color"] - o["color"]).max() <= OUT_ATOL and np.abs(hb["others"] - o["others"]).max() <= OUT_ATOL)
            npx = int((np.abs(h["color"] - o["color"]).max(axis=0) > OUT_ATOL).sum())
            kind_s = "CULLING BUG" if cured and str(ex) in ("color", "others") else "threshold flip" if str(ex) in ("color", "others") else "other"
            bad.append((tag, str(ex), kind_s, npx))
            print("MISMATCH", tag, ex, kind_s, f"{npx} pixels", flush=True)
    if os.environ.get("FUZZ_BIG") or (seed - first) % 100 == 99:
        print(f"{seed - first + 1} scenes, {len(bad)} mismatches", flush=True)
_lib.set_option("bwd_hot_threshold", _lib.OPTION_UNSET)
print(f"done: {count} scenes x 2 backward kernels, {len(bad)} mismatches")
for b in bad:
    print(" ", b)
