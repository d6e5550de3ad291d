"""distCUDA2 (HIP) against the brute-force oracle over many point distributions and sizes.
    python tools/knn_sweep.py [count]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.oracle as oracle_mod  # noqa: E402
from g4splat_amd.simple_knn._C import distCUDA2  # noqa: E402

count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for seed in range(count):
    rng = np.random.default_rng(seed)
    P = int(rng.choice([1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 3000, 20000, 70000]))
    kind = seed % 6
    if kind == 0:
        pts = rng.normal(size=(P, 3))
    elif kind == 1:  # a few tight clusters far apart
        c = rng.normal(size=(5, 3)) * 1000
        pts = c[rng.integers(0, 5, P)] + rng.normal(size=(P, 3)) * 1e-3
    elif kind == 2:  # many exact duplicates
        base = rng.normal(size=(max(P // 7, 1), 3))
        pts = base[rng.integers(0, len(base), P)]
    elif kind == 3:  # on a line / in a plane
        pts = np.zeros((P, 3))
        pts[:, 0] = rng.uniform(-5, 5, P)
        if seed % 12 == 3:
            pts[:, 1] = rng.uniform(-5, 5, P)
    elif kind == 4:  # huge dynamic range
        pts = rng.normal(size=(P, 3)) * np.exp(rng.uniform(-8, 8, (P, 1)))
    else:  # regular grid (ties everywhere)
        n = max(int(round(P ** (1 / 3))), 1)
        g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3)
        pts = g[:P] if len(g) >= P else np.concatenate([g, g[: P - len(g)]])
        P = len(pts)
    pts = pts.astype(np.float32)
    got = distCUDA2(torch.as_tensor(pts, device="cuda")).cpu().numpy()
    want = oracle_mod.distCUDA2(pts)
    if not np.array_equal(got, want):
        bad += 1
        d = np.abs(got.astype(np.float64) - want)
        print(f"MISMATCH seed {seed} kind {kind} P={P}: {int((got != want).sum())} of {P} differ, worst abs {d.max():.3e} rel {np.nanmax(d / np.maximum(np.abs(want), 1e-30)):.3e}", flush=True)
print(f"done: {count} point sets, {bad} mismatches")
