"""Fraction of the Gaussians visible on SOME rank in one step of bench.py's weak-scaling view assignment (rank r renders view
(r + i N) % 8 in step i), for N = 1, 2, 4, 8 -- what decides whether the fallback exchange (parallel.RowSparseAllReduce) packs
rows or all-reduces the whole bucket, and how many rows the owner exchange's all_to_all carries.
    python tools/union_fractions.py [s3|s2|s5]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from g4splat_amd.diff_surfel_rasterization import _C  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "s3"
dev = torch.device("cuda", 0)
scene, cams, d, dcams, (P, W, H, D) = bench.build_scene(wl, dev)
e, bg = torch.empty(0, device=dev), torch.zeros(3, device=dev)
vis = []
for c in dcams:
    f = _C.rasterize_gaussians(bg, d["means3D"], e, d["opacity"], d["scales"], d["rotations"], 1.0, e, c["view"], c["proj"],
                               c["tanfovx"], c["tanfovy"], H, W, d["sh"], D, c["campos"], False, False)
    vis.append(f[3] > 0)
print(f"{wl}: P = {P}; visible per view: " + ", ".join(f"{float(v.float().mean()):.3f}" for v in vis))
for N in (1, 2, 4, 8):
    fr = []
    for i in range(8):
        u = torch.zeros(P, dtype=torch.bool, device=dev)
        for r in range(N):
            u |= vis[(r + i * N) % len(vis)]
        fr.append(float(u.float().mean()))
    print(f"N = {N}: union of the step's views over 8 steps: min {min(fr):.3f} mean {sum(fr) / len(fr):.3f} max {max(fr):.3f}")
