#!/usr/bin/env python
"""Measures what the HIP path achieves against the oracle (run on the GPU box): the numbers the guard bars of
tests/common.py are set from.  One line per case: largest output error on pixels that sit on no decision threshold,
tensor-level and row-level gradient errors over unexplained rows, number of threshold flips.

    python tools/parity_report.py [small] [s2] [s3] [s5]     (default: all)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

from common import EMPTY, cotangents, parity_report, run_hip, run_oracle, scene_inputs
from g4splat_amd import synthetic
from oracle import oracle as om


def room_inputs(P, W, H, view, nviews, D=3, bg=(0.3, 0.1, 0.2)):
    scene = synthetic.scene_room(P, seed=0)
    cam = synthetic.room_cameras(nviews, W, H, fovx_deg=90.0)[view]
    return dict(bg=np.array(bg, np.float32), means3D=scene.means3D, colors=EMPTY, opacity=scene.opacities,
                scales=scene.scales, rotations=scene.rotations, scale_modifier=1.0, transMat=EMPTY,
                view=cam.world_view_transform, proj=cam.full_proj_transform, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                H=H, W=W, sh=scene.shs, D=D, campos=cam.camera_center)


def summarize(tag, rep):
    g = rep.get("grads", {})
    line = dict(case=tag, N=rep["N"], R=rep["R"], suspect=rep["suspect_pixels"], flipped=rep["flipped_pixels"],
                flipped_worst=rep["flipped_worst"], id_mismatch=rep["id_mismatch"],
                id_mismatch_unexplained=rep["id_mismatch_unexplained"], out_err_unexplained=rep["out_err_unexplained"],
                out_err_all=rep["out_err_all"], explained_rows=rep.get("explained_rows"),
                grad_rel_unexplained=max([v["rel_unexplained"] for v in g.values()] or [0.0]),
                grad_rel_all=max([v["rel_all"] for v in g.values()] or [0.0]),
                grad_row_rel_unexplained=max([v["row_rel_unexplained"] for v in g.values()] or [0.0]),
                grad_l2=max([v["l2"] for v in g.values()] or [0.0]),
                worst_tensor=max(g, key=lambda k: g[k]["row_rel_unexplained"]) if g else None,
                per_map=[float("%.3g" % x) for x in rep["out_err_per_map_unexplained"]])
    print(json.dumps(line), flush=True)
    return line


def main():
    which = set(sys.argv[1:]) or {"small", "s2", "s3", "s5"}
    worst = dict(out=0.0, grad=0.0, row=0.0)

    def run(tag, inp, gr):
        o = run_oracle(om, inp, gr)
        h = run_hip(inp, gr)
        assert h["R"] == o["R"], tag
        assert np.array_equal(h["radii"], o["radii"]), tag
        line = summarize(tag, parity_report(h, o, inp, om))
        worst["out"] = max(worst["out"], line["out_err_unexplained"])
        worst["grad"] = max(worst["grad"], line["grad_rel_unexplained"])
        worst["row"] = max(worst["row"], line["grad_row_rel_unexplained"])

    if "small" in which:
        for D in range(4):
            inp = scene_inputs(P=10000, W=256, H=256, seed=0, D=D, bg=(0.4, 0.2, 0.9))
            run(f"s1 D={D}", inp, cotangents(256, 256))
        for seed in range(1000, 1060):
            rng = np.random.default_rng(seed)
            W = int(rng.choice([7, 16, 33, 100, 161, 250])); H = int(rng.choice([5, 16, 47, 96, 130]))
            P = int(rng.choice([17, 300, 2000, 6000])); D = int(rng.integers(0, 4))
            inp = scene_inputs(P=P, W=W, H=H, seed=seed, D=D, bg=tuple(rng.uniform(0, 1, 3)),
                               scale_mul=float(rng.choice([0.05, 0.5, 1.0, 4.0, 20.0])),
                               opacity_max=float(rng.choice([0.02, 0.3, 1.0])),
                               scale_modifier=float(rng.choice([1.0, 1.0, 0.7, 1.6])), fov_deg=float(rng.uniform(25, 110)))
            run(f"fuzz {seed} P={P} {W}x{H} D={D}", inp, cotangents(H, W, seed=seed))
    if "s2" in which:
        for view, D in ((0, 3), (2, 0), (4, 3)):
            run(f"s2 view {view} D={D}", room_inputs(300_000, 1200, 680, view, 5, D=D), cotangents(680, 1200, seed=3))
    if "s3" in which:
        for view in (5, 1):
            run(f"s3 view {view}", room_inputs(1_500_000, 1600, 1200, view, 8), cotangents(1200, 1600, seed=3))
    if "s5" in which:
        run("s5 view 0", room_inputs(3_000_000, 1200, 680, 0, 8), cotangents(680, 1200, seed=3))
    print(json.dumps(dict(worst=worst)))


if __name__ == "__main__":
    main()
