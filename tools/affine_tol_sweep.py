#!/usr/bin/env python
"""What a given AFFINE_TOL (csrc/g4s_device.h: the bound the REC_AFFINE certificate puts on the affine form's alpha)
costs and buys -- for the library that is currently installed (run on the GPU box; tools/rounds/r06_affine_tol.sh installs
the variants one after the other):

    python tools/affine_tol_sweep.py <margin> [s1] [s3] [s3t]

One JSON line per frame: share of the binned instances certified REC_AFFINE; the gate's figures with MARGIN = <margin>
(pixels within the margin of a threshold, flipped pixels, worst output error on pixels on no threshold, worst distance
of a flipped pixel to the oracle's alternative, gradient errors over unexplained rows); the contract-literal figures
(pixels beyond 1e-4 abs against the oracle, worst of them); and the same library against ITSELF with option
no_fastpath (every pair through the reference's arithmetic): the largest output difference over all pixels and over
the pixels that sit on no threshold -- what the affine form changes, measured directly."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np

import common
from common import EMPTY, cotangents, hip_state, parity_report, run_hip, run_oracle, scene_inputs
from g4splat_amd import _lib, synthetic
from oracle import oracle as om
from parity_report import room_inputs

margin = float(sys.argv[1])
which = set(sys.argv[2:]) or {"s1", "s3", "s3t"}
common.MARGIN = margin
S3T_CACHE = "/tmp/g4s_s3t_scene.npz"


def s3t_inputs(view):
    """The trained scene is a function of the library that trained it; the sweep evaluates every variant on ONE scene,
    trained by the first library it runs with and kept in /tmp for the others."""
    if os.path.exists(S3T_CACHE):
        d = np.load(S3T_CACHE)
        sc = synthetic.Scene(means3D=d["means3D"], scales=d["scales"], rotations=d["rotations"], opacities=d["opacities"], shs=d["shs"])
    else:
        from g4splat_amd import trained_scene
        sc, _info = trained_scene.scene_trained(seed=0, iters=1000, P=1_500_000, width=1600, height=1200)
        np.savez(S3T_CACHE, means3D=sc.means3D, scales=sc.scales, rotations=sc.rotations, opacities=sc.opacities, shs=sc.shs)
    cam = synthetic.room_cameras(8, 1600, 1200, fovx_deg=90.0)[view]
    return dict(bg=np.array((0.3, 0.1, 0.2), np.float32), means3D=sc.means3D, colors=EMPTY, opacity=sc.opacities,
                scales=sc.scales, rotations=sc.rotations, scale_modifier=1.0, transMat=EMPTY, view=cam.world_view_transform,
                proj=cam.full_proj_transform, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, H=1200, W=1600, sh=sc.shs, D=3,
                campos=cam.camera_center)


def frame(tag, inp, seed=3):
    gr = cotangents(inp["H"], inp["W"], seed=seed)
    o = run_oracle(om, inp, gr)
    h = run_hip(inp, gr)
    st = hip_state(h, inp)
    vis = h["radii"] > 0
    aff = (st["rec_u32"][:, 3] >> 31).astype(bool) & vis
    tt = st["tiles_touched"].astype(np.int64)
    rep = parity_report(h, o, inp, om)
    N = rep["N"]
    d = np.concatenate([np.abs(h["color"] - o["color"]), np.abs(h["others"] - o["others"])], 0).reshape(10, N).max(axis=0)
    with _lib.option("no_fastpath", 1):
        g = run_hip(inp)
    dg = np.concatenate([np.abs(h["color"] - g["color"]), np.abs(h["others"] - g["others"])], 0).reshape(10, N).max(axis=0)
    _p, _g, margins = om.skip_suspects(o["oracle"], margin, with_margins=True)
    calm = margins.min(axis=0) >= margin
    gg = rep.get("grads", {})
    gm = rep.get("grads_masked") or {}
    print(json.dumps(dict(
        frame=tag, build=_lib.load().g4s_version().decode(), margin=margin, N=N, R=rep["R"],
        affine_share_of_instances=round(float(tt[aff].sum() / max(1, tt.sum())), 4),
        pixels_within_margin=rep["suspect_pixels"], flipped=rep["flipped_pixels"], flipped_worst=rep["flipped_worst"],
        flipped_alt_err=rep["flipped_alt_err"], flipped_unmatched=rep["flipped_unmatched"],
        out_err_off_threshold=rep["out_err_unexplained"], id_mismatch_off_threshold=rep["id_mismatch_unexplained"],
        grad_rel_unexplained=max([v["rel_unexplained"] for v in gg.values()] or [0.0]),
        grad_row_rel_unexplained=max([v["row_rel_unexplained"] for v in gg.values()] or [0.0]),
        grad_rel_masked_all_rows=max([v["rel"] for v in gm.values()] or [0.0]),
        grad_row_rel_masked_all_rows=max([v["row_rel"] for v in gm.values()] or [0.0]),
        skip_suspect_pairs=rep.get("skip_suspect_pairs"), explained_rows=rep.get("explained_rows"),
        pixels_beyond_1e_4=int((d > 1e-4).sum()), worst_abs=float(d.max()),
        vs_no_fastpath_all=float(dg.max()), vs_no_fastpath_off_threshold=float(dg[calm].max()) if calm.any() else 0.0,
        vs_no_fastpath_pixels_beyond_2e_5=int((dg > 2e-5).sum()))), flush=True)


if "s1" in which:
    frame("s1 D=3", scene_inputs(P=10000, W=256, H=256, seed=0, D=3, bg=(0.4, 0.2, 0.9)), seed=1)
if "s3" in which:
    for v in (5, 0):
        frame(f"s3 view {v}", room_inputs(1_500_000, 1600, 1200, v, 8))
if "s3t" in which:
    frame("s3t view 2", s3t_inputs(2))
