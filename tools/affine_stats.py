#!/usr/bin/env python
"""Share of the splats / binned instances that carry REC_AFFINE (csrc/g4s_device.h) in a frame (run on the GPU box):
    python tools/affine_stats.py [s1] [s2] [s3] [s5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np

from common import hip_state, run_hip, scene_inputs
from parity_report import room_inputs


def report(tag, inp):
    h = run_hip(inp)
    st = hip_state(h, inp)
    vis = h["radii"] > 0
    aff = (st["rec_u32"][:, 3] >> 31).astype(bool) & vis
    tt = st["tiles_touched"].astype(np.int64)
    binned = tt > 0
    print(f"{tag}: visible {int(vis.sum())}, binned {int(binned.sum())}, affine {int(aff.sum())} "
          f"= {aff.sum() / max(1, binned.sum()):.4f} of the binned splats, {tt[aff].sum() / max(1, tt.sum()):.4f} of the instances; "
          f"binned into ONE tile: {(tt == 1).sum() / max(1, binned.sum()):.4f} of the binned splats, into <= 4: "
          f"{((tt > 0) & (tt <= 4)).sum() / max(1, binned.sum()):.4f}; instances per binned splat {tt.sum() / max(1, binned.sum()):.1f}", flush=True)


which = set(sys.argv[1:]) or {"s1", "s2", "s3"}
if "s1" in which:
    report("s1", scene_inputs(P=10000, W=256, H=256, seed=0, D=3))
if "s2" in which:
    for v in (0, 2):
        report(f"s2 view {v}", room_inputs(300_000, 1200, 680, v, 5))
if "s3" in which:
    for v in (0, 6):
        report(f"s3 view {v}", room_inputs(1_500_000, 1600, 1200, v, 8))
if "s5" in which:
    report("s5 view 0", room_inputs(3_000_000, 1200, 680, 0, 8))
