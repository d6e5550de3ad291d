"""Lane-utilisation model of the blend backward (verdict r3 item 6): on a real frame of a workload, how many wave visits
would different STATIC work units need, and what share of the lanes would do useful work?  Pure CPU (the oracle's forward
+ oracle_lane_model); run it on the GPU box for its 256 host cores:

    python tools/lane_util_model.py [s3|s2|s5] [view ...]  > profiles/r04_lane_util_model.txt

Cost model of the shipped kernel (LAB_NOTES.md): 111 VALU instructions per (entry, unit) visit + 78 per entry
(18 accumulators, the 64-lane reduction of 16 terms, the record)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as om  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "s3"
views = [int(v) for v in sys.argv[2:]] or [0, 6]
import torch  # noqa: E402
scene, cams, _d, _dc, (P, W, H, D) = bench.build_scene(wl, torch.device("cpu"))
e = np.zeros((0,), np.float32)
print(f"workload {wl}: {P} surfels, {W}x{H}; cost model 111 instructions per visit + 78 per entry")
for v in views:
    cam = cams[v % len(cams)]
    o = om.Oracle()
    t0 = time.perf_counter()
    R, _, _, radii = o.rasterize_gaussians(np.zeros(3, np.float32), scene.means3D, e, scene.opacities, scene.scales,
                                           scene.rotations, 1.0, e, cam.world_view_transform, cam.full_proj_transform,
                                           cam.tanfovx, cam.tanfovy, H, W, scene.shs, D, cam.camera_center)
    m = om.lane_model(o)
    dt = time.perf_counter() - t0
    E, pairs = m["entries"], m["pairs"]
    base = 111 * m["visits_quadrants"] + 78 * E
    print(f"\nview {v}: {int((radii > 0).sum())} visible, {R} instances, {int(E)} entries blended by some pixel, "
          f"{int(pairs)} blending (entry, pixel) pairs ({pairs / E:.1f} pixels per entry)   [{dt:.1f} s]")
    print(f"  {'work unit':58s} {'visits':>10s} {'per entry':>9s} {'x shipped':>9s} {'useful lanes':>12s} {'model time':>10s}")

    def row(name, visits, feasible=True, extra_per_visit=0):
        t = (111 + extra_per_visit) * visits + 78 * E
        print(f"  {name:58s} {int(visits):10d} {visits / E:9.2f} {visits / m['visits_quadrants']:9.2f} "
              f"{100 * pairs / (64 * visits):11.1f}% {t / base:10.3f}" + ("" if feasible else "   (not buildable, see note)"))

    row("8x8 quadrants (shipped)", m["visits_quadrants"])
    row("16x4 row strips (lane = same position in each strip)", m["visits_row_strips"])
    row("4x16 column strips", m["visits_col_strips"])
    row("8x4 halves, each half-wave picks its quadrant freely", m["visits_free_halves"], feasible=False)
    row("4x4 cells, each 16-lane row picks its quadrant freely", m["visits_free_cells"], feasible=False)
    row("two consecutive entries with disjoint masks share a visit", m["visits_quadrants"] - m["visits_saved_by_entry_pairing"],
        extra_per_visit=0)
    row("perfect packing (lower bound: ceil(pixels / 64) per entry)", m["visits_lower_bound"], feasible=False)
    print(f"  quadrant visits whose blending lanes fit one 8x4 half: {100 * m['visits_in_one_half'] / m['visits_quadrants']:.1f} %, "
          f"one 16-lane row (8x2): {100 * m['visits_in_one_row'] / m['visits_quadrants']:.1f} %  (worth nothing: the SIMD does not "
          f"skip masked-off rows or halves, tools/micro/exec_rows.hip)")
    print(f"  quadrant visits with <= 16 blending lanes: {100 * m['visits_le16_lanes'] / m['visits_quadrants']:.1f} % "
          f"(they hold {100 * m['pairs_in_le16'] / pairs:.1f} % of the blending pixels), <= 8 lanes: "
          f"{100 * m['visits_le8_lanes'] / m['visits_quadrants']:.1f} %  -- gfx950 runs a VALU instruction with <= 16 enabled lanes in a "
          f"slow mode (profiles/r04_exec_lane_threshold.txt)")
print("""
note: a lane's per-pixel state (T, the two recurrences, nine cotangents: 12 registers per pixel, 4 pixels per lane) is
addressed by STATIC register names; a visit in which the two half-waves (or the four rows) work on different quadrants
would need each instruction to name a different register in different lanes.  Those rows are upper bounds of what a
finer unit could give, not designs.""")
