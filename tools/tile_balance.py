"""How well can the one-wave blend backward be balanced -- in a slot model with a fixed speed per wave (which the hardware
does not follow: LAB_NOTES section 5 has the measurement that contradicts the 1.2 x this prints for view 0)?  Per-tile cost model of the kernel (111 instructions per visit +
78 per contributing entry, the key the tiles are ordered by) from the forward's contribution masks of a real frame, then a
list-scheduling simulation: tiles in descending order onto `slots` wave slots (what the dispatcher does with the ordered
grid) against the ideal total / slots.        python tools/tile_balance.py [view] [slots]"""
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import hip_state, run_hip  # noqa: E402
from g4splat_amd import synthetic  # noqa: E402

view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
P, W, H = 1_500_000, 1600, 1200
scene = synthetic.scene_room(P, seed=0)
cam = synthetic.room_cameras(8, W, H, fovx_deg=90.0)[view]
E0 = np.zeros(0, np.float32)
inp = dict(bg=np.zeros(3, np.float32), means3D=scene.means3D, colors=E0, opacity=scene.opacities, scales=scene.scales,
           rotations=scene.rotations, scale_modifier=1.0, transMat=E0, view=cam.world_view_transform,
           proj=cam.full_proj_transform, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, H=H, W=W, sh=scene.shs, D=3,
           campos=cam.camera_center)
h = run_hip(inp)
st = hip_state(h, inp)
q = st["qhit"].astype(np.uint8)
visits_per_entry = ((q & 1) + ((q >> 1) & 1) + ((q >> 2) & 1) + ((q >> 3) & 1)).astype(np.int64)
cv = np.concatenate([[0], np.cumsum(visits_per_entry)])
ce = np.concatenate([[0], np.cumsum(q != 0)])
r = st["ranges"].astype(np.int64)
visits = cv[r[:, 1]] - cv[r[:, 0]]
entries = ce[r[:, 1]] - ce[r[:, 0]]
cost = 111 * visits + 78 * entries
order = np.argsort(-cost, kind="stable")
heap = [0] * slots
for c in cost[order]:
    heapq.heapreplace(heap, heap[0] + int(c)) if False else heapq.heappush(heap, heapq.heappop(heap) + int(c))
makespan, total = max(heap), int(cost.sum())
print(f"S3 view {view}: {len(cost)} tiles, cost (instructions) mean {cost.mean():.0f}, median {np.median(cost):.0f}, "
      f"p90 {np.quantile(cost, 0.9):.0f}, p99 {np.quantile(cost, 0.99):.0f}, max {cost.max()}")
print(f"{slots} slots: ideal {total / slots:.0f}, list schedule in descending order {makespan} = {makespan / (total / slots):.3f} x ideal; "
      f"heaviest tile alone {cost.max() / (total / slots):.3f} x ideal")
for s2 in (5120, 6144, 8192):
    hp = [0] * s2
    for c in cost[order]:
        heapq.heappush(hp, heapq.heappop(hp) + int(c))
    print(f"{s2} slots: {max(hp) / (total / s2):.3f} x ideal")


def schedule(jobs, n):
    hp = [0.0] * n
    for c in jobs:
        heapq.heappush(hp, heapq.heappop(hp) + c)
    return max(hp)


# tiles above a cost threshold handed to the four-wave kernel (1.9 x the work, four waves: 0.475 x the time on four slots),
# started first and running BESIDE the one-wave kernel
ideal = total / slots
for frac in (1.0, 0.9, 0.8, 0.7, 0.6, 0.5):
    thr = frac * ideal
    hot = cost > thr
    jobs = []
    for c in np.sort(cost[hot])[::-1]:
        jobs += [0.475 * c] * 4
    jobs += list(np.sort(cost[~hot])[::-1].astype(float))
    ms = schedule(jobs, slots)
    print(f"threshold {frac:.1f} x ideal: {int(hot.sum())} hot tiles ({hot.mean() * 100:.1f} %), makespan {ms / ideal:.3f} x today's ideal "
          f"(work {(1.9 * cost[hot].sum() + cost[~hot].sum()) / total:.3f} x)")
