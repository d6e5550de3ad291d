"""How the (entry, quadrant) work of the blend kernels pairs up -- sizing of the packed-FP32 variants (run on the GPU box).

Reads the forward's qhit bytes (which 8x8 quadrants of its tile blended an instance) for one view of a workload and prints
  * the histogram of the sixteen hit patterns and the mean number of quadrants per contributing entry,
  * pixel-pair packing: quadrants paired (0,1),(2,3) [horizontal neighbours] or (0,2),(1,3) [vertical]: how many
    visits would be packed (both quadrants hit) and how many stay single,
  * entry-pair packing: consecutive contributing entries of a tile (back to front, as the backward stages them) paired:
    per quadrant, pairs where both entries hit it and pairs where one does.
    python tools/qhit_stats.py [s3|s2|s5|s1] [view]
"""
import sys
import numpy as np
import torch

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import hip_state
import bench
from g4splat_amd.diff_surfel_rasterization import _C

wl = sys.argv[1] if len(sys.argv) > 1 else "s3"
view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
scene, cams, d, dcams, (P, W, H, D) = bench.build_scene(wl, dev)
c = dcams[view % len(dcams)]
empty = torch.empty(0, device=dev); bg = torch.zeros(3, device=dev)
f = _C.rasterize_gaussians(bg, d["means3D"], empty, d["opacity"], d["scales"], d["rotations"], 1.0, empty, c["view"], c["proj"],
                           c["tanfovx"], c["tanfovy"], H, W, d["sh"], D, c["campos"], False, False)
st = hip_state(dict(R=f[0], geom=f[4], binning=f[5], img=f[6]), dict(means3D=scene.means3D, W=W, H=H))
qh = st["qhit"].astype(np.uint8)
nz = qh[qh != 0]
hist = np.bincount(nz, minlength=16)
bits = np.array([bin(i).count("1") for i in range(16)])
print("workload %s view %d: binned %d, contributing %d (%.1f %%), quadrants per contributing entry %.3f" %
      (wl, view, len(qh), len(nz), 100.0 * len(nz) / max(len(qh), 1), (hist * bits).sum() / max(len(nz), 1)))
print("pattern histogram (bit q = quadrant q; q0 = top-left, q1 = top-right, q2 = bottom-left, q3 = bottom-right):")
for i in range(1, 16):
    print("   %s %9d  %5.1f %%" % (format(i, "04b"), hist[i], 100.0 * hist[i] / len(nz)))
visits = int((hist * bits).sum())
for name, pairs in (("horizontal (0,1),(2,3)", ((0, 1), (2, 3))), ("vertical (0,2),(1,3)", ((0, 2), (1, 3))),
                    ("diagonal (0,3),(1,2)", ((0, 3), (1, 2)))):
    packed = single = 0
    for pat in range(1, 16):
        for a, b in pairs:
            ha, hb = (pat >> a) & 1, (pat >> b) & 1
            packed += hist[pat] * (ha & hb)
            single += hist[pat] * (ha ^ hb)
    print("pixel pairs %-24s packed visits %9d (cover %.1f %% of the %d quadrant visits), single %9d" %
          (name, packed, 200.0 * packed / visits, visits, single))
# entry pairs: per tile, contributing entries back to front, batches of 48 list positions as the backward stages them
r0 = st["ranges"][:, 0].astype(np.int64); r1 = st["ranges"][:, 1].astype(np.int64)
both = one = 0
rng = np.random.default_rng(0)
tiles = rng.choice(len(r0), size=min(len(r0), 1500), replace=False)
vis_s = 0
for t in tiles:
    q = qh[r0[t]:r1[t]][::-1]
    q = q[q != 0]
    if len(q) == 0:
        continue
    if len(q) & 1:
        q = np.concatenate([q, [0]])
    a, b = q[0::2], q[1::2]
    both += int(np.unpackbits((a & b).astype(np.uint8)[:, None], axis=1).sum())
    one += int(np.unpackbits((a ^ b).astype(np.uint8)[:, None], axis=1).sum())
    vis_s += int(np.unpackbits(q.astype(np.uint8)[:, None], axis=1).sum())
print("entry pairs (sample of %d tiles): quadrant visits %d -> packed %d (cover %.1f %%), single %d" %
      (len(tiles), vis_s, both, 200.0 * both / max(vis_s, 1), one))
