"""How many binned (tile, Gaussian) instances can reach NO pixel of their tile (run on the GPU box).

The emit kernel bins by the alpha-cutoff bounding box; the blend kernels then test the exact region (low-pass disk +
cutoff ellipse) per 8x8 quadrant.  This restates that test (blend.hip quad_misses_box / quad_misses_region) in numpy on the
forward's records and lists and prints, for one view, the share of binned instances whose region misses all four
quadrants of their tile, next to the share the forward actually blended (qhit != 0).
    python tools/relmask_stats.py [s3|s2|s5|s1] [view]
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
ent = st["entries"]
tile = (ent >> np.uint64(32)).astype(np.int64)
idx = (ent & np.uint64(0xFFFFFFFF)).astype(np.int64)
tiles_x = (W + 15) // 16
rec = st["rec"][idx].astype(np.float32)
f32 = np.float32


def region_miss(qx, qy, size):
    """blend.hip quad_misses_box || quad_misses_region for the square [qx, qx+size] x [qy, qy+size] (float32 arithmetic)."""
    x1, y1 = qx + f32(size), qy + f32(size)
    bw = rec[:, 20:24].copy().view(np.uint32)  # bounds in 8x8-pixel quadrants: x0, x1, -, y0 | y1 << 16 (g4s_internal.h)
    bx0, bx1, by0, by1 = (8 * bw[:, 0]).astype(f32), (8 * bw[:, 1] + 7).astype(f32), (8 * (bw[:, 3] & 0xFFFF)).astype(f32), (8 * (bw[:, 3] >> 16) + 7).astype(f32)
    miss_box = (bx0 > x1) | (bx1 < qx) | (by0 > y1) | (by1 < qy)
    cx, cy = rec[:, 0], rec[:, 1]
    ddx = np.maximum(np.maximum(qx - cx, cx - x1), f32(0)); ddy = np.maximum(np.maximum(qy - cy, cy - y1), f32(0))
    in_disk = ddx * ddx + ddy * ddy <= rec[:, 30]
    ex, ey, ux, uy, ia, ib = (rec[:, 24 + i] for i in range(6))
    no_ell = ia == 0
    ax0, ax1, ay0, ay1 = qx - ex, x1 - ex, qy - ey, y1 - ey
    inside = (ax0 <= 0) & (ax1 >= 0) & (ay0 <= 0) & (ay1 >= 0)
    with np.errstate(all="ignore"):
        m11 = ia * ux * ux + ib * uy * uy; m22 = ia * uy * uy + ib * ux * ux; m12 = (ia - ib) * ux * uy
        r22, r11 = m12 / m22, m12 / m11

        def form(dx, dy):
            t1 = dx * ux + dy * uy; t2 = dy * ux - dx * uy
            return ia * t1 * t1 + ib * t2 * t2
        f0 = form(ax0, np.clip(-r22 * ax0, ay0, ay1)); f1 = form(ax1, np.clip(-r22 * ax1, ay0, ay1))
        f2 = form(np.clip(-r11 * ay0, ax0, ax1), ay0); f3 = form(np.clip(-r11 * ay1, ax0, ax1), ay1)
        miss_ell = np.minimum(np.minimum(f0, f1), np.minimum(f2, f3)) > 1
    miss_region = ~in_disk & ~no_ell & ~inside & miss_ell
    return miss_box | miss_region, miss_box


tx = (tile % tiles_x).astype(np.float32) * f32(16); ty = (tile // tiles_x).astype(np.float32) * f32(16)
relmask = np.zeros(len(ent), np.uint8); boxmask = np.zeros(len(ent), np.uint8)
for q in range(4):
    m, mb = region_miss(tx + f32((q & 1) * 8), ty + f32((q >> 1) * 8), 7)
    relmask |= (~m).astype(np.uint8) << q; boxmask |= (~mb).astype(np.uint8) << q
tile_miss, _ = region_miss(tx, ty, 15)
qh = st["qhit"].astype(np.uint8)
n = len(ent)
pc = lambda v: "%9d (%.1f %%)" % (int(v), 100.0 * v / n)
bits = np.array([bin(i).count("1") for i in range(16)])
print("workload %s view %d: binned instances %d" % (wl, view, n))
print("  no quadrant inside the bounding box          %s" % pc((boxmask == 0).sum()))
print("  region misses all four quadrants (relmask=0) %s" % pc((relmask == 0).sum()))
print("  region misses the 16x16 tile (one test)      %s" % pc(tile_miss.sum()))
print("  blended by some pixel (qhit != 0)            %s" % pc((qh != 0).sum()))
print("  reachable but not blended                    %s" % pc(((relmask != 0) & (qh == 0)).sum()))
print("  quadrant visits: reachable %d, blended %d" % (bits[relmask].sum(), bits[qh & 15].sum()))
assert not ((qh & ~relmask & 15) != 0).any(), "a blended quadrant the region test calls unreachable"
no_ell = rec[:, 28] == 0
print("  instances without an ellipse                 %s" % pc(no_ell.sum()))
# per tile: list length before / after dropping relmask == 0
r0 = st["ranges"][:, 0].astype(np.int64); r1 = st["ranges"][:, 1].astype(np.int64)
keep = np.concatenate([[0], np.cumsum(relmask != 0)])
kept = keep[r1] - keep[r0]
print("  list length per tile: mean %.0f -> %.0f, max %d -> %d" % ((r1 - r0).mean(), kept.mean(), (r1 - r0).max(), kept.max()))
