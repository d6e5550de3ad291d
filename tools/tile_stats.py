import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import hip_state
from g4splat_amd import synthetic
from g4splat_amd.diff_surfel_rasterization import _C
P, W, H, D = 1_500_000, 1600, 1200, 3
scene = synthetic.scene_room(P, seed=0)
cams = synthetic.room_cameras(8, W, H, fovx_deg=90.0)
dev = "cuda:0"
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
empty = torch.empty(0, device=dev); bg = torch.zeros(3, device=dev)
args = dict(m=t(scene.means3D), o=t(scene.opacities), s=t(scene.scales), r=t(scene.rotations), sh=t(scene.shs))
for ci in (0, 3):
    cam = cams[ci]
    f = _C.rasterize_gaussians(bg, args["m"], empty, args["o"], args["s"], args["r"], 1.0, empty, t(cam.world_view_transform),
                               t(cam.full_proj_transform), cam.tanfovx, cam.tanfovy, H, W, args["sh"], D, t(cam.camera_center), False, False)
    st = hip_state(dict(R=f[0], geom=f[4], binning=f[5], img=f[6]), dict(means3D=scene.means3D, W=W, H=H))
    n = (st["ranges"][:, 1] - st["ranges"][:, 0]).astype(np.int64)
    last = st["n_contrib"][0].reshape(H, W)
    tl = last.reshape(H // 16, 16, W // 16, 16).max(axis=(1, 3)).reshape(-1)
    print("view", ci, "R_ref", f[0], "binned", int(n.sum()), "tile list len: mean %.0f p50 %d p99 %d max %d" % (n.mean(), np.median(n), np.percentile(n, 99), n.max()),
          "| n_live: mean %.0f p99 %d max %d" % (tl.mean(), np.percentile(tl, 99), tl.max()))
    # work per tile for the backward blend: staged entries with a contribution, and (entry, quadrant) visits
    qh = st["qhit"].astype(np.uint8)
    pc = np.unpackbits(qh[:, None], axis=1).sum(1).astype(np.int64)
    cs_v = np.concatenate([[0], np.cumsum(pc)]); cs_e = np.concatenate([[0], np.cumsum(qh != 0)])
    r0, r1 = st["ranges"][:, 0].astype(np.int64), st["ranges"][:, 1].astype(np.int64)
    visits, ent = cs_v[r1] - cs_v[r0], cs_e[r1] - cs_e[r0]
    instr = ent * 110 + visits * 95   # rough wave instructions per tile
    print("   bwd entries/tile mean %.0f max %d | visits/tile mean %.0f max %d | est. instr: total %.0fM, max tile %.0fK -> alone on a SIMD %.2f ms; "
          "all tiles over 1024 SIMDs %.2f ms" % (ent.mean(), ent.max(), visits.mean(), visits.max(), instr.sum() / 1e6, instr.max() / 1e3,
          instr.max() * 4 / 2.4e6, instr.sum() * 4 / 1024 / 2.4e6))
    top = np.sort(instr)[::-1][:8]
    print("   top tiles (K instr):", (top / 1e3).astype(int))
    order = np.argsort(-n, kind="stable")   # the launch order (longest list first)
    ci = np.cumsum(instr[order])
    for k in (1024, 2048, 3072, 4096, 5500, len(n)):
        print("   first %5d workgroups: %.0fM instr (%.0f%%), largest %dK smallest %dK" % (k, ci[k - 1] / 1e6, 100 * ci[k - 1] / ci[-1], instr[order][:k].max() / 1e3, instr[order][k - 1] / 1e3))
    print("   rank of tiles by work vs by list length: corr %.3f; top-work tile has launch rank %d" % (np.corrcoef(instr, n)[0, 1], int(np.where(order == np.argmax(instr))[0][0])))
