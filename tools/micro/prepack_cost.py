"""What it costs the backward to write the owner exchange's send rows itself (g4s_rasterizer_backward_accumulate_packed,
OwnerReduce.prepack) at the metric size, against the pack launch it replaces (run on the GPU box):
    python tools/micro/prepack_cost.py [view]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from g4splat_amd import _lib, synthetic  # noqa: E402
from g4splat_amd.diff_surfel_rasterization import _C  # noqa: E402

view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda", 0)
lib = _lib.load()
P, W, H, D = 1_500_000, 1600, 1200, 3
scene = synthetic.scene_room(P, seed=0)
cam = synthetic.room_cameras(8, W, H, fovx_deg=90.0)[view]
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
m3, sc, rot, opa, sh = t(scene.means3D), t(scene.scales), t(scene.rotations), t(scene.opacities), t(scene.shs)
vm, pm, cp = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
bg, empty = torch.zeros(3, device=dev), torch.empty(0, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
gc, go = torch.randn((3, H, W), device=dev, generator=g), torch.randn((7, H, W), device=dev, generator=g)
R, _c, _o, radii, geom, binning, img = _C.rasterize_gaussians(bg, m3, empty, opa, sc, rot, 1.0, empty, vm, pm, cam.tanfovx,
                                                              cam.tanfovy, H, W, sh, D, cp, False, False)
vis = radii > 0
n = int(vis.sum())
grads = dict(dL_dmeans3D=torch.zeros(P, 3, device=dev), dL_dsh=torch.zeros(P, 16, 3, device=dev),
             dL_dopacity=torch.zeros(P, 1, device=dev), dL_dscales=torch.zeros(P, 2, device=dev),
             dL_drotations=torch.zeros(P, 4, device=dev))
side = torch.zeros(P, 2, device=dev)
rows = [v.view(P, -1) for v in grads.values()] + [side]
widths = [r.shape[1] for r in rows]
packed = torch.empty(P, sum(widths) + 1, device=dev)
pad = torch.zeros((P + 255) // 256 * 256, dtype=torch.int32, device=dev)
pad[:P] = vis
counts = pad.view(-1, 256).sum(1, dtype=torch.int32)
offs = (torch.cumsum(counts, 0) - counts).to(torch.int32)
ws = torch.empty(lib.g4s_rasterizer_backward_workspace(P, int(R)), dtype=torch.uint8, device=dev)
idx = vis.nonzero(as_tuple=True)[0]
buf = torch.empty(n, sum(widths) + 1, device=dev)
ptrs = (ctypes.c_void_p * len(rows))(*[r.data_ptr() for r in rows])
wid = (ctypes.c_int * len(rows))(*widths)


def backward(with_packed):
    out = dict(grads, accumulate="first", view_stats=side, workspace=ws)
    if with_packed:
        out["packed"] = (packed, offs)
    _C.rasterize_gaussians_backward(bg, m3, radii, empty, sc, rot, 1.0, empty, vm, pm, cam.tanfovx, cam.tanfovy, gc, go, sh, D,
                                    cp, geom, R, binning, img, False, out=out)


def pack():
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.g4s_pack_rows(len(rows), ptrs, wid, ctypes.c_void_p(idx.data_ptr()), n, ctypes.c_void_p(buf.data_ptr()), 10, stream) == 0


def kernel_ms(fn, reps=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    lib.g4s_profile_enable(1)
    lib.g4s_profile_reset()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    per = {}
    for k in range(lib.g4s_profile_kernels()):
        ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
        lib.g4s_profile_read(k, ctypes.byref(ms), ctypes.byref(cnt))
        if cnt.value:
            per[lib.g4s_profile_name(k).decode()] = ms.value / cnt.value
    lib.g4s_profile_enable(0)
    return a.elapsed_time(b) / reps, per


print(f"S3 view {view}: {n} visible rows of {sum(widths) + 1} floats = {n * (sum(widths) + 1) * 4 / 1e6:.1f} MB")
for rep in range(2):
    t0, k0 = kernel_ms(lambda: backward(False))
    t1, k1 = kernel_ms(lambda: backward(True))
    tp, _ = kernel_ms(pack)
    print(f"backward {t0:.4f} ms (preprocess_bwd {k0.get('preprocess_bwd', 0):.4f}); with the packed rows {t1:.4f} ms "
          f"(preprocess_bwd {k1.get('preprocess_bwd', 0):.4f}); the pack launch it replaces {tp:.4f} ms")
backward(True)
pack()
torch.cuda.synchronize()
print("rows identical to the pack launch's:", bool(torch.equal(packed[:n].view(torch.int32), buf.view(torch.int32))))
