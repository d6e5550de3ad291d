"""Sizing experiment (run on the GPU box): one workgroup sorts 2 048 / 4 096 (depth key, index) pairs entirely in LDS (stable,
four 8-bit passes; tools/micro/lds_sort.hip).  A sample sort of the depth keys (PSRS: sort chunks, regular samples ->
splitters, partition, scatter, sort buckets; buckets are bounded by twice the chunk size) would launch this kernel twice plus
four small kernels, instead of the twelve launches of four global radix passes (0.080 ms at S3).  Times the kernel on the
keys of a real frame and checks every chunk against torch's stable sort.

    python tools/lds_sort_bench.py [s3|s2|s5] [view]"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import hip_state  # noqa: E402
import bench  # noqa: E402
from g4splat_amd.diff_surfel_rasterization import _C  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "s3"
view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
so = "/tmp/liblds_sort.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                       os.path.join(ROOT, "tools", "micro", "lds_sort.hip"), "-o", so])
lib = ctypes.CDLL(so)
dev = torch.device("cuda", 0)
scene, cams, d, dcams, (P, W, H, D) = bench.build_scene(wl, dev)
c = dcams[view % len(dcams)]
empty = torch.empty(0, device=dev); bg = torch.zeros(3, device=dev)
f = _C.rasterize_gaussians(bg, d["means3D"], empty, d["opacity"], d["scales"], d["rotations"], 1.0, empty, c["view"], c["proj"],
                           c["tanfovx"], c["tanfovy"], H, W, d["sh"], D, c["campos"], False, False)
st = hip_state(dict(R=f[0], geom=f[4], binning=f[5], img=f[6]), dict(means3D=scene.means3D, W=W, H=H))
emit = torch.tensor(st["tiles_touched"] > 0, device=dev)
v = c["view"].reshape(4, 4)
z = d["means3D"] @ v[:3, 2] + v[3, 2]
idx = torch.nonzero(emit).squeeze(1).to(torch.int32)          # the emitting Gaussians in index order (what the library packs)
keys = z[emit].contiguous().view(torch.int32)
n = int(keys.numel())
ptr = lambda t: ctypes.c_void_p(t.data_ptr())
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
print(f"workload {wl} view {view}: {n} emitting Gaussians")
for items in (8, 16):
    tk = 256 * items
    ko, vo = torch.empty_like(keys), torch.empty_like(idx)
    for _ in range(3):
        lib.lds_sort(items, ptr(keys), ptr(idx), ptr(ko), ptr(vo), n, stream)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.lds_sort(items, ptr(keys), ptr(idx), ptr(ko), ptr(vo), n, stream); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    # check: every chunk is the stable sort of its input (depth bits of positive floats order like unsigned integers)
    kh, ih, koh, voh = keys.cpu().numpy().view(np.uint32), idx.cpu().numpy(), ko.cpu().numpy().view(np.uint32), vo.cpu().numpy()
    bad = 0
    for b in range(0, n, tk):
        o = np.argsort(kh[b:b + tk], kind="stable")
        bad += int(not (np.array_equal(koh[b:b + tk], kh[b:b + tk][o]) and np.array_equal(voh[b:b + tk], ih[b:b + tk][o])))
    print(f"  {tk} pairs per workgroup, {(n + tk - 1) // tk} workgroups: median {np.median(ts):.1f} us (min {min(ts):.1f}); chunks not matching "
          f"the stable sort: {bad}")
