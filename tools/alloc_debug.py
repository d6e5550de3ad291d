"""Which steps of the headline loop make torch's allocator call hipMalloc, and for what (run on the GPU box):
    python tools/alloc_debug.py [steps]
The bench's N = 1 step (reference-shaped forward + plain backward) over S3's eight views; after every step the change of
num_device_alloc / reserved bytes, and the sizes _C.py requested for its scratch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from g4splat_amd.diff_surfel_rasterization import _C

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 48
device = torch.device("cuda", 0)
scene, cams, dev, dcams, (P, W, H, D) = bench.build_scene("s3", device)
bg = torch.zeros(3, device=device)
empty = torch.empty(0, device=device)
g = torch.Generator(device=device).manual_seed(1)
gc_ = torch.randn((3, H, W), device=device, generator=g)
go_ = torch.randn((7, H, W), device=device, generator=g)
req = []
orig = _C._stable_size


def spy(key, n):
    r = orig(key, n)
    req.append((key[-1] if key else None, int(n) >> 20, r >> 20))
    return r


_C._stable_size = spy
prev = torch.cuda.memory_stats(device)
for i in range(steps):
    cam = dcams[i % len(dcams)]
    req.clear()
    fw = _C.rasterize_gaussians(bg, dev["means3D"], empty, dev["opacity"], dev["scales"], dev["rotations"], 1.0, empty, cam["view"],
                                cam["proj"], cam["tanfovx"], cam["tanfovy"], H, W, dev["sh"], D, cam["campos"], False, False)
    R, color, others, radii, geom, binning, img = fw
    out = _C.rasterize_gaussians_backward(bg, dev["means3D"], radii, empty, dev["scales"], dev["rotations"], 1.0, empty, cam["view"],
                                          cam["proj"], cam["tanfovx"], cam["tanfovy"], gc_, go_, dev["sh"], D, cam["campos"], geom, R,
                                          binning, img, False)
    del fw, color, others, radii, geom, binning, img, out
    st = torch.cuda.memory_stats(device)
    da = st["num_device_alloc"] - prev["num_device_alloc"]
    df = st.get("num_device_free", 0) - prev.get("num_device_free", 0)
    print(f"step {i:3d} view {i % len(dcams)} R={R}: hipMalloc +{da} hipFree +{df} reserved {st['reserved_bytes.all.current'] >> 20} MiB "
          f"active {st['active_bytes.all.current'] >> 20} MiB; requests (kind, asked MiB, given MiB): {req}", flush=True)
    prev = st
