"""Experiment: one training view's forward (presized entry point) + backward captured in a HIP graph and replayed,
against the same calls issued eagerly.  Same kernels, same results; what changes is who pays for the ~30 launches.

    python tools/graph_step.py [workload] [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from g4splat_amd.diff_surfel_rasterization import _C  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "s3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
device = torch.device("cuda", 0)
scene, cams, dev, dcams, (P, W, H, D) = bench.build_scene(wl, device)
bg = torch.zeros(3, device=device)
empty = torch.empty(0, device=device)
g = torch.Generator(device=device).manual_seed(1)
dL_dcolor = torch.randn((3, H, W), device=device, generator=g)
dL_dothers = torch.randn((7, H, W), device=device, generator=g)


def fwd_std(cam):
    return _C.rasterize_gaussians(bg, dev["means3D"], empty, dev["opacity"], dev["scales"], dev["rotations"], 1.0, empty,
                                  cam["view"], cam["proj"], cam["tanfovx"], cam["tanfovy"], H, W, dev["sh"], D,
                                  cam["campos"], False, False)


def fwd_pre(state, cam):
    return _C.rasterize_gaussians_presized(state, bg, dev["means3D"], empty, dev["opacity"], dev["scales"],
                                           dev["rotations"], 1.0, empty, cam["view"], cam["proj"], cam["tanfovx"],
                                           cam["tanfovy"], H, W, dev["sh"], D, cam["campos"], False, False)


def bwd(fw, cam):
    R, color, others, radii, geom, binning, img = fw
    return _C.rasterize_gaussians_backward(bg, dev["means3D"], radii, empty, dev["scales"], dev["rotations"], 1.0, empty,
                                           cam["view"], cam["proj"], cam["tanfovx"], cam["tanfovy"], dL_dcolor, dL_dothers,
                                           dev["sh"], D, cam["campos"], geom, R, binning, img, False)


Rmax = max(int(fwd_std(c)[0]) for c in dcams)
state = _C.PresizedState(P, W, H, int(Rmax * 1.25) + 4096, device)
# static camera buffers: a replay renders whatever view was copied into them
cam_s = dict(view=dcams[0]["view"].clone(), proj=dcams[0]["proj"].clone(), campos=dcams[0]["campos"].clone(),
             tanfovx=dcams[0]["tanfovx"], tanfovy=dcams[0]["tanfovy"])


def set_cam(c):
    cam_s["view"].copy_(c["view"]); cam_s["proj"].copy_(c["proj"]); cam_s["campos"].copy_(c["campos"])


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def eager_std(i):
    c = dcams[i % len(dcams)]
    bwd(fwd_std(c), c)


def eager_pre(i):
    c = dcams[i % len(dcams)]
    bwd(fwd_pre(state, c), c)


for i in range(8):
    eager_std(i); eager_pre(i)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for i in range(3):
        set_cam(dcams[i]); bwd(fwd_pre(state, cam_s), cam_s)
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    fw_g = fwd_pre(state, cam_s)
    grads_g = bwd(fw_g, cam_s)


def replay(i):
    set_cam(dcams[i % len(dcams)])
    graph.replay()


# same results?
c = dcams[3]
ref = bwd(fwd_std(c), c)
ref = [x.clone() for x in ref]
set_cam(c); graph.replay(); torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(ref, grads_g))
print("graph replay == eager, bit for bit:", same, " status:", state.status.tolist())
for name, fn in (("eager, reference-shaped forward", eager_std), ("eager, presized forward", eager_pre), ("graph replay", replay)):
    fn(0)
    print(f"{name:36s} {min(timed(fn, steps) for _ in range(3)):.3f} ms/step")
