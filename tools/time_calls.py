import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from g4splat_amd.diff_surfel_rasterization import _C
dev = torch.device("cuda:0")
scene, cams, d, dcams, (P, W, H, D) = bench.build_scene("s3", dev)
bg = torch.zeros(3, device=dev); e = torch.empty(0, device=dev)
gc = torch.randn((3, H, W), device=dev); go = torch.randn((7, H, W), device=dev)
cam = dcams[0]
def sync(): torch.cuda.synchronize()
for it in range(6):
    sync(); t0 = time.perf_counter()
    fw = _C.rasterize_gaussians(bg, d["means3D"], e, d["opacity"], d["scales"], d["rotations"], 1.0, e, cam["view"], cam["proj"], cam["tanfovx"], cam["tanfovy"], H, W, d["sh"], D, cam["campos"], False, False)
    t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
    R, color, others, radii, geom, binning, img = fw
    g = _C.rasterize_gaussians_backward(bg, d["means3D"], radii, e, d["scales"], d["rotations"], 1.0, e, cam["view"], cam["proj"], cam["tanfovx"], cam["tanfovy"], gc, go, d["sh"], D, cam["campos"], geom, R, binning, img, False)
    t3 = time.perf_counter(); sync(); t4 = time.perf_counter()
    print(f"fwd host {1e3*(t1-t0):.2f} ms, fwd total {1e3*(t2-t0):.2f}; bwd host {1e3*(t3-t2):.2f}, bwd total {1e3*(t4-t2):.2f}", flush=True)
