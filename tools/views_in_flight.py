"""Throughput of the rasterizer step with several INDEPENDENT views in flight on one GPU (run on the GPU box):

    python tools/views_in_flight.py [--workload s3] [--steps 200] [--max 3]

A multi-view batch (gradient accumulation, or SURVEY.md 8(e)'s 8 views over fewer than 8 GPUs) leaves the views of one
optimiser step independent of each other, so view j+1's launch-bound binning phase and HBM-bound per-Gaussian kernels
can run beside view j's VALU-bound blend kernels.  K HIP streams, each with its own PresizedState (no host read-back:
g4s_rasterizer_forward_presized), backward workspace and output tensors; step i goes to stream i % K; the host thread
issues all launches (bench.run_views_in_flight; bench.py reports the same figures as `views_in_flight` next to its
headline).  Prints Gaussians/s for K = 1..max and checks that a view's outputs and gradients are bit-identical
whatever K is.  This is NOT bench.py's headline (one view per step, one step after the other, like the reference's loop).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from g4splat_amd import _lib  # noqa: E402
from g4splat_amd.diff_surfel_rasterization import _C  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="s3")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--max", type=int, default=3)
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    lib = _lib.load()
    scene, cams, dev, dcams, (P, W, H, D) = bench.build_scene(a.workload, device)
    bg = torch.zeros(3, device=device)
    empty = torch.empty(0, device=device)
    g = torch.Generator(device=device).manual_seed(1)
    gc_ = torch.randn((3, H, W), device=device, generator=g)
    go_ = torch.randn((7, H, W), device=device, generator=g)
    Rs, Vs = {}, {}
    for i, c in enumerate(dcams):  # capacity + visible counts from one reference-shaped pass over the views
        fw = _C.rasterize_gaussians(bg, dev["means3D"], empty, dev["opacity"], dev["scales"], dev["rotations"], 1.0, empty,
                                    c["view"], c["proj"], c["tanfovx"], c["tanfovy"], H, W, dev["sh"], D, c["campos"], False, False)
        Rs[i], Vs[i] = int(fw[0]), int((fw[3] > 0).sum())
    out, ref = {}, None
    for K in range(1, a.max + 1):
        ms, gps, digest = bench.run_views_in_flight(lib, _C, device, dev, dcams, P, W, H, D, Vs, Rs, K, a.steps, gc_, go_)
        ref = ref or digest
        assert digest == ref, (K, digest, ref)
        out[str(K)] = {"ms_per_view": round(ms, 4), "gaussians_per_s": gps}
        print(f"views in flight {K}: {ms:.3f} ms per view, {gps:.4g} rasterized Gaussians/s", flush=True)
    print(json.dumps({"workload": a.workload, "steps": a.steps, "forward": "presized", "views_in_flight": out}))


if __name__ == "__main__":
    main()
