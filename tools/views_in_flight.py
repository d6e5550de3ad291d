"""Throughput of the rasterizer step with several INDEPENDENT views in flight on one GPU (run on the GPU box):

    python tools/views_in_flight.py [--workload s3] [--steps 200]

A multi-view batch (gradient accumulation, or SURVEY.md 8(e)'s 8 views over fewer than 8 GPUs) leaves the views of one
optimiser step independent of each other, so view j+1's launch-bound binning phase and HBM-bound per-Gaussian kernels
can run beside view j's VALU-bound blend kernels.  K HIP streams, each with its own PresizedState (no host read-back:
g4s_rasterizer_forward_presized), backward workspace and output tensors; step i goes to stream i % K; the host thread
issues all launches.  Prints Gaussians/s for K = 1, 2, 3 and checks that a view's outputs and gradients are bit-identical
whatever K is.  This is NOT bench.py's headline (one view per step, one step after the other, like the reference's loop).
"""
import argparse
import json
import os
import sys
import time

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
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    scene, cams, d, dcams, (P, W, H, D) = bench.build_scene(a.workload, dev)
    bg = torch.zeros(3, device=dev)
    empty = torch.empty(0, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    gc_ = torch.randn((3, H, W), device=dev, generator=g)
    go_ = torch.randn((7, H, W), device=dev, generator=g)
    # capacity + visible counts from one reference-shaped pass over the views
    Rs, Vs = [], []
    for c in dcams:
        fw = _C.rasterize_gaussians(bg, d["means3D"], empty, d["opacity"], d["scales"], d["rotations"], 1.0, empty, c["view"],
                                    c["proj"], c["tanfovx"], c["tanfovy"], H, W, d["sh"], D, c["campos"], False, False)
        Rs.append(int(fw[0]))
        Vs.append(int((fw[3] > 0).sum()))
    cap = int(max(Rs) * 1.25) + 4096
    ws_bytes = lib.g4s_rasterizer_backward_workspace(P, cap)
    results = {}
    ref = None
    for K in (1, 2, 3):
        streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
        states, works = [], []
        for s in streams:
            with torch.cuda.stream(s):
                states.append(_C.PresizedState(P, W, H, cap, dev))
                works.append(torch.empty(ws_bytes, dtype=torch.uint8, device=dev))
        torch.cuda.synchronize()

        def step(i, keep=False):
            k = i % K
            c = dcams[i % len(dcams)]
            with torch.cuda.stream(streams[k]):
                fw = _C.rasterize_gaussians_presized(states[k], bg, d["means3D"], empty, d["opacity"], d["scales"],
                                                     d["rotations"], 1.0, empty, c["view"], c["proj"], c["tanfovx"],
                                                     c["tanfovy"], H, W, d["sh"], D, c["campos"], False, False)
                gr = _C.rasterize_gaussians_backward(bg, d["means3D"], fw[3], empty, d["scales"], d["rotations"], 1.0, empty,
                                                     c["view"], c["proj"], c["tanfovx"], c["tanfovy"], gc_, go_, d["sh"], D,
                                                     c["campos"], fw[4], fw[0], fw[5], fw[6], False,
                                                     out={"workspace": works[k]})
            return (fw, gr) if keep else None

        for i in range(2 * K + 2):
            step(i)
        torch.cuda.synchronize()
        # bit-identity of view 1 whatever runs beside it
        fw, gr = step(1, keep=True)
        step(2)
        torch.cuda.synchronize()
        digest = [fw[1].double().sum().item(), fw[2].double().sum().item()] + [x.double().sum().item() for x in gr if torch.is_tensor(x)]
        if ref is None:
            ref = digest
        assert digest == ref, (K, digest, ref)
        assert all(int(s.status[3].item()) == 0 for s in states)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        units = sum(Vs[i % len(dcams)] for i in range(a.steps))
        results[K] = {"ms_per_view": round(dt / a.steps * 1e3, 4), "gaussians_per_s": units / dt}
        print(f"views in flight {K}: {dt / a.steps * 1e3:.3f} ms per view, {units / dt:.4g} rasterized Gaussians/s", flush=True)
        del states, works, streams
    print(json.dumps({"workload": a.workload, "steps": a.steps, "forward": "presized", "views_in_flight": results}))


if __name__ == "__main__":
    main()
