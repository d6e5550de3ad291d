"""distCUDA2 timing: python tools/knn_bench.py [lib.so ...]   (run on the GPU box; default = the in-tree library)

For every library given (e.g. var/knn_r05.so = round 5's one-thread-per-point search, built by tools/build_variant.sh)
and every point set: ms per call (HIP events, median of 7 after 2 warm-ups), million queries per second, the input +
output bytes per second, equality of the results across the libraries, exactness against the brute-force oracle on
4 000 sampled queries, and the oracle's own CPU time for those queries scaled to the whole set."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from g4splat_amd import synthetic  # noqa: E402
import oracle.oracle as oracle_mod  # noqa: E402

libs = sys.argv[1:] or [os.path.join(ROOT, "g4splat_amd", "libg4s_hip.so")]


def point_sets():
    rng = np.random.default_rng(0)
    yield "room 300k (config 2/4)", synthetic.scene_room(300_000, seed=0).means3D.copy()
    yield "room 1.5M (config 3)", synthetic.scene_room(1_500_000, seed=0).means3D.copy()
    yield "room 3M (config 5)", synthetic.scene_room(3_000_000, seed=0).means3D.copy()
    yield "uniform cube 1.5M", rng.uniform(-1, 1, (1_500_000, 3)).astype(np.float32)
    c = rng.normal(size=(200, 3)) * 10
    yield "200 clusters 1.5M", (c[rng.integers(0, 200, 1_500_000)] + rng.normal(size=(1_500_000, 3)) * np.exp(rng.uniform(-4, 0, (1_500_000, 1)))).astype(np.float32)


def run(lib, pts_d, P):
    nbytes = lib.g4s_knn_workspace(P)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = torch.zeros(P, dtype=torch.float32, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    times = []
    for it in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.g4s_knn_mean_dist(P, ctypes.c_void_p(pts_d.data_ptr()), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()),
                                   ctypes.c_size_t(nbytes), stream)
        e1.record()
        assert rc == 0
        torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e1))
    return float(np.median(times)), out.cpu().numpy()


handles = []
for path in libs:
    h = ctypes.CDLL(os.path.abspath(path))
    h.g4s_knn_workspace.restype = ctypes.c_size_t
    h.g4s_knn_workspace.argtypes = [ctypes.c_int]
    h.g4s_knn_mean_dist.restype = ctypes.c_int
    handles.append((path, h))

for name, pts in point_sets():
    P = len(pts)
    pts_d = torch.as_tensor(pts, device="cuda")
    rng = np.random.default_rng(1)
    q = rng.choice(P, 4000, replace=False).astype(np.int32)
    t0 = time.time()
    want = oracle_mod.distCUDA2_queries(pts, q)
    cpu_s = (time.time() - t0) * P / len(q)
    results = []
    for path, h in handles:
        ms, got = run(h, pts_d, P)
        exact = bool(np.array_equal(got[q], want))
        results.append(got)
        print(f"{name:24s} {os.path.basename(path):20s} {ms:9.3f} ms  {P / ms / 1e3:8.1f} Mquery/s  {16 * P / ms / 1e6:7.2f} GB/s (16 B/point)  "
              f"exact on {len(q)} sampled queries: {exact}", flush=True)
    same = all(np.array_equal(results[0], r) for r in results[1:])
    print(f"{name:24s} oracle (brute force, OpenMP, {os.cpu_count()} threads) {cpu_s:9.1f} s for the whole set (scaled from {len(q)} queries)"
          + (f"; libraries agree bit for bit: {same}" if len(results) > 1 else ""), flush=True)
