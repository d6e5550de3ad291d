"""Local (non-link) cost of OwnerReduce's SPARSE gather at the metric size, as one rank of `world` sees it on its own GPU: the
owner packs the rows of its shard that some rank saw (g4s_pack_rows mode 2), and unpacks the other owners' rows where they
belong (mode 3, one launch per owner); the all_gather between them is not run.  Against the bytes it saves.
    python tools/micro/sparse_gather_local_cost.py [world] [union fraction ...]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from g4splat_amd import _lib  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fracs = [float(x) for x in sys.argv[2:]] or [0.39, 0.70, 0.86, 0.95]
dev = torch.device("cuda", 0)
lib = _lib.load()
P, widths = 1_500_000, (3, 48, 1, 2, 4, 2)
W = sum(widths)
rows = [torch.randn(P, w, device=dev) for w in widths]
k = len(rows)
ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in rows])
wid = (ctypes.c_int * k)(*widths)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
shard = P // world


def kernel(idx, n, buf, mode):
    assert lib.g4s_pack_rows(k, ptrs, wid, ctypes.c_void_p(idx.data_ptr()), int(n), ctypes.c_void_p(buf.data_ptr()), mode, stream) == 0


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for f in fracs:
    union = torch.rand(P, device=dev) < f
    idx = torch.nonzero(union).view(-1)
    parts = [idx[(idx >= d * shard) & (idx < (d + 1) * shard)].contiguous() for d in range(world)]
    maxc = max(int(p.numel()) for p in parts)
    gin = torch.empty(maxc, W, device=dev)
    gout = torch.randn(world * maxc, W, device=dev)

    def local():
        kernel(parts[0], parts[0].numel(), gin, 2)
        for s_ in range(1, world):
            kernel(parts[s_], parts[s_].numel(), gout[s_ * maxc:s_ * maxc + parts[s_].numel()], 3)
    ms = t(local)
    dense_bytes = (P - shard) * W * 4
    sparse_bytes = (world - 1) * maxc * W * 4
    link = 7 * 153e9 if world == 8 else (world - 1) * 153e9
    print(f"world {world}, union {f:.2f} of the rows: pack + {world - 1} unpacks {ms:.3f} ms locally; gather bytes received {dense_bytes / 1e6:.0f} MB dense -> "
          f"{sparse_bytes / 1e6:.0f} MB sparse = {1e3 * (dense_bytes - sparse_bytes) / link:.3f} ms saved at 100 % of {link / 1e9:.0f} GB/s, "
          f"{1e3 * (dense_bytes - sparse_bytes) / link / 0.8:.3f} ms at 80 %", flush=True)
