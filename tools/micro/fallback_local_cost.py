"""Local (non-link) cost of the FALLBACK exchange -- parallel.RowSparseAllReduce, what bench.py runs when the owner
exchange's uneven all_to_all is refused in warm-up -- at the metric size, on one GPU with the collectives replaced by
nothing: the radii MAX all-reduce and the SUM all-reduce are stubbed, everything else (union mask -> index list with its
host read-back, pack, unpack; or nothing at all on the dense path) runs as in production.
    python tools/micro/fallback_local_cost.py
Prints, per union fraction (what `world` views of S3 cover: ~0.5 at 2 ranks ... ~1.0 at 8), the local ms per step and the
bytes the SUM all-reduce would carry, so that DESIGN.md section 5 can price the path that actually runs if all_to_all is
unavailable:  t_exchange = local + 2 (N-1)/N * bytes / busbw."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from g4splat_amd import parallel  # noqa: E402

dev = torch.device("cuda", 0)
P, widths = 1_500_000, (3, 48, 1, 2, 4, 2)
W = sum(widths)
flat = torch.randn(P * W, device=dev)
rows, o = [], 0
for w in widths:
    rows.append(flat[o:o + P * w].view(P, w))
    o += P * w
dist.all_reduce = lambda *a, **k: None  # the stub: no collective is issued
red = parallel.RowSparseAllReduce(flat, rows, compact_below=0.7)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"P = {P}, {W} floats per row; dense bucket {P * W * 4 / 1e6:.0f} MB")
for frac in (0.28, 0.45, 0.6, 0.69, 0.8, 1.0):
    union = torch.rand(P, device=dev) < frac
    ms = t(lambda: red.reduce(union))
    nbytes = (red.last_rows if not red.last_dense else P) * W * 4
    print(f"union {frac:4.2f}: local {ms:6.3f} ms/step  ({'dense: all-reduce of the whole bucket' if red.last_dense else 'packed rows'}), "
          f"all-reduce payload {nbytes / 1e6:6.1f} MB")
