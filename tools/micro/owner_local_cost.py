"""Local (non-link) cost of one OwnerReduce exchange at the metric size, as rank 0 of `world` ranks would see it on its
own GPU: everything except the two collectives, which are replaced by nothing (buffers are just allocated).
    python tools/micro/owner_local_cost.py [world]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from g4splat_amd import _lib  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
lib = _lib.load()
P, widths = 1_500_000, (3, 48, 1, 2, 4, 2)
W = sum(widths)
rows = [torch.randn(P, w, device=dev) for w in widths]
vis = torch.rand(P, device=dev) < 0.28
shard = P // world
edges = torch.arange(0, world + 1, device=dev, dtype=torch.int64) * shard
k = len(rows)
ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in rows])
wid = (ctypes.c_int * k)(*widths)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def kernel(idx, n, buf, mode):
    ip = ctypes.c_void_p(idx.data_ptr() if idx is not None else 0)
    assert lib.g4s_pack_rows(k, ptrs, wid, ip, int(n), ctypes.c_void_p(buf.data_ptr()), mode, stream) == 0


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def begin():
    idx = torch.nonzero_static(vis, size=P, fill_value=P).view(-1)
    pos = torch.searchsorted(idx, edges)
    return idx, (pos[1:] - pos[:-1])


idx, counts = begin()
c = counts.tolist()
n_all = sum(c)
send_idx = idx[c[0]:n_all].contiguous()   # rank 0: everything but its own shard
n = n_all - c[0]
# (rows travel with their index as a trailing int32 column: modes 10 = pack, 15 = accumulate with indices from the buffer)
buf = torch.empty(n, W + 1, device=dev)
per_src = n // (world - 1) if world > 1 else 0
in_rows = torch.randn(max(1, per_src), W + 1, device=dev)
in_idx = torch.randperm(shard, device=dev)[:max(1, per_src)].sort().values
in_rows[:, W] = in_idx.to(torch.int32).view(torch.float32)


# all sources in one launch (g4s_accumulate_rows): the received rows of the world - 1 sources back to back, each source's
# indices ascending inside this rank's shard [0, shard)
all_rows = torch.randn(max(1, per_src * (world - 1)), W + 1, device=dev)
for s_ in range(world - 1):
    ii = torch.randperm(shard, device=dev)[:per_src].sort().values
    all_rows[s_ * per_src:(s_ + 1) * per_src, W] = ii.to(torch.int32).view(torch.float32)
src_off = (ctypes.c_int * max(1, world - 1))(*[s_ * per_src for s_ in range(world - 1)])
src_cnt = (ctypes.c_int * max(1, world - 1))(*[per_src] * (world - 1))


def accumulate_one_launch():
    assert lib.g4s_accumulate_rows(k, ptrs, wid, world - 1, src_off, src_cnt, ctypes.c_void_p(all_rows.data_ptr()), 0, shard,
                                   stream) == 0


def finish_local(prepacked=False):  # persistent buffers: nothing is allocated here
    if n and not prepacked:
        kernel(send_idx, n, buf, 10)
    if world - 1 >= 4:  # OwnerReduce.ONE_LAUNCH_SOURCES
        accumulate_one_launch()
    else:
        for _s in range(world - 1):
            kernel(None, per_src, in_rows, 15)


print(f"world {world}: {n} rows to send, {per_src} rows per source to accumulate")
print("begin (index list + counts)      %.3f ms" % t(begin))
print("pack (one launch)                %.3f ms" % t(lambda: kernel(send_idx, n, buf, 10)))
print("accumulate (world-1 launches)    %.3f ms" % t(lambda: [kernel(None, per_src, in_rows, 15) for _ in range(world - 1)]))
print("accumulate (one launch)          %.3f ms" % t(accumulate_one_launch))
print("finish, local part               %.3f ms   (pack + the accumulation OwnerReduce picks: one launch from four sources on)" % t(finish_local))
print("finish, local part, prepacked    %.3f ms   (the backward wrote the send rows: + %s ms in its per-Gaussian kernel,"
      " profiles/r05_prepack_cost.txt)" % (t(lambda: finish_local(True)), "0.027-0.044"))
