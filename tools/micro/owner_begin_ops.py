"""Micro-timing of the torch ops inside OwnerReduce.begin at the metric size (one GPU, no collective)."""
import time
import torch
dev = torch.device("cuda", 0)
P, world = 1_500_000, 8
shard = (P + world - 1) // world
vis = torch.rand(P, device=dev) < 0.28


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


idx = torch.nonzero_static(vis, size=P, fill_value=P).view(-1)
print("nonzero_static        %.3f ms" % t(lambda: torch.nonzero_static(vis, size=P, fill_value=P)))


def counts_scatter():
    owner = torch.div(idx, shard, rounding_mode="floor")
    owner = torch.where(idx >= P, torch.full_like(owner, world), owner)
    c = torch.zeros(world + 1, dtype=torch.int64, device=dev)
    c.scatter_add_(0, owner, torch.ones_like(owner))
    return c[:world]


edges = torch.arange(0, world + 1, device=dev, dtype=torch.int64) * shard
edges[-1] = P


def counts_search():
    pos = torch.searchsorted(idx, edges)
    return pos[1:] - pos[:-1]


assert torch.equal(counts_scatter(), counts_search())
print("counts: scatter_add   %.3f ms" % t(counts_scatter))
print("counts: searchsorted  %.3f ms" % t(counts_search))
