"""Times the pieces of the multi-GPU gradient exchange on whatever ranks it is
launched on, one JSON line from rank 0 -- the table for 1/2/4/8 GPUs is four runs of one command:

    for n in 1 2 4 8; do
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29544 \\
          tools/time_allreduce.py [P] [visible fraction]
    done

Pieces (milliseconds, max over ranks, synthetic gradients of the metric scene's shape: 58 + 2 floats per Gaussian in
six row tensors, a random `visible` set of the given fraction per rank):
  dense_all_reduce     one RCCL SUM all-reduce of the flat 60-float bucket (round 1's exchange)
  radii_max            the MAX all-reduce of the radii (-> max_radii2D), needed by every variant
  owner_begin          OwnerReduce.begin: index list + counts + ONE all-reduce(MAX) carrying radii and count matrix
  owner_finish         OwnerReduce.finish: pack (+ index column), ONE all_to_all, owner accumulation, in-place all_gather
  owner_total          begin + finish back to back (in training begin() hides behind the backward)
  zero1_finish         finish(gather=False) + ShardedAdam.step: owner-applied Adam, the gather carries parameters
and the byte budget per rank next to them.  Output format (one line):
  {"world": N, "P": ..., "visible": ..., "ms": {...}, "MB_per_rank": {"all_to_all": ..., "all_gather": ..., "dense_all_reduce": ...}}
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
from g4splat_amd.parallel import OwnerReduce, ShardedAdam  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_500_000
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.28
widths = (3, 48, 1, 2, 4, 2)
W = sum(widths)
flat = torch.zeros(P * W, device=dev)
rows, o = [], 0
for w in widths:
    rows.append(flat[o:o + P * w].view(P, w))
    o += P * w
g = torch.Generator(device=dev).manual_seed(7 + rank)
vis = torch.rand(P, device=dev, generator=g) < frac
src = [torch.randn(P, w, device=dev, generator=g) * vis[:, None] for w in widths]
radii = (vis * 7).to(torch.int32)


def refill():
    for r, s in zip(rows, src):
        r.copy_(s)


def timed(fn, n=10, warm=3):
    out = []
    for i in range(warm + n):
        refill()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        if i >= warm:
            out.append(float(dt.item()) * 1e3)
    out.sort()
    return round(out[len(out) // 2], 4)


red = OwnerReduce(rows)
ms = {}
ms["dense_all_reduce"] = timed(lambda: dist.all_reduce(flat))
ms["radii_max"] = timed(lambda: dist.all_reduce(radii, op=dist.ReduceOp.MAX))


# begin / finish separately: begin() of the timed finish() runs outside the clock
def timed_finish(fin, n=10, warm=3):
    out = []
    for i in range(warm + n):
        refill()
        red.begin(vis, radii=radii)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fin()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        if i >= warm:
            out.append(float(dt.item()) * 1e3)
    out.sort()
    return round(out[len(out) // 2], 4)


ms["owner_finish"] = timed_finish(red.finish)
ms["owner_total"] = timed(lambda: (red.begin(vis, radii=radii), red.finish()))
ms["owner_begin"] = round(ms["owner_total"] - ms["owner_finish"], 4)
# ZeRO-1: the owner steps its shard, the gather carries parameters (+ the two statistics columns)
params = [torch.randn(P, w, device=dev, generator=g) for w in widths[:5]]
opt = ShardedAdam(params, rows[:5], red, (1.6e-4, 2.5e-3, 0.05, 0.005, 0.001), eps=1e-15)
ms["zero1_finish"] = timed_finish(lambda: (red.finish(gather=False), opt.step(extra=[rows[5]])))
# correctness on the spot: owner-reduce == dense all-reduce (to the order of <= N additions)
refill()
dense = flat.clone()
dist.all_reduce(dense)
refill()
red.begin(vis, radii=radii)
red.finish()
rmax = radii.clone()
dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
assert torch.equal(red.max_radii, rmax)
err = float((flat - dense).abs().max() / dense.abs().max().clamp_min(1e-30))
nv = int(vis.sum())
sent = int(red.last_rows_sent)
if rank == 0:
    print(json.dumps({
        "world": world, "P": P, "visible": round(nv / P, 4), "rows_sent_to_other_owners": sent,
        "gather": "in place" if red.even else "staged (P not divisible by the world size)",
        "coalesced_gathers": bool(red._coalesce), "max_rel_diff_vs_dense": err, "ms": ms,
        "collectives_per_step": 3, "buffer_allocations": red.allocations,
        "MB_per_rank": {"all_to_all": round(sent * 4 * (W + 1) / 1e6, 1),
                        "all_gather_received": round(P * (world - 1) / world * 4 * W / 1e6, 1),
                        "dense_all_reduce": round(2 * (world - 1) / world * P * 4 * W / 1e6, 1)}}))
dist.destroy_process_group()
