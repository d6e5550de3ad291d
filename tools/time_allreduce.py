import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
P = 1_500_000
flat = torch.randn(P * 58, device=dev)
parts = [torch.randn(P, k, device=dev) for k in (3, 48, 1, 2, 4)]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("flat 348MB all_reduce ms:", t(lambda: dist.all_reduce(flat)))
def multi():
    ws = [dist.all_reduce(p, async_op=True) for p in parts]
    for w in ws: w.wait()
print("5 tensors async all_reduce ms:", t(multi))
radii = torch.randint(0, 50, (P,), device=dev, dtype=torch.int32)
print("radii MAX ms:", t(lambda: dist.all_reduce(radii, op=dist.ReduceOp.MAX)))
gm2 = torch.randn(P, 3, device=dev)
print("stats build ms:", t(lambda: torch.stack([gm2[:, :2].norm(dim=1), (radii > 0).float()], 1)))
dist.destroy_process_group()
