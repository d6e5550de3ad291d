"""The HIP side of the multi-GPU gradient exchange on the one GPU a test box has: two `gloo` ranks share
cuda:0, so OwnerReduce runs its device path -- row-major g4s_pack_rows, accumulating unpack per source -- with a real
second rank on the other end of the all_to_all.  (RCCL refuses two ranks on one device; the collectives themselves are
torch.distributed's either way.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    root = os.path.dirname(HERE)
    for p in (root, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from g4splat_amd.parallel import OwnerReduce
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    out = []
    widths = (3, 48, 1, 2, 4, 2)
    for P, frac in ((20000, 0.3), (20001, 0.3), (4096, 1.0), (4096, 0.0)):
        g = torch.Generator().manual_seed(1000 * rank + P)
        vis = (torch.rand(P, generator=g) < frac).to(dev)
        rows = []
        for w in widths:
            t = torch.zeros(P, w, device=dev)
            t[vis] = torch.randn(int(vis.sum()), w, generator=g).to(dev)
            rows.append(t)
        dense = [r.clone() for r in rows]
        for d in dense:
            dist.all_reduce(d)
        red = OwnerReduce(rows)
        assert red.hip and not red.rccl
        red.begin(vis)
        red.finish()
        torch.cuda.synchronize()
        same = all(torch.equal(a, b) for a, b in zip(rows, dense))  # two addends commute: bit-identical
        out.append((P, frac, bool(same), int(red.last_rows_sent), int(vis.sum())))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_owner_reduce_device_path_with_two_ranks_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        for P, frac, same, sent, nvis in res[r]:
            assert same, (r, P, frac)
            assert 0 <= sent <= nvis
            if frac == 0.3:
                assert 0.3 * nvis < sent < 0.7 * nvis  # about half of a rank's visible rows belong to the other owner


def test_bench_two_ranks_control_flow_on_one_gpu():
    """bench.py's N > 1 path end to end (launcher, warm-up fallback logic, owner-reduce exchange inside the timed steps,
    max-over-ranks timing, the one JSON line) with two gloo ranks sharing cuda:0 -- the bring-up mode bench.py documents
    (G4S_BENCH_BACKEND / G4S_BENCH_ONE_DEVICE).  The numbers of such a run mean nothing; the control flow is what the
    driver's multi-GPU run will execute."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, G4S_BENCH_BACKEND="gloo", G4S_BENCH_ONE_DEVICE="1")
    port = 36500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "s2", "--no-cpu-baseline"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert "owner-reduce" in d["config"]["parallelism"], d["config"]["parallelism"]
    assert d["config"]["exchanged_rows_per_step"] > 0
    assert d["config"]["exchange_ms_per_step"] > 0
    assert d["cpu_baseline"] is None and d["roofline"] is not None
