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
    # ---- ZeRO-1 on the device: the owner steps its shard with g4s_adam_step on row slices, parameters are gathered;
    # against FusedAdam over the gathered gradients (the replicated optimiser) -- bit-identical (Adam is element-wise)
    from g4splat_amd.optim import FusedAdam
    from g4splat_amd.parallel import ShardedAdam
    zero1 = []
    lrs = (1.6e-4, 2.5e-3, 0.05, 0.005, 0.001)
    zw = (3, 48, 1, 2, 4)
    for P in (20000, 20001):
        gen = torch.Generator().manual_seed(5)
        pa = [torch.nn.Parameter(torch.randn(P, w, generator=gen).to(dev)) for w in zw]
        pb = [p.detach().clone() for p in pa]
        ga = [torch.zeros(P, w, device=dev) for w in zw]
        gb = [torch.zeros(P, w, device=dev) for w in zw]
        for p_, g_ in zip(pa, ga):
            p_.grad = g_
        full = FusedAdam([{"params": [p_], "lr": lr} for p_, lr in zip(pa, lrs)], lr=0.0, eps=1e-15)
        ra, rb = OwnerReduce(ga), OwnerReduce(gb)
        sh = ShardedAdam(pb, gb, rb, lrs, eps=1e-15)
        for it in range(3):
            g = torch.Generator().manual_seed(100 * it + rank)
            vis = (torch.rand(P, generator=g) < 0.3).to(dev)
            radii = (vis * 7).to(torch.int32)
            for a_, b_, w in zip(ga, gb, zw):
                vals = torch.randn(P, w, generator=g).to(dev)
                a_.zero_(); b_.zero_()
                a_[vis] = vals[vis]; b_[vis] = vals[vis]
            ra.begin(vis, radii=radii); ra.finish(); full.step()
            rb.begin(vis, radii=radii); rb.finish(gather=False); sh.step()
            if it == 0:
                first = (ra.allocations, rb.allocations)
        torch.cuda.synchronize()
        # persistent buffers only: whatever the first step allocated (send / receive buffers, the per-width staging of
        # ragged shards) may at most be regrown once by a larger visible set; nothing is allocated per step
        zero1.append((P, all(torch.equal(a_.detach(), b_) for a_, b_ in zip(pa, pb)), ra.allocations - first[0],
                      rb.allocations - first[1]))
    q.put((rank, out, zero1))
    dist.barrier()
    dist.destroy_process_group()


def test_owner_reduce_device_path_with_two_ranks_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    res = {g[0]: g[1] for g in got}
    zero1 = {g[0]: g[2] for g in got}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        for P, same_params, alloc_a, alloc_b in zero1[r]:
            assert same_params, (r, P, "owner-applied Adam differs from the replicated FusedAdam")
            assert alloc_a <= 2 and alloc_b <= 2, (alloc_a, alloc_b)  # after step one: send / receive regrown at most once
        for P, frac, same, sent, nvis in res[r]:
            assert same, (r, P, frac)
            assert 0 <= sent <= nvis
            if frac == 0.3:
                assert 0.3 * nvis < sent < 0.7 * nvis  # about half of a rank's visible rows belong to the other owner


def _prepack_worker(rank, world, port, q):
    root = os.path.dirname(HERE)
    for p in (root, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from g4splat_amd import synthetic
    from g4splat_amd.diff_surfel_rasterization import _C
    from g4splat_amd.parallel import OwnerReduce
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    P, W, H, D = 40_001, 320, 200, 3   # (a ragged last 256-block and a ragged last shard)
    scene = synthetic.scene_room(P, seed=2)
    cam = synthetic.room_cameras(4, W, H, fovx_deg=90.0)[rank]  # every rank its own view of the replicated scene
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    m3, sc, rot, opa, sh = t(scene.means3D), t(scene.scales), t(scene.rotations), t(scene.opacities), t(scene.shs)
    view, proj, campos = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
    bg, empty = torch.zeros(3, device=dev), torch.empty(0, device=dev)
    g = torch.Generator(device=dev).manual_seed(11)
    gc, go = torch.randn((3, H, W), device=dev, generator=g), torch.randn((7, H, W), device=dev, generator=g)
    fw = _C.rasterize_gaussians(bg, m3, empty, opa, sc, rot, 1.0, empty, view, proj, cam.tanfovx, cam.tanfovy, H, W, sh, D,
                                campos, False, False)
    R, _c, _o, radii, geom, binning, img = fw
    vis = radii > 0

    def exchange(prepacked):
        grads = dict(dL_dmeans3D=torch.zeros(P, 3, device=dev), dL_dsh=torch.zeros(P, 16, 3, device=dev),
                     dL_dopacity=torch.zeros(P, 1, device=dev), dL_dscales=torch.zeros(P, 2, device=dev),
                     dL_drotations=torch.zeros(P, 4, device=dev))
        side = torch.zeros(P, 2, device=dev)
        rows = [v.view(P, -1) for v in grads.values()] + [side]
        red = OwnerReduce(rows)
        red.begin(vis, radii=radii)
        out = dict(grads, accumulate="first", view_stats=side)
        if prepacked:
            out["packed"] = red.prepack(vis)
        _C.rasterize_gaussians_backward(bg, m3, radii, empty, sc, rot, 1.0, empty, view, proj, cam.tanfovx, cam.tanfovy, gc, go,
                                        sh, D, campos, geom, R, binning, img, False, out=out)
        packed_ok = None
        if prepacked:  # the rows the kernel wrote == what the pack launch makes of the tensors (mode 10), bit for bit
            n = int(vis.sum())
            idx = vis.nonzero(as_tuple=True)[0]
            want = torch.empty(n, red.width + 1, device=dev)
            red._rows_kernel(idx, n, want, 10)
            packed_ok = bool(torch.equal(out["packed"][0][:n].view(torch.int32), want.view(torch.int32))) and n > 1000
        red.finish(prepacked=prepacked)
        torch.cuda.synchronize()
        return rows, packed_ok, red

    rows_a, _, red_a = exchange(False)
    rows_b, packed_ok, red_b = exchange(True)
    same = all(bool(torch.equal(a.view(torch.int32), b.view(torch.int32))) for a, b in zip(rows_a, rows_b))
    q.put((rank, same, packed_ok, bool(red_b.last_prepacked), int(red_a.last_rows_sent), int(red_b.last_rows_sent),
           float(rows_b[0].abs().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_backward_writes_the_exchange_rows_itself_and_the_exchange_is_unchanged():
    """Verdict r4 item 2(a): with `out["packed"]` the per-Gaussian kernel of the backward leaves this rank's visible rows in
    the owner exchange's send layout (g4s_rasterizer_backward_accumulate_packed), and OwnerReduce.finish(prepacked=True)
    sends from there without its pack launch.  Two gloo ranks on cuda:0, each with its own view of a replicated 40 k-surfel
    room: the rows the kernel wrote equal the pack launch's output bit for bit, and after the exchange every gradient
    tensor and the statistics hold the same bits as with the pack launch."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36900 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_prepack_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, packed_ok, flagged, sent_a, sent_b, mass in got:
        assert packed_ok, rank
        assert same, rank
        assert flagged and sent_a == sent_b > 0 and mass > 0


def test_all_sources_accumulated_in_one_launch_equal_one_launch_per_source():
    """g4s_accumulate_rows (the owner's side of the exchange as ONE launch: a workgroup per 64 destination rows, the
    sources' rows of the chunk found by a 32-ary search over their ascending index columns) against what it replaces,
    one g4s_pack_rows(mode 15) launch per source in the same order: bit-identical, and equal to the sequential float32
    sums computed on the host.  Sources with no row, with one row, with every row of the shard, with rows at both ends of
    the shard; shards whose length is not a multiple of 64; one to nine sources (nine = two launches)."""
    import ctypes
    import numpy as np
    from g4splat_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    widths = (3, 48, 1, 2, 4, 2)
    W = sum(widths)
    k = len(widths)
    wid = (ctypes.c_int * k)(*widths)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(5)
    for P, lo, hi, nsrc in ((50_000, 18_750, 25_000, 7), (50_000, 0, 6_250, 7), (50_001, 43_750, 50_001, 7),
                            (1_000, 100, 137, 3), (1_000, 0, 1_000, 1), (30_000, 10_000, 20_000, 9), (300, 64, 128, 8)):
        base = [torch.randn(P, w, device=dev) for w in widths]
        counts = []
        for s_ in range(nsrc):
            n = hi - lo
            counts.append([0, 1, n, int(n * 0.28), int(n * 0.28), 2, int(n * 0.5), int(n * 0.9), 3][s_ % 9])
        idx = [np.sort(rng.choice(np.arange(lo, hi), c, replace=False)) for c in counts]
        if counts[1 % nsrc] == 1 and nsrc > 1:
            idx[1] = np.array([hi - 1])
        if nsrc > 5:
            idx[5] = np.array([lo, hi - 1])
        m = sum(counts)
        buf = torch.randn(max(m, 1), W + 1, device=dev)
        flat = np.concatenate(idx) if m else np.zeros(0, np.int64)
        if m:
            buf[:m, W] = torch.as_tensor(flat.astype(np.int32), device=dev).view(torch.float32)
        offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int).tolist()

        def per_source(rows):
            ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in rows])
            for o, c in zip(offs, counts):
                if c:
                    assert lib.g4s_pack_rows(k, ptrs, wid, None, c, ctypes.c_void_p(buf[o:o + c].data_ptr()), 15, stream) == 0

        def one_launch(rows):
            ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in rows])
            live = [(o, c) for o, c in zip(offs, counts) if c]
            n = len(live)
            rc = lib.g4s_accumulate_rows(k, ptrs, wid, n, (ctypes.c_int * n)(*[o for o, _ in live]),
                                         (ctypes.c_int * n)(*[c for _, c in live]), ctypes.c_void_p(buf.data_ptr()), lo, hi,
                                         stream)
            assert rc == 0, _lib.last_error()

        a = [r.clone() for r in base]
        b = [r.clone() for r in base]
        per_source(a)
        one_launch(b)
        torch.cuda.synchronize()
        host = [r.cpu().numpy().copy() for r in base]
        hb = buf.cpu().numpy()
        for o, c, ix in zip(offs, counts, idx):
            off = 0
            for r, w in zip(host, widths):
                r[ix] = r[ix] + hb[o:o + c, off:off + w]  # (distinct indices inside a source)
                off += w
        for x, y, z in zip(a, b, host):
            assert torch.equal(x, y), (P, lo, hi, nsrc)
            assert np.array_equal(y.cpu().numpy(), z), (P, lo, hi, nsrc)
        # g4s_accumulate_rows_ordered: the owner's own rows at position k of the order of additions -- every element is
        # ((0 + s_0 + ... + s_{k-1}) + own) + s_k + ..., against the same sequence of float32 additions on the host
        live = [(o, c, ix) for o, c, ix in zip(offs, counts, idx) if c]
        if 1 <= len(live) <= 8:
            n = len(live)
            for own_pos in sorted({0, 1, n // 2, n}):
                c_rows = [r.clone() for r in base]
                ptrs = (ctypes.c_void_p * k)(*[r.data_ptr() for r in c_rows])
                rc = lib.g4s_accumulate_rows_ordered(k, ptrs, wid, n, (ctypes.c_int * n)(*[o for o, _, _ in live]),
                                                     (ctypes.c_int * n)(*[c for _, c, _ in live]), ctypes.c_void_p(buf.data_ptr()),
                                                     lo, hi, own_pos, stream)
                assert rc == 0, _lib.last_error()
                torch.cuda.synchronize()
                want = [r.cpu().numpy().copy() for r in base]
                acc = [np.zeros((hi - lo, w), np.float32) for w in widths]
                for pos in range(n + 1):
                    if pos == own_pos:
                        for a_, r in zip(acc, want):
                            a_ += r[lo:hi]
                    if pos == n:
                        break
                    o, c, ix = live[pos]
                    off = 0
                    for a_, w in zip(acc, widths):
                        a_[ix - lo] = a_[ix - lo] + hb[o:o + c, off:off + w]
                        off += w
                for a_, r in zip(acc, want):
                    r[lo:hi] = a_
                for y, z in zip(c_rows, want):
                    assert np.array_equal(y.cpu().numpy(), z), (P, lo, hi, nsrc, own_pos)


# (strong scaling at two ranks -- four views per rank accumulated locally -- moved behind `-m "gpu and exhaustive"` in round 5:
# the default suite runs strong scaling at eight ranks now, test_bench_eight_ranks_control_flow_on_one_gpu)
@pytest.mark.parametrize("scaling", ["weak", pytest.param("strong", marks=pytest.mark.exhaustive)])
def test_bench_two_ranks_control_flow_on_one_gpu(scaling):
    """`python bench.py --gpus 2` as ONE command (it starts its own ranks through torch.distributed.run): the N > 1 path
    end to end (launcher, warm-up fallback logic, owner-reduce exchange inside the timed steps, max-over-ranks timing,
    per-step events, the one JSON line) with two gloo ranks sharing cuda:0 -- the bring-up mode bench.py documents
    (G4S_BENCH_BACKEND / G4S_BENCH_ONE_DEVICE) -- in both scaling forms: weak (one view per rank per step) and strong
    (8 views per step, 4 per rank accumulated locally with views in flight).  The numbers of such a run
    mean nothing; the control flow is what the driver's multi-GPU run will execute."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, G4S_BENCH_BACKEND="gloo", G4S_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "s2", "--no-cpu-baseline", "--scaling", scaling, "--sustained-seconds", "0.1"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == scaling and d["value"] > 0
    assert "owner-reduce" in d["config"]["parallelism"], d["config"]["parallelism"]
    assert d["config"]["views_per_step"] == (8 if scaling == "strong" else 2)
    assert d["config"]["exchanged_rows_per_step"] > 0
    assert d["config"]["exchange_ms_per_step"] > 0
    ex = d["exchange"]
    assert ex["rccl_ranks"] == 2 and ex["backend"] == "gloo" and ex["ran"].startswith("owner-reduce") and ex["why"] == "default"
    assert ex["prepacked"] is (scaling == "weak") and ex["replicas_identical"] is True
    assert set(ex["ms_pieces"]) >= {"begin_local", "max_all_reduce", "all_to_all", "accumulate", "all_gather"}
    assert ("pack" in ex["ms_pieces"]) == (scaling == "strong")  # weak: the backward wrote the send rows itself
    assert ex["bytes_per_rank"]["all_to_all_sent"] > 0 and ex["bytes_per_rank"]["all_gather_received"] > 0
    t = d["timing"]
    assert len(t["per_step_ms"]) == 3 and t["mean_ms"] == pytest.approx(d["ms_per_step"], rel=1e-3)
    assert 0 <= t["undisturbed_passes"] <= t["passes"] == len(t["attempts"])  # (two gloo ranks on one GPU jitter)
    assert t["disturbed"] == (t["undisturbed_passes"] == 0)
    assert t["min_ms"] <= t["median_ms"] <= t["max_ms"] and isinstance(t["disturbed"], bool)
    assert d["build_id"] and d["build_id"] != "unknown"
    assert d["cpu_baseline"] is None and d["roofline"] is not None


@pytest.mark.parametrize("scaling,exchange", [("weak", "owner"), ("strong", "allreduce"),
                                              pytest.param("strong", "owner", marks=pytest.mark.exhaustive),
                                              pytest.param("weak", "allreduce", marks=pytest.mark.exhaustive)])
def test_bench_eight_ranks_control_flow_on_one_gpu(scaling, exchange):
    """The N = 8 shape of `python bench.py --gpus 8` on ONE device (verdict r4 item 3: until the first 8-GPU lease no
    device had run the eight-rank code paths): eight gloo ranks sharing cuda:0 at configuration 4's size (S2, 300 k
    surfels, 8 views), weak (one view per rank) and strong (8 views per step = one per rank) scaling, the owner exchange
    and the all-reduce fallback.  One JSON line; the communicator reports eight ranks and says which exchange ran and why;
    nothing of the exchange is (re)allocated after the warm-up; every rank ends with the same bits in its gradient bucket."""
    import json
    import subprocess
    root = os.path.dirname(HERE)
    env = dict(os.environ, G4S_BENCH_BACKEND="gloo", G4S_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "G4S_BENCH_EXCHANGE"):
        env.pop(k, None)
    if exchange != "owner":
        env["G4S_BENCH_EXCHANGE"] = exchange
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--workload", "s2",
           "--no-cpu-baseline", "--scaling", scaling, "--sustained-seconds", "0", "--no-settle", "--attempts", "1",
           "--views-in-flight", "0"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["views_per_step"] == 8
    ex = d["exchange"]
    assert ex["rccl_ranks"] == 8 and ex["backend"] == "gloo"
    if exchange == "owner":
        assert ex["ran"].startswith("owner-reduce") and ex["why"] == "default"
        assert set(ex["ms_pieces"]) >= {"begin_local", "max_all_reduce", "all_to_all", "accumulate", "all_gather"}
        assert ex["prepacked"] is (scaling == "weak")
        assert ex["bytes_per_rank"]["all_to_all_sent"] > 0 and ex["bytes_per_rank"]["all_gather_received"] > 0
        assert ex["buffer_allocations_after_warmup"] == 0, ex
    else:
        assert ex["ran"].startswith("visible-rows all-reduce") and ex["why"] == "requested"
    assert ex["replicas_identical"] is True, ex
    assert d["config"]["exchanged_rows_per_step"] > 0


def _rccl_worker(q, P=30000):
    root = os.path.dirname(HERE)
    for p in (root, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(38500 + (os.getpid() % 2000))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from g4splat_amd.parallel import OwnerReduce, ShardedAdam
    widths = (3, 48, 1, 2, 4, 2)
    g = torch.Generator().manual_seed(3)
    vis = (torch.rand(P, generator=g) < 0.3).to(dev)
    rows = [torch.zeros(P, w, device=dev) for w in widths]
    for r in rows:
        r[vis] = torch.randn(int(vis.sum()), r.shape[1], generator=g).to(dev)
    want = [r.clone() for r in rows]
    radii = (vis * 9).to(torch.int32)
    red = OwnerReduce(rows)
    assert red.rccl
    dev_allocs = []
    for _ in range(4):  # persistent buffers: the same collectives on the same memory every step
        red.begin(vis, radii=radii)
        red.finish()
        torch.cuda.synchronize()
        dev_allocs.append((torch.cuda.memory_stats(dev)["num_device_alloc"], red.allocations))
    params = [torch.randn(P, w, generator=g).to(dev) for w in widths[:5]]
    before = [p.clone() for p in params]
    opt = ShardedAdam(params, rows[:5], red, (1e-3,) * 5)
    red.begin(vis, radii=radii)
    red.finish(gather=False)
    opt.step(extra=[rows[5]])
    torch.cuda.synchronize()
    q.put((all(torch.equal(a, b) for a, b in zip(rows, want)), bool(torch.equal(red.max_radii, radii)), red._coalesce,
           red.allocations, all(not torch.equal(a, b) for a, b in zip(params, before)),
           all(bool(torch.isfinite(p).all()) for p in params), dev_allocs))
    dist.destroy_process_group()


@pytest.mark.parametrize("P", [30000, 1_500_000])
def test_owner_reduce_and_sharded_adam_on_the_rccl_backend(P):
    """Verdict r2 item 3d / r3 item 3: the whole exchange on the **nccl** (= RCCL) backend, at the world size a one-GPU box
    allows, at a small size and at S3's row count (1.5 M rows x 61 floats): the fused [P + 1] int32 MAX collective, the
    uneven all_to_all (zero-row splits: every row is self-owned), the in-place all_gather issued as one RCCL group, and
    the owner-applied Adam with its parameter gather are executed BY RCCL on the persistent buffers, four steps in a
    row; results are the identity on one rank, and after the first step neither the exchange nor torch's allocator
    obtains device memory (hipMalloc count constant)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q, P))
    p.start()
    same, radii_ok, coalesced, allocs, stepped, finite, dev_allocs = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same and radii_ok and stepped and finite
    assert allocs <= 2, allocs
    assert isinstance(coalesced, bool)
    assert all(a == dev_allocs[1] for a in dev_allocs[1:]), dev_allocs  # nothing allocated after the warm-up step


def test_views_in_flight_on_separate_streams_are_bit_identical(hip_lib):
    """Multi-view batches (bench.py's `views_in_flight`): K independent views on K HIP streams,
    each with its own PresizedState, workspace and outputs, issued round-robin by one host thread.  A view's outputs and
    gradients must not depend on what runs beside it: the digest of view 1 is the same for K = 1, 2 and 3."""
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from g4splat_amd import _lib
    from g4splat_amd.diff_surfel_rasterization import _C
    device = torch.device("cuda", 0)
    scene, cams, dev, dcams, (P, W, H, D) = bench.build_scene("s2", device)
    bg = torch.zeros(3, device=device)
    empty = torch.empty(0, device=device)
    g = torch.Generator(device=device).manual_seed(1)
    gc_ = torch.randn((3, H, W), device=device, generator=g)
    go_ = torch.randn((7, H, W), device=device, generator=g)
    Rs, Vs = {}, {}
    for i, c in enumerate(dcams):
        fw = _C.rasterize_gaussians(bg, dev["means3D"], empty, dev["opacity"], dev["scales"], dev["rotations"], 1.0, empty,
                                    c["view"], c["proj"], c["tanfovx"], c["tanfovy"], H, W, dev["sh"], D, c["campos"], False, False)
        Rs[i], Vs[i] = int(fw[0]), int((fw[3] > 0).sum())
    ref, rates = None, []
    for K in (1, 2, 3):
        ms, gps, digest = bench.run_views_in_flight(_lib.load(), _C, device, dev, dcams, P, W, H, D, Vs, Rs, K, 24, gc_, go_)
        ref = ref or digest
        assert digest == ref, (K, digest, ref)
        rates.append(gps)
    assert all(r > 0 for r in rates)


@pytest.mark.parametrize("split_sh", [False, True])
def test_accumulating_backward_equals_the_sequential_sum_bit_for_bit(hip_lib, split_sh):
    """g4s_rasterizer_backward_accumulate (several views per GPU per step, accumulated locally): the first
    view's backward writes the running sums, the following views ADD to them.  Against separate backwards summed by torch
    in the same order -- (g0 + g1) + g2 + ... -- every parameter gradient is BIT-identical, rows no view sees stay exactly
    zero, the per-view outputs (dL_dmeans2D) are those of the separate call; and the same with the views in flight on
    three HIP streams, only the accumulating kernel ordered by the previous view's event (`out["after"]`)."""
    root = os.path.dirname(HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from g4splat_amd.diff_surfel_rasterization import _C
    from g4splat_amd.pipeline import ViewPipeline
    device = torch.device("cuda", 0)
    scene, cams, dev, dcams, (P, W, H, D) = bench.build_scene("s2", device)
    dcams = dcams[:5]
    bg = torch.zeros(3, device=device)
    empty = torch.empty(0, device=device)
    g = torch.Generator(device=device).manual_seed(1)
    gc_ = torch.randn((3, H, W), device=device, generator=g)
    go_ = torch.randn((7, H, W), device=device, generator=g)
    sh = (dev["sh"][:, :1].contiguous(), dev["sh"][:, 1:].contiguous()) if split_sh else dev["sh"]
    names = ["dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations"] + (["dL_dsh_dc", "dL_dsh_rest"] if split_sh else ["dL_dsh"])

    def pick(gr):  # the returned tuple -> {name: tensor} of the parameter gradients
        d = {"dL_dmeans3D": gr[3], "dL_dopacity": gr[2], "dL_dscales": gr[6], "dL_drotations": gr[7]}
        if split_sh:
            d["dL_dsh_dc"], d["dL_dsh_rest"] = gr[5]
        else:
            d["dL_dsh"] = gr[5]
        return d

    def fwd(c, state=None):
        a = (bg, dev["means3D"], empty, dev["opacity"], dev["scales"], dev["rotations"], 1.0, empty, c["view"], c["proj"],
             c["tanfovx"], c["tanfovy"], H, W, sh, D, c["campos"], False, False)
        return _C.rasterize_gaussians_presized(state, *a) if state is not None else _C.rasterize_gaussians(*a)

    def bwd(c, fw, out=None):
        return _C.rasterize_gaussians_backward(bg, dev["means3D"], fw[3], empty, dev["scales"], dev["rotations"], 1.0, empty,
                                               c["view"], c["proj"], c["tanfovx"], c["tanfovy"], gc_, go_, sh, D, c["campos"],
                                               fw[4], fw[0], fw[5], fw[6], False, out=out)

    # separate backwards, summed by torch in view order
    want, mean2d, seen, Rmax = None, [], torch.zeros(P, dtype=torch.bool, device=device), 0
    count = torch.zeros(P, device=device)
    for c in dcams:
        fw = fwd(c)
        Rmax = max(Rmax, int(fw[0]))
        gr = bwd(c, fw)
        seen |= fw[3] > 0
        count += (fw[3] > 0).float()
        mean2d.append(gr[0].clone())
        d = pick(gr)
        want = {k: v.clone() for k, v in d.items()} if want is None else {k: want[k] + d[k] for k in want}
    assert 0 < int(seen.sum()) < P

    # (a) one stream: first view overwrites, the others accumulate
    sums = {k: torch.full_like(v, float("nan")) for k, v in want.items()}  # (the first view must overwrite every element)
    for j, c in enumerate(dcams):
        fw = fwd(c)
        out = dict(sums)
        if j:
            out["accumulate"] = True
        gr = bwd(c, fw, out=out)
        assert torch.equal(gr[0], mean2d[j])
    for k in names:
        assert torch.equal(sums[k], want[k]), k
        assert not bool(sums[k][~seen].any()), k

    # (b) three views in flight; only the accumulating kernel waits for the previous view
    pipe = ViewPipeline(P, W, H, int(Rmax * 1.25) + 4096, device, k=3)
    for rep in range(2):
        sums2 = {k: torch.full_like(v, float("nan")) for k, v in want.items()}
        stats = torch.full((P, 2), float("nan"), device=device)
        for j, c in enumerate(dcams):
            with pipe.slot(j) as (state, work):
                fw = fwd(c, state)
                out = dict(sums2)
                out["workspace"] = work
                out["view_stats"] = stats
                out["accumulate"] = True if j else "first"   # the same entry point starts the sums
                if j:
                    out["after"] = pipe.previous_view_done
                gr = bwd(c, fw, out=out)
        pipe.join()
        torch.cuda.synchronize()
        assert not pipe.overflowed()
        for k in names:
            assert torch.equal(sums2[k], want[k]), (rep, k)
        # the fused densification statistics: sum over the views of the per-view ||dL_dmeans2D.xy||, and the view count
        want_norm = torch.zeros(P, device=device)
        for m in mean2d:
            want_norm += torch.sqrt(m[:, 0] * m[:, 0] + m[:, 1] * m[:, 1])
        assert torch.allclose(stats[:, 0], want_norm, rtol=2e-6, atol=0) and float(want_norm.max()) > 0
        assert torch.equal(stats[:, 1], count)

    # refusals: the running sums must be handed in, and `after` belongs to an accumulating call
    fw = fwd(dcams[0])
    with pytest.raises(RuntimeError, match="running sums"):
        bwd(dcams[0], fw, out={"accumulate": True})
    with pytest.raises(RuntimeError, match="only applies"):
        bwd(dcams[0], fw, out={"after": torch.cuda.Event()})
    with pytest.raises(RuntimeError, match="only applies"):
        bwd(dcams[0], fw, out={"view_stats": torch.zeros(P, 2, device=device)})


def test_multi_view_batch_through_render_and_autograd_on_two_streams():
    """A multi-view batch (8 views over fewer than 8 GPUs, accumulated locally) through the reference's
    own call sequence -- gaussian_renderer.render(), a loss, loss.backward() -- with each view inside a ViewPipeline slot:
    the autograd node picks the slot's PresizedState up by itself (diff_surfel_rasterization.presized), forward and
    backward of a view run on the slot's stream, the parameters' .grad accumulate across views -- in program order, which
    ViewPipeline.order_accumulation enforces (two backward() calls on two streams are unrelated graphs to autograd and
    would add into .grad concurrently).  Must equal the same batch rendered view after view on the default stream, bit for
    bit, images and summed gradients, five times in a row."""
    import contextlib
    import math
    from types import SimpleNamespace
    import numpy as np
    from g4splat_amd import _lib, synthetic
    from g4splat_amd.gaussian_model import GaussianModel
    from g4splat_amd.gaussian_renderer import render
    from g4splat_amd.pipeline import ViewPipeline
    dev = torch.device("cuda", 0)
    P, W, H = 120_000, 640, 400
    scene = synthetic.scene_room(P, seed=2)
    cams = []
    for c in synthetic.room_cameras(4, W, H, fovx_deg=90.0):
        cams.append(SimpleNamespace(image_width=W, image_height=H, FoVx=c.FoVx, FoVy=c.FoVy,
                                    world_view_transform=torch.tensor(c.world_view_transform, device=dev),
                                    full_proj_transform=torch.tensor(c.full_proj_transform, device=dev),
                                    camera_center=torch.tensor(c.camera_center, device=dev), znear=0.01, zfar=100.0))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    model = GaussianModel(sh_degree=3)
    model.create_from_parameters(t(scene.means3D), t(scene.scales), t(scene.rotations),
                                 t(np.clip(scene.shs[:, 0, :] * 0.28209479177387814 + 0.5, 0, 1).astype(np.float32)))
    model.active_sh_degree = 3
    params = [model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation, model._opacity]
    for p in params:
        p.requires_grad_(True)
        p.grad = torch.zeros_like(p)
    cfg = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    g = torch.Generator(device=dev).manual_seed(5)
    wc = [torch.randn((3, H, W), device=dev, generator=g) for _ in cams]
    wn = [torch.randn((3, H, W), device=dev, generator=g) for _ in cams]

    def view_loss(out, i):
        return (out["render"] * wc[i]).sum() + (out["rend_normal"] * wn[i]).sum() + 10.0 * out["rend_dist"].mean() \
            + out["surf_depth"].mean()

    def run(pipe):
        for p in params:
            p.grad.zero_()
        images = []
        for i, cam in enumerate(cams):
            ctx = pipe.slot(i) if pipe is not None else contextlib.nullcontext()
            with ctx:
                out = render(cam, model, cfg, bg)
                view_loss(out, i).backward()
                images.append(out["render"].detach())
        if pipe is not None:
            pipe.join()
        torch.cuda.synchronize()
        return [im.cpu() for im in images], [p.grad.detach().cpu().clone() for p in params]

    lib = _lib.load()
    im0, g0 = run(None)
    # capacity: the largest instance count of the four views, with headroom (the host never sees the counts in a slot)
    from g4splat_amd.diff_surfel_rasterization import _C
    R, empty = 0, torch.empty(0, device=dev)
    for cam in cams:
        fw = _C.rasterize_gaussians(bg, model.get_xyz.detach(), empty, model.get_opacity.detach(), model.get_scaling.detach(),
                                    model.get_rotation.detach(), 1.0, empty, cam.world_view_transform, cam.full_proj_transform,
                                    math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), H, W, model.get_features.detach(), 3,
                                    cam.camera_center, False, False)
        R = max(R, int(fw[0]))
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    pipe = ViewPipeline(P, W, H, int(R * 1.25), dev, k=2)
    pipe.order_accumulation(params)
    for rep in range(5):
        im1, g1 = run(pipe)
        assert not pipe.overflowed()
        for a, b in zip(im0, im1):
            assert torch.equal(a, b)
        for name, a, b in zip(("xyz", "f_dc", "f_rest", "scaling", "rotation", "opacity"), g0, g1):
            assert a.abs().max() > 0, name
            assert torch.equal(a, b), (rep, name, (a - b).abs().max().item() / (a.abs().max().item() + 1e-30))
    pipe.release_hooks()
    # and the slot really was used: the second run's forward states are the pipeline's
    assert lib is not None and all(int(st.status[1].item()) > 0 for st in pipe.states)


def test_view_parallel_accumulate_pipelines_by_default_and_matches_the_sequential_loop():
    """Verdict r3 item 4: ViewParallel.accumulate() puts a rank's views of a step through a ViewPipeline by default (two in
    flight, accumulation ordered).  20 optimiser steps (FusedAdam) of 4 views each, through the reference's call
    sequence (render() -> loss -> backward()): parameters, Adam moments and densification statistics equal those of the
    plain sequential loop (in_flight=1) BIT FOR BIT; the pipeline really ran (its states hold frames), the first step at a
    size runs sequentially to learn the instance capacity, and a dropped .grad buffer is refused."""
    from types import SimpleNamespace
    import numpy as np
    from g4splat_amd import synthetic
    from g4splat_amd.gaussian_model import GaussianModel
    from g4splat_amd.gaussian_renderer import render
    from g4splat_amd.parallel import ViewParallel
    dev = torch.device("cuda", 0)
    P, W, H = 100_000, 640, 400
    scene = synthetic.scene_room(P, seed=4)
    cams = []
    for c in synthetic.room_cameras(4, W, H, fovx_deg=90.0):
        cams.append(SimpleNamespace(image_width=W, image_height=H, FoVx=c.FoVx, FoVy=c.FoVy,
                                    world_view_transform=torch.tensor(c.world_view_transform, device=dev),
                                    full_proj_transform=torch.tensor(c.full_proj_transform, device=dev),
                                    camera_center=torch.tensor(c.camera_center, device=dev), znear=0.01, zfar=100.0))
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    cfg = SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False)
    bg = torch.zeros(3, device=dev)
    g = torch.Generator(device=dev).manual_seed(5)
    target = [torch.rand((3, H, W), device=dev, generator=g) for _ in cams]

    def train(in_flight, shrink_capacity_after_first_step=False):
        model = GaussianModel(sh_degree=3)
        model.create_from_parameters(t(scene.means3D), t(scene.scales), t(scene.rotations),
                                     t(np.clip(scene.shs[:, 0, :] * 0.28209479177387814 + 0.5, 0, 1).astype(np.float32)))
        model.active_sh_degree = 3
        model.training_setup()
        vp = ViewParallel(model.parameters())
        idx = {id(c): i for i, c in enumerate(cams)}

        def view_step(cam):
            out = render(cam, model, cfg, bg)
            loss = (out["render"] - target[idx[id(cam)]]).abs().mean() + 0.05 * out["rend_dist"].mean() \
                + 0.01 * (1 - out["rend_alpha"]).mean()
            loss.backward()
            return out

        for it in range(20):
            vp.accumulate(cams, view_step, in_flight=in_flight)
            if it == 0 and shrink_capacity_after_first_step:
                # as if the first step's cameras had seen half of what the later ones bin (training draws
                # other cameras every step): the second step's views outgrow the pipeline that is built from this
                (key, cap), = vp._capacity.items()
                vp._capacity[key] = cap // 3
            stats = vp.all_reduce()
            model.optimizer.step()
            if it == 19:
                keep = {k: v.clone() for k, v in stats.items()}
            vp.zero()
        torch.cuda.synchronize()
        state = [model.optimizer.state[p] for p in model.parameters()]
        return ([p.detach().clone() for p in model.parameters()],
                [st[k].clone() for st in state for k in ("exp_avg", "exp_avg_sq")], keep, vp, model)

    pa, ma, sa, vpa, _ = train(1)
    pb, mb, sb, vpb, model_b = train(2)
    assert vpa._pipe is None
    assert vpb._pipe is not None and vpb._pipe.k == 2 and all(int(st.status[1].item()) > 0 for st in vpb._pipe.states)
    for name, a, b in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), pa, pb):
        assert torch.equal(a, b), (name, (a - b).abs().max().item())
    for a, b in zip(ma, mb):
        assert torch.equal(a, b)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert float(sa["vis_count"].max()) >= 2.0 and float(sa["grad_norm_sum"].max()) > 0
    # A view that outgrows the pipeline's capacity: the step is discarded, the capacity grows to what
    # was seen, the step is redone view after view, the NEXT step runs pipelined again at the new size -- and nothing of
    # that shows in the result.
    pc, mc, sc, vpc, _ = train(2, shrink_capacity_after_first_step=True)
    assert vpc.regrown == 1 and vpb.regrown == 0
    assert vpc._pipe is not None and vpc._pipe.states[0].capacity > 2 * (next(iter(vpb._capacity.values())) // 3)
    for name, a, c in zip(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"), pa, pc):
        assert torch.equal(a, c), (name, (a - c).abs().max().item())
    for a, c in zip(ma, mc):
        assert torch.equal(a, c)
    for k in sa:
        assert torch.equal(sa[k], sc[k]), k
    vpc._pipe.release_hooks()
    # a dropped gradient buffer is refused before a view is issued
    model_b.optimizer.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="no .grad buffer"):
        vpb.accumulate(cams, lambda cam: None, in_flight=2)
    vpb.zero()  # re-attaches the bucket views
    vpb._pipe.release_hooks()


def test_backward_refuses_a_presized_state_that_rendered_another_view_since():
    """The forward's scratch of a view rendered inside a slot IS the slot's PresizedState: forward(A), forward(B) through
    the same state, backward(A) would read B's lists.  The autograd node keeps the state's generation and raises."""
    import math
    from types import SimpleNamespace
    import numpy as np
    from g4splat_amd import synthetic
    from g4splat_amd.diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer, presized, _C
    dev = torch.device("cuda", 0)
    P, W, H = 5000, 160, 96
    scene, cam = synthetic.scene_random(P, seed=1, width=W, height=H)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    xyz = t(scene.means3D).requires_grad_(True)
    args = dict(means2D=torch.zeros_like(xyz), opacities=t(scene.opacities), shs=t(scene.shs), scales=t(scene.scales),
                rotations=t(scene.rotations))
    rs = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                                       bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=t(cam.world_view_transform),
                                       projmatrix=t(cam.full_proj_transform), sh_degree=3, campos=t(cam.camera_center),
                                       prefiltered=False, debug=False)
    rast = GaussianRasterizer(rs)
    state = _C.PresizedState(P, W, H, 400_000, dev)
    with presized(state):
        c1, _r1, _d1 = rast(xyz, **args)
        c2, _r2, _d2 = rast(xyz, **args)
        with pytest.raises(RuntimeError, match="used by another forward since"):
            c1.sum().backward()
        c2.sum().backward()  # the latest forward's backward is fine
    assert xyz.grad is not None and torch.isfinite(xyz.grad).all() and xyz.grad.abs().max() > 0


# ---- densification under data parallelism on the device (verdict r5 item 3) ---------------------------------
DZ_P, DZ_W, DZ_H, DZ_VIEWS, DZ_STEPS = 40_000, 320, 208, 8, 3


def _dz_scene(dev):
    """-> (model, cameras, targets, cfg, bg): a small room, eight cameras, one target image per camera."""
    from types import SimpleNamespace
    import numpy as np
    from g4splat_amd import synthetic
    from g4splat_amd.gaussian_model import GaussianModel
    scene = synthetic.scene_room(DZ_P, seed=9)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), device=dev)
    cams = []
    for c in synthetic.room_cameras(DZ_VIEWS, DZ_W, DZ_H, fovx_deg=90.0):
        cams.append(SimpleNamespace(image_width=DZ_W, image_height=DZ_H, FoVx=c.FoVx, FoVy=c.FoVy,
                                    world_view_transform=t(c.world_view_transform), full_proj_transform=t(c.full_proj_transform),
                                    camera_center=t(c.camera_center), znear=0.01, zfar=100.0))
    model = GaussianModel(sh_degree=3)
    model.create_from_parameters(t(scene.means3D), t(scene.scales), t(scene.rotations),
                                 t(np.clip(scene.shs[:, 0, :] * 0.28209479177387814 + 0.5, 0, 1).astype(np.float32)))
    with torch.no_grad():
        model._opacity[::9] -= 7.0   # translucent rows: pruned
    model.active_sh_degree = 3
    model.training_setup()
    g = torch.Generator(device="cpu").manual_seed(5)
    targets = [torch.rand((3, DZ_H, DZ_W), generator=g).to(dev) for _ in cams]
    return model, cams, targets, SimpleNamespace(depth_ratio=0.0, compute_cov3D_python=False), torch.zeros(3, device=dev)


def _dz_view_step(model, cam, target, cfg, bg):
    from g4splat_amd.gaussian_renderer import render
    out = render(cam, model, cfg, bg)
    loss = (out["render"] - target).abs().mean() + 0.05 * out["rend_dist"].mean() + 0.01 * (1 - out["rend_alpha"]).mean()
    loss.backward()
    return out


def _dz_densify_args(model):
    grads = (model.xyz_gradient_accum / model.denom).nan_to_num(0.0)
    extent = float(model.get_scaling.detach().max(dim=1).values.median()) / 0.01   # half of the selected rows clone, half split
    return dict(max_grad=float(grads[grads > 0].median()), min_opacity=0.005, extent=extent, max_screen_size=20)


def _dz_pack(model, exp_avg, exp_avg_sq, history):
    # (numpy, pickled by value: a torch tensor in a multiprocessing queue travels as a shared-memory handle that dies with
    # the rank that sent it)
    c = lambda ts: torch.cat([t.detach().reshape(-1) for t in ts]).cpu().numpy()
    n = lambda t: t.detach().cpu().numpy()
    return dict(params=c(model.parameters()), exp_avg=c(exp_avg), exp_avg_sq=c(exp_avg_sq), accum=n(model.xyz_gradient_accum),
                denom=n(model.denom), radii=n(model.max_radii2D), history=history)


def _dz_worker(rank, world, port, q, backend="gloo"):
    root = os.path.dirname(HERE)
    for p in (root, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from g4splat_amd.parallel import ReplicatedDensification, ViewParallel
    model, cams, targets, cfg, bg = _dz_scene(dev)
    vp = ViewParallel(model.parameters(), exchange="owner")
    opt = vp.sharded_adam([g["lr"] for g in model.optimizer.param_groups])
    dz = ReplicatedDensification(model, vp, sharded=opt, base_seed=11)
    history = []
    for it in range(1, 2 * DZ_STEPS + 1):
        out = _dz_view_step(model, cams[rank], targets[rank], cfg, bg)        # one view per rank per step
        vp.record_view(out["viewspace_points"], out["visibility_filter"], out["radii"])
        vp.reduce_to_owners()
        opt.step(extra=[vp.side])
        dz.add_stats(vp.stats_after_owner_step())
        vp.zero()
        if it == DZ_STEPS:
            before = model._xyz.shape[0]
            dz.densify_and_prune(it, **_dz_densify_args(model))
            history.append((before, model._xyz.shape[0]))
    ea, es = opt.full_state()
    torch.cuda.synchronize()
    out = _dz_pack(model, ea, es, history)
    out["rccl"] = bool(vp._ensure_owner().rccl)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def _dz_single_process(world):
    """The same schedule in ONE process with no exchange code at all: the `world` views rendered one after the other, their
    gradients ACCUMULATED by autograd into one bucket (and the per-view statistics summed) in view order, FusedAdam over the
    full tensors, densify_and_prune with the iteration's generator on the model's own optimiser state.  The owner exchange
    sums every row in rank order (OwnerReduce.rank_order), which is this order -- so the comparison is bit for bit."""
    from g4splat_amd.parallel import ViewParallel, densify_generator
    dev = torch.device("cuda", 0)
    model, cams, targets, cfg, bg = _dz_scene(dev)
    vp = ViewParallel(model.parameters())
    history = []
    for it in range(1, 2 * DZ_STEPS + 1):
        for r in range(world):
            out = _dz_view_step(model, cams[r], targets[r], cfg, bg)
            vp.record_view(out["viewspace_points"], out["visibility_filter"], out["radii"])
        with torch.no_grad():
            model.optimizer.step()
            model.xyz_gradient_accum += vp.grad_norm_sum
            model.denom += vp.vis_count
            model.max_radii2D = torch.maximum(model.max_radii2D, vp.max_radii)
            vp.zero()
            if it == DZ_STEPS:
                before = model._xyz.shape[0]
                model.densify_and_prune(generator=densify_generator(dev, it, 11), **_dz_densify_args(model))
                vp.rebind(model.parameters())
                history.append((before, model._xyz.shape[0]))
    st = [model.optimizer.state[p] for p in model.parameters()]
    torch.cuda.synchronize()
    return _dz_pack(model, [x["exp_avg"] for x in st], [x["exp_avg_sq"] for x in st], history)


@pytest.mark.parametrize("world", [8, pytest.param(2, marks=pytest.mark.exhaustive)])
def test_densify_and_prune_on_every_replica_with_eight_ranks_on_one_gpu(world):
    """Verdict r5 item 3(b): BASELINE config 4's shape -- eight training views, one per rank -- with the ranks sharing the
    one GPU of the test box (gloo; the device path of the exchange, g4s_adam_step on the owners' shards, the HIP
    compaction kernels of densify.py, torch.normal on the device from the iteration's generator).  Three ZeRO-1 steps, ONE
    densify_and_prune on every replica (clone + split + prune + screen-size limit; the Adam moments gathered for the edit
    and re-sharded for the new row count), three more steps.  Parameters, moments and statistics: the same bits on all
    eight ranks, and THE SAME BITS AS ONE PROCESS THAT ACCUMULATES THE EIGHT VIEWS ONE AFTER THE OTHER (plain autograd
    accumulation, FusedAdam, no exchange code): the owner exchange sums every row in rank order."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 43500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_dz_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = _dz_single_process(world)
    (before, after), = ref["history"]
    assert before == DZ_P and abs(after - before) > 100, ref["history"]
    for r in range(world):
        assert res[r]["history"] == ref["history"], (r, res[r]["history"], ref["history"])
        for key in ("params", "exp_avg", "exp_avg_sq", "accum", "denom", "radii"):
            assert np.array_equal(res[r][key], res[0][key]), f"rank {r} differs from rank 0 in {key}"
            assert np.array_equal(res[r][key], ref[key]), (f"rank {r} differs from the single-process run in {key}",
                                                           float(np.abs(res[r][key] - ref[key]).max()))


def test_densify_and_prune_through_the_rccl_backend():
    """The same schedule with every collective executed BY RCCL (backend nccl, the world size a one-GPU box allows): the
    radii MAX, the zero-split all_to_all, the owner step, the grouped in-place all_gather of the parameters, the all-gather of
    the Adam moments for the edit and their re-sharding run on RCCL's persistent-buffer path; the result equals the
    single-process run bit for bit."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_dz_worker, args=(0, 1, 44500 + (os.getpid() % 2000), q, "nccl"))
    p.start()
    _rank, res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and res["rccl"]
    ref = _dz_single_process(1)
    assert res["history"] == ref["history"] and ref["history"][0][0] != ref["history"][0][1]
    for key in ("params", "exp_avg", "exp_avg_sq", "accum", "denom", "radii"):
        assert np.array_equal(res[key], ref[key]), key
