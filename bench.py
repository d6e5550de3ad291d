#!/usr/bin/env python
"""bench.py -- rasterized Gaussians/s, forward+backward, on MI355X.

A "step" is one pass of the hot path over one training view: one
`_C.rasterize_gaussians` + one `_C.rasterize_gaussians_backward` on
synthetic data already resident in HBM; with N > 1 ranks every rank renders its own view
of the replicated scene and the step also exchanges the parameter gradients over RCCL
(SURVEY.md 8(e)).

Workload at N=1 (BASELINE.json metric "rasterized Gaussians/s fwd+bwd @1600x1200"):
S3 = configs[2] stand-in: 1.5 M surfels on the faces of a 6x4x3 m room, 1600x1200, SH degree 3.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (re-exec through
torch.distributed.run on 127.0.0.1); under torch.distributed.run it uses the ranks it is given.

  --scaling weak    (default) one view per rank per step: the job renders N views per step.
  --scaling strong  SURVEY.md 8(e)'s form: 8 views per optimiser step in total, 8/N per rank, accumulated
                    locally (views in flight on HIP streams, gradients summed in program order), then ONE exchange.

Timing: a pass is EXACTLY K steps between barrier + synchronize on both sides (host clock, MAX over the ranks), with one
HIP event on the launch stream behind every step.  A pass whose mean step exceeds its median step by more than 10 % is
`disturbed` (something other than the work landed in it; the steps that carry the excess are named).  Passes are
repeated until three are undisturbed (at most --attempts); `ms_per_step` / `value` are those of the MEDIAN undisturbed pass
-- the contract's quotient, units of the K steps / time of the K steps -- and `timing` lists every pass, the per-step
durations and SURVEY.md 8(d)'s median step.  If no pass was clean the line says `"disturbed": true`.  The warm-up is by
state, not by count: after the W steps the driver asks for, stepping continues until three consecutive windows of ~10 steps
agree within 2 % (at most ~1 s).

prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
`roofline` for the dominant kernel and `cpu_baseline` (the CPU oracle timed on the host
cores on a bounded sample; reported baseline, not a target).
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from g4splat_amd.benchlib import (HBM_PEAK_GBS, SIMDS, STRONG_VIEWS, WORKLOADS, algorithmic_bytes, assemble_line,  # noqa: E402,F401
                                   exchange_report, pick_headline, roofline_report, summarize_steps)
from g4splat_amd import benchlib  # noqa: E402

TRAINED_INFO = {}      # workload -> what trained_scene.scene_trained reported (printed under config.trained_scene)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="s3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--sustained-seconds", type=float, default=1.0,
                    help="after the headline pass, keep stepping for at least this long (and >= 500 steps) and report "
                         "the settled throughput as `sustained` (0 = skip)")
    ap.add_argument("--no-gc-freeze", action="store_true",
                    help="leave Python's cyclic garbage collector as it is (default: gc.collect() + gc.freeze() before "
                         "the warm-up, so that a full collection over the interpreter's ~10^6 long-lived objects cannot "
                         "land inside a timed pass; host-side hygiene, the GPU work is unchanged)")
    ap.add_argument("--no-settle", action="store_true", help="skip the state-based part of the warm-up")
    ap.add_argument("--attempts", type=int, default=7,
                    help="upper bound on the headline passes (K steps each; repeated until three are undisturbed)")
    ap.add_argument("--views-in-flight", type=int, default=3,
                    help="N = 1 only, reported next to the headline (never as `value`): throughput with up to this many "
                         "INDEPENDENT views in flight on separate HIP streams (multi-view batches; 0 / 1 = skip)")
    ap.add_argument("--presized", action="store_true",
                    help="use g4s_rasterizer_forward_presized (no host read-back; extension) instead of the "
                         "reference-shaped forward -- for the step-time comparison in DESIGN.md, not the default")
    return ap.parse_args(argv)


def self_spawn(args):
    """`python bench.py --gpus N` outside torch.distributed.run: become the launcher (one rank per GPU, rendezvous on
    127.0.0.1, a free port), exactly the command the driver would have issued."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (task environment)
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def build_scene(name, device):
    import numpy as np
    import torch
    from g4splat_amd import synthetic
    P, W, H, D, nviews = WORKLOADS[name]
    if name == "s1":
        scene, cam = synthetic.scene_random(P, seed=0, width=W, height=H)
        cams = [cam]
    elif name == "s3t":
        from g4splat_amd import trained_scene
        scene, TRAINED_INFO["s3t"] = trained_scene.scene_trained(seed=0, iters=1000, P=P, width=W, height=H, nviews=nviews,
                                                                 device=device)
        P = int(scene.means3D.shape[0])
        cams = synthetic.room_cameras(nviews, W, H, fovx_deg=90.0)
    else:
        scene = synthetic.scene_room(P, seed=0)
        cams = synthetic.room_cameras(nviews, W, H, fovx_deg=90.0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)
    dev = dict(means3D=t(scene.means3D), scales=t(scene.scales), rotations=t(scene.rotations),
               opacity=t(scene.opacities), sh=t(scene.shs))
    dcams = [dict(view=t(c.world_view_transform), proj=t(c.full_proj_transform), campos=t(c.camera_center),
                  tanfovx=c.tanfovx, tanfovy=c.tanfovy) for c in cams]
    return scene, cams, dev, dcams, (P, W, H, D)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)  # does not return

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # bring-up aids (never set by the driver): G4S_BENCH_BACKEND=gloo + G4S_BENCH_ONE_DEVICE=1 run several ranks on
    # ONE GPU to exercise the N > 1 control flow where only a single-GPU box is available (RCCL refuses two ranks
    # on one device); the numbers of such a run mean nothing.
    backend = os.environ.get("G4S_BENCH_BACKEND", "nccl")
    if os.environ.get("G4S_BENCH_ONE_DEVICE"):
        local_rank = 0
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    if args.scaling == "strong" and STRONG_VIEWS % world:
        raise SystemExit(f"--scaling strong shards {STRONG_VIEWS} views per step: N must divide {STRONG_VIEWS}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("G4S_FORCE_DIST"):  # G4S_FORCE_DIST: exercise the RCCL path on one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")  # (torch.distributed.run sets both; this is for G4S_FORCE_DIST)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from g4splat_amd import _lib, build
    if rank == 0:
        build.build()
    if dist is not None:
        dist.barrier()
    lib = _lib.load()
    build_id = lib.g4s_version().decode().split("build ")[-1]
    # bring-up aid (never set by the driver): G4S_BENCH_OPTIONS="side_stream=0,..." -> g4s_set_option, for A/B runs
    lib_options = {}
    for kv in filter(None, os.environ.get("G4S_BENCH_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        _lib.set_option(k.strip(), int(v))
        lib_options[k.strip()] = int(v)
    from g4splat_amd.diff_surfel_rasterization import _C

    scene, cams, dev, dcams, (P, W, H, D) = build_scene(args.workload, device)
    N = W * H
    strong = args.scaling == "strong"
    if strong and len(dcams) != STRONG_VIEWS:
        raise SystemExit(f"--scaling strong needs a workload with {STRONG_VIEWS} views")
    views_per_rank = STRONG_VIEWS // world if strong else 1
    bg = torch.zeros(3, device=device)
    empty = torch.empty(0, device=device)
    g = torch.Generator(device=device).manual_seed(1)
    dL_dcolor = torch.randn((3, H, W), device=device, generator=g)
    dL_dothers = torch.randn((7, H, W), device=device, generator=g)

    # one persistent flat fp32 bucket holds the parameter gradients of this rank's view(s) (xyz, SH,
    # opacity, scale, rotation = 58 floats per Gaussian at SH degree 3) plus the densification side channel (per-view
    # ||grad_means2D||, visibility); the backward writes straight into views of it (`out=`).  The exchange is
    # g4splat_amd.parallel.OwnerReduce (three collectives per step on persistent buffers: radii MAX + sizes, all_to_all
    # of visible rows to their owners, grouped in-place all_gather of the reduced shards) or, as the fallback,
    # RowSparseAllReduce (MAX all-reduce of the radii, then one SUM all-reduce of the rows visible on some rank).
    # G4S_BENCH_EXCHANGE=owner (default) | allreduce.  If the owner exchange fails on this machine's RCCL during
    # warm-up the bench falls back to the all-reduce -- agreed on by all ranks -- and says so (`exchange.why`).
    exchange = os.environ.get("G4S_BENCH_EXCHANGE", "owner")
    exchange_why = "requested" if "G4S_BENCH_EXCHANGE" in os.environ else "default"
    M = int(dev["sh"].shape[1])
    shapes = [("dL_dmeans3D", (P, 3)), ("dL_dsh", (P, M, 3)), ("dL_dopacity", (P, 1)), ("dL_dscales", (P, 2)),
              ("dL_drotations", (P, 4)), ("side", (P, 2))]

    def make_bucket():
        offs, o = [], 0
        for _n, shp in shapes:
            offs.append(o)
            o += (int(np.prod(shp)) + 63) // 64 * 64  # 256-B aligned views
        flat = torch.zeros(o, device=device)
        views = {n: flat[b:b + int(np.prod(shp))].view(shp) for (n, shp), b in zip(shapes, offs)}
        return flat, views

    grad_out = side = rmax = reducer = bucket = None
    if dist is not None or strong:
        bucket, views = make_bucket()
        side = views.pop("side")
        grad_out = views
        rmax = torch.zeros((P,), dtype=torch.int32, device=device)
    if dist is not None:
        from g4splat_amd.parallel import OwnerReduce, RowSparseAllReduce
        row_views = [v.view(P, -1) for v in grad_out.values()] + [side]
        reducer = OwnerReduce(row_views) if exchange == "owner" else RowSparseAllReduce(bucket, row_views)
    exchanged_rows = []
    exchange_events = None  # list of (start, stop) events while the instrumented pass runs
    # weak scaling: one view per exchange, so the backward can write the send rows itself (OwnerReduce.prepack);
    # G4S_BENCH_PREPACK=0 keeps the pack launch (A/B)
    prepack = (dist is not None and exchange == "owner" and not strong and os.environ.get("G4S_BENCH_PREPACK", "1") != "0")

    pstate = None
    pipe = None
    vis_union = torch.zeros((P,), dtype=torch.bool, device=device) if strong else None

    def views_of_step(i):
        """The camera indices THIS rank renders in step i."""
        return benchlib.views_of_step(i, rank, world, len(dcams), strong, views_per_rank)

    def exchange_step(radii):
        """weak scaling: the collectives (the owner exchange's begin() has been issued right after the forward; the view's
        statistics were written into `side` by the backward)."""
        nonlocal exchange_events
        ev = None
        if exchange_events is not None:  # instrumented pass only: GPU time of the exchange, this rank
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if exchange == "owner":
            # (reducer.max_radii = MAX over the ranks of the radii, from begin()'s collective)
            reducer.finish(prepacked=prepack)
            exchanged_rows.append(reducer.last_rows_sent)
        else:
            rmax.copy_(radii)
            dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
            reducer.reduce(rmax > 0)
            exchanged_rows.append(reducer.last_rows)
        if ev is not None:
            ev[1].record()
            exchange_events.append(ev)

    def step_weak(i):
        nonlocal pstate
        cam = dcams[views_of_step(i)[0]]
        if args.presized and pstate is not None:
            fw = _C.rasterize_gaussians_presized(pstate, bg, dev["means3D"], empty, dev["opacity"], dev["scales"],
                                                 dev["rotations"], 1.0, empty, cam["view"], cam["proj"], cam["tanfovx"],
                                                 cam["tanfovy"], H, W, dev["sh"], D, cam["campos"], False, False)
        else:
            fw = _C.rasterize_gaussians(bg, dev["means3D"], empty, dev["opacity"], dev["scales"], dev["rotations"], 1.0,
                                        empty, cam["view"], cam["proj"], cam["tanfovx"], cam["tanfovy"], H, W, dev["sh"],
                                        D, cam["campos"], False, False)
        R, color, others, radii, geom, binning, img = fw
        if dist is not None and exchange == "owner":
            # sizes travel to the host while the backward runs; the radii MAX (-> max_radii2D) rides in the same collective
            reducer.begin(radii > 0, radii=radii)
        out = grad_out
        if dist is not None:
            # the densification statistics of the view (||dL_dmeans2D.xy||, visible) come out of the per-Gaussian kernel
            # itself (g4s_rasterizer_backward_accumulate, first_view: every row written like the plain call): no torch
            # kernels between the backward and the exchange
            out = dict(grad_out, accumulate="first", view_stats=side)
            if prepack:  # ... and so do the rows this rank sends to the owners, already in the all_to_all's layout
                out["packed"] = reducer.prepack(radii > 0)
        _C.rasterize_gaussians_backward(bg, dev["means3D"], radii, empty, dev["scales"], dev["rotations"],
                                        1.0, empty, cam["view"], cam["proj"], cam["tanfovx"], cam["tanfovy"],
                                        dL_dcolor, dL_dothers, dev["sh"], D, cam["campos"], geom, R, binning,
                                        img, False, out=out)
        if dist is not None:
            exchange_step(radii)
        return [(R, radii)]

    def step_strong(i):
        """One optimiser step of SURVEY.md 8(e)'s form: this rank's 8/N views go through a ViewPipeline (presized forward,
        up to three views in flight on HIP streams); the first view's backward writes the step's gradient bucket, the
        following ones add their visible rows to it in program order (accumulating backward: only its per-Gaussian kernel
        waits for the previous view, so the sums are bit-identical to the sequential loop's); the exchange's begin() is issued right after the LAST view's forward (the
        union of the views' visible sets is known then) so that its collective hides behind that view's backward."""
        nonlocal exchange_events
        vs = views_of_step(i)
        L = len(vs)
        fwd_done, radii_all, res = [], [], []
        for j, c in enumerate(vs):
            cam = dcams[c]
            with pipe.slot(j) as (state, work):
                fw = _C.rasterize_gaussians_presized(state, bg, dev["means3D"], empty, dev["opacity"], dev["scales"],
                                                     dev["rotations"], 1.0, empty, cam["view"], cam["proj"],
                                                     cam["tanfovx"], cam["tanfovy"], H, W, dev["sh"], D, cam["campos"],
                                                     False, False)
                radii = fw[3]
                e = torch.cuda.Event()
                e.record()
                fwd_done.append(e)
                radii_all.append(radii)
                if j == L - 1:
                    cur = torch.cuda.current_stream(device)
                    for e2 in fwd_done[:-1]:
                        cur.wait_event(e2)  # forwards issued long ago: no stall
                    torch.gt(radii_all[0], 0, out=vis_union)
                    rmax.copy_(radii_all[0])
                    for r2 in radii_all[1:]:
                        vis_union.logical_or_(r2 > 0)
                        torch.maximum(rmax, r2, out=rmax)
                    if dist is not None and exchange == "owner":
                        reducer.begin(vis_union, radii=rmax)
                # Every view goes through g4s_rasterizer_backward_accumulate: the first one of the step starts the sums
                # (every row of the step's bucket written, dL_dsh cleared on the side), the following ones ADD their
                # visible rows; the per-view densification statistics (norm per view BEFORE any reduction,
                # gaussian_model.py:649-651) are summed by the same kernel.  Only that per-Gaussian kernel waits for
                # the previous view's slot; the blend kernels of the views overlap.
                out = dict(grad_out)
                out["workspace"] = work
                out["view_stats"] = side
                out["accumulate"] = True if j > 0 else "first"
                if j > 0:
                    out["after"] = pipe.previous_view_done
                _C.rasterize_gaussians_backward(bg, dev["means3D"], radii, empty, dev["scales"], dev["rotations"],
                                                1.0, empty, cam["view"], cam["proj"], cam["tanfovx"],
                                                cam["tanfovy"], dL_dcolor, dL_dothers, dev["sh"], D, cam["campos"],
                                                fw[4], fw[0], fw[5], fw[6], False, out=out)
                res.append((fw[0], radii))
        pipe.join()
        if dist is not None:
            ev = None
            if exchange_events is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if exchange == "owner":
                reducer.finish()
                exchanged_rows.append(reducer.last_rows_sent)
            else:
                dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
                reducer.reduce(rmax > 0)
                exchanged_rows.append(reducer.last_rows)
            if ev is not None:
                ev[1].record()
                exchange_events.append(ev)
        return res

    step = step_strong if strong else step_weak

    # Host hygiene: a generation-2 collection of CPython's cyclic GC walks every long-lived object of the process (torch,
    # numpy, the scene: ~50 ms here) and lands wherever the allocation counters trip -- measured as ONE 53-ms stall in a
    # 500-step pass.  Long-running loops freeze the startup heap; so does this one -- BEFORE the warm-up steps, so that no
    # host pause sits between the warm-up and the timed pass.  The collections that still happen are timed and reported.
    import gc
    gc_ms = [0.0, 0]
    gc_t0 = [0.0]

    def _gc_cb(phase, info):
        if phase == "start":
            gc_t0[0] = time.perf_counter()
        else:
            gc_ms[0] += (time.perf_counter() - gc_t0[0]) * 1e3
            gc_ms[1] += 1

    gc.callbacks.append(_gc_cb)
    if not args.no_gc_freeze:
        gc.collect()
        gc.freeze()

    # ---- warm-up (also measures V and R per view outside the timed region) ------------------------------------------
    Vs, Rs = {}, {}

    def note(i, res):
        for c, (R, radii) in zip(views_of_step(i), res):
            if c not in Vs:
                Vs[c] = int((radii > 0).sum().item())
                Rs[c] = int(R)

    if strong:
        # the pipeline's capacity comes from a reference-shaped forward of every view this rank renders
        for c in views_of_step(0):
            cam = dcams[c]
            fw = _C.rasterize_gaussians(bg, dev["means3D"], empty, dev["opacity"], dev["scales"], dev["rotations"], 1.0,
                                        empty, cam["view"], cam["proj"], cam["tanfovx"], cam["tanfovy"], H, W, dev["sh"],
                                        D, cam["campos"], False, False)
            Vs[c], Rs[c] = int((fw[3] > 0).sum().item()), int(fw[0])
            del fw
        from g4splat_amd.pipeline import ViewPipeline
        pipe = ViewPipeline(P, W, H, int(max(Rs.values()) * 1.25) + 4096, device, k=min(3, views_per_rank))
        torch.cuda.synchronize()
    if dist is not None and exchange == "owner":
        try:
            step(0)
            torch.cuda.synchronize()
            ok = torch.ones(1, device=device)
        except Exception as ex:  # e.g. an RCCL build without uneven all_to_all
            print(f"[bench] owner exchange failed on rank {rank}: {ex}; falling back to all-reduce", file=sys.stderr)
            ok = torch.zeros(1, device=device)
            exchange_why = f"owner exchange failed in warm-up on some rank ({type(ex).__name__}: {str(ex)[:200]})"
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0:
            exchange = "allreduce"
            prepack = False
            if exchange_why in ("default", "requested"):
                exchange_why = "owner exchange failed in warm-up on another rank"
            reducer = RowSparseAllReduce(bucket, row_views)
    for i in range(max(args.warmup, 1)):
        note(i, step(i))
    for i in range(min(len(dcams), args.steps)):  # make sure every view that the timed steps use has its V known
        if any(c not in Vs for c in views_of_step(i)):
            note(i, step(i))
    if strong:
        assert not pipe.overflowed(), "presized capacity overflow"

    if args.presized and not strong:  # capacity from the warm-up's instance counts, with head-room
        pstate = _C.PresizedState(P, W, H, int(max(Rs.values()) * 1.25) + 4096, device)
        step(0)
        torch.cuda.synchronize()
        assert pstate.status.tolist()[3] == 0

    def agree_max(x):
        """MAX over the ranks of a host float (every rank gets the same value back)."""
        if dist is None:
            return x
        tt = torch.tensor([x], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # Warm-up by state: a cold box keeps changing for a while after the first kernels (clocks, the allocator's pools,
    # page tables) -- step in windows of ~10 steps until three in a row agree within 2 %, at most ~1 s.
    settle = {"extra_steps": 0, "windows_ms_per_step": [], "settled": None}
    if not args.no_settle:
        t_begin = time.perf_counter()
        wins = []
        k = 0
        # a window holds the same mix of views every time: one full cycle of the views (weak), a few whole steps (strong)
        wlen = max(2, 10 // views_per_rank) if strong else (len(dcams) if len(dcams) >= 4 else 10)
        while True:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(wlen):
                step(k)
                k += 1
            torch.cuda.synchronize()
            wins.append(agree_max((time.perf_counter() - t0) / wlen * 1e3))
            last = wins[-3:]
            if len(last) == 3 and (max(last) - min(last)) <= 0.02 * min(last):
                settle["settled"] = True
                break
            if agree_max(time.perf_counter() - t_begin) > 1.0 or len(wins) >= 60:
                settle["settled"] = False
                break
        settle["extra_steps"] = k
        settle["windows_ms_per_step"] = [round(w, 4) for w in wins]

    timing = not args.no_kernel_timing
    lib.g4s_profile_reset()
    mem0 = torch.cuda.memory_stats(device)
    exchange_allocs0 = getattr(reducer, "allocations", None)  # (re)allocations of the exchange's buffers up to here

    def timed_steps(with_events):
        """EXACTLY args.steps steps between barrier + synchronize on both sides; one HIP event on the launch stream
        behind every step (one packet per step: ~2 us of a ~2 ms step) gives the per-step GPU durations.
        with_events: the library ALSO brackets every kernel group with a pair of HIP events (per-kernel durations for
        the roofline); those ~20 extra packets per step cost GPU time themselves, so the headline pass runs without
        them and the instrumented pass repeats the same steps afterwards.
        -> (host seconds over the region, MAX over the ranks; per-step ms, element-wise MAX over the ranks)."""
        lib.g4s_profile_enable(1 if with_events else 0)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(args.steps):
            step(i)
            evs[i + 1].record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lib.g4s_profile_enable(0)
        per = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
        if dist is not None:
            tt = torch.tensor([dt] + per, device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt, per = float(tt[0].item()), [float(x) for x in tt[1:].tolist()]
        return dt, per

    # ---- the headline: the contract's K steps, as several passes -------------------------------------------------------
    # Every pass is EXACTLY K steps between barrier + synchronize.  Passes are repeated until three of them are
    # undisturbed (at most --attempts); the headline is the MEDIAN pass of the undisturbed ones (by its mean step time =
    # host clock over the region / K, the contract's quotient) -- or, if no pass was clean, the median pass of all of
    # them, and the line says `disturbed`.  Every pass is listed.
    attempts = []
    for _a in range(max(1, args.attempts)):
        elapsed, per_step = timed_steps(False)
        summ = summarize_steps(per_step, elapsed / args.steps * 1e3)
        summ["per_step_ms"] = [round(t, 4) for t in per_step]
        attempts.append(summ)
        if sum(1 for a in attempts if not a["disturbed"]) >= 3:
            break
    head = pick_headline(attempts)
    mem1 = torch.cuda.memory_stats(device)

    # Sustained throughput: the same step, issued back to back right behind the headline pass (no pause anywhere), for
    # >= --sustained-seconds and >= 500 steps; the engine clock is sampled from sysfs on the way.
    sustained = None
    if args.sustained_seconds > 0:
        sclk = []
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        gc_ms[0], gc_ms[1] = 0.0, 0
        t0 = time.perf_counter()
        n_s, units_s = 0, 0
        # every rank must run the same number of steps (each one holds collectives): the count comes from the slowest
        # rank's headline time; the pass repeats the headline's K steps (same views) over and over
        n_target = max(500 // views_per_rank, int(math.ceil(args.sustained_seconds / max(head["median_ms"] * 1e-3, 1e-6))))
        reader = None
        marks = []  # host time every 50 steps: the reference-shaped forward waits for its read-back, so the host runs at
        every = max(1, 50 // views_per_rank)
        for i in range(n_target):   # most one step ahead of the GPU and these times follow the GPU's progress
            if i % every == 0:
                marks.append(time.perf_counter())
            if i == (3 * n_target) // 4 and rank == 0:
                # one sysfs read of the engine clock, on a side thread: the read blocks for tens of ms inside the driver
                # (the GIL is released meanwhile), and it must happen while the GPU is under load
                import threading

                def _read():
                    mhz = read_sclk_mhz(local_rank)
                    if mhz is not None:
                        sclk.append(mhz)
                reader = threading.Thread(target=_read)
                reader.start()
            step(i % args.steps)
            units_s += sum(Vs[c] for c in views_of_step(i % args.steps))
            n_s += 1
        if reader is not None:
            reader.join()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt_s = agree_max(time.perf_counter() - t0)
        if dist is not None:
            uu = torch.tensor([units_s], device=device, dtype=torch.float64)
            dist.all_reduce(uu)
            units_s = int(uu.item())
        sustained = {"steps": n_s, "seconds": round(dt_s, 3), "ms_per_step": round(dt_s / n_s * 1e3, 4),
                     "value": units_s / dt_s, "unit": "Gaussians/s",
                     "effective_clock_GHz": (round(sum(sclk) / len(sclk) / 1e3, 3) if sclk else None),
                     "clock_source": ("sysfs pp_dpm_sclk (the DPM level in force, not a cycle count) read once on a side "
                                      "thread three quarters into the pass" if sclk else
                                      "sysfs pp_dpm_sclk not readable on this box"),
                     f"ms_per_step_by_{every}_steps": [round((b - a) / every * 1e3, 4) for a, b in zip(marks, marks[1:])],
                     "host_gc": {"collections": gc_ms[1], "ms_total": round(gc_ms[0], 2),
                                 "startup_heap": "as is" if args.no_gc_freeze else "frozen before the warm-up (gc.freeze)"},
                     "note": "same step as the headline, issued back to back right behind it (no pause)"}

    exchange_ms = None
    exchange_pieces = None
    if timing:                              # same steps again, instrumented: kernels_ms / roofline (+ the exchange at N > 1)
        exchange_events = [] if dist is not None else None
        if dist is not None and exchange == "owner":
            reducer.timing = True
        # (runs right behind the sustained pass: the per-kernel durations are those of the settled clock)
        elapsed_events, _per = timed_steps(True)
        if exchange_events:
            exchange_ms = sum(a.elapsed_time(b) for a, b in exchange_events) / len(exchange_events)
        exchange_events = None
        if dist is not None and exchange == "owner":
            exchange_pieces = {k: round(v, 4) for k, v in reducer.read_timers().items()}
            reducer.timing = False
    else:
        elapsed_events = None

    units = sum(Vs[c] for i in range(args.steps) for c in views_of_step(i))
    inst = sum(Rs[c] for i in range(args.steps) for c in views_of_step(i))
    if dist is not None:
        uu = torch.tensor([units, inst], device=device, dtype=torch.float64)
        dist.all_reduce(uu)
        units, inst = int(uu[0].item()), int(uu[1].item())

    exchange_info = None
    replicas_identical = None
    if dist is not None:
        # After the last exchange every rank must hold the SAME BITS in its gradient bucket (the owners' sums gathered
        # everywhere): two checksums of the bit patterns, compared across the ranks.
        bits = bucket.view(torch.int32).to(torch.int64)
        weights = torch.arange(bits.numel(), device=device, dtype=torch.int64) % 1021 + 1
        mine = torch.stack([bits.sum(), (bits * weights).sum()])
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        replicas_identical = all(bool(torch.equal(e, everyone[0])) for e in everyone)
        try:
            rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            rccl_version = None
        exchange_info = exchange_report(backend, rccl_version, dist.get_world_size(), exchange, exchange_why, reducer,
                                        exchange_ms, exchange_pieces, exchanged_rows, args.steps, exchange_allocs0,
                                        replicas_identical)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    import ctypes
    kernels_ms = {}
    if timing:
        for k in range(lib.g4s_profile_kernels()):
            ms, cnt = ctypes.c_double(0), ctypes.c_int(0)
            lib.g4s_profile_read(k, ctypes.byref(ms), ctypes.byref(cnt))
            if cnt.value:
                kernels_ms[lib.g4s_profile_name(k).decode()] = ms.value / cnt.value
        lib.g4s_profile_reset()

    step_ms = head["mean_ms"]
    # roofline of the dominant kernel (rank 0's view mix): SURVEY.md 8(d) bytes of one launch / its HIP-event duration,
    # measured in this run, against the HBM peak; the blend kernels are NOT bound by HBM but by VALU issue, and `valu`
    # says so with numbers -- wave-level VALU instructions per launch come from the committed rocprofv3 PMC passes (they
    # cannot be read in-process; the file carries the build id of the library it was collected on and
    # `traffic_matches_build` says whether that is the library timed here), the duration is this run's
    roofline = None
    if kernels_ms:
        vlist = [c for i in range(args.steps) for c in views_of_step(i)]
        Vm = sum(Vs[c] for c in vlist) / len(vlist)
        Rm = sum(Rs[c] for c in vlist) / len(vlist)
        roofline = roofline_report(kernels_ms, pmc_profile(args.workload), build_id, P, Vm, Rm, W, H, D, step_ms)

    vif = None
    if world == 1 and not strong and args.views_in_flight > 1 and len(dcams) > 1:
        per_k, ref_digest = {}, None
        for nv in range(1, args.views_in_flight + 1):
            ms, gps, digest = run_views_in_flight(lib, _C, device, dev, dcams, P, W, H, D, Vs, Rs, nv,
                                                  max(100, 5 * args.steps), dL_dcolor, dL_dothers)
            ref_digest = ref_digest or digest
            assert digest == ref_digest, "a view's results changed with its neighbours"
            per_k[str(nv)] = {"ms_per_view": round(ms, 4), "value": gps}
        vif = {"unit": "Gaussians/s", "by_views_in_flight": per_k, "forward": "presized (no host read-back)",
               "note": "NOT the headline: K independent views of one multi-view batch on K HIP streams (each with its own "
                       "state, workspace and outputs; one host thread); every view's outputs and gradients are "
                       "bit-identical whatever runs beside it.  The reference's loop is one view per optimiser step."}

    cpu_baseline = None
    if not args.no_cpu_baseline and world == 1:
        cpu_baseline = run_cpu_baseline(scene, cams[0], P, W, H, D)
        ref_maps = cpu_baseline.pop("_maps", None)
        if ref_maps is not None:
            # The oracle has just rendered the WHOLE scene from view 0 (its timed sample): the checker's use of it.  The
            # contract read literally -- north_star: "RGB/depth/normal to <= 1e-4 abs" -- on that frame: how many pixels
            # of the HIP forward lie beyond 1e-4 in any of the ten maps, and the worst of them (they are threshold flips:
            # tests/common.py proves each one equal to the oracle's own pixel with that decision taken the other way).
            c0 = dcams[0]
            fw = _C.rasterize_gaussians(torch.zeros(3, device=device), dev["means3D"], empty, dev["opacity"], dev["scales"],
                                        dev["rotations"], 1.0, empty, c0["view"], c0["proj"], c0["tanfovx"], c0["tanfovy"], H, W,
                                        dev["sh"], D, c0["campos"], False, False)
            import numpy as np
            dmax = np.maximum(np.abs(fw[1].cpu().numpy() - ref_maps[0]).max(axis=0), np.abs(fw[2].cpu().numpy() - ref_maps[1]).max(axis=0))
            cpu_baseline["checker"] = {"frame": f"view 0, {W}x{H}, all {P} surfels", "pixels": int(dmax.size),
                                       "pixels_beyond_1e-4_abs": int((dmax > 1e-4).sum()),
                                       "frac_beyond_1e-4_abs": float((dmax > 1e-4).mean()), "worst_abs": float(dmax.max()),
                                       "median_abs": float(np.median(dmax)), "radii_equal": bool(np.array_equal(fw[3].cpu().numpy(), ref_maps[2]))}
            del ref_maps
        if args.workload == "s3":
            # SURVEY.md 8(d) asks for the CPU restatement on S1 and S2: both whole, in the same run, next to the sample
            # of the headline's own workload
            others = {}
            for wl in ("s1", "s2"):
                sc2, cams2, _d, _dc, (P2, W2, H2, D2) = build_scene(wl, torch.device("cpu"))
                b = _cpu_baseline_once(sc2, cams2[0], P2, W2, H2, D2, P2)
                for k_ in ("seconds", "_maps", "_n"):
                    b.pop(k_)
                others[wl] = b
            cpu_baseline["other_workloads"] = others

    out = assemble_line(dict(
        workload=args.workload, P=P, W=W, H=H, D=D, n_views=len(dcams), world=world, views_per_rank=views_per_rank,
        strong=strong, steps=args.steps, warmup=args.warmup, scaling=args.scaling, backend=backend, exchange=exchange,
        presized=args.presized, units=units, inst=inst, Vs=Vs, attempts=attempts, head=head, settle=settle,
        device_allocations=int(mem1.get("num_device_alloc", 0) - mem0.get("num_device_alloc", 0)), kernels_ms=kernels_ms,
        elapsed_events_s=elapsed_events, sustained=sustained, exchange_info=exchange_info, exchanged_rows=exchanged_rows,
        exchange_ms=exchange_ms, views_in_flight=vif, roofline=roofline, cpu_baseline=cpu_baseline, build_id=build_id,
        trained_scene=TRAINED_INFO.get(args.workload)))
    if lib_options:
        out["library_options"] = lib_options  # (an A/B run: not the library's defaults)
    print(json.dumps(out))
    sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


def run_views_in_flight(lib, _C, device, dev, dcams, P, W, H, D, Vs, Rs, K, steps, dL_dcolor, dL_dothers):
    """Throughput with K independent views in flight (tools/views_in_flight.py has the stand-alone form): K HIP streams,
    each with its own PresizedState (no host read-back), backward workspace and outputs; step i goes to stream i % K.
    Views of one multi-view batch (gradient accumulation; SURVEY.md 8(e)'s 8 views over fewer than 8 GPUs) are
    independent, so one view's launch-bound binning and HBM-bound per-Gaussian kernels run beside another view's
    VALU-bound blend kernels.  Returns (ms per view, Gaussians/s, digest of view 1's results)."""
    import torch
    from g4splat_amd.pipeline import ViewPipeline
    bg = torch.zeros(3, device=device)
    empty = torch.empty(0, device=device)
    cap = int(max(Rs.values()) * 1.25) + 4096
    pipe = ViewPipeline(P, W, H, cap, device, k=K)

    def one(i, keep=False):
        c = dcams[i % len(dcams)]
        with pipe.slot(i) as (state, work):
            fw = _C.rasterize_gaussians_presized(state, bg, dev["means3D"], empty, dev["opacity"], dev["scales"],
                                                 dev["rotations"], 1.0, empty, c["view"], c["proj"], c["tanfovx"],
                                                 c["tanfovy"], H, W, dev["sh"], D, c["campos"], False, False)
            gr = _C.rasterize_gaussians_backward(bg, dev["means3D"], fw[3], empty, dev["scales"], dev["rotations"], 1.0,
                                                 empty, c["view"], c["proj"], c["tanfovx"], c["tanfovy"], dL_dcolor,
                                                 dL_dothers, dev["sh"], D, c["campos"], fw[4], fw[0], fw[5], fw[6], False,
                                                 out={"workspace": work})
        return (fw, gr) if keep else None

    for i in range(2 * K + 2):
        one(i)
    torch.cuda.synchronize()
    fw, gr = one(1 % len(dcams), keep=True)  # the same view whatever K is: its results must not depend on its neighbours
    one(2)
    torch.cuda.synchronize()
    digest = [fw[1].double().sum().item(), fw[2].double().sum().item()] + [x.double().sum().item() for x in gr if torch.is_tensor(x)]
    assert not pipe.overflowed(), "presized capacity overflow"
    del fw, gr
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    units = sum(Vs[i % len(dcams)] for i in range(steps))
    return dt / steps * 1e3, units / dt, digest


def read_sclk_mhz(device_index=0):
    """Current engine clock of torch device `device_index` from sysfs (the line of pp_dpm_sclk marked '*'), or None.
    The box exposes the whole node's cards in sysfs: the one that belongs to the device is found by its PCI address."""
    import glob
    import re
    import torch
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x." % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None
    for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        if bdf not in os.path.realpath(os.path.dirname(path)):
            continue
        try:
            m = re.search(r"(\d+)\s*[Mm][Hh]z\s*\*", open(path).read())
        except OSError:
            return None
        return float(m.group(1)) if m else None
    return None


def pmc_profile(workload):
    """The committed rocprofv3 PMC summary of this workload (profiles/r<NN>_traffic_<workload>.json, newest round
    first; written by tools/profile_gpu.sh + tools/summarize_prof.py on the GPU box): FETCH_SIZE / WRITE_SIZE in KiB
    (separate passes; HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction of MI355X_MICROARCH.md),
    SQ_INSTS_VALU and GRBM_GUI_ACTIVE per launch, and `build_id` = g4s_version()'s id of the library the counters were
    collected on.  These counters cannot be collected inside this process, so the bench line labels them."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_traffic_{workload}.json")), reverse=True)
    cands.append(os.path.join(ROOT, "profiles", f"traffic_{workload}.json"))
    for path in cands:
        try:
            d = json.load(open(path))
            d.setdefault("provenance", f"{os.path.relpath(path, ROOT)} (static: rocprofv3 --pmc passes, not measured in this run)")
            return d
        except Exception:
            continue
    return None


def run_cpu_baseline(scene, cam, P, W, H, D, target_gaussians=150_000, target_seconds=15.0):
    first = _cpu_baseline_once(scene, cam, P, W, H, D, target_gaussians)
    if first["seconds"] < 0.5 * target_seconds and target_gaussians < P:
        n2 = min(P, int(target_gaussians * target_seconds / max(first["seconds"], 1e-3)))
        if n2 > 0.8 * P:
            n2 = P  # (nearly everything fits the time budget: take the whole scene, whose render is also the checker's frame)
        if n2 > 1.5 * target_gaussians:
            first = _cpu_baseline_once(scene, cam, P, W, H, D, n2)
    first.pop("seconds")
    if first.get("_n") != P:
        first.pop("_maps", None)   # (a subset was timed: its render is not the scene's)
    first.pop("_n", None)
    return first


def _cpu_baseline_once(scene, cam, P, W, H, D, target_gaussians):
    """CPU restatement of the reference algorithm (oracle/, 'port') on the host cores, bounded sample:
    a seeded subset of the same scene, same camera and resolution."""
    import numpy as np
    from oracle import oracle as om
    om.build()
    n = min(P, target_gaussians)
    idx = np.random.default_rng(0).choice(P, n, replace=False) if n < P else np.arange(P)
    e = np.zeros((0,), np.float32)
    rng = np.random.default_rng(1)
    gc = rng.normal(size=(3, H, W)).astype(np.float32)
    go = rng.normal(size=(7, H, W)).astype(np.float32)
    o = om.Oracle()
    t0 = time.perf_counter()
    R, color, others, radii = o.rasterize_gaussians(np.zeros(3, np.float32), scene.means3D[idx], e, scene.opacities[idx],
                                           scene.scales[idx], scene.rotations[idx], 1.0, e, cam.world_view_transform,
                                           cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, scene.shs[idx], D,
                                           cam.camera_center)
    o.rasterize_gaussians_backward(gc, go)
    dt = time.perf_counter() - t0
    V = int((radii > 0).sum())
    return {"seconds": dt, "_n": n, "_maps": (color, others, radii), "value": V / dt, "unit": "Gaussians/s", "cores": os.cpu_count(), "kind": "port",
            "sample": (f"{n} of {P} surfels" + (" (seeded subset)" if n < P else " (the whole scene)") +
                       f", view 0 at {W}x{H}, 1 fwd+bwd, {V} visible, {R} instances, {dt:.2f} s, OpenMP over all host "
                       f"cores (oracle/surfel_oracle.c)")}


if __name__ == "__main__":
    main()
