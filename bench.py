#!/usr/bin/env python
"""bench.py -- rasterized Gaussians/s, forward+backward, on MI355X.

A "step" is one pass of the hot path over one training view: one
`_C.rasterize_gaussians` + one `_C.rasterize_gaussians_backward` on
synthetic data already resident in HBM; with N > 1 ranks every rank renders its own view
of the replicated scene and the step also all-reduces the parameter gradients over RCCL
(one view per GPU, weak scaling).

Workload at N=1 (BASELINE.json metric "rasterized Gaussians/s fwd+bwd @1600x1200"):
S3 = configs[2] stand-in: 1.5 M surfels on the faces of a 6x4x3 m room, 1600x1200, SH degree 3.

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
`roofline` for the dominant kernel and `cpu_baseline` (the CPU oracle timed on the host
cores on a bounded sample; reported baseline, not a target).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

WORKLOADS = {
    # name: (P, W, H, SH degree, #views)
    "s3": (1_500_000, 1600, 1200, 3, 8),   # ScanNet++-like, BASELINE configs[2] (metric resolution)
    "s2": (300_000, 1200, 680, 3, 8),      # Replica-room0-like, configs[1]/[3]
    "s1": (10_000, 256, 256, 3, 1),        # configs[0]
    "s5": (3_000_000, 1200, 680, 3, 8),    # DeepBlending-like, configs[4]
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS = 1024           # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9       # engine clock


def build_scene(name, device):
    from g4splat_amd import synthetic
    P, W, H, D, nviews = WORKLOADS[name]
    if name == "s1":
        scene, cam = synthetic.scene_random(P, seed=0, width=W, height=H)
        cams = [cam]
    else:
        scene = synthetic.scene_room(P, seed=0)
        cams = synthetic.room_cameras(nviews, W, H, fovx_deg=90.0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=device)
    dev = dict(means3D=t(scene.means3D), scales=t(scene.scales), rotations=t(scene.rotations),
               opacity=t(scene.opacities), sh=t(scene.shs))
    dcams = [dict(view=t(c.world_view_transform), proj=t(c.full_proj_transform), campos=t(c.camera_center),
                  tanfovx=c.tanfovx, tanfovy=c.tanfovy) for c in cams]
    return scene, cams, dev, dcams, (P, W, H, D)


def algorithmic_bytes(kernel, P, V, R, N, K, M, tiles, tile_bits):
    """SURVEY.md 8(d) per-kernel algorithmic bytes of one forward+backward."""
    p_s = (32 + tile_bits + 7) // 8  # the reference sorts 64-bit (tile | depth) keys over 32 + bit bits
    return {
        "preprocess_fwd": P * (44 + 4 + 4 + 8) + V * (12 * K + 76),
        "blend_fwd": R * 76 + N * 60,
        "blend_bwd": R * 76 + N * 60 + V * 72,
        "preprocess_bwd": V * (44 + 12 * K + 72 + 36 + 3) + P * (12 + 12 + 8 + 16 + 4) + P * 12 * M + V * 12 * K,
        "tile_sort": R * 24 * p_s,
        "emit": R * 12,
        "tile_ranges": R * 8 + tiles * 8,
    }.get(kernel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="s3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP-event timing")
    ap.add_argument("--sustained-seconds", type=float, default=1.0,
                    help="after the headline pass, keep stepping for at least this long (and >= 500 steps) and report "
                         "the settled throughput as `sustained` (0 = skip)")
    ap.add_argument("--no-gc-freeze", action="store_true",
                    help="leave Python's cyclic garbage collector as it is (default: gc.collect() + gc.freeze() after the "
                         "warm-up, so that a full collection over the interpreter's ~10^6 long-lived objects cannot land "
                         "inside a timed pass; host-side hygiene, the GPU work is unchanged)")
    ap.add_argument("--views-in-flight", type=int, default=3,
                    help="N = 1 only, reported next to the headline (never as `value`): throughput with up to this many "
                         "INDEPENDENT views in flight on separate HIP streams (multi-view batches; 0 / 1 = skip)")
    ap.add_argument("--presized", action="store_true",
                    help="use g4s_rasterizer_forward_presized (no host read-back; extension) instead of the "
                         "reference-shaped forward -- for the step-time comparison in DESIGN.md, not the default")
    args = ap.parse_args()

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
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
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
    build.build()
    lib = _lib.load()
    from g4splat_amd.diff_surfel_rasterization import _C

    scene, cams, dev, dcams, (P, W, H, D) = build_scene(args.workload, device)
    N = W * H
    bg = torch.zeros(3, device=device)
    empty = torch.empty(0, device=device)
    g = torch.Generator(device=device).manual_seed(1)
    dL_dcolor = torch.randn((3, H, W), device=device, generator=g)
    dL_dothers = torch.randn((7, H, W), device=device, generator=g)

    # one persistent flat fp32 bucket holds the parameter gradients of this rank's view (xyz, SH,
    # opacity, scale, rotation = 58 floats per Gaussian at SH degree 3) plus the densification side channel (per-view
    # ||grad_means2D||, visibility); the backward writes straight into views of it (`out=`).  The exchange is
    # g4splat_amd.parallel.OwnerReduce (three collectives per step on persistent buffers: radii MAX + sizes, all_to_all
    # of visible rows to their owners, grouped in-place all_gather of the reduced shards) or, as the fallback,
    # RowSparseAllReduce (MAX all-reduce of the radii, then one SUM all-reduce of the rows visible on some rank).
    grad_out = side = rmax = reducer = None
    # G4S_BENCH_EXCHANGE=owner (default): OwnerReduce -- all_to_all of this rank's visible rows to index-shard owners +
    # all_gather of the reduced shards (g4splat_amd/parallel.py); =allreduce: the visible-rows /
    # dense SUM all-reduce of round 1.  If the owner exchange fails on this machine's RCCL during warm-up the bench
    # falls back to the all-reduce and says so in config.parallelism.
    exchange = os.environ.get("G4S_BENCH_EXCHANGE", "owner")
    if dist is not None:
        from g4splat_amd.parallel import OwnerReduce, RowSparseAllReduce
        M = int(dev["sh"].shape[1])
        shapes = [("dL_dmeans3D", (P, 3)), ("dL_dsh", (P, M, 3)), ("dL_dopacity", (P, 1)), ("dL_dscales", (P, 2)),
                  ("dL_drotations", (P, 4)), ("side", (P, 2))]
        offs, o = [], 0
        for _n, shp in shapes:
            offs.append(o)
            o += (int(np.prod(shp)) + 63) // 64 * 64  # 256-B aligned views
        bucket = torch.zeros(o, device=device)
        views = {n: bucket[b:b + int(np.prod(shp))].view(shp) for (n, shp), b in zip(shapes, offs)}
        side = views.pop("side")
        grad_out = views
        rmax = torch.zeros((P,), dtype=torch.int32, device=device)
        row_views = [v.view(P, -1) for v in grad_out.values()] + [side]
        reducer = OwnerReduce(row_views) if exchange == "owner" else RowSparseAllReduce(bucket, row_views)
    exchanged_rows = []
    exchange_events = None  # list of (start, stop) events while the instrumented pass runs

    pstate = None

    def step(i):
        nonlocal pstate
        cam = dcams[(rank + i * world) % len(dcams)]
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
        grads = _C.rasterize_gaussians_backward(bg, dev["means3D"], radii, empty, dev["scales"], dev["rotations"],
                                                1.0, empty, cam["view"], cam["proj"], cam["tanfovx"], cam["tanfovy"],
                                                dL_dcolor, dL_dothers, dev["sh"], D, cam["campos"], geom, R, binning,
                                                img, False, out=grad_out)
        if dist is not None:
            ev = None
            if exchange_events is not None:  # instrumented pass only: GPU time of the exchange, this rank
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            gm2 = grads[0]
            torch.linalg.vector_norm(gm2[:, :2], dim=1, out=side[:, 0])
            side[:, 1] = radii > 0
            if exchange == "owner":
                reducer.finish()  # (reducer.max_radii = MAX over the ranks of the radii, from begin()'s collective)
                exchanged_rows.append(reducer.last_rows_sent)
            else:
                rmax.copy_(radii)
                dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
                reducer.reduce(rmax > 0)
                exchanged_rows.append(reducer.last_rows)
            if ev is not None:
                ev[1].record()
                exchange_events.append(ev)
        return R, radii

    # Host hygiene: a generation-2 collection of CPython's cyclic GC walks every long-lived object of the process (torch,
    # numpy, the scene: ~50 ms here) and lands wherever the allocation counters trip -- measured as ONE 53-ms stall in a
    # 500-step pass, i.e. the whole difference between the 20-step headline and the "sustained" rate of round 2.  Long-
    # running loops freeze the startup heap; so does this one -- BEFORE the warm-up steps, so that no host pause sits between
    # the warm-up and the timed pass (a GPU left idle for the 0.1 s a collection takes starts the timed steps from a lower
    # power state).  The collections that still happen are timed and reported.
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

    # warm-up (also measures V and R per view outside the timed region)
    Vs, Rs = {}, {}
    if dist is not None and exchange == "owner":
        try:
            step(0)
            torch.cuda.synchronize()
            ok = torch.ones(1, device=device)
        except Exception as ex:  # e.g. an RCCL build without uneven all_to_all
            print(f"[bench] owner exchange failed on rank {rank}: {ex}; falling back to all-reduce", file=sys.stderr)
            ok = torch.zeros(1, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0:
            exchange = "allreduce(fallback)"
            reducer = RowSparseAllReduce(bucket, row_views)
    for i in range(max(args.warmup, 1)):
        R, radii = step(i)
        c = (rank + i * world) % len(dcams)
        Vs[c] = int((radii > 0).sum().item())
        Rs[c] = int(R)
    for i in range(len(dcams)):  # make sure every view that the timed steps use has its V known
        c = (rank + i * world) % len(dcams)
        if c not in Vs and i < args.steps:
            R, radii = step(i)
            Vs[c] = int((radii > 0).sum().item())
            Rs[c] = int(R)

    if args.presized:  # capacity from the warm-up's instance counts, with head-room
        pstate = _C.PresizedState(P, W, H, int(max(Rs.values()) * 1.25) + 4096, device)
        step(0)
        torch.cuda.synchronize()
        assert pstate.status.tolist()[3] == 0
    timing = not args.no_kernel_timing
    lib.g4s_profile_reset()

    def timed_steps(with_events):
        """EXACTLY args.steps steps between barrier + synchronize on both sides.  with_events: the library brackets
        every kernel group with a pair of HIP events on the launch stream (per-kernel durations for the roofline);
        those ~20 extra packets per step cost GPU time themselves, so the headline pass runs without them and the
        instrumented pass repeats the same steps afterwards."""
        lib.g4s_profile_enable(1 if with_events else 0)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        per_step = []
        for i in range(args.steps):
            ts = time.perf_counter()
            step(i)
            if os.environ.get("G4S_BENCH_PER_STEP"):
                torch.cuda.synchronize()
                per_step.append(round((time.perf_counter() - ts) * 1e3, 2))
        torch.cuda.synchronize()
        if per_step and rank == 0:
            print("per-step ms:", per_step, file=sys.stderr)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        lib.g4s_profile_enable(0)
        return dt

    elapsed = timed_steps(False)            # the headline: value / ms_per_step (the contract's K steps)

    # Sustained throughput.  The headline's K steps last ~40 ms; the blend kernels hold the VALU at its issue limit and
    # the part lowers its clock after ~0.2 s of that.  The reference's loop is thousands of iterations, so the settled
    # rate is reported next to the headline: the same step, issued back to back right behind the headline pass (no
    # pause anywhere), for >= --sustained-seconds and >= 500 steps; the engine clock is sampled from sysfs on the way.
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
        el = elapsed
        if dist is not None:
            tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        n_target = max(500, int(math.ceil(args.sustained_seconds / max(el / args.steps, 1e-6))))
        first = []
        reader = None
        marks = []  # host time every 50 steps: the reference-shaped forward waits for its read-back, so the host runs at
        for i in range(n_target):   # most one step ahead of the GPU and these times follow the GPU's progress
            if i % 50 == 0:
                marks.append(time.perf_counter())
            if i < 10:
                first.append(time.perf_counter())
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
            units_s += Vs[(rank + (i % args.steps) * world) % len(dcams)]
            n_s += 1
        if reader is not None:
            reader.join()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt_s], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_s = float(tt.item())
            uu = torch.tensor([units_s], device=device, dtype=torch.float64)
            dist.all_reduce(uu)
            units_s = int(uu.item())
        sustained = {"steps": n_s, "seconds": round(dt_s, 3), "ms_per_step": round(dt_s / n_s * 1e3, 4),
                     "value": units_s / dt_s, "unit": "Gaussians/s",
                     "effective_clock_GHz": (round(sum(sclk) / len(sclk) / 1e3, 3) if sclk else None),
                     "clock_source": ("sysfs pp_dpm_sclk (the DPM level in force, not a cycle count) read once on a side "
                                      "thread three quarters into the pass" if sclk else
                                      "sysfs pp_dpm_sclk not readable on this box"),
                     "ms_first_steps": [round((b - a) * 1e3, 3) for a, b in zip(first, first[1:])],
                     "ms_per_step_by_50_steps": [round((b - a) / 50 * 1e3, 4) for a, b in zip(marks, marks[1:])],
                     "host_gc": {"collections": gc_ms[1], "ms_total": round(gc_ms[0], 2),
                                 "startup_heap": "as is" if args.no_gc_freeze else "frozen after the warm-up (gc.freeze)"},
                     "note": "same step as the headline, issued back to back right behind it (no pause)"}

    exchange_ms = None
    if timing:                              # same steps again, instrumented: kernels_ms / roofline (+ the exchange at N > 1)
        exchange_events = [] if dist is not None else None
        # (runs right behind the sustained pass: the per-kernel durations are those of the settled clock)
        elapsed_events = timed_steps(True)
        if exchange_events:
            exchange_ms = sum(a.elapsed_time(b) for a, b in exchange_events) / len(exchange_events)
        exchange_events = None
    else:
        elapsed_events = None

    units = sum(Vs[(rank + i * world) % len(dcams)] for i in range(args.steps))
    inst = sum(Rs[(rank + i * world) % len(dcams)] for i in range(args.steps))
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        uu = torch.tensor([units, inst], device=device, dtype=torch.float64)
        dist.all_reduce(uu)
        units, inst = int(uu[0].item()), int(uu[1].item())

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

    # roofline of the dominant kernel (rank 0's view mix)
    roofline = None
    if kernels_ms:
        dom = max((k for k in kernels_ms if algorithmic_bytes(k, 1, 1, 1, 1, 1, 1, 1, 8) is not None),
                  key=lambda k: kernels_ms[k])
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        tile_bits = max(1, math.ceil(math.log2(tiles + 1)))
        views = [(0 + i * world) % len(dcams) for i in range(args.steps)]
        Vm = sum(Vs[c] for c in views) / len(views)
        Rm = sum(Rs[c] for c in views) / len(views)
        K = (D + 1) ** 2
        B = algorithmic_bytes(dom, P, Vm, Rm, N, K, 16, tiles, tile_bits)
        achieved = B / (kernels_ms[dom] * 1e-3) / 1e9
        # What binds the kernel.  `achieved` / `peak` / `frac` are the contract's HBM figures (SURVEY.md 8(d) bytes of
        # one launch / its HIP-event duration, measured in this run).  The blend kernels are NOT bound by HBM but by
        # VALU issue: `valu` says so with numbers -- wave-level VALU instructions per launch come from the committed
        # rocprofv3 PMC passes of this build (they cannot be read in-process; `source` names the file and the commit it
        # was taken at), the duration is this run's.
        pmc = pmc_profile(args.workload)
        pk = (pmc or {}).get("kernels", {}).get(dom)
        valu = None
        if pk and "SQ_INSTS_VALU" in pk:
            insts = float(pk["SQ_INSTS_VALU"])
            t = kernels_ms[dom] * 1e-3
            # Calibration (tools/micro/valu_rate.hip, profiles/r03_valu_rate.txt): a wave64 VALU instruction holds its
            # SIMD for 4 cycles (v_fma/mul/mov_f32; packed FP32 ~5, transcendentals ~6-9), an instruction issued under
            # an empty EXEC mask retires in ~1.  `cycles_per_instruction_profiled` is the launch's own quotient --
            # SIMD cycles (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) / counted instructions -- so nothing is assumed
            # about the clock; at or below ~4.2 the SIMDs do nothing but issue VALU instructions.
            cpi = (pk["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS / insts) if pk.get("GRBM_GUI_ACTIVE") else None
            valu = {"wave_instructions_per_launch": int(insts),
                    "cycles_per_instruction_profiled": round(cpi, 3) if cpi else None,
                    "full_rate_cycles_per_instruction": 4.0,
                    # share of the SIMDs' issue slots the launch used, clock-free: 1.0 when cpi <= 4
                    "simd_issue_utilisation": round(min(1.0, 4.0 / cpi), 4) if cpi else None,
                    "ns_per_instruction_this_run": round(t * SIMDS / insts * 1e9, 4),
                    "microbenchmark_ns_per_v_fma_f32": [1.73, 1.97],
                    "calibration": "tools/micro/valu_rate.hip on MI355X (profiles/r03_valu_rate.txt): back-to-back "
                                   "v_fma_f32 1.73-1.97 ns per wave instruction and SIMD (4 cycles at the clock the part "
                                   "sustains under that load), v_pk_fma_f32 2.45 ns"}
        if valu and valu["simd_issue_utilisation"] is not None:
            # the roof the kernel is closer to -- or neither: a small frame (S1: one wave per SIMD) runs at a third of
            # the issue rate and 2 % of the HBM peak; that is latency, not a roofline
            bound = "valu" if valu["simd_issue_utilisation"] > achieved / HBM_PEAK_GBS else "hbm"
            if max(valu["simd_issue_utilisation"], achieved / HBM_PEAK_GBS) < 0.5:
                bound = "latency"
        else:
            bound = "unknown"  # no counters for this workload / kernel: the HBM fraction below is all this run can say
        traffic = int((2.0 * pk["FETCH_SIZE"] + pk["WRITE_SIZE"]) * 1024) if pk and "FETCH_SIZE" in pk and "WRITE_SIZE" in pk else None
        # whole step against the HBM roofline: every kernel's 8(d) bytes / the step time of this run
        B_step = (P * 60 + Vm * (12 * K + 76) + Rm * 12 + Rm * 24 * ((32 + tile_bits + 7) // 8) + Rm * 8 + tiles * 8
                  + 2 * (Rm * 76 + N * 60) + Vm * 72 + Vm * (44 + 12 * K + 72 + 36 + 3) + P * 52 + P * 12 * 16 + Vm * 12 * K)
        step_ms = elapsed / args.steps * 1e3
        roofline = {"kernel": dom, "bound": bound, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": (pmc or {}).get("provenance"),
                    "algorithmic_bytes_per_launch": int(B), "avg_launch_ms": round(kernels_ms[dom], 4),
                    "valu": valu,
                    "whole_step": {"algorithmic_bytes": int(B_step), "GBps": round(B_step / (step_ms * 1e-3) / 1e9, 1),
                                   "frac_of_hbm_peak": round(B_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "SURVEY.md 8(d) bytes of all kernels, priced as the table prices them -- the reference's "
                                           "64-bit (tile | depth) key sort, p_s = ceil((32 + bit) / 8) passes over all "
                                           "instances -- / this run's step time (this build moves fewer bytes: a 2-pass tile "
                                           "partition of depth-ordered instances)"}}

    vif = None
    if world == 1 and args.views_in_flight > 1 and len(dcams) > 1:
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

    ms_per_step = elapsed / args.steps * 1e3
    out = {
        "metric": "rasterized Gaussians/s fwd+bwd @1600x1200" if args.workload == "s3"
        else f"rasterized Gaussians/s fwd+bwd @{W}x{H}",
        "value": units / elapsed, "unit": "Gaussians/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {P} surfels (room box), {W}x{H}, SH degree {D}, "
                               f"{len(dcams)} views, 1 view/GPU/step", "P": P, "width": W, "height": H,
                   "sh_degree": D, "visible_per_view": round(units / args.steps / world),
                   "instances_per_view": round(inst / args.steps / world),
                   "forward": "presized (no host read-back)" if args.presized else "reference-shaped",
                   "parallelism": f"view-dp{world}" + ((("+rccl" if backend == "nccl" else "+" + backend) +
                                                                   ("-owner-reduce(all_to_all+all_gather)" if exchange == "owner"
                                                                    else "-visible-rows-" + exchange)) if world > 1 else ""),
                   "exchanged_rows_per_step": (round(sum(exchanged_rows[-args.steps:]) / args.steps) if exchanged_rows
                                               else None),
                   # GPU time of the gradient exchange per step on rank 0 (statistics + radii MAX + owner-reduce), from the
                   # instrumented pass; it is part of every timed step at N > 1
                   "exchange_ms_per_step": round(exchange_ms, 4) if exchange_ms is not None else None},
        "gaussians_total_per_s": P * args.steps * world / elapsed,
        "instances_per_s": inst / elapsed,
        "kernels_ms": {k: round(v, 4) for k, v in kernels_ms.items()},
        # per-kernel durations come from a second pass over the same K steps with a HIP-event pair around every kernel
        # group on the launch stream; the events cost GPU time themselves, so that pass is not the headline
        "kernel_timing": ({"pass": "same steps repeated with HIP events around each kernel group",
                           "ms_per_step_with_events": round(elapsed_events / args.steps * 1e3, 4),
                           "clock_state": ("settled (behind the sustained pass)" if sustained else "as found")}
                          if elapsed_events is not None else None),
        "sustained": sustained,
        "views_in_flight": vif,
        "roofline": roofline, "cpu_baseline": cpu_baseline,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def run_views_in_flight(lib, _C, device, dev, dcams, P, W, H, D, Vs, Rs, K, steps, dL_dcolor, dL_dothers):
    """Throughput with K independent views in flight (tools/views_in_flight.py has the stand-alone form): K HIP streams,
    each with its own PresizedState (no host read-back), backward workspace and outputs; step i goes to stream i % K.
    Views of one multi-view batch (gradient accumulation; SURVEY.md 8(e)'s 8 views over fewer than 8 GPUs) are
    independent, so one view's launch-bound binning and HBM-bound per-Gaussian kernels run beside another view's
    VALU-bound blend kernels.  Returns (ms per view, Gaussians/s, digest of view 1's results)."""
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
    SQ_INSTS_VALU and GRBM_GUI_ACTIVE per launch.  These counters cannot be collected inside this process, so the
    bench line labels them with `provenance` (file + the commit the profiled library was built from)."""
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
        if n2 > 1.5 * target_gaussians:
            first = _cpu_baseline_once(scene, cam, P, W, H, D, n2)
    first.pop("seconds")
    return first


def _cpu_baseline_once(scene, cam, P, W, H, D, target_gaussians):
    """CPU restatement of the reference algorithm (oracle/, 'port') on the host cores, bounded sample:
    a seeded subset of the same scene, same camera and resolution."""
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
    R, _, _, radii = o.rasterize_gaussians(np.zeros(3, np.float32), scene.means3D[idx], e, scene.opacities[idx],
                                           scene.scales[idx], scene.rotations[idx], 1.0, e, cam.world_view_transform,
                                           cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, scene.shs[idx], D,
                                           cam.camera_center)
    o.rasterize_gaussians_backward(gc, go)
    dt = time.perf_counter() - t0
    V = int((radii > 0).sum())
    return {"seconds": dt, "value": V / dt, "unit": "Gaussians/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"workload of this line, bounded: {n} of {P} surfels (seeded subset), same view 0 at {W}x{H}, 1 fwd+bwd, "
                      f"{V} visible, {R} instances, {dt:.2f} s, OpenMP over all host cores (oracle/surfel_oracle.c; "
                      f"SURVEY.md 8(d) asks for S1/S2: run --workload s1 / s2 for those)"}


if __name__ == "__main__":
    main()
