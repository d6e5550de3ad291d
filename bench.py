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
    p_s = (tile_bits + 7) // 8
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
    # g4splat_amd.parallel.RowSparseAllReduce: MAX all-reduce of the radii (-> max_radii2D and the union
    # visibility), then ONE RCCL SUM all-reduce over xGMI of the rows visible on some rank (the whole bucket when
    # that is most of them).  Persistent buffers keep torch's caching allocator out of the timed region.
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
            reducer.begin(radii > 0)  # sizes travel to the host while the backward runs
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
            rmax.copy_(radii)
            dist.all_reduce(rmax, op=dist.ReduceOp.MAX)
            if exchange == "owner":
                reducer.finish()
                exchanged_rows.append(reducer.last_rows_sent)
            else:
                reducer.reduce(rmax > 0)
                exchanged_rows.append(reducer.last_rows)
            if ev is not None:
                ev[1].record()
                exchange_events.append(ev)
        return R, radii

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

    elapsed = timed_steps(False)            # the headline: value / ms_per_step
    exchange_ms = None
    if timing:                              # same steps again, instrumented: kernels_ms / roofline (+ the exchange at N > 1)
        exchange_events = [] if dist is not None else None
        time.sleep(0.5)  # the blend kernels hold the VALU at its limit: give the part's clock the same start as the first pass
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
            valu = {"wave_instructions_per_launch": int(insts),
                    # share of the SIMDs' issue cycles one launch uses: every wave64 VALU instruction holds its SIMD
                    # for 4 cycles; 1024 SIMDs at the 2.4 GHz engine clock (MI355X_MICROARCH.md)
                    "issue_frac_this_run": round(insts * 4.0 / (SIMDS * CLOCK_HZ * t), 4),
                    "assumes": f"{SIMDS} SIMDs x {CLOCK_HZ / 1e9:.1f} GHz, 4 cycles per wave64 VALU instruction",
                    # the same ratio with the cycle count of the profiled launch itself (no clock assumption)
                    "issue_frac_profiled": (round(insts * 4.0 / (pk["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS), 4)
                                            if pk.get("GRBM_GUI_ACTIVE") else None)}
        bound = "valu" if valu and valu["issue_frac_this_run"] > achieved / HBM_PEAK_GBS else "hbm"
        traffic = int((2.0 * pk["FETCH_SIZE"] + pk["WRITE_SIZE"]) * 1024) if pk and "FETCH_SIZE" in pk and "WRITE_SIZE" in pk else None
        # whole step against the HBM roofline: every kernel's 8(d) bytes / the step time of this run
        B_step = (P * 60 + Vm * (12 * K + 76) + Rm * 12 + Rm * 24 * ((tile_bits + 7) // 8) + Rm * 8 + tiles * 8
                  + 2 * (Rm * 76 + N * 60) + Vm * 72 + Vm * (44 + 12 * K + 72 + 36 + 3) + P * 52 + P * 12 * 16 + Vm * 12 * K)
        step_ms = elapsed / args.steps * 1e3
        roofline = {"kernel": dom, "bound": bound, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": (pmc or {}).get("provenance"),
                    "algorithmic_bytes_per_launch": int(B), "avg_launch_ms": round(kernels_ms[dom], 4),
                    "valu": valu,
                    "whole_step": {"algorithmic_bytes": int(B_step), "GBps": round(B_step / (step_ms * 1e-3) / 1e9, 1),
                                   "frac_of_hbm_peak": round(B_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "SURVEY.md 8(d) bytes of all kernels (reference algorithm: 64-bit-key sort over "
                                           "all instances) / this run's step time"}}

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
                           "ms_per_step_with_events": round(elapsed_events / args.steps * 1e3, 4)}
                          if elapsed_events is not None else None),
        "roofline": roofline, "cpu_baseline": cpu_baseline,
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


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
