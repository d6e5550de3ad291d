"""The host-side logic of bench.py, free of the GPU: how a timed pass is judged and which one becomes the headline, which
views a rank renders in a step, SURVEY.md 8(d)'s byte model, the roofline / VALU report, the exchange report and the
assembly of the ONE JSON line.  bench.py measures and hands the numbers over; everything that selects or derives what is
printed lives here and is unit-tested on fixed inputs (tests/test_bench_logic.py; tests/golden/bench_line_fake.json is
the line for a fixed fake measurement, bit for bit).

(Verdict r4 item 9: bench.py was 1 000 lines with the reporting spread over closures of main() and 47 lines of tests.)"""
import math
import statistics

WORKLOADS = {
    # name: (P, W, H, SH degree, #views)
    "s3": (1_500_000, 1600, 1200, 3, 8),   # ScanNet++-like, BASELINE configs[2] (metric resolution)
    "s2": (300_000, 1200, 680, 3, 8),      # Replica-room0-like, configs[1]/[3]
    "s1": (10_000, 256, 256, 3, 1),        # configs[0]
    "s5": (3_000_000, 1200, 680, 3, 8),    # DeepBlending-like, configs[4]
    # S3 with a TRAINED distribution (verdict r4 item 5): the room re-learnt from its own renders through the product's
    # training path for 1 000 iterations of the reference's densify / prune / SH / opacity-reset schedule
    # (g4splat_amd/trained_scene.py); the surfel count is what the training leaves (about 1 M)
    "s3t": (1_500_000, 1600, 1200, 3, 8),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS = 1024           # 256 CUs x 4 SIMDs
STRONG_VIEWS = 8       # C4 = 8 training views per optimiser step over 1/2/4/8 GPUs
# Issue cost of a wave64 VALU instruction, nominal cycles per SIMD:
GUIDE_FMA_CPI = 2.0    # MI355X_MICROARCH.md: the part's 157 TFLOP/s FP32 = one v_fma_f32 per SIMD every 2 cycles
FMA_CPI = 2.77         # measured, independent v_fma_f32 streams at 4 waves per SIMD (tools/micro/exec_rows.hip)
DEP_CPI_4WAVES = 5.3   # measured, four waves of one dependent chain each


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


def whole_step_bytes(P, V, R, N, K, tiles, tile_bits):
    """SURVEY.md 8(d)'s B: the algorithmic bytes of every kernel of one forward + backward (M = 16)."""
    return (P * 60 + V * (12 * K + 76) + R * 12 + R * 24 * ((32 + tile_bits + 7) // 8) + R * 8 + tiles * 8
            + 2 * (R * 76 + N * 60) + V * 72 + V * (44 + 12 * K + 72 + 36 + 3) + P * 52 + P * 12 * 16 + V * 12 * K)


def summarize_steps(per_step_ms, mean_ms):
    """median / mean / max of one timed pass and whether something other than the work itself landed in it."""
    med = statistics.median(per_step_ms)
    disturbed = mean_ms > 1.1 * med
    return {"median_ms": med, "mean_ms": mean_ms, "max_ms": max(per_step_ms), "min_ms": min(per_step_ms),
            "disturbed": bool(disturbed),
            # the steps that carry the excess: more than 1.5x the median
            "disturbed_steps": [i for i, t in enumerate(per_step_ms) if t > 1.5 * med] if disturbed else []}


def pick_headline(attempts):
    """The pass the line reports: the median (by mean step time) of the undisturbed passes -- of all passes if none was
    clean, in which case the chosen pass itself says `disturbed`."""
    clean = [a for a in attempts if not a["disturbed"]] or attempts
    return sorted(clean, key=lambda a: a["mean_ms"])[(len(clean) - 1) // 2]


def median_of_all_passes(attempts):
    """The lower-median pass by mean step time over ALL passes, disturbed ones included (the headline discards
    slow passes, so the unfiltered figure is printed next to it)."""
    return sorted(attempts, key=lambda a: a["mean_ms"])[(len(attempts) - 1) // 2]


def views_of_step(i, rank, world, n_views, strong, views_per_rank):
    """The camera indices rank `rank` renders in step i.  strong: the step's STRONG_VIEWS views are dealt
    round robin -- rank r takes r, r + world, ... -- and every step renders the same eight; weak: one view per rank per
    step, the ranks walking the camera ring together."""
    if strong:
        return [(rank + j * world) % n_views for j in range(views_per_rank)]
    return [(rank + i * world) % n_views]


def hbm_bytes(pk):
    """HBM bytes of one launch from its PMC row (MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide read on gfx950)."""
    if pk and "FETCH_SIZE" in pk and "WRITE_SIZE" in pk:
        return int((2.0 * pk["FETCH_SIZE"] + pk["WRITE_SIZE"]) * 1024)
    return None


def valu_report(pk, kernel_ms, lane):
    """What binds a VALU-bound kernel, from its PMC row `pk` (committed rocprofv3 passes), this run's duration and the
    useful-lane share of its visits.  None without an instruction count.

    Calibration (tools/micro/exec_rows.hip, profiles/r04_exec_lane_threshold.txt; long kernels -- round 3's "4 cycles"
    carried ~86 us of per-launch start-up): at 4 waves per SIMD independent v_fma_f32 issue at 2.77 nominal cycles per wave
    instruction (2.0 = the part's 157 TFLOP/s, the figure of MI355X_MICROARCH.md), a transcendental costs ~13.5, and ONE
    wave issues a DEPENDENT instruction every ~21 cycles -- four waves of dependent code reach 5.3 cycles per
    instruction.  `cycles_per_instruction_profiled` is the launch's own quotient -- SIMD cycles (GRBM_GUI_ACTIVE / 8 XCDs
    x 1024 SIMDs) / counted VALU instructions -- so nothing is assumed about the clock.  The blend kernels sit between the
    two: bound by how fast four (seven) waves of mostly dependent code can issue, scalar instructions and branches
    included, not by the FMA rate."""
    if not pk or "SQ_INSTS_VALU" not in pk:
        return None
    insts = float(pk["SQ_INSTS_VALU"])
    t = kernel_ms * 1e-3
    cpi = (pk["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS / insts) if pk.get("GRBM_GUI_ACTIVE") else None
    issue = round(min(1.0, FMA_CPI / cpi), 4) if cpi else None
    guide = round(min(1.0, GUIDE_FMA_CPI / cpi), 4) if cpi else None
    extra = {}
    if all(c in pk for c in ("SQ_INSTS_SALU", "SQ_INSTS_LDS")) and pk.get("GRBM_GUI_ACTIVE"):
        # Round 6 (LAB_NOTES section 6): the waves' scalar, LDS and memory instructions take issue slots too, and the blend
        # kernels sit at the SIMD's issue limit for their TOTAL count -- ~2.9 cycles per instruction of any kind against the
        # 2.77 of independent FMA streams; a second independent chain per wave (software pipelining) bought 2.5 %
        anyk = insts + float(pk["SQ_INSTS_SALU"]) + float(pk["SQ_INSTS_LDS"]) + float(pk.get("SQ_INSTS_VMEM_RD", 0)) + float(pk.get("SQ_INSTS_VMEM_WR", 0))
        extra = {"wave_instructions_of_any_kind_per_launch": int(anyk),
                 "cycles_per_instruction_of_any_kind": round(pk["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS / anyk, 3),
                 "frac_of_issue_limit_any_kind": round(min(1.0, FMA_CPI / (pk["GRBM_GUI_ACTIVE"] / 8.0 * SIMDS / anyk)), 4)}
    return {"wave_instructions_per_launch": int(insts),
            **extra,
            "cycles_per_instruction_profiled": round(cpi, 3) if cpi else None,
            "full_rate_cycles_per_instruction": FMA_CPI,
            "guide_cycles_per_instruction": GUIDE_FMA_CPI,
            "dependent_code_cycles_per_instruction_at_4_waves": DEP_CPI_4WAVES,
            # share of the SIMDs' FMA issue rate the launch used, clock-free: against the rate measured on this part
            # (2.77) and against the guide's 2 cycles per wave64 v_fma_f32 (verdict r4 item 8)
            "simd_issue_utilisation": issue,
            "frac_of_scalar_issue": issue,
            "frac_of_guide_issue": guide,
            # lanes with a blending pixel / 64 per visit of the entry loop (profiles/r04_lane_util_model.txt)
            "useful_lane_frac": lane,
            # guide issue share x useful lanes: an upper bound of the FP32 peak fraction (it counts every instruction as an FMA)
            "frac_of_fp32_peak": (round(guide * lane, 4) if (guide is not None and lane is not None) else None),
            "ns_per_instruction_this_run": round(t * SIMDS / insts * 1e9, 4),
            "calibration": "tools/micro/exec_rows.hip on MI355X (profiles/r04_exec_lane_threshold.txt): independent "
                           "v_fma_f32 2.77 cycles per wave instruction at 4 waves per SIMD, the blend mix (6 fma + exp + "
                           "rcp) 5.46, one dependent chain per wave 5.3; MI355X_MICROARCH.md: 2.0"}


def roofline_report(kernels_ms, pmc, build_id, P, Vm, Rm, W, H, D, step_ms):
    """`roofline` of the bench line: the dominant kernel (largest average launch among those SURVEY.md 8(d) prices)
    against the HBM peak -- the contract's figures -- plus what actually binds it (`valu`) and the whole step.
    pmc: the committed PMC table of the workload (profiles/rNN_traffic_<workload>.json) or None."""
    if not kernels_ms:
        return None
    N = W * H
    dom = max((k for k in kernels_ms if algorithmic_bytes(k, 1, 1, 1, 1, 1, 1, 1, 8) is not None), key=lambda k: kernels_ms[k])
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    tile_bits = max(1, math.ceil(math.log2(tiles + 1)))
    K = (D + 1) ** 2
    B = algorithmic_bytes(dom, P, Vm, Rm, N, K, 16, tiles, tile_bits)
    achieved = B / (kernels_ms[dom] * 1e-3) / 1e9
    pk = (pmc or {}).get("kernels", {}).get(dom)
    valu = valu_report(pk, kernels_ms[dom], (pmc or {}).get("useful_lane_frac", {}).get(dom))
    if valu and valu["simd_issue_utilisation"] is not None:
        # the roof the kernel is closer to -- or neither: a small frame (S1: one wave per SIMD) runs at a third of the
        # issue rate and 2 % of the HBM peak; that is latency, not a roofline.  "valu" = instruction issue / latency of
        # the waves' own code (see `valu`), as opposed to memory
        bound = "valu" if valu["simd_issue_utilisation"] > achieved / HBM_PEAK_GBS else "hbm"
        if max(valu["simd_issue_utilisation"], achieved / HBM_PEAK_GBS) < 0.5:
            bound = "latency"
    else:
        bound = "unknown"  # no counters for this workload / kernel: the HBM fraction is all the run can say
    B_step = whole_step_bytes(P, Vm, Rm, N, K, tiles, tile_bits)
    moved = [hbm_bytes(q) for q in (pmc or {}).get("kernels", {}).values()]
    moved = sum(m for m in moved if m) if moved and all(m is not None for m in moved) else None
    pmc_build = (pmc or {}).get("build_id")
    return {"kernel": dom, "bound": bound, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": hbm_bytes(pk),
            "traffic_source": (pmc or {}).get("provenance"),
            "traffic_build_id": pmc_build, "traffic_matches_build": (pmc_build == build_id) if pmc else None,
            "algorithmic_bytes_per_launch": int(B), "avg_launch_ms": round(kernels_ms[dom], 4),
            "valu": valu,
            "whole_step": {"algorithmic_bytes": int(B_step), "GBps": round(B_step / (step_ms * 1e-3) / 1e9, 1),
                           "frac_of_hbm_peak": round(B_step / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "moved_bytes": moved,
                           "moved_frac_of_hbm_peak": (round(moved / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if moved else None),
                           "note": "algorithmic_bytes: SURVEY.md 8(d) bytes of all kernels, priced as the table prices "
                                   "them (the reference's 64-bit (tile | depth) key sort, p_s = ceil((32 + bit) / 8) "
                                   "passes over all instances) / this run's step time; moved_bytes: HBM bytes the "
                                   "kernels of this build moved per step (PMC, the kernels in the traffic table). "
                                   "north_star's 40 % of the HBM peak is not reachable while the two blend kernels "
                                   "-- 70 % of the step -- are VALU-issue-bound"}}


def exchange_report(backend, rccl_version, world, exchange, why, reducer, exchange_ms, pieces, rows_sent, steps,
                    allocations_before, replicas_identical):
    """`exchange` of the bench line at N > 1: which exchange ran and why, its per-step GPU time and pieces, the bytes
    through a rank's links, and the two invariants a lease must show -- persistent buffers, identical replicas."""
    owner = exchange == "owner"
    allocations = getattr(reducer, "allocations", None)
    return {
        "backend": backend, "rccl_version": rccl_version,
        "rccl_ranks": world,  # from the communicator, not from the command line
        "ran": ("owner-reduce: MAX all-reduce [P + N^2] int32 + uneven all_to_all of visible rows + grouped in-place "
                "all_gather of the reduced shards" if owner else
                "visible-rows all-reduce: MAX all-reduce of the radii + one SUM all-reduce of the union's rows"),
        "why": why,
        "coalesced_gather": bool(getattr(reducer, "_coalesce", False)) if owner else None,
        # dense: every owner's whole shard; sparse: only the rows some rank saw (a minority of the scene on few ranks)
        "gather": getattr(reducer, "last_gather", None) if owner else None,
        # True: the backward's per-Gaussian kernel wrote the send rows itself (no pack launch behind the backward)
        "prepacked": getattr(reducer, "last_prepacked", None) if owner else None,
        "ms_per_step": round(exchange_ms, 4) if exchange_ms is not None else None,
        "ms_pieces": pieces,
        "pieces_note": ("HIP-event pairs on rank 0 in the instrumented pass; begin_local + max_all_reduce are issued "
                        "right after the forward and overlap the backward, `ms_per_step` covers what follows the "
                        "backward (statistics, pack, all_to_all, accumulate, all_gather)"),
        "bytes_per_rank": dict(getattr(reducer, "last_bytes", {})) or None,
        "rows_sent_per_step": (round(sum(rows_sent[-steps:]) / steps) if rows_sent else None),
        "buffer_allocations": allocations,
        # persistent buffers: nothing of the exchange is (re)allocated once the warm-up is over
        "buffer_allocations_after_warmup": (allocations - allocations_before if allocations_before is not None else None),
        "replicas_identical": replicas_identical,
    }


def assemble_line(m):
    """The ONE JSON line (as a dict) from the measurement `m` bench.py collected on rank 0:

      workload, P, W, H, D, n_views, world, views_per_rank, strong, steps, warmup, scaling, backend, exchange, presized
      units, inst            Gaussians with radii > 0 / (Gaussian, tile) instances over the K steps, all ranks
      Vs                     {camera index: visible Gaussians} on rank 0
      attempts, head         every timed pass (summarize_steps + per_step_ms) and the headline pass (pick_headline)
      settle, device_allocations, kernels_ms, elapsed_events_s, sustained, exchange_info, exchanged_rows, exchange_ms,
      views_in_flight, roofline, cpu_baseline, build_id, trained_scene"""
    steps, world, vpr, strong = m["steps"], m["world"], m["views_per_rank"], m["strong"]
    head, attempts = m["head"], m["attempts"]
    step_ms = head["mean_ms"]
    every = median_of_all_passes(attempts)
    units, inst, P, W, H, D = m["units"], m["inst"], m["P"], m["W"], m["H"], m["D"]
    kernels_ms = m["kernels_ms"]
    kernels_sum = sum(kernels_ms.values()) if kernels_ms else None
    sustained, settle = m["sustained"], m["settle"]
    backend, exchange = m["backend"], m["exchange"]
    exchanged_rows, exchange_ms = m["exchanged_rows"], m["exchange_ms"]
    return {
        "metric": "rasterized Gaussians/s fwd+bwd @1600x1200" if m["workload"] == "s3"
        else f"rasterized Gaussians/s fwd+bwd @{W}x{H}",
        # value = units of the K steps / the time of the K steps (barrier + synchronize on both sides, MAX over ranks) of
        # the median undisturbed pass (see `timing`); *_all_passes = the same quotient of the median over ALL passes
        "value": units / (steps * step_ms * 1e-3), "unit": "Gaussians/s", "n_gpus": world, "steps": steps,
        "warmup": m["warmup"], "ms_per_step": step_ms, "higher_is_better": True, "scaling": m["scaling"],
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_all_passes": units / (steps * every["mean_ms"] * 1e-3), "ms_per_step_all_passes": every["mean_ms"],
        "config": {"workload": f"{m['workload']}: {P} surfels (room box" + (", re-learnt: trained distribution" if m["trained_scene"] else "")
                               + f"), {W}x{H}, SH degree {D}, {m['n_views']} views, "
                               + (f"{STRONG_VIEWS} views per step over {world} GPU(s): {vpr} per GPU, accumulated "
                                  f"locally ({min(3, vpr)} in flight)" if strong else "1 view/GPU/step"),
                   "P": P, "width": W, "height": H,
                   "trained_scene": m["trained_scene"],
                   "sh_degree": D, "visible_per_view": round(units / steps / world / vpr),
                   "instances_per_view": round(inst / steps / world / vpr),
                   "views_per_step": world * vpr,
                   "visible_by_view_rank0": {str(c): m["Vs"][c] for c in sorted(m["Vs"])},
                   "forward": "presized (no host read-back)" if (m["presized"] or strong) else "reference-shaped",
                   "parallelism": f"view-dp{world}" + ((("+rccl" if backend == "nccl" else "+" + backend) +
                                                        ("-owner-reduce(all_to_all+all_gather)" if exchange == "owner"
                                                         else "-visible-rows-" + exchange)) if world > 1 else ""),
                   "exchanged_rows_per_step": (round(sum(exchanged_rows[-steps:]) / steps) if exchanged_rows else None),
                   # GPU time of the gradient exchange per step on rank 0 (what follows the backward), from the
                   # instrumented pass; it is part of every timed step at N > 1
                   "exchange_ms_per_step": round(exchange_ms, 4) if exchange_ms is not None else None},
        "build_id": m["build_id"],
        "timing": {"protocol": "passes of exactly K steps between barrier + synchronize (host clock, MAX over ranks), one HIP "
                               "event behind every step; repeated until 3 passes are undisturbed (mean <= 1.1 x median step); "
                               "ms_per_step = mean step time of the MEDIAN undisturbed pass; `median_ms` = SURVEY.md 8(d)'s "
                               "median step of that pass (the steps cycle through views of different cost, so the median "
                               "of the mix sits ~2 % above its mean); ms_per_step_all_passes = the median pass with "
                               "nothing discarded",
                   "passes": len(attempts), "undisturbed_passes": sum(1 for a in attempts if not a["disturbed"]),
                   "mean_ms": round(head["mean_ms"], 4), "median_ms": round(head["median_ms"], 4),
                   "max_ms": round(head["max_ms"], 4), "min_ms": round(head["min_ms"], 4),
                   "per_step_ms": head["per_step_ms"],
                   "value_from_median_step": units / (steps * head["median_ms"] * 1e-3),
                   "disturbed": head["disturbed"], "disturbed_steps": head["disturbed_steps"],
                   "attempts": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in a.items()
                                 if k != "per_step_ms" or a["disturbed"]} for a in attempts],
                   "warmup_extra_steps": settle["extra_steps"], "warmup_settled": settle["settled"],
                   "warmup_windows_ms_per_step": settle["windows_ms_per_step"],
                   "device_allocations_in_timed_region": m["device_allocations"],
                   "vs_sustained": (round(step_ms / sustained["ms_per_step"], 4) if sustained else None),
                   "vs_kernels_sum": (round(step_ms / kernels_sum, 4) if (kernels_sum and not strong and world == 1) else None)},
        "gaussians_total_per_s": P * steps * world * vpr / (steps * step_ms * 1e-3),
        "instances_per_s": inst / (steps * step_ms * 1e-3),
        "kernels_ms": {k: round(v, 4) for k, v in kernels_ms.items()},
        # per-kernel durations come from a second pass over the same K steps with a HIP-event pair around every kernel
        # group on the launch stream; the events cost GPU time themselves, so that pass is not the headline
        "kernel_timing": ({"pass": "same steps repeated with HIP events around each kernel group",
                           "ms_per_step_with_events": round(m["elapsed_events_s"] / steps * 1e3, 4),
                           "clock_state": ("settled (behind the sustained pass)" if sustained else "as found")}
                          if m["elapsed_events_s"] is not None else None),
        "sustained": sustained,
        "exchange": m["exchange_info"],
        "views_in_flight": m["views_in_flight"],
        "roofline": m["roofline"], "cpu_baseline": m["cpu_baseline"],
    }
