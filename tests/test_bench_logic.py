"""bench.py's host-side logic (g4splat_amd/benchlib.py; no GPU): how a timed pass is judged, which pass becomes the
headline, which views a rank renders, the 8(d) byte model, what the roofline report calls the bound, the exchange report,
and the assembly of the one JSON line -- bit for bit against tests/golden/bench_line_fake.json for a fixed fake measurement."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402
from g4splat_amd import benchlib  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "bench_line_fake.json")


def test_a_stall_inside_a_pass_is_flagged_and_named():
    steps = [2.0, 2.1, 1.7, 2.0] * 5
    ok = bench.summarize_steps(steps, sum(steps) / len(steps))
    assert not ok["disturbed"] and ok["disturbed_steps"] == [] and ok["median_ms"] == 2.0
    stalled = list(steps)
    stalled[7] += 120.0  # round 3's driver run: ~190 ms of something else in a 40-ms region
    bad = bench.summarize_steps(stalled, sum(stalled) / len(stalled))
    assert bad["disturbed"] and bad["disturbed_steps"] == [7]
    assert bad["median_ms"] == 2.0 and bad["mean_ms"] > 7.0 and bad["max_ms"] > 120.0
    # a stall BETWEEN the steps (host clock only: the per-step events do not see it) is caught by the mean as well
    between = bench.summarize_steps(steps, sum(steps) / len(steps) + 1.0)
    assert between["disturbed"] and between["disturbed_steps"] == []


def test_the_headline_is_the_median_undisturbed_pass_and_the_unfiltered_median_stands_next_to_it():
    mk = lambda mean, disturbed: {"mean_ms": mean, "median_ms": mean, "disturbed": disturbed}
    passes = [mk(8.03, True), mk(1.99, False), mk(1.97, False), mk(2.05, False)]
    assert bench.pick_headline(passes)["mean_ms"] == 1.99            # the disturbed pass is not the headline ...
    assert bench.pick_headline([mk(8.0, True), mk(9.0, True), mk(7.0, True)])["mean_ms"] == 8.0  # ... unless all are
    assert bench.pick_headline([mk(2.0, False)])["mean_ms"] == 2.0
    assert bench.pick_headline([mk(2.0, False), mk(2.2, False)])["mean_ms"] == 2.0  # (even count: the lower median)
    # the median over ALL passes, nothing discarded
    assert benchlib.median_of_all_passes(passes)["mean_ms"] == 1.99
    assert benchlib.median_of_all_passes([mk(8.0, True), mk(2.0, False), mk(9.0, True)])["mean_ms"] == 8.0
    assert benchlib.median_of_all_passes([mk(2.0, False)])["mean_ms"] == 2.0


def test_algorithmic_bytes_follow_survey_8d():
    P, V, R, N, K, M, tiles, bits = 1_500_000, 423_755, 5_549_638, 1_920_000, 16, 16, 7500, 13
    b = lambda k: bench.algorithmic_bytes(k, P, V, R, N, K, M, tiles, bits)
    assert b("blend_bwd") == R * 76 + N * 60 + V * 72 == 567_482_848  # (the bench line of round 3 quoted 567 482 888: per-step averages of V and R)
    assert b("blend_fwd") == R * 76 + N * 60
    assert b("tile_sort") == R * 24 * 6                                # p_s = ceil((32 + 13) / 8)
    assert b("no_such_kernel") is None
    # the whole step = the sum of the kernels the table prices + K2's and K3's terms, M = 16
    parts = sum(b(k) for k in ("preprocess_fwd", "emit", "tile_sort", "tile_ranges", "blend_fwd", "blend_bwd", "preprocess_bwd"))
    assert benchlib.whole_step_bytes(P, V, R, N, K, tiles, bits) == parts
    assert 2.7e9 < parts < 2.9e9  # 2.81 GB per S3 step


def test_command_line_defaults_are_the_drivers_contract():
    a = bench.parse_args([])
    assert (a.gpus, a.workload, a.scaling) == (1, "s3", "weak") and a.steps == 20 and a.warmup == 5
    a = bench.parse_args(["--gpus", "8", "--steps", "50", "--warmup", "10", "--scaling", "strong"])
    assert (a.gpus, a.steps, a.warmup, a.scaling) == (8, 50, 10, "strong")
    assert set(bench.WORKLOADS) == {"s1", "s2", "s3", "s3t", "s5"} and bench.WORKLOADS["s3"][:3] == (1_500_000, 1600, 1200)


def test_view_sharding_weak_and_strong():
    """strong = the step's eight views dealt round robin over the ranks, the same eight every step; weak =
    one view per rank per step, the ranks walking the camera ring together."""
    for world in (1, 2, 4, 8):
        per = 8 // world
        dealt = [benchlib.views_of_step(3, r, world, 8, True, per) for r in range(world)]
        assert sorted(v for d in dealt for v in d) == list(range(8)) and all(len(d) == per for d in dealt)
        assert dealt == [benchlib.views_of_step(0, r, world, 8, True, per) for r in range(world)]
        for i in (0, 1, 5):
            ring = [benchlib.views_of_step(i, r, world, 8, False, 1) for r in range(world)]
            assert ring == [[(r + i * world) % 8] for r in range(world)]
    assert benchlib.views_of_step(7, 0, 1, 1, False, 1) == [0]  # S1: one camera


def _pmc(cycles_per_inst):
    insts = 5.0e8
    return {"build_id": "abc", "provenance": "fake", "useful_lane_frac": {"blend_bwd": 0.5},
            "kernels": {"blend_bwd": {"SQ_INSTS_VALU": insts, "GRBM_GUI_ACTIVE": cycles_per_inst * insts / 1024 * 8,
                                      "FETCH_SIZE": 300000.0, "WRITE_SIZE": 400000.0}}}


def test_roofline_report_names_the_bound():
    k = {"blend_bwd": 0.8, "blend_fwd": 0.4, "count_scan": 5.0}  # (count_scan is not priced by 8(d): never the dominant one)
    args = (1_500_000, 423_755.0, 5_549_638.0, 1600, 1200, 3, 1.7)
    r = benchlib.roofline_report(k, _pmc(3.5), "abc", *args)
    assert r["kernel"] == "blend_bwd" and r["bound"] == "valu" and r["traffic_matches_build"] is True
    assert r["algorithmic_bytes_per_launch"] == 567_482_848 and abs(r["frac"] - 567_482_848 / 0.8e-3 / 8e12) < 1e-5
    v = r["valu"]
    assert v["cycles_per_instruction_profiled"] == 3.5
    assert v["simd_issue_utilisation"] == round(2.77 / 3.5, 4) and v["frac_of_guide_issue"] == round(2.0 / 3.5, 4)  # verdict r4 item 8
    assert v["frac_of_fp32_peak"] == round(v["frac_of_guide_issue"] * 0.5, 4)
    assert r["traffic"] == int((2 * 300000.0 + 400000.0) * 1024)
    # neither roof half reached: latency
    assert benchlib.roofline_report(k, _pmc(9.0), "abc", *args)["bound"] == "latency"
    # no counters: the HBM fraction is all the run can say
    r0 = benchlib.roofline_report(k, None, "abc", *args)
    assert r0["bound"] == "unknown" and r0["valu"] is None and r0["traffic"] is None and r0["traffic_matches_build"] is None
    # counters of another build are flagged
    assert benchlib.roofline_report(k, _pmc(3.5), "other", *args)["traffic_matches_build"] is False
    # an HBM-bound dominant kernel
    kb = {"preprocess_bwd": 0.12, "blend_bwd": 0.1}
    pm = {"kernels": {"preprocess_bwd": {"SQ_INSTS_VALU": 1e7, "GRBM_GUI_ACTIVE": 4.0e5 * 8, "FETCH_SIZE": 1.0, "WRITE_SIZE": 1.0}}}
    rb = benchlib.roofline_report(kb, pm, "abc", *args)
    assert rb["kernel"] == "preprocess_bwd" and rb["bound"] == "hbm" and rb["frac"] > 0.5
    assert benchlib.roofline_report({}, None, "abc", *args) is None


class _Reducer:
    allocations = 5
    _coalesce = True
    last_gather = "dense"
    last_bytes = {"all_to_all_sent": 90, "all_gather_received": 315}


def test_exchange_report_bookkeeping():
    e = benchlib.exchange_report("nccl", "2.22.3", 8, "owner", "default", _Reducer(), 0.61234, {"pack": 0.08}, [100, 200, 300, 500], 2, 5, True)
    assert e["rccl_ranks"] == 8 and e["ran"].startswith("owner-reduce") and e["why"] == "default" and e["gather"] == "dense"
    assert e["rows_sent_per_step"] == 400 and e["ms_per_step"] == 0.6123 and e["coalesced_gather"] is True
    assert e["buffer_allocations"] == 5 and e["buffer_allocations_after_warmup"] == 0 and e["replicas_identical"] is True
    assert e["bytes_per_rank"] == {"all_to_all_sent": 90, "all_gather_received": 315}
    f = benchlib.exchange_report("gloo", None, 2, "allreduce", "requested", object(), None, None, [], 2, None, False)
    assert f["ran"].startswith("visible-rows all-reduce") and f["gather"] is None and f["coalesced_gather"] is None
    assert f["rows_sent_per_step"] is None and f["buffer_allocations_after_warmup"] is None and f["replicas_identical"] is False


def _fake_measurement():
    attempts = []
    for mean, steps in ((8.03, [2.0, 2.1, 120.0, 1.9]), (1.99, [2.0, 2.1, 1.9, 1.96]), (1.97, [1.9, 2.0, 2.0, 1.98]),
                        (2.05, [2.1, 2.0, 2.05, 2.05])):
        a = benchlib.summarize_steps(steps, mean)
        a["per_step_ms"] = steps
        attempts.append(a)
    kernels = {"preprocess_fwd": 0.102, "depth_sort": 0.074, "count_scan": 0.013, "emit": 0.031, "tile_sort": 0.074,
               "tile_ranges": 0.010, "blend_fwd": 0.401, "blend_bwd": 0.785, "preprocess_bwd": 0.204}
    roof = benchlib.roofline_report(kernels, _pmc(3.2), "abc", 1_500_000, 423_755.0, 5_549_638.0, 1600, 1200, 3, 1.99)
    return dict(workload="s3", P=1_500_000, W=1600, H=1200, D=3, n_views=8, world=1, views_per_rank=1, strong=False, steps=4,
                warmup=2, scaling="weak", backend=None, exchange="owner", presized=False, units=4 * 423_755, inst=4 * 5_549_638,
                Vs={0: 584_578, 1: 400_000, 2: 300_000, 3: 410_442}, attempts=attempts, head=benchlib.pick_headline(attempts),
                settle={"extra_steps": 16, "windows_ms_per_step": [1.9, 1.91, 1.9], "settled": True}, device_allocations=2,
                kernels_ms=kernels, elapsed_events_s=0.0078, sustained={"ms_per_step": 1.7}, exchange_info=None,
                exchanged_rows=[], exchange_ms=None, views_in_flight=None, roofline=roof, cpu_baseline={"value": 1.0e5},
                build_id="abc", trained_scene=None)


def test_the_line_for_a_fixed_fake_measurement_is_the_committed_one_bit_for_bit():
    line = json.dumps(benchlib.assemble_line(_fake_measurement()))
    if os.environ.get("G4S_WRITE_GOLDEN"):
        with open(GOLDEN, "w") as f:
            f.write(line + "\n")
    with open(GOLDEN) as f:
        assert line == f.read().rstrip("\n")
    d = json.loads(line)
    # the contract's keys, and what selects them
    assert d["metric"] == "rasterized Gaussians/s fwd+bwd @1600x1200" and d["unit"] == "Gaussians/s" and d["higher_is_better"] is True
    assert d["ms_per_step"] == 1.99 and d["value"] == 4 * 423_755 / (4 * 1.99e-3)          # the median undisturbed pass
    assert d["ms_per_step_all_passes"] == 1.99 and d["timing"]["passes"] == 4 and d["timing"]["undisturbed_passes"] == 3
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["config"]["parallelism"] == "view-dp1"
    assert d["timing"]["vs_kernels_sum"] == round(1.99 / sum(_fake_measurement()["kernels_ms"].values()), 4)
    assert [a.get("per_step_ms") for a in d["timing"]["attempts"]] == [[2.0, 2.1, 120.0, 1.9], None, None, None]  # only the disturbed pass lists its steps


def test_the_line_at_eight_ranks_strong_scaling():
    m = _fake_measurement()
    m.update(world=8, strong=True, scaling="strong", backend="nccl", views_per_rank=1, workload="s2", W=1200, H=680,
             exchange_info={"rccl_ranks": 8}, exchanged_rows=[10, 20, 30, 40], exchange_ms=0.5)
    d = benchlib.assemble_line(m)
    assert d["metric"] == "rasterized Gaussians/s fwd+bwd @1200x680" and d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["config"]["views_per_step"] == 8 and d["config"]["forward"].startswith("presized")
    assert d["config"]["parallelism"] == "view-dp8+rccl-owner-reduce(all_to_all+all_gather)"
    assert d["config"]["exchanged_rows_per_step"] == 25 and d["config"]["exchange_ms_per_step"] == 0.5
    assert d["timing"]["vs_kernels_sum"] is None and d["exchange"] == {"rccl_ranks": 8}
    assert "8 views per step over 8 GPU(s): 1 per GPU" in d["config"]["workload"]
    m.update(exchange="allreduce", backend="gloo")
    assert benchlib.assemble_line(m)["config"]["parallelism"] == "view-dp8+gloo-visible-rows-allreduce"
