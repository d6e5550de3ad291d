"""bench.py's host-side logic (no GPU): how a timed pass is judged, which pass becomes the headline, the 8(d) byte model."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


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


def test_the_headline_is_the_median_undisturbed_pass():
    mk = lambda mean, disturbed: {"mean_ms": mean, "median_ms": mean, "disturbed": disturbed}
    passes = [mk(8.03, True), mk(1.99, False), mk(1.97, False), mk(2.05, False)]
    assert bench.pick_headline(passes)["mean_ms"] == 1.99            # the disturbed pass is not the headline ...
    assert bench.pick_headline([mk(8.0, True), mk(9.0, True), mk(7.0, True)])["mean_ms"] == 8.0  # ... unless all are
    assert bench.pick_headline([mk(2.0, False)])["mean_ms"] == 2.0
    assert bench.pick_headline([mk(2.0, False), mk(2.2, False)])["mean_ms"] == 2.0  # (even count: the lower median)


def test_algorithmic_bytes_follow_survey_8d():
    P, V, R, N, K, M, tiles, bits = 1_500_000, 423_755, 5_549_638, 1_920_000, 16, 16, 7500, 13
    b = lambda k: bench.algorithmic_bytes(k, P, V, R, N, K, M, tiles, bits)
    assert b("blend_bwd") == R * 76 + N * 60 + V * 72 == 567_482_848  # (the bench line of round 3 quoted 567 482 888: per-step averages of V and R)
    assert b("blend_fwd") == R * 76 + N * 60
    assert b("tile_sort") == R * 24 * 6                                # p_s = ceil((32 + 13) / 8)
    assert b("no_such_kernel") is None


def test_command_line_defaults_are_the_drivers_contract():
    a = bench.parse_args([])
    assert (a.gpus, a.workload, a.scaling) == (1, "s3", "weak") and a.steps == 20 and a.warmup == 5
    a = bench.parse_args(["--gpus", "8", "--steps", "50", "--warmup", "10", "--scaling", "strong"])
    assert (a.gpus, a.steps, a.warmup, a.scaling) == (8, 50, 10, "strong")
