"""ONE command for the 1 / 2 / 4 / 8-GPU table of the multi-GPU path:

    python tools/scaling_table.py [--gpus 1,2,4,8] [--steps 20] [--warmup 5] [--workload s3] [--scaling weak,strong]

A thin loop over `python bench.py --gpus N --scaling S` -- bench.py starts its own ranks (one per GPU over RCCL) and prints
everything the table needs: whole-job Gaussians/s, ms per step, the exchange that ran (and why), its pieces
(begin / MAX all-reduce / pack / all_to_all / accumulate / all_gather) and bytes per rank.  Sizes that exceed the visible
GPU count are skipped (and said so).  The raw JSON lines follow the table."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, scaling, a):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(a.steps), "--warmup", str(a.warmup),
           "--workload", a.workload, "--scaling", scaling, "--no-cpu-baseline", "--views-in-flight", "0"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=3600)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    if res.returncode != 0 or not lines:
        sys.stderr.write(res.stderr[-3000:])
        return None
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="s3")
    ap.add_argument("--scaling", default="weak,strong")
    a = ap.parse_args()
    import torch
    have = torch.cuda.device_count()
    raw = []
    for scaling in a.scaling.split(","):
        base = None
        print(f"\n--scaling {scaling}\n  N |  Gaussians/s |  x N=1 | ms/step | views/step | exchange ms (after the backward) | pieces ms "
              "(begin_local, max_all_reduce | pack, all_to_all, accumulate, all_gather) | MB sent a2a / received gather | ran")
        for n in [int(x) for x in a.gpus.split(",")]:
            if n > have:
                print(f"{n:3d} | skipped ({have} GPU(s) visible)")
                continue
            b = run(n, scaling, a)
            raw.append(b)
            if b is None:
                print(f"{n:3d} | bench.py failed")
                continue
            base = base or b["value"]
            ex = b.get("exchange") or {}
            pc = ex.get("ms_pieces") or {}
            by = ex.get("bytes_per_rank") or {}
            pieces = " ".join(f"{pc.get(k, float('nan')):.3f}" for k in ("begin_local", "max_all_reduce", "pack", "all_to_all",
                                                                          "accumulate", "all_gather")) if pc else "-"
            print(f"{n:3d} | {b['value']:12.4g} | {b['value'] / base:6.2f} | {b['ms_per_step']:7.3f} | "
                  f"{b['config']['views_per_step']:10d} | {ex.get('ms_per_step') if ex else None} | {pieces} | "
                  f"{by.get('all_to_all_sent', 0) / 1e6:.1f} / {by.get('all_gather_received', 0) / 1e6:.1f} | "
                  f"{(ex.get('ran') or '-')[:24]} ({ex.get('why', '-')})"
                  + ("  DISTURBED" if b["timing"]["disturbed"] else ""))
    print()
    for r in raw:
        if r is not None:
            print(json.dumps(r))


if __name__ == "__main__":
    main()
