"""ONE command for the 1 / 2 / 4 / 8-GPU table of the multi-GPU path:

    python tools/scaling_table.py [--gpus 1,2,4,8] [--steps 20] [--warmup 5] [--workload s3]

For every N it launches bench.py exactly as the driver does (torch.distributed.run, one rank per GPU over RCCL) and
tools/time_allreduce.py on the same ranks, and prints one row per N:

    N | Gaussians/s (whole job) | x vs N=1 | ms/step | exchange ms/step | rows sent | MB/rank all_to_all + all_gather |
      dense all-reduce ms | owner begin / finish ms | ZeRO-1 finish ms | owner == dense (max rel. diff)

followed by the raw JSON lines.  Sizes that exceed the visible GPU count are skipped (and said so)."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(n, script, args, port):
    cmd = [sys.executable, os.path.join(ROOT, script)] + args
    if n > 1 or script.endswith("time_allreduce.py"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, script)] + args
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=3600)
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
    a = ap.parse_args()
    import torch
    have = torch.cuda.device_count()
    rows, raw = [], []
    base = None
    for n in [int(x) for x in a.gpus.split(",")]:
        if n > have:
            print(f"N={n}: skipped ({have} GPU(s) visible)")
            continue
        b = run(n, "bench.py", ["--gpus", str(n), "--steps", str(a.steps), "--warmup", str(a.warmup), "--workload",
                                 a.workload, "--no-cpu-baseline"], 29600 + n)
        t = run(n, "tools/time_allreduce.py", [], 29700 + n)
        raw += [b, t]
        if b is None:
            print(f"N={n}: bench.py failed")
            continue
        base = base or b["value"]
        cfg = b["config"]
        rows.append((n, b["value"], b["value"] / base, b["ms_per_step"], cfg.get("exchange_ms_per_step"),
                     cfg.get("exchanged_rows_per_step"), t))
    print("\n  N |  Gaussians/s |  x N=1 | ms/step | exch ms | rows sent | MB a2a + gather | dense AR ms | begin / finish ms | "
          "ZeRO-1 ms | owner==dense")
    for n, v, x, ms, ex, rws, t in rows:
        mb = t["MB_per_rank"] if t else {}
        tm = t["ms"] if t else {}
        print(f"{n:3d} | {v:12.4g} | {x:6.2f} | {ms:7.3f} | {ex if ex is not None else float('nan'):7.3f} | "
              f"{rws if rws is not None else 0:9d} | {mb.get('all_to_all', 0):6.1f} + {mb.get('all_gather_received', 0):6.1f} | "
              f"{tm.get('dense_all_reduce', float('nan')):11.3f} | {tm.get('owner_begin', float('nan')):6.3f} / "
              f"{tm.get('owner_finish', float('nan')):6.3f} | {tm.get('zero1_finish', float('nan')):9.3f} | "
              f"{t['max_rel_diff_vs_dense'] if t else float('nan'):.1e}")
    print()
    for r in raw:
        if r is not None:
            print(json.dumps(r))


if __name__ == "__main__":
    main()
