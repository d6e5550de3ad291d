"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a small text table."""
import csv, glob, os, sys, collections

out = sys.argv[1]
def short(n):
    n = n.split("(")[0]
    return n.replace("g4s::", "").replace("void ", "")[:40]

# kernel stats
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print(f"{short(r['Name']):42s} calls {r['Calls']:>6s} total_ns {r['TotalDurationNs']:>12s} avg_ns {float(r['AverageNs']):12.0f} pct {r['Percentage']}")
# PMC
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
print("== PMC (mean per dispatch)")
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", 0)):
    if not any(s in k for s in ("blend", "preprocess", "radix", "emit", "tile_", "count", "scan", "fold", "grad_slots", "knn")):
        continue
    print(k)
    for c in sorted(agg[k]):
        print(f"    {c:28s} {agg[k][c] / max(cnt[k][c], 1):16.1f}")
