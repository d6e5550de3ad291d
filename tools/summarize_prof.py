"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a small text table.

    summarize_prof.py <dir> [traffic.json workload]   # also writes the FETCH_SIZE/WRITE_SIZE table bench.py reads
"""
import csv, glob, json, os, sys, collections

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

# HBM traffic table for bench.py's roofline.traffic (profiles/traffic_<workload>.json)
if len(sys.argv) >= 4:
    names = {"blend_fwd_kernel": "blend_fwd", "blend_bwd_kernel": "blend_bwd", "preprocess_fwd_kernel": "preprocess_fwd",
             "preprocess_bwd_kernel": "preprocess_bwd", "emit_kernel": "emit", "tile_ranges_kernel": "tile_ranges"}
    kern = {}
    for k in agg:
        key = names.get(k) or names.get(k.split("<")[0]) or ("tile_sort" if k.startswith("radix_scatter_kernel<unsigned long") else None)
        if key and "FETCH_SIZE" in agg[k] and "WRITE_SIZE" in agg[k]:
            kern[key] = {"FETCH_SIZE": round(agg[k]["FETCH_SIZE"] / max(cnt[k]["FETCH_SIZE"], 1), 1),
                         "WRITE_SIZE": round(agg[k]["WRITE_SIZE"] / max(cnt[k]["WRITE_SIZE"], 1), 1)}
            # instruction counts by kind + active cycles: the blend kernels are bound by how many instructions of ANY kind
            # their waves issue (LAB_NOTES section 6), not by HBM
            for c in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
                if c in agg[k]:
                    kern[key][c] = round(agg[k][c] / max(cnt[k][c], 1), 1)
    # frames small enough for the four-wave backward throughout (<= 768 tiles, e.g. S1) never launch blend_bwd_kernel:
    # there the four-wave kernel IS the blend backward
    hot = "blend_bwd_hot_kernel"
    hk = next((k for k in agg if k.split("(")[0].endswith(hot) or k.startswith(hot)), None)
    if "blend_bwd" not in kern and hk and "FETCH_SIZE" in agg[hk] and "WRITE_SIZE" in agg[hk]:
        kern["blend_bwd"] = {c: round(agg[hk][c] / max(cnt[hk][c], 1), 1)
                             for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE") if c in agg[hk]}
        kern["blend_bwd"]["kernel"] = hot
    try:  # the id of the library the counters were just collected on (g4s_version(): digest of its sources)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from g4splat_amd import _lib
        build_id = _lib.load().g4s_version().decode().split("build ")[-1]
    except Exception as ex:  # noqa: BLE001
        build_id = None
        print("build id unavailable:", ex)
    json.dump({"workload": sys.argv[3], "build_id": build_id,
               "provenance": (sys.argv[4] if len(sys.argv) >= 5 else "rocprofv3 --pmc passes (commit not recorded)") +
                             " -- static: collected by tools/profile_gpu.sh, not measured in the bench run that quotes it",
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, mean per dispatch, KiB "
                         "(tools/profile_gpu.sh)",
               "correction": "HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: on gfx950 FETCH_SIZE under-reports coalesced "
                             "reads by ~2x (MI355X_MICROARCH.md; re-checked on kernels with known byte counts: "
                             "radix_hist<u32> reads 6.0 MB -> FETCH 2.94 MB (x2.04), preprocess_fwd writes 170 MB -> "
                             "WRITE_SIZE 170 MB (x1.00))",
               "kernels": kern}, open(sys.argv[2], "w"), indent=1)
