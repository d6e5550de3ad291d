#!/bin/bash
# Runs on the GPU box (via gpurun) as the FIRST GPU work of a fresh lease: the driver's N=1 command three times in a row
# (separate processes), then once more with a 1-Hz amd-smi / sysfs poll beside it (what a monitoring side-car does),
# to see whether the headline survives a cold box and a poller.  Output: gpurun_out/<tag>/cold_{1,2,3}.json, polled.json
TAG=${1:-r04_cold}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/cold_$i.json 2> $OUT/cold_$i.err
done
( while true; do amd-smi metric > /dev/null 2>&1; cat /sys/class/drm/card*/device/pp_dpm_sclk > /dev/null 2>&1; sleep 1; done ) &
POLL=$!
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --views-in-flight 0 > $OUT/polled.json 2> $OUT/polled.err
kill $POLL 2>/dev/null
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as ex:
        print(f, "NO LINE", ex); continue
    t = d["timing"]
    print(f.split("/")[-1], "value %.4g" % d["value"], "median", t["median_ms"], "mean", t["mean_ms"], "max", t["max_ms"],
          "disturbed", t["disturbed"], t["disturbed_steps"], "passes", t["passes"], t["undisturbed_passes"], [a["mean_ms"] for a in t["attempts"]], "extra warmup", t["warmup_extra_steps"],
          t["warmup_settled"], "allocs", t["device_allocations_in_timed_region"], "vs_sustained", t["vs_sustained"], "vs_kernels", t["vs_kernels_sum"])
    print("   windows", t["warmup_windows_ms_per_step"])
    print("   per-step", t["per_step_ms"])
PY
