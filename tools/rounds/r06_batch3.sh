#!/bin/bash
mkdir -p gpurun_out/r06
python tools/alloc_debug.py 40 2>&1 | grep -v amdgpu.ids | cut -c1-110 > gpurun_out/r06/alloc_debug.txt; tail -12 gpurun_out/r06/alloc_debug.txt
( time python -m pytest tests -q -m gpu --durations=15 ) > gpurun_out/r06/gpu_tests.txt 2>&1
grep -v "UserWarning\|Consider using\|extent = " gpurun_out/r06/gpu_tests.txt | tail -45 | cut -c1-600
python bench.py > gpurun_out/r06/bench_s3.json 2> gpurun_out/r06/bench_s3.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r06/bench_s3.json") if l.startswith("{")][-1])
print("value %.4g ms/step %.4f"%(d["value"], d["ms_per_step"]), d["kernels_ms"])
print("allocations in timed region:", d["timing"].get("device_allocations_in_timed_region"), "checker:", d["cpu_baseline"].get("checker"))
PY
tail -3 gpurun_out/r06/bench_s3.err
