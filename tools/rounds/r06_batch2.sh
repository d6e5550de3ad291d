#!/bin/bash
# round 6, batch 2: the GPU suite under the re-tightened gate (MARGIN 1e-5, AFFINE_TOL 2e-5), the DP densification test,
# the bench line (allocations in the timed region, contract-literal checker figures)
mkdir -p gpurun_out/r06
( time python -m pytest tests -q -m gpu -x --durations=15 ) > gpurun_out/r06/gpu_tests.txt 2>&1
tail -40 gpurun_out/r06/gpu_tests.txt | cut -c1-400
python bench.py > gpurun_out/r06/bench_s3.json 2> gpurun_out/r06/bench_s3.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r06/bench_s3.json") if l.startswith("{")][-1])
print("value %.4g ms/step %.4f"%(d["value"], d["ms_per_step"]), d["kernels_ms"])
print("allocations in timed region:", d["timing"].get("device_allocations_in_timed_region"), "checker:", d["cpu_baseline"].get("checker"))
PY
tail -3 gpurun_out/r06/bench_s3.err
