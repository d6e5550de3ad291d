#!/bin/bash
# round-4 evidence batch: rocprofv3 stats + PMC tables for S3 (and S1/S2/S5 benches), the exhaustive parity views
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/r04
ID=$(python -c "from g4splat_amd import _lib; print(_lib.load().g4s_version().decode().split('build ')[-1])")
bash tools/profile_gpu.sh r04_s3 s3 "library build $ID" > gpurun_out/r04/profile_s3.log 2>&1
python bench.py > gpurun_out/r04/bench_s3.json 2> gpurun_out/r04/bench_s3.err
for wl in s1 s2 s5; do python bench.py --workload $wl > gpurun_out/r04/bench_$wl.json 2> gpurun_out/r04/bench_$wl.err; done
python bench.py --scaling strong --no-cpu-baseline > gpurun_out/r04/bench_s3_strong_n1.json 2> gpurun_out/r04/bench_s3_strong_n1.err
python tools/lane_util_model.py s3 0 6 > gpurun_out/r04/lane_util_model.txt 2>&1
( time python -m pytest tests -q -m "gpu and exhaustive" ) > gpurun_out/r04/gpu_tests_exhaustive.txt 2>&1
tail -4 gpurun_out/r04/gpu_tests_exhaustive.txt
python - <<PY
import json
for wl in ("s3","s1","s2","s5","s3_strong_n1"):
    try:
        d=json.loads([l for l in open("gpurun_out/r04/bench_%s.json"%wl) if l.startswith("{")][-1])
        print(wl, "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "passes", d["timing"]["passes"], "disturbed", d["timing"]["disturbed"], d["kernels_ms"], (d["roofline"] or {}).get("traffic_matches_build"))
    except Exception as ex: print(wl, "FAILED", ex)
PY
tail -30 gpurun_out/r04/profile_s3.log
