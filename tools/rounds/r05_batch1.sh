#!/bin/bash
# round-5 evidence batch (GPU box): rocprofv3 stats + PMC tables of the final build for S3 / S1 / S2 / S5, the bench lines,
# the default and the exhaustive GPU suites (timed), the parity report, the REC_AFFINE shares
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/r05
ID=$(python -c "from g4splat_amd import _lib; print(_lib.load().g4s_version().decode().split('build ')[-1])")
echo "library build $ID"
bash tools/profile_gpu.sh r05_s3 s3 "library build $ID" > gpurun_out/r05/profile_s3.log 2>&1
cp gpurun_out/prof_r05_s3/traffic_s3.json profiles/r05_traffic_s3.json 2>/dev/null   # so that the bench lines below quote THIS build's counters
python bench.py > gpurun_out/r05/bench_s3.json 2> gpurun_out/r05/bench_s3.err
for wl in s1 s2 s5; do
  bash tools/profile_gpu.sh r05_$wl $wl "library build $ID" > gpurun_out/r05/profile_$wl.log 2>&1
  cp gpurun_out/prof_r05_$wl/traffic_$wl.json profiles/r05_traffic_$wl.json 2>/dev/null
  python bench.py --workload $wl > gpurun_out/r05/bench_$wl.json 2> gpurun_out/r05/bench_$wl.err
done
python bench.py --workload s3t --no-cpu-baseline > gpurun_out/r05/bench_s3t.json 2> gpurun_out/r05/bench_s3t.err
python bench.py --scaling strong --no-cpu-baseline > gpurun_out/r05/bench_s3_strong_n1.json 2> gpurun_out/r05/bench_s3_strong_n1.err
python tools/affine_stats.py s1 s2 s3 s5 > gpurun_out/r05/affine_share.txt 2>&1
( time python -m pytest tests -q -m gpu --durations=12 ) > gpurun_out/r05/gpu_tests.txt 2>&1
tail -22 gpurun_out/r05/gpu_tests.txt | cut -c1-300
( time python -m pytest tests -q -m "gpu and exhaustive" ) > gpurun_out/r05/gpu_tests_exhaustive.txt 2>&1
tail -4 gpurun_out/r05/gpu_tests_exhaustive.txt
python tools/parity_report.py > gpurun_out/r05/parity_report.txt 2>&1
python tools/parity_worst.py < gpurun_out/r05/parity_report.txt > gpurun_out/r05/parity_worst.txt; cat gpurun_out/r05/parity_worst.txt
python - <<PY
import json
for wl in ("s3","s1","s2","s5","s3t","s3_strong_n1"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05/bench_%s.json"%wl) if l.startswith("{")][-1])
        r=d["roofline"] or {}
        print(wl, "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "passes", d["timing"]["passes"], "disturbed", d["timing"]["disturbed"], d["kernels_ms"], r.get("traffic_matches_build"), r.get("frac"), (r.get("valu") or {}).get("cycles_per_instruction_profiled"))
    except Exception as ex: print(wl, "FAILED", ex)
PY
cat gpurun_out/r05/affine_share.txt | grep -v amdgpu
tail -25 gpurun_out/r05/profile_s3.log
