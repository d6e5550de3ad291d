#!/bin/bash
# round-6 evidence batch (GPU box): rocprofv3 stats + PMC tables of the final build for S3 (S1 / S2 / S5: bench lines), the
# GPU suites (timed), the parity report, the REC_AFFINE shares, distCUDA2 under rocprofv3, the exchange's local costs
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/r06
ID=$(python -c "from g4splat_amd import _lib; print(_lib.load().g4s_version().decode().split('build ')[-1])")
echo "library build $ID"
bash tools/profile_gpu.sh r06_s3 s3 "library build $ID" > gpurun_out/r06/profile_s3.log 2>&1
cp gpurun_out/prof_r06_s3/traffic_s3.json profiles/r06_traffic_s3.json 2>/dev/null   # so that the bench lines below quote THIS build's counters
python bench.py > gpurun_out/r06/bench_s3.json 2> gpurun_out/r06/bench_s3.err
for wl in s1 s2 s5; do
  python bench.py --workload $wl > gpurun_out/r06/bench_$wl.json 2> gpurun_out/r06/bench_$wl.err
done
python bench.py --workload s3t --no-cpu-baseline > gpurun_out/r06/bench_s3t.json 2> gpurun_out/r06/bench_s3t.err
python bench.py --scaling strong --no-cpu-baseline > gpurun_out/r06/bench_s3_strong_n1.json 2> gpurun_out/r06/bench_s3_strong_n1.err
python tools/affine_stats.py s1 s2 s3 s5 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/affine_share.txt
# distCUDA2: per-kernel times under rocprofv3 (kernel trace + stats only)
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_knn && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_knn -- python $REPO/tools/knn_bench.py $REPO/g4splat_amd/libg4s_hip.so > /tmp/prof_knn.log 2>&1; find /tmp/prof_knn -name "*kernel_stats.csv" -exec cp {} $REPO/gpurun_out/r06/knn_kernel_stats.csv \; ; grep -v amdgpu.ids /tmp/prof_knn.log > $REPO/gpurun_out/r06/knn_under_rocprof.txt )
python tools/knn_bench.py g4splat_amd/libg4s_hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/knn_bench.txt
python tools/micro/sparse_gather_local_cost.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/sparse_gather_local_cost.txt
python tools/micro/sparse_gather_local_cost.py 2 0.39 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06/sparse_gather_local_cost.txt
python tools/micro/owner_local_cost.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/owner_local_cost.txt
( time python -m pytest tests -q -m gpu --durations=12 ) > gpurun_out/r06/gpu_tests.txt 2>&1
grep -v "UserWarning\|Consider using\|extent = " gpurun_out/r06/gpu_tests.txt | tail -30 | cut -c1-300
( time python -m pytest tests -q -m "gpu and exhaustive" ) > gpurun_out/r06/gpu_tests_exhaustive.txt 2>&1
tail -4 gpurun_out/r06/gpu_tests_exhaustive.txt
python tools/parity_report.py > gpurun_out/r06/parity_report.txt 2>&1
python tools/parity_worst.py < gpurun_out/r06/parity_report.txt > gpurun_out/r06/parity_worst.txt; cat gpurun_out/r06/parity_worst.txt
python - <<PY
import json
for wl in ("s3","s1","s2","s5","s3t","s3_strong_n1"):
    try:
        d=json.loads([l for l in open("gpurun_out/r06/bench_%s.json"%wl) if l.startswith("{")][-1])
        r=d["roofline"] or {}
        print(wl, "%.4g"%d["value"], "ms/step %.4f"%d["ms_per_step"], "passes", d["timing"]["passes"], "disturbed", d["timing"]["disturbed"], "allocs", d["timing"].get("device_allocations_in_timed_region"), d["kernels_ms"], r.get("traffic_matches_build"), r.get("frac"), (r.get("valu") or {}).get("cycles_per_instruction_profiled"), (d.get("cpu_baseline") or {}).get("checker"))
    except Exception as ex: print(wl, "FAILED", ex)
PY
cat gpurun_out/r06/affine_share.txt gpurun_out/r06/sparse_gather_local_cost.txt gpurun_out/r06/owner_local_cost.txt
cat gpurun_out/r06/knn_bench.txt
tail -25 gpurun_out/r06/profile_s3.log
