#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + stats, then PMC passes, of a short bench run.
# Usage: tools/profile_gpu.sh <tag> [workload] [provenance text, e.g. the commit the library was built from]
set -u
TAG=${1:-r01}
WL=${2:-s3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
# raw rocprofv3 output goes to /tmp (tens of MB); only the summaries are copied into gpurun_out/prof_<tag>/ (gpurun merges
# at most 64 MiB back)
OUT=/tmp/prof_$TAG
KEEP=$REPO/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $KEEP
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --views-in-flight 0 --sustained-seconds 0.2"
# kernel trace + stats of the command the driver runs (default steps / warm-up, sustained pass included) minus two legs: the
# CPU baseline launches no kernels, and the views-in-flight leg runs K steps CONCURRENTLY on K streams -- its kernels overlap
# and would inflate every per-kernel average (preprocess_bwd 0.20 -> 0.36 ms) without being part of the headline
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --workload $WL --no-cpu-baseline --views-in-flight 0 > $OUT/stats.log 2>&1
grep '^{"metric"' $OUT/stats.log > $OUT/bench_under_rocprof.json
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/pmc_$name.log 2>&1
done
python $REPO/tools/summarize_prof.py $OUT $OUT/traffic_$WL.json $WL "${3:-}" > $OUT/summary.txt 2>&1
cp $OUT/summary.txt $OUT/traffic_$WL.json $OUT/bench_under_rocprof.json $KEEP/ 2>/dev/null
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $KEEP/kernel_stats.csv \; 2>/dev/null
tail -5 $OUT/stats.log > $KEEP/stats_tail.log 2>/dev/null
cat $OUT/summary.txt
