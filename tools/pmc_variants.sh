#!/bin/bash
# One PMC pass per library variant, kernels whose name contains <pattern>:
#   tools/pmc_variants.sh <pattern> var/A.so var/B.so ...     (run on the GPU box; counters in their own pass, kernel trace only)
set -u
PAT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
keep=$(mktemp); cp $REPO/g4splat_amd/libg4s_hip.so "$keep"
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --sustained-seconds 0 --views-in-flight 0"
for v in "$@"; do
  cp $REPO/$v $REPO/g4splat_amd/libg4s_hip.so; touch $REPO/g4splat_amd/libg4s_hip.so
  OUT=$REPO/gpurun_out/pmcv_$(basename $v .so); mkdir -p $OUT
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_a -- $BENCH > $OUT/a.log 2>&1
  echo "== $v"
  python $REPO/tools/summarize_prof.py $OUT 2>/dev/null | awk -v pat="$PAT" '/^[a-z_]/ {show = index($0, pat) > 0} show'
done
cp "$keep" $REPO/g4splat_amd/libg4s_hip.so
