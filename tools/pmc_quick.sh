#!/bin/bash
# Quick PMC pass on the GPU box: tools/pmc_quick.sh <tag> [lib.so]   -> gpurun_out/pmcq_<tag>.txt
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
[ $# -ge 2 ] && cp "$2" $REPO/g4splat_amd/libg4s_hip.so && touch $REPO/g4splat_amd/libg4s_hip.so
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --sustained-seconds 0 --views-in-flight 0"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_a -- $BENCH > $OUT/a.log 2>&1
python $REPO/tools/summarize_prof.py $OUT 2>/dev/null | grep -A8 "^blend_" > $REPO/gpurun_out/pmcq_$TAG.txt
cat $REPO/gpurun_out/pmcq_$TAG.txt
