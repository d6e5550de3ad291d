#!/bin/bash
# Issue / stall / traffic counters of the kernels whose name contains <pattern>:
#   tools/pmc_kernel.sh <tag> <pattern> [lib.so] [workload]   -> gpurun_out/pmck_<tag>.txt   (run on the GPU box)
# Counters are collected in separate --pmc passes with --kernel-trace only (no other trace domain).
set -u
TAG=${1:-x}; PAT=${2:-blend}; WL=${4:-s3}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
[ $# -ge 3 ] && [ -n "$3" ] && cp "$3" $REPO/g4splat_amd/libg4s_hip.so && touch $REPO/g4splat_amd/libg4s_hip.so
OUT=$REPO/gpurun_out/pmck_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --workload $WL --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_a -- $BENCH > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_b -- $BENCH > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_c -- $BENCH > $OUT/c.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_f -- $BENCH > $OUT/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_w -- $BENCH > $OUT/w.log 2>&1
python $REPO/tools/summarize_prof.py $OUT 2>/dev/null | awk -v pat="$PAT" '/^[a-z_]/ {show = index($0, pat) > 0} show' > $REPO/gpurun_out/pmck_$TAG.txt
cat $REPO/gpurun_out/pmck_$TAG.txt
