#!/bin/bash
# Issue-stall counters for the blend kernels: tools/pmc_stall.sh <tag> [lib.so] -> gpurun_out/pmcs_<tag>.txt
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
[ $# -ge 2 ] && cp "$2" $REPO/g4splat_amd/libg4s_hip.so && touch $REPO/g4splat_amd/libg4s_hip.so
OUT=/tmp/pmcs_$TAG  # raw rocprofv3 output stays off gpurun_out (64 MiB merge limit)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing --sustained-seconds 0 --views-in-flight 0"
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_a -- $BENCH > $OUT/a.log 2>&1
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_b -- $BENCH > $OUT/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_c -- $BENCH > $OUT/c.log 2>&1
python $REPO/tools/summarize_prof.py $OUT 2>/dev/null | grep -A26 "^blend_bwd_kernel$" > $REPO/gpurun_out/pmcs_$TAG.txt
cat $REPO/gpurun_out/pmcs_$TAG.txt
