#!/bin/bash
# one GPU-box batch of round 4: lane model, partial-EXEC microbenchmark, the default suite (timed), profiles
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/r04
python tools/lane_util_model.py s3 0 6 > gpurun_out/r04/lane_util_model.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > gpurun_out/r04/valu_rate.txt 2>&1
( time python -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/r04/gpu_tests.txt 2>&1
tail -25 gpurun_out/r04/gpu_tests.txt
tail -8 gpurun_out/r04/valu_rate.txt
cat gpurun_out/r04/lane_util_model.txt
