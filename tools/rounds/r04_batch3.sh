#!/bin/bash
# round-4 evidence batch 3: PMC tables for S1/S2/S5 (build id inside), training-iteration figures, exchange local costs
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/r04
ID=$(python -c "from g4splat_amd import _lib; print(_lib.load().g4s_version().decode().split('build ')[-1])")
for wl in s1 s2 s5; do bash tools/profile_gpu.sh r04_$wl $wl "library build $ID" > gpurun_out/r04/profile_$wl.log 2>&1; done
python tools/train_iter_bench.py --iters 40 > gpurun_out/r04/train_iter.txt 2>&1
python tools/multi_view_train_bench.py --views 4 --k 2 --steps 10 > gpurun_out/r04/multi_view_train.txt 2>&1
python tools/micro/owner_local_cost.py 8 > gpurun_out/r04/owner_local_cost.txt 2>&1
python tools/micro/owner_local_cost.py 2 >> gpurun_out/r04/owner_local_cost.txt 2>&1
python tools/micro/fallback_local_cost.py > gpurun_out/r04/fallback_local_cost.txt 2>&1
tail -6 gpurun_out/r04/train_iter.txt; tail -6 gpurun_out/r04/multi_view_train.txt; cat gpurun_out/r04/owner_local_cost.txt | grep -v amdgpu; ls gpurun_out/prof_r04_s1 gpurun_out/prof_r04_s2 gpurun_out/prof_r04_s5
