#!/bin/bash
# round 6, batch 1: the rewritten distCUDA2 search
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_parity.py -q -x -k "distCUDA2" 2>&1 | tail -5 > gpurun_out/r06/knn_tests.txt
python -m pytest tests/test_gpu_abi_client.py -q -x 2>&1 | tail -3 >> gpurun_out/r06/knn_tests.txt
timeout 900 python tools/knn_sweep.py 180 2>&1 | tail -8 > gpurun_out/r06/knn_sweep.txt
timeout 900 python tools/knn_bench.py g4splat_amd/libg4s_hip.so var/knn_r05.so > gpurun_out/r06/knn_bench.txt 2>&1
cat gpurun_out/r06/knn_tests.txt gpurun_out/r06/knn_sweep.txt gpurun_out/r06/knn_bench.txt
