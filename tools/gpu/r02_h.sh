#!/bin/bash
# round 2, GPU session H: ablation of the one-wave GEMM loop (what keeps the matrix pipe at 85 %?)
set -u
mkdir -p gpurun_out/r02h
for d in 0 1 2 3; do
  echo "== EAP_GEMM_DEBUG=$d" >> gpurun_out/r02h/ablation.txt
  EAP_GEMM_DEBUG=$d timeout 300 python tools/gemm_only.py 8 first >> gpurun_out/r02h/ablation.txt 2>&1
done
timeout 300 python -m pytest tests/test_gpu_lists_and_modules.py -m gpu -x -q > gpurun_out/r02h/pytest.log 2>&1
cat gpurun_out/r02h/ablation.txt; tail -3 gpurun_out/r02h/pytest.log
