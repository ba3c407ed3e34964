#!/bin/bash
set -u
mkdir -p gpurun_out/r02k
timeout 300 python -m pytest tests/test_gpu_lists_and_modules.py tests/test_gpu_parity.py -m gpu -x -q -k "gemm" > gpurun_out/r02k/pytest.log 2>&1
for x in 1 0; do
 for d in 0 1 2; do
  echo "== X16=$x DEBUG=$d" >> gpurun_out/r02k/ablation.txt
  EAP_GEMM_X16=$x EAP_GEMM_DEBUG=$d timeout 300 python tools/gemm_only.py 8 first 2>&1 | grep gemm_dma >> gpurun_out/r02k/ablation.txt
 done
done
tail -3 gpurun_out/r02k/pytest.log; cat gpurun_out/r02k/ablation.txt
