#!/bin/bash
set -u
mkdir -p gpurun_out/r02j
for d in 4 5 6; do
  echo "== EAP_GEMM_DEBUG=$d" >> gpurun_out/r02j/ablation.txt
  EAP_GEMM_DEBUG=$d timeout 300 python tools/gemm_only.py 8 first 2>&1 | grep gemm_dma >> gpurun_out/r02j/ablation.txt
done
cat gpurun_out/r02j/ablation.txt
