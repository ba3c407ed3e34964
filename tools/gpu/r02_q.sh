#!/bin/bash
# session Q: where the backward grouping's time goes -- memory locality experiment + wave-count MFMA microbenchmark
mkdir -p gpurun_out/r02q
cd /root/repo
timeout 300 tools/microbench/mfma_waves > gpurun_out/r02q/mfma_waves.txt 2>&1
timeout 600 python tools/inv_locality_experiment.py > gpurun_out/r02q/inv_locality.txt 2>&1
cat gpurun_out/r02q/mfma_waves.txt gpurun_out/r02q/inv_locality.txt
