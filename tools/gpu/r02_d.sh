#!/bin/bash
# round 2, GPU session D: DMA GEMM with the DMA issue spread over the MFMA groups; dY row-pitch experiment;
# strided path tests.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02d
timeout 600 python tools/gemm_only.py > gpurun_out/r02d/gemm_only.txt 2>&1
timeout 600 python tools/inv_pitch_experiment.py > gpurun_out/r02d/inv_pitch.txt 2>&1
( time timeout 900 python -m pytest tests/test_gpu_lists_and_modules.py -m gpu -x -q ) > gpurun_out/r02d/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02d/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err
cat gpurun_out/r02d/gemm_only.txt gpurun_out/r02d/inv_pitch.txt; tail -5 gpurun_out/r02d/pytest.log
