#!/bin/bash
# round 2, GPU session C: DMA-ring GEMM -- parity, isolated timing vs the older kernel and the library, bench.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
( time timeout 900 python -m pytest tests/test_gpu_lists_and_modules.py tests/test_gpu_parity.py -m gpu -x -q -k "gemm or golden or end_to_end" ) > gpurun_out/r02c/pytest_gemm.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02c/pytest_gemm.log
timeout 600 python tools/gemm_only.py > gpurun_out/r02c/gemm_only.txt 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err
EAP_LIBRARY_GEMMS=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02c/bench_library.json 2> gpurun_out/r02c/bench_library.err
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02c/pytest_all.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02c/pytest_all.log
tail -3 gpurun_out/r02c/pytest_gemm.log; cat gpurun_out/r02c/gemm_only.txt; tail -3 gpurun_out/r02c/pytest_all.log
