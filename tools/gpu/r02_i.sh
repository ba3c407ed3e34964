#!/bin/bash
set -u
mkdir -p gpurun_out/r02i
timeout 300 python -m pytest tests/test_gpu_lists_and_modules.py tests/test_gpu_parity.py -m gpu -x -q -k "gemm or golden or end_to_end" > gpurun_out/r02i/pytest.log 2>&1
timeout 600 python tools/gemm_only.py > gpurun_out/r02i/gemm_only.txt 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err
tail -3 gpurun_out/r02i/pytest.log; cat gpurun_out/r02i/gemm_only.txt
