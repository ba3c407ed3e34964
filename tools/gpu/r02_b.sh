#!/bin/bash
# round 2, GPU session B: device-side inverse lists + new B1 modules; full GPU suite; bench.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02b
( time timeout 1500 python -m pytest tests/test_gpu_lists_and_modules.py tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q --durations=8 ) > gpurun_out/r02b/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02b/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err
timeout 600 python tools/torch_ops_profile.py > gpurun_out/r02b/torch_ops.txt 2>&1
tail -5 gpurun_out/r02b/pytest.log
