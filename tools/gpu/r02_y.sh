#!/bin/bash
# session Y: permuted-pose kernels after the rotated-offset / packed-weight change
mkdir -p gpurun_out/r02y
cd /root/repo
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q ) > gpurun_out/r02y/pytest.log 2>&1
tail -4 gpurun_out/r02y/pytest.log
timeout 600 python tools/permuted_pose_time.py > gpurun_out/r02y/permuted.txt 2>&1; cat gpurun_out/r02y/permuted.txt
