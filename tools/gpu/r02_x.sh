#!/bin/bash
mkdir -p gpurun_out/r02x
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_lists_and_modules.py -m gpu -x -q ) > gpurun_out/r02x/pytest.log 2>&1
tail -6 gpurun_out/r02x/pytest.log
