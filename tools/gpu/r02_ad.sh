#!/bin/bash
mkdir -p gpurun_out/r02ad
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zpconv" ) > gpurun_out/r02ad/pytest.log 2>&1; tail -3 gpurun_out/r02ad/pytest.log; grep -n "^E  " gpurun_out/r02ad/pytest.log | head -5
