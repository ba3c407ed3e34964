#!/bin/bash
mkdir -p gpurun_out/r02x
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_lists_and_modules.py -m gpu -x -q -k "pose_head" ) > gpurun_out/r02x/pytest.log 2>&1
tail -6 gpurun_out/r02x/pytest.log; grep -n "^E  *Assertion\|^E  *assert\|Error" gpurun_out/r02x/pytest.log | head -5
