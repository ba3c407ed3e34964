#!/bin/bash
mkdir -p gpurun_out/r02u
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zpconv" ) > gpurun_out/r02u/pytest.log 2>&1
tail -3 gpurun_out/r02u/pytest.log
for d in 0 1 2 4 7; do echo "EAP_ZP_DEBUG=$d"; EAP_ZP_DEBUG=$d timeout 300 python tools/zpconv_roofline.py 64 2>&1 | grep forward; done > gpurun_out/r02u/ablate.txt 2>&1
cat gpurun_out/r02u/ablate.txt
