#!/bin/bash
mkdir -p gpurun_out/r02ad
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zpconv" ) > gpurun_out/r02ad/pytest.log 2>&1; tail -2 gpurun_out/r02ad/pytest.log
timeout 600 python - <<'PY' 2>&1 | tail -3
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/equi-articulated-pose_amd')
import bench
for _ in range(2):
    z = bench.zpconv_roofline(torch.device('cuda:0'), 4096)
    print('zpconv fwd', round(z['ms'], 2), round(z['frac'], 4), 'bwd', round(z['backward']['ms'], 2), round(z['backward']['frac'], 4))
PY
