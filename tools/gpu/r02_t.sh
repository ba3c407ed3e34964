#!/bin/bash
# session T: zpconv forward on the matrix cores -- parity tests, roofline tool, per-kernel times
mkdir -p gpurun_out/r02t
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zpconv" ) > gpurun_out/r02t/pytest.log 2>&1
tail -5 gpurun_out/r02t/pytest.log
timeout 600 python tools/zpconv_roofline.py 64 > gpurun_out/r02t/zpconv_roofline.txt 2>&1
cat gpurun_out/r02t/zpconv_roofline.txt
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02t/prof -o zp --output-format csv -- python /root/repo/tools/zpconv_roofline.py 64 > /root/repo/gpurun_out/r02t/prof.log 2>&1
cd /root/repo
head -8 gpurun_out/r02t/prof/zp_kernel_stats.csv | cut -c1-160
