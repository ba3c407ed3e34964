#!/bin/bash
mkdir -p gpurun_out/r02v
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "zpconv" ) > gpurun_out/r02v/pytest.log 2>&1
tail -3 gpurun_out/r02v/pytest.log
timeout 600 python tools/zpconv_roofline.py 64 > gpurun_out/r02v/zpconv_roofline.txt 2>&1
cat gpurun_out/r02v/zpconv_roofline.txt
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02v/prof -o zp --output-format csv -- python /root/repo/tools/zpconv_roofline.py 64 > /root/repo/gpurun_out/r02v/prof.log 2>&1
cd /root/repo
grep -v "at::native" gpurun_out/r02v/prof/zp_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2, $3, $4}' | head -9
