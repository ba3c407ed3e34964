#!/bin/bash
mkdir -p gpurun_out/r02ae
cd /root/repo
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q ) > gpurun_out/r02ae/pytest.log 2>&1; tail -2 gpurun_out/r02ae/pytest.log
for r in 1 2; do timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(round(b['value'],2), round(b['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in list(b['kernels'].items())[:3]})"; done > gpurun_out/r02ae/b.txt 2>&1
cat gpurun_out/r02ae/b.txt
