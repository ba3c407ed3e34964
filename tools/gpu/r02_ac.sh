#!/bin/bash
mkdir -p gpurun_out/r02ac
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or entry_list or layer" ) > gpurun_out/r02ac/pytest.log 2>&1; tail -2 gpurun_out/r02ac/pytest.log
for r in 1 2; do timeout 300 python bench.py --steps 4 --warmup 2 --fwd-only --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; b=json.loads(sys.stdin.read()); print(round(b['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in list(b['kernels'].items())[:3]})"; done > gpurun_out/r02ac/fwd.txt 2>&1
cat gpurun_out/r02ac/fwd.txt
