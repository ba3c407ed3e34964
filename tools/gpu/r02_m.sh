#!/bin/bash
set -u
mkdir -p gpurun_out/r02m
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q -k "entry_list or golden or 4096_layers_identity or 8192 or full_size or end_to_end or equivariance" ) > gpurun_out/r02m/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02m/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r02m/bench_v2.json 2> gpurun_out/r02m/bench_v2.err
tail -4 gpurun_out/r02m/pytest.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02m/bench_v2.json').read())
print(d['value'], d['ms_per_step'], {k:(round(x['ms_per_step'],2), x['tflops'] and round(x['tflops'],1)) for k,x in list(d['kernels'].items())[:5]})
PY
