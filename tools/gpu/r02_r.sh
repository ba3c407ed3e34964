#!/bin/bash
# session R: lists kernel with the VALU diet -- parity tests that go through it, the standalone launch, a short bench
mkdir -p gpurun_out/r02r
cd /root/repo
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q ) > gpurun_out/r02r/pytest.log 2>&1
tail -5 gpurun_out/r02r/pytest.log
timeout 600 python tools/inv_locality_experiment.py > gpurun_out/r02r/inv_locality.txt 2>&1
cat gpurun_out/r02r/inv_locality.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > gpurun_out/r02r/bench.json 2> gpurun_out/r02r/bench.err
python - <<'PY'
import json
b = json.load(open('gpurun_out/r02r/bench.json'))
print(b['value'], b['ms_per_step'], b['roofline']['frac'])
for k, v in list(b['kernels'].items())[:6]: print(k, v)
PY
