#!/bin/bash
# round 2, GPU session L: second-generation lists kernel (matrix waves + loader waves): parity, then A/B bench
set -u
mkdir -p gpurun_out/r02l
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02l/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02l/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r02l/bench_v2.json 2> gpurun_out/r02l/bench_v2.err
EAP_LISTS_V2=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/r02l/bench_v1.json 2> gpurun_out/r02l/bench_v1.err
tail -5 gpurun_out/r02l/pytest.log
python - <<'PY'
import json
for v in ('v2','v1'):
    try:
        d=json.loads(open(f'gpurun_out/r02l/bench_{v}.json').read())
        print(v, d['value'], d['ms_per_step'], {k:(round(x['ms_per_step'],2), x['tflops'] and round(x['tflops'],1)) for k,x in list(d['kernels'].items())[:5]})
    except Exception as e: print(v, 'failed', e)
PY
