#!/bin/bash
# session S: wide GEMM tile for M <= 128, full GPU suite on the new grouping kernel, bench
mkdir -p gpurun_out/r02s
cd /root/repo
timeout 600 python tools/gemm_only.py 8 > gpurun_out/r02s/gemm_only.txt 2>&1
head -12 gpurun_out/r02s/gemm_only.txt
( timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02s/pytest.log 2>&1
tail -4 gpurun_out/r02s/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r02s/bench.json 2> gpurun_out/r02s/bench.err
python - <<'PY'
import json
b = json.load(open('gpurun_out/r02s/bench.json'))
print(b['value'], b['ms_per_step'], b['roofline']['entry'], b['roofline']['frac'])
for k, v in list(b['kernels'].items())[:8]: print(k, v)
for o in b['other_configs']: print(o['name'], o['value'], o['ms_per_step'])
PY
