#!/bin/bash
# session W: full GPU suite + bench on the build with the zpconv matrix paths
mkdir -p gpurun_out/r02w
cd /root/repo
( timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r02w/pytest.log 2>&1
tail -3 gpurun_out/r02w/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02w/smoke.log 2>&1; tail -1 gpurun_out/r02w/smoke.log
timeout 1200 python bench.py --steps 10 --warmup 3 > gpurun_out/r02w/bench.json 2> gpurun_out/r02w/bench.err
tail -3 gpurun_out/r02w/bench.err
python - <<'PY'
import json
b = json.load(open('gpurun_out/r02w/bench.json'))
print(b['value'], b['ms_per_step'], b['roofline']['entry'], b['roofline']['frac'])
for k, v in list(b['kernels'].items())[:4]: print(k, v)
z = b['zpconv_roofline']; print('zpconv fwd', z['ms'], z['frac'], 'bwd', z['backward']['ms'], z['backward']['frac'])
PY
