#!/bin/bash
mkdir -p gpurun_out/r02ab
cd /root/repo
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q ) > gpurun_out/r02ab/pytest.log 2>&1; tail -2 gpurun_out/r02ab/pytest.log
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export EAP_LISTS_NOREV=1; else unset EAP_LISTS_NOREV; fi
  echo "NOREV=$v"; timeout 300 python tools/inv_locality_experiment.py 2>&1 | grep "real"
done > gpurun_out/r02ab/rev.txt 2>&1
cat gpurun_out/r02ab/rev.txt
