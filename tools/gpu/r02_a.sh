#!/bin/bash
# round 2, GPU session A: full GPU test suite (with the new bench-shape parity tests), counter list,
# kernel trace + MFMA-utilisation counters of one bench step.
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=25 ) > gpurun_out/r02a/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02a/pytest.log
rocprofv3 -L > gpurun_out/r02a/counters.txt 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
EAP_X_LAYOUT=blocked EAP_LIBRARY_GEMMS=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02a/bench_own_gemm.json 2> gpurun_out/r02a/bench_own_gemm.err
cd /tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  EAP_X_LAYOUT=blocked EAP_LIBRARY_GEMMS=0 timeout 600 rocprofv3 --kernel-trace --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02a/pmc_$tag.log 2>&1
done
ls -R $GRAFT_REPO_ROOT/gpurun_out/r02a | head -50
tail -5 $GRAFT_REPO_ROOT/gpurun_out/r02a/pytest.log
