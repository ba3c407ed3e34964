#!/bin/bash
# round 2, GPU session G: one-wave-per-SIMD DMA GEMM (parity + timing + counters), heads kernels, bench.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
( time timeout 900 python -m pytest tests/test_gpu_lists_and_modules.py tests/test_gpu_parity.py -m gpu -x -q -k "gemm or golden or end_to_end or heads or attention or slot_masked or strided" ) > gpurun_out/r02g/pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02g/pytest.log
timeout 600 python tools/gemm_only.py > gpurun_out/r02g/gemm_only.txt 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r02g/bench.json 2> gpurun_out/r02g/bench.err
cd /tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-30)
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/r02g/dma_$tag -o pmc --output-format csv -- python $R/tools/gemm_pmc_target.py dma > $R/gpurun_out/r02g/dma_$tag.log 2>&1
done
python $R/tools/pmc_summary.py $R/gpurun_out/r02g/dma_* > $R/gpurun_out/r02g/summary_dma.json 2>&1
cd $R
tail -4 gpurun_out/r02g/pytest.log; cat gpurun_out/r02g/gemm_only.txt; cat gpurun_out/r02g/summary_dma.json | head -60
