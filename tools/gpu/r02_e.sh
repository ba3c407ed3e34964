#!/bin/bash
# round 2, GPU session E: counters of the three GEMM implementations on the same contraction.
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02e
cd /tmp
for impl in dma lib old; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-30)
    timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/r02e/${impl}_$tag -o pmc --output-format csv -- python $R/tools/gemm_pmc_target.py $impl > $R/gpurun_out/r02e/${impl}_$tag.log 2>&1
  done
  python $R/tools/pmc_summary.py $R/gpurun_out/r02e/${impl}_* > $R/gpurun_out/r02e/summary_$impl.json 2>&1
done
cat $R/gpurun_out/r02e/summary_*.json | head -150
