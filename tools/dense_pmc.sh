#!/bin/bash
# counters of the dense product kernel alone (tools/gpu/dense_time.py): matrix-pipe busy, clock, wave states, LDS
export TMPDIR=/tmp FORMS=1 REPS=2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$tag -o pmc --output-format csv -- python $R/tools/gpu/dense_time.py > $O/pmc_$tag.log 2>&1
done
cd $R
python tools/pmc_summary.py $O/pmc_* > $O/dense_pmc_summary.json
rm -rf $O/pmc_*/
python - <<PY
import json
d = json.load(open('$O/dense_pmc_summary.json'))
for k, e in d.items():
    if 'kc_gemm' in k or 'dense' in k:
        print(k, {n: (round(v, 4) if isinstance(v, float) else v) for n, v in e.items() if n.startswith('frac_') or n in ('avg_ms', 'launches', 'mfma_util', 'shader_clock_ghz', 'SQ_LDS_BANK_CONFLICT', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM')})
PY
