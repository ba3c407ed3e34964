#!/bin/bash
# counters of the split contraction kernel alone (tools/gemm_only.py 8 first): matrix-pipe busy, clock, wait states
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$tag -o pmc --output-format csv -- python $R/tools/gemm_only.py 2 first > $O/pmc_$tag.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int); dur = collections.defaultdict(float)
for f in glob.glob('$O/pmc_*/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '')
        if 'gemm' not in k: continue
        k = k[k.index('gemm'):][:44]
        per[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (f, r['Dispatch_Id']) not in seen:
            seen.add((f, r['Dispatch_Id'])); dur[(k, r['Counter_Name'])] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, c in per.items():
    print(k)
    for name, v in sorted(c.items()):
        print(f'   {name:28s} {v:.4g}   (ns {dur[(k, name)]:.4g})')
    if 'GRBM_GUI_ACTIVE' in c:
        cyc = c['GRBM_GUI_ACTIVE'] / 8
        print('   mfma_util', c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), ' clock GHz', cyc / dur[(k, 'GRBM_GUI_ACTIVE')])
PY
rm -rf $O/pmc_*/
