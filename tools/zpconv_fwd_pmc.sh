#!/bin/bash
# counters of the two zpconv forward matrix kernels (tools/zpconv_fwd_ab.py)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_$tag -o pmc --output-format csv -- python $R/tools/zpconv_fwd_ab.py > $O/pmc_$tag.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set)); dur = collections.defaultdict(float)
for f in glob.glob('$O/pmc_*/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        if 'zpconv_mfma' not in k: continue
        k = k[:k.index('(')] if '(' in k else k
        per[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k][r['Counter_Name']].add(r['Dispatch_Id'])
        if (f, r['Dispatch_Id']) not in seen:
            seen.add((f, r['Dispatch_Id'])); dur[(k, r['Counter_Name'])] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, c in per.items():
    print(k)
    for name, v in sorted(c.items()):
        n = len(cnt[k][name])
        extra = ''
        if name == 'FETCH_SIZE': extra = f'  = {v / n * 1024 * 2 / 1e9:.2f} GB (x2)'
        if name == 'WRITE_SIZE': extra = f'  = {v / n * 1024 / 1e9:.2f} GB'
        print(f'   {name:28s} {v / n:.4g} per launch  ({n} launches, {dur[(k, name)] / n / 1e6:.2f} ms each){extra}')
    if 'GRBM_GUI_ACTIVE' in c:
        cyc = c['GRBM_GUI_ACTIVE'] / 8
        print('   mfma_util', c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), ' clock GHz', cyc / dur[(k, 'GRBM_GUI_ACTIVE')])
    if 'TCC_HIT_sum' in c: print('   l2 hit', c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']))
PY
rm -rf $O/pmc_*/
