#!/bin/bash
# tools/gpu/zphot_ablation.sh TAG "BITS ..." -- the on-chip zpconv backward under its timing ablations (library rebuilt with
# ABLATION=1 on the box) + rocprofv3 kernel statistics of one un-ablated run (which kernel takes what).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
(make -C $R/equi-articulated-pose_amd/csrc clean -s; make -C $R/equi-articulated-pose_amd/csrc -j48 ABLATION=1 -s) > $O/build.log 2>&1
for d in $2; do echo "EAP_ZPHOT_DEBUG=$d"; EAP_ZPHOT_DEBUG=$d timeout 120 python $R/tools/zpconv_bwd_ab.py "on chip, 8" 2>&1 | tail -1; done > $O/ablation.txt 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o zp --output-format csv -- python $R/tools/zpconv_bwd_ab.py "on chip, 8" > $O/prof.log 2>&1
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv 2>/dev/null
rm -rf $O/prof
cat $O/ablation.txt; head -12 $O/kernel_stats.csv | cut -c1-150
