#!/bin/bash
# rocprofv3 kernel statistics of the config-3 composite step (tools/config3_profile.py) -> gpurun_out/$1/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o c3 --output-format csv -- python $R/tools/config3_profile.py 2 > $O/config3.log 2>&1
cd $R
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/config3_kernel_stats.csv \;
rm -rf $O/prof
grep "^{" $O/config3.log | tail -1 > $O/config3_line.json
head -25 $O/config3_kernel_stats.csv | cut -c1-200
python -c "
import json; d = json.load(open('$O/config3_line.json')); print({k: d[k] for k in ('value', 'ms_per_step', 'peak_memory_GB')})"
