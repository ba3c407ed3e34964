#!/bin/bash
# tools/gpu/config3_profile.sh TAG: rocprofv3 kernel statistics of the config-3 composite step
set -u
export TMPDIR=/tmp
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o c3 --output-format csv -- python $R/tools/config3_profile.py 2 > $O/config3.json 2> $O/config3.err
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/config3_kernel_stats.csv
rm -rf $O/prof
cat $O/config3.json
head -50 $O/config3_kernel_stats.csv | cut -c1-170
