#!/bin/bash
# tools/gpu/split_ablation.sh TAG: builds the library with ABLATION=1 in a scratch copy on the GPU box and runs
# tools/gemm_split_ablation.py for two and for three planes
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
rm -rf /tmp/abl && mkdir -p /tmp/abl && cp -r $R/equi-articulated-pose_amd $R/include $R/tools $R/oracle /tmp/abl/ 2>/dev/null
cp $R/*.py /tmp/abl/ 2>/dev/null
make -C /tmp/abl/equi-articulated-pose_amd/csrc clean > /dev/null
make -C /tmp/abl/equi-articulated-pose_amd/csrc -j32 -s ABLATION=1 > $O/build.log 2>&1
for p in 2 3; do
  echo "planes $p" >> $O/ablation.txt
  timeout 300 python /tmp/abl/tools/gemm_split_ablation.py $p 2>&1 | grep -v amdgpu.ids >> $O/ablation.txt
done
cat $O/ablation.txt
