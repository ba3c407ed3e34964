#!/bin/bash
# tools/gpu/experiments_build_run.sh SCRIPT [ARGS]: builds the library with EXPERIMENTS=1 in a scratch copy on the GPU box and runs tools/SCRIPT there
set -u
R=$GRAFT_REPO_ROOT
rm -rf /tmp/exp && mkdir -p /tmp/exp && cp -r $R/equi-articulated-pose_amd $R/include $R/tools $R/oracle /tmp/exp/ 2>/dev/null
cp $R/*.py /tmp/exp/ 2>/dev/null
make -C /tmp/exp/equi-articulated-pose_amd/csrc clean > /dev/null
make -C /tmp/exp/equi-articulated-pose_amd/csrc -j32 -s EXPERIMENTS=1 2>&1 | grep -i "error" | head
S=$1; shift
timeout 600 python /tmp/exp/tools/$S "$@" 2>&1 | grep -v amdgpu.ids
