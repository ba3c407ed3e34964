#!/bin/bash
# tools/gpu/extras.sh TAG -- the bench legs the driver's default command does not run: the separable glb_backbone forward, the
# N > 1 code path as a functional check on ONE device (2 ranks over gloo: SyncBN moments, hook-driven gradient all-reduce,
# pose all-gather), rocprofv3 kernel statistics of the config-3 composite step, the split kernel's layouts in isolation
set -u
export TMPDIR=/tmp
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 600 python bench.py --separable --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/bench_separable_fwd.json 2> $O/bench_separable.err
EAP_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --batch 4 --no-cpu-baseline > $O/bench_two_ranks_one_device_gloo.json 2> $O/bench_two_ranks.err
python tools/split_modes_timing.py 4 2>&1 | grep -v amdgpu.ids > $O/split_modes_timing.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o c3 --output-format csv -- python $R/tools/config3_profile.py 2 > $O/config3.json 2> $O/config3.err
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/config3_kernel_stats.csv
rm -rf $O/prof
cd $R
cut -c1-400 $O/bench_separable_fwd.json; echo; cut -c1-400 $O/bench_two_ranks_one_device_gloo.json; echo; tail -3 $O/bench_two_ranks.err; cat $O/config3.json | cut -c1-600
