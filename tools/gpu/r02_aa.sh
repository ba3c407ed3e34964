#!/bin/bash
# N > 1 path on a 1-GPU box: two ranks on device 0 over gloo (functional check of SyncBN moments, gradient all-reduce, pose all-gather)
mkdir -p gpurun_out/r02aa
cd /root/repo
EAP_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-other-configs > gpurun_out/r02aa/bench2.json 2> gpurun_out/r02aa/bench2.err
echo rc $?; tail -3 gpurun_out/r02aa/bench2.err; cut -c1-400 gpurun_out/r02aa/bench2.json
timeout 600 python bench.py --gpus 1 --steps 3 --warmup 1 --batch 2 --no-cpu-baseline --no-other-configs 2>/dev/null | cut -c1-300
