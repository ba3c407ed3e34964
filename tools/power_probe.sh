#!/bin/bash
# tools/power_probe.sh: package power and shader clock (rocm-smi, sampled) while one kernel family runs in a loop -- the split
# contraction, the fp32-MFMA contraction and the backward grouping (is the split kernel's 1.58 GHz a power limit?)
R=$GRAFT_REPO_ROOT
cd $R
for what in split2 split3 fp32 grouping; do
  python - "$what" <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'equi-articulated-pose_amd'))
import torch
from vgtk import _hip
what = sys.argv[1]
dev = torch.device('cuda:0')
B, PA = 4, 4096 * 60
W = torch.randn(512, 3072, device=dev) * 0.05
XT = torch.randn(B, PA, 3072, device=dev)
y = torch.empty(B, 512, PA, device=dev)
if what == 'grouping':
    import bench
    import synth_clouds
    model = bench.Backbone(4096).to(dev)
    xyz, _, pose = synth_clouds.laptop_batch(0, 8, 4096)
    xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
_hip.SPLIT_BF16_CONTRACTION = what != 'fp32'
_hip.SPLIT_PLANES = 3 if what == 'split3' else 2          # split2: two fp16 planes (default), split3: three bf16 planes
bound = (_hip.absmax_rows(XT, B, PA, 3072, 3072, PA * 3072), 1, 1.0) if what == 'split2' else None     # (the pass over B stays out of the loop)
t0 = time.time()
while time.time() - t0 < 6.0:
    if what == 'grouping':
        model(xyz, pose).square().mean().backward()
    else:
        for _ in range(10):
            _hip.gemm(0, 1, 512, PA, 3072, W, 3072, 0, XT, 3072, PA * 3072, y, PA, 512 * PA, B, b_bound=bound)
    torch.cuda.synchronize()
PY
  pid=$!
  sleep 3.5
  echo "== $what"
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -3; sleep 0.4; done
  wait $pid
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -2
