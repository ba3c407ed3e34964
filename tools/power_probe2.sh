#!/bin/bash
# tools/power_probe2.sh: package power / shader clock while the deepest layer's BACKWARD (inverse-list grouping + small GEMMs) or its
# forward grouping + split contraction runs in a loop
R=$GRAFT_REPO_ROOT
cd $R
for what in backward forward; do
  python - "$what" <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'equi-articulated-pose_amd'))
import torch, synth_clouds
import vgtk.so3conv as sptk, vgtk.spconv as zptk
import vgtk.so3conv.functional as L
what = sys.argv[1]
B, P = 8, 4096
dev = torch.device('cuda:0')
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
c, o, r, s = synth_clouds.backbone_layers(P)[2]
conv = sptk.InterSO3PoseConv(c, o, 1, 1, r, s, 64, kanchor=60, permute_modes=1).to(dev)
f = torch.randn(B, c, P, 60, device=dev).requires_grad_(True)
gy = torch.randn(B, o, P, 60, device=dev)
y = conv(zptk.SphericalPointCloudPose(xyz, f, None, pose))[3].feats
t0 = time.time()
while time.time() - t0 < 6.0:
    for _ in range(5):
        if what == 'backward':
            torch.autograd.grad(y, [f], gy, retain_graph=True)
        else:
            with torch.no_grad():
                conv(zptk.SphericalPointCloudPose(xyz, f, None, pose))
    torch.cuda.synchronize()
PY
  pid=$!
  sleep 3.5
  echo "== deepest layer, $what"
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | head -3; sleep 0.4; done
  wait $pid
done
