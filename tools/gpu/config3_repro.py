"""tools/gpu/config3_repro.py: the forward of the config-3 composite N times in one process: are the losses bit-equal?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench  # noqa: F401
import synth_clouds
import config3_step as C3
dev = torch.device('cuda:0')
P, B = 4096, 16
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
torch.manual_seed(2913)
model = C3.Config3Model(P).to(dev)
vals = []
with torch.no_grad():
    for i in range(int(os.environ.get('N', 8))):
        out = model(xyz, pose)
        vals.append((float(out[0]), float(out[1]['scores'].double().sum()), float(out[1]['recon'].double().sum())))
        del out
print(vals)
print('all equal:', all(v == vals[0] for v in vals))
