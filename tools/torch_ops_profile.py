"""tools/torch_ops_profile.py -- which torch ops launch the non-C-ABI kernels of one bench step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch, bench, synth_clouds
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
B = 8
model = bench.Backbone(4096).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
xyz, _, pose = synth_clouds.laptop_batch(0, B, 4096)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
def step():
    opt.zero_grad(set_to_none=True)
    f = model(xyz, pose)
    R, T = model.hypotheses(f)
    loss = f.square().mean() + R.square().mean() + T.square().mean()
    loss.backward()
    opt.step()
step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='cuda_time_total', row_limit=40, max_name_column_width=60, max_shapes_column_width=70))
