"""tools/config3_torch_profile.py: torch.profiler over ONE steady-state config-3 composite step -- which torch operators
(by input shapes and Python call site) account for the device time that is not in the C-ABI kernels."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from torch.profiler import profile, ProfilerActivity
import bench  # noqa: F401  (puts the package on sys.path)
import synth_clouds
import config3_step as C3

dev = torch.device('cuda:0')
batch, points = 16, 4096
torch.manual_seed(2913)
model = C3.Config3Model(points).to(dev)
opt = torch.optim.Adam(model.trained_parameters(), lr=1e-4)
xyz_np, _, pose_np = synth_clouds.laptop_batch(0, batch, points)
xyz, pose = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(pose_np).to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _ = model(xyz, pose)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=45, max_name_column_width=60, max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by='self_cuda_time_total', row_limit=40, max_name_column_width=50, max_src_column_width=110))
