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
    loss = bench.StandInLoss.apply(f, model.pose_head.weight, model.pose_head.bias)
    loss.backward()
    opt.step()
step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.self_device_time_total > 0 and not e.key.startswith('void (anonymous') and 'so3_' not in e.key and 'gemm_f32' not in e.key and 'bn_' not in e.key]
rows.sort(key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print('non-C-ABI device time per step: %.2f ms' % (tot / 1e3))
for e in rows[:28]:
    print('%8.3f ms  x%-4d %s' % (e.self_device_time_total / 1e3, e.count, e.key[:110]))
