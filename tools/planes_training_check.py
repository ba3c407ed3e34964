"""tools/planes_training_check.py [steps]: the bench's training step (8 x 4096, fwd + bwd + Adam) run for a few steps from the same
initial weights with the forward contraction on two fp16 planes, three bf16 planes and the fp32 matrix pipe -- losses per step and
the distance of the trained weights from the fp32-pipe run (how much of a training trajectory the arithmetic of the contraction moves)."""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
import synth_clouds
from vgtk import _hip

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device('cuda:0')
xyz_np, _, pose_np = synth_clouds.laptop_batch(0, 8, 4096)
xyz, pose = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(pose_np).to(dev)
out = {}
for name, (planes, split) in {'fp32 pipe': (3, False), 'three bf16 planes': (3, True), 'two fp16 planes': (2, True), 'fp32 pipe again': (3, False)}.items():
    _hip.SPLIT_PLANES, _hip.SPLIT_BF16_CONTRACTION = planes, split
    torch.manual_seed(2913)
    model = bench.Backbone(4096).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        feats = model(xyz, pose)
        loss = bench.StandInLoss.apply(feats, model.pose_head.weight, model.pose_head.bias)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    out[name] = (losses, [p.detach().double().clone() for p in model.parameters()])
    del model, opt
_hip.SPLIT_PLANES, _hip.SPLIT_BF16_CONTRACTION = 2, True
ref_l, ref_p = out['fp32 pipe']
for name, (losses, params) in out.items():
    num = sum(((a - b) ** 2).sum().item() for a, b in zip(params, ref_p)) ** 0.5
    den = sum((b ** 2).sum().item() for b in ref_p) ** 0.5
    print(json.dumps({'contraction': name, 'losses': [round(l, 9) for l in losses],
                      'max_rel_loss_difference_to_fp32_pipe': max(abs(a - b) / abs(b) for a, b in zip(losses, ref_l)),
                      'weights_after_training_rel_l2_distance_to_fp32_pipe': num / den}))
