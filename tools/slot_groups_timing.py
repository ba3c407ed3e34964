"""tools/slot_groups_timing.py: the two pose heads of config 3 (16 x 4096 points, 512 channels, width 256) forward + backward,
masked form on the full clouds (slots x P points of work) against the compacted slot groups (P points per cloud), for a
balanced and for a degenerate assignment of the points to the two slots."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import numpy as np
import torch
import vgtk.so3conv as sptk
import vgtk.so3conv.functional as L

dev = torch.device('cuda:0')
B, C, P, A, S = 16, 512, 4096, 60, 2
torch.manual_seed(0)
heads = [sptk.SO3OutBlockRTWithMaskSep({'dim_in': C, 'mlp': [256, 256], 'kanchor': A, 'temperature': 3.0}, norm=1, pooling_method='mean',
                                       pred_axis=True, pred_central_points=True).to(dev) for _ in range(S)]
feats = torch.randn(B, C, P, A, device=dev, requires_grad=True)
xyz = torch.randn(B, 3, P, device=dev) * 0.3
anchors = torch.from_numpy(np.ascontiguousarray(L.get_anchors(A))).to(dev)


def run(form, labels):
    if form == 'masked':
        outs = []
        for s_, head in enumerate(heads):
            member = labels == s_
            outs.append(sptk.pose_head_over_subsets(head, feats, xyz, member | (member.sum(1, keepdim=True) == 0), anchors))
    else:
        outs = sptk.pose_head_over_slot_groups(heads, feats, xyz, labels, anchors)
    loss = sum(o['T'].square().mean() + o['R'].square().mean() for o in outs)
    loss.backward()
    feats.grad = None


for name, frac in (('balanced (45-55 % of the points in slot 0)', (0.45, 0.55)), ('degenerate (0-100 %)', (0.0, 1.0))):
    share = torch.linspace(frac[0], frac[1], B, device=dev).view(B, 1)
    labels = (torch.rand(B, P, device=dev) > share).long()
    for form in ('masked', 'groups'):
        run(form, labels); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(form, labels); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f'{name:45s} {form:7s} {sorted(ts)[1]:8.1f} ms  (peak {torch.cuda.max_memory_allocated() / 2**30:.0f} GB)', flush=True)
