"""tools/gpu/zpconv_verify_ab.py: does the asynchronous re-check of the zpconv backward's remembered verdict (vgtk/cuda/zpconv.py
_check_pending_verdicts: a 32-byte device-to-host copy + an event per call) cost launch time?  The backward at the bench workload with and
without it, interleaved, medians of event-timed groups."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import numpy as np
import torch
import synth_clouds
import vgtk.cuda.zpconv as Z
import vgtk.cuda.grouping as G
dev = torch.device('cuda:0')
clouds, points, channels, NN, NA, KS = 8, 4096, 64, 64, 60, 24
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, clouds, points)[0]).to(dev)
ball = G.ball_query(xyz, xyz, synth_clouds.backbone_layers(points)[1][2], NN)
idx = ball[:, :, None, None, :].expand(clouds, points, NA, KS, NN).contiguous()
w = torch.rand(clouds, points, NA, KS, NN, device=dev)
grad = torch.randn(clouds, channels, KS, points, NA, device=dev)
feats = torch.randn(clouds, channels, points, NA, device=dev)
byts = 4.0 * clouds * (2.0 * points * NA * KS * NN + channels * points * NA + channels * KS * points * NA)


def group(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


bwd = lambda: Z.inter_zpconv_backward(idx, w, grad, points)
fwd = lambda: Z.inter_zpconv_forward(idx, w, feats)
for _ in range(3):
    bwd(); fwd()
res = {True: [], False: [], 'fwd': []}
for i in range(8):
    for v in (True, False):
        Z.VERIFY_REMEMBERED = v
        res[v].append(group(bwd))
    res['fwd'].append(group(fwd))
Z.VERIFY_REMEMBERED = True
for k, v in res.items():
    v = sorted(v)
    print(f'{"forward" if k == "fwd" else ("backward, re-check " + ("on" if k else "off"))}: median {v[len(v) // 2]:.2f} ms = {byts / v[len(v) // 2] / 1e6 / 8000:.3f} of the HBM roofline; groups {[round(x, 2) for x in v]}')
