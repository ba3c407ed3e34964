"""tools/zpconv_bwd_ab.py -- the native zpconv backward at the bench workload (8 x 4096, C = 64, layer-1 radius): rows on chip
(csrc/zpconv_bwd_hot.hip) against the product pipeline (csrc/zpconv_bwd.hip), whole batch and a slice of two clouds.
Median of 5 interleaved rounds; fractions of the 8 TB/s roofline in ALGORITHMIC bytes (idx + w + grad read, gfeats written)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.cuda.zpconv as Z
import vgtk.cuda.grouping as G

B, P, A, K, NN, C = 8, 4096, 60, 24, 64, 64
dev = torch.device('cuda:0')
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, B, P)[0]).to(dev)
radius = synth_clouds.backbone_layers(P)[1][2]
ball = G.ball_query(xyz, xyz, radius, NN)
print('referenced rows per cloud:', [int(torch.unique(ball[i]).numel()) for i in range(B)], flush=True)
idx = ball[:, :, None, None, :].expand(B, P, A, K, NN).contiguous()
w = torch.rand(B, P, A, K, NN, device=dev)
g = torch.randn(B, C, K, P, A, device=dev)
only = sys.argv[1:]
cases = [('on chip, 8 clouds', True, B), ('products, 8 clouds', False, B), ('on chip, 2 clouds', True, 2), ('products, 2 clouds', False, 2)]
if only:
    cases = [c for c in cases if any(o in c[0] for o in only)]
res = {c[0]: [] for c in cases}
for _ in range(6):
    for name, hot, nb in cases:
        Z.ON_CHIP_BACKWARD = hot
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); Z.inter_zpconv_backward(idx[:nb], w[:nb], g[:nb], P); e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1))
for name, hot, nb in cases:
    t = sorted(res[name][1:])
    byts = 4.0 * nb * (2.0 * P * A * K * NN + C * P * A + C * K * P * A)
    print(f'{name:22s}: median {t[2]:7.2f} ms = {byts / t[2] / 1e6 / 8000:.3f} of the HBM roofline ({byts / 1e9:.1f} GB algorithmic)', flush=True)
