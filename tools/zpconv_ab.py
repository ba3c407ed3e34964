"""tools/zpconv_ab.py BITS... -- zpconv forward at the bench workload (8 x 4096, C = 64, layer-1 radius) under EAP_ZP_DEBUG
variants (library built with ABLATION=1).  Median of 5 interleaved rounds."""
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
idx = ball[:, :, None, None, :].expand(B, P, A, K, NN).contiguous()
w = torch.rand(B, P, A, K, NN, device=dev)
feats = torch.randn(B, C, P, A, device=dev)
byts = 4.0 * B * (2.0 * P * A * K * NN + C * P * A + C * K * P * A)
variants = [int(v) for v in sys.argv[1:]] or [0]
res = {v: [] for v in variants}
for _ in range(6):
    for v in variants:
        os.environ['EAP_ZP_DEBUG'] = str(v)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); Z.inter_zpconv_forward(idx, w, feats); e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1))
for v in variants:
    t = sorted(res[v][1:])
    print(f'EAP_ZP_DEBUG={v:2d}: median {t[2]:6.2f} ms = {byts / t[2] / 1e6 / 8000:.3f} of the HBM roofline', flush=True)
