"""tools/zpconv_roofline.py -- HBM roofline of the standalone native zpconv op:
bytes = 4*B*[2*PAKN + C*Q*A + C*K*PA] (idx + w read once, feats, out written once)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.cuda.zpconv as Z
import vgtk.cuda.grouping as G

B, P, A, K, NN, C = 2, 4096, 60, 24, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda:0')
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, B, P)[0]).to(dev)
for radius in (0.08, 0.32):
    ball = G.ball_query(xyz, xyz, radius, NN)
    idx = ball[:, :, None, None, :].expand(B, P, A, K, NN).contiguous()
    w = torch.rand(B, P, A, K, NN, device=dev)
    feats = torch.randn(B, C, P, A, device=dev)
    byts = 4.0 * B * (2.0 * P * A * K * NN + C * P * A + C * K * P * A)
    for name, fn in (('inter_zpconv_forward', lambda: Z.inter_zpconv_forward(idx, w, feats)),):
        out = fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print(f'{name} r={radius} C={C}: {ms:.2f} ms, {byts / 1e9:.2f} GB algorithmic -> {byts / ms / 1e6:.0f} GB/s = {byts / ms / 1e6 / 8000:.1%} of 8 TB/s', flush=True)
    g = torch.randn_like(out)
    Z.inter_zpconv_backward(idx, w, g, P); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); Z.inter_zpconv_backward(idx, w, g, P); e1.record(); torch.cuda.synchronize()
    print(f'inter_zpconv_backward r={radius}: {e0.elapsed_time(e1):.2f} ms', flush=True)
    del idx, w, feats, out, g
