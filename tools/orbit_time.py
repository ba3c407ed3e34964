"""tools/orbit_time.py -- time of the orbit-selection distances at the reference's sizes (B=8, S=2, A=60, M=256, N=4096)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
from extensions.chamfer_dist import orbit_reconstruction_distances
dev = torch.device('cuda:0')
b, s, a, m, n = 8, 2, 60, 256, 4096
recon = (torch.randn(b, s, a, m, 3, device=dev) * 0.3).requires_grad_(True)
ori = torch.randn(b, 3, n, device=dev) * 0.3
labels = torch.nn.functional.one_hot(torch.randint(0, s, (b, n), device=dev), s).float()
for it in range(3):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    out = orbit_reconstruction_distances(recon, ori, labels)
    e1.record()
    (out[0].sum() + out[1].sum() + out[2].sum()).backward()
    e2.record(); torch.cuda.synchronize()
print(f'forward {e0.elapsed_time(e1):.2f} ms, backward {e1.elapsed_time(e2):.2f} ms; the reference tensor [B,S,A,M,N] would be {b*s*a*m*n*4/1e9:.1f} GB; '
      f'distance evaluations {2*b*s*a*m*n/1e9:.1f} G')
