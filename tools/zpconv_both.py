"""tools/zpconv_both.py -- the native zpconv forward and backward at the bench workload (8 x 4096, C = 64, layer-1 radius),
as bench.py's zpconv_roofline times them: 2 warm-up + 5 timed calls each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import bench
z = bench.zpconv_roofline(torch.device('cuda:0'), 4096)
print(f"forward  {z['ms']:.2f} ms = {z['frac']:.3f} of the HBM roofline; backward {z['backward']['ms']:.2f} ms = {z['backward']['frac']:.3f}")
