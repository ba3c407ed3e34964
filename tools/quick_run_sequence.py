"""tools/quick_run_sequence.py: bench.quick_run legs in the bench's order -- is a leg's step time independent of the legs before it?"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
dev = torch.device('cuda:0')
for pts, kw in ((4096, dict(fwd_only=True)), (8192, dict(partial=True)), (8192, dict()), (4096, dict(plan_points=512)), (4096, dict(part_poses=True)), (4096, dict(part_poses=True))):
    r = bench.quick_run(dev, 8, pts, **kw)
    print(pts, kw, round(r['ms_per_step'], 1), r['top_kernels_ms_per_step'], f'reserved {torch.cuda.memory_reserved() / 2**30:.1f} GB', flush=True)
