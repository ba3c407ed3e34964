import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, bench
dev = torch.device('cuda:0')
for c in bench.other_configs(dev):
    print(c['name'][:50], round(c['ms_per_step'], 1), c['top_kernels_ms_per_step'], flush=True)
