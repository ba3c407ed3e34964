"""tools/config3_profile.py [steps]: the config-3 composite step (bench.config3_step) alone -- for
`rocprofv3 --kernel-trace --stats -- python tools/config3_profile.py` (which torch kernels sit between the hot-path launches)."""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
print(json.dumps(bench.config3_step(torch.device('cuda:0'), steps=steps, warmup=1)))
