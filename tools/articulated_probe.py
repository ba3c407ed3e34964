"""tools/articulated_probe.py: the bench's articulated-input configuration (one rotation per rigid part) with every C-ABI entry's
time per step, for EAP_SPLIT_PLANES=2 / 3 A/B runs."""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from vgtk import _hip

dev = torch.device('cuda:0')
orig = bench.summarize_kernels
full = {}


def keep_all(records):
    kern, shapes = orig(records)
    full.update({n: round(k['ms'], 2) for n, k in kern.items()})
    return kern, shapes


bench.summarize_kernels = keep_all
part = len(sys.argv) < 2 or sys.argv[1] != 'identity'
r = bench.quick_run(dev, 8, 4096, part_poses=part, steps=3)
print(json.dumps({'planes': _hip.SPLIT_PLANES, 'poses': r['poses'], 'ms_per_step': r['ms_per_step'], 'sum_of_entries_ms_per_step': sum(full.values()) / 3,
                  'entries_ms_per_3_steps': dict(sorted(full.items(), key=lambda kv: -kv[1]))}))
