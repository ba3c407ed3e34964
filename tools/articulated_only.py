"""tools/articulated_only.py: the articulated-input configuration of bench.py alone (one rotation per rigid part), with the per-part dense
product (vgtk.so3conv.functional.DENSE_PARTS) on and off."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch, bench
import vgtk.so3conv.functional as L
dev = torch.device('cuda:0')
for on in (True, False, True):
    L.DENSE_PARTS = on
    r = bench.quick_run(dev, 8, 4096, part_poses=True, steps=4)
    print(f'per-part dense product {on}: {r["value"]:.1f} clouds/s, {r["ms_per_step"]:.1f} ms per step; regimes {[(x["channels"], x["regime"], x.get("parts")) for x in r["backward_regimes"]]}', flush=True)
    print('   ', r['top_kernels_ms_per_step'], flush=True)
