"""tools/gemm_split_ablation.py [planes] -- where the time of the split contraction (csrc/gemm_bf16x3.hip) goes: the deepest
layer's forward contraction (B = 8) with parts switched off (EAP_GEMM_SPLIT_DEBUG; library built with ABLATION=1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
from vgtk import _hip
B, O, CK, PA = 8, 512, 3072, 4096 * 60
PLANES = int(sys.argv[1]) if len(sys.argv) > 1 else 2          # 2: two fp16 planes (three products), 3: three bf16 planes (six)
_hip.SPLIT_PLANES = PLANES
dev = torch.device('cuda:0')
W = torch.randn(O, CK, device=dev); XT = torch.randn(B, PA, CK, device=dev); Y = torch.empty(B, O, PA, device=dev)
BOUND = (_hip.absmax_rows(XT, B, PA, CK, CK, PA * CK), 1, 1.0) if PLANES == 2 else None      # (the pass over B stays outside the timed launches)
CASES = [('full kernel', 0), ('no global loads in the loop', 1), ('no split / park', 2), ('no loads, no split', 3), ('no fragment reads', 4),
         ('MFMAs + barrier', 7), ('MFMAs only', 15)]
res = {k: [] for k, _ in CASES}
for _ in range(6):
    for k, bits in CASES:
        os.environ['EAP_GEMM_SPLIT_DEBUG'] = str(bits)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _hip.gemm(0, 1, O, PA, CK, W, CK, 0, XT, CK, CK * PA, Y, PA, O * PA, B, b_bound=BOUND); e1.record(); torch.cuda.synchronize()
        res[k].append(e0.elapsed_time(e1))
fl = 2.0 * O * CK * PA * B
for k, bits in CASES:
    v = sorted(res[k][1:])
    print(f'{k:30s} (bits {bits:2d}): median {v[2]:6.2f} ms = {fl / v[2] / 1e9:6.1f} TFLOP/s of fp32-equivalent products', flush=True)
