"""tools/lists2_ablation.py [B] -- where the time of the two-tile grouping kernel (csrc/so3_inter_lists2.hip) goes: the
deepest layer's backward (O = 512, real inverse lists) and forward (C = 128) launches with parts of the kernel switched
off (EAP_LISTS2_DEBUG bits; needs a library built with `make -C equi-articulated-pose_amd/csrc clean && make ... ABLATION=1`;
results of the ablated runs are wrong by design).  Also the XCD map A/B (mode 1 / 2).  Median of 5 interleaved rounds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
import synth_clouds
import vgtk.cuda.grouping as G
import vgtk.so3conv as sptk
import vgtk.so3conv.functional as L
from vgtk import _hip

B, P, NN, NA, KS = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 4096, 64, 60, 24
dev = torch.device('cuda:0')
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, B, P)[0]).to(dev)
c, o, r, s = synth_clouds.backbone_layers(P)[2]
conv = sptk.InterSO3PoseConv(c, 8, 1, 1, r, s, NN, kanchor=NA, permute_modes=1).to(dev)
idx = G.ball_query(xyz, xyz, r, NN)
gx, nonident = _hip.so3_prep(xyz, xyz, idx, None, None, conv.anchors, 29)
rk = L.rotated_kernels(conv.anchors, conv.kernels)
rows, off, cnt, ent_p, ent_gx, rcap, _ = L._inverse_lists(idx, gx, P, 29, nonident)
gy = torch.randn(B, o, P, NA, device=dev)
feats = torch.randn(B, c, P, NA, device=dev)
CASES = [('full kernel', 0), ('no feature DMA', 1), ('feature DMA of the same 8 rows (hits only)', 512), ('constant weights', 2), ('no row-end stores', 4), ('row-end stores in address order (wrong places)', 1024), ('no LDS operand reads', 16),
         ('no DMA, no barrier', 9), ('no DMA, constant weights', 3), ('no DMA, no weights, no LDS reads', 19), ('MFMAs + barrier only', 23), ('MFMAs only', 31)]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


def inv():
    return _hip.so3_inter_group_inv(gy, rows, off, cnt, ent_p, ent_gx, rk, None, s, NN)


def fwd():
    return _hip.so3_inter_group_fwd(feats, idx, gx, rk, None, s, blocked=2)


for name, fn, fl in (('backward Z, O = 512', inv, 2.0 * B * o * KS * P * NN * NA), ('forward X (transposed), C = 128', fwd, 2.0 * B * c * KS * P * NN * NA)):
    res = {k: [] for k, _ in CASES}
    for _ in range(6):
        for k, bits in CASES:
            os.environ['EAP_LISTS2_DEBUG'] = str(bits)
            res[k].append(timed(fn))
    os.environ['EAP_LISTS2_DEBUG'] = '0'
    for k, bits in CASES:
        v = sorted(res[k][1:])
        print(f'{name}: {k:44s} (bits {bits:2d}): median {v[2]:7.2f} ms = {fl / v[2] / 1e9 / 157.3:.3f} of peak (algorithmic)', flush=True)
    which = 1 if fn is inv else 0
    for mode in (1, 2):
        _hip.lib.eap_so3_group_lists_xcd_map(which, mode)
        v = sorted(timed(fn) for _ in range(6))[:5]
        print(f'{name}: XCD map {mode}: median {v[2]:7.2f} ms = {fl / v[2] / 1e9 / 157.3:.3f} of peak', flush=True)
    _hip.lib.eap_so3_group_lists_xcd_map(which, 1)
