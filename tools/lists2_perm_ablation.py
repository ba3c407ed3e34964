"""tools/lists2_perm_ablation.py -- where the extra time of the PERMUTED two-tile grouping kernel (csrc/so3_inter_lists2.hip,
PERM) goes: the deepest layer's backward launch (O = 512, real inverse lists, random per-point poses) with parts switched
off (EAP_LISTS2_DEBUG bits; library built with `make ABLATION=1`; ablated results are wrong by design), next to the
identity-pose launch on the same lists."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import numpy as np, torch
import synth_clouds
import vgtk.cuda.grouping as G
import vgtk.so3conv as sptk
import vgtk.so3conv.functional as L
from vgtk import _hip

B, P, NN, NA, KS = 8, 4096, 64, 60, 24
dev = torch.device('cuda:0')
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, B, P)[0]).to(dev)
c, o, r, s = synth_clouds.backbone_layers(P)[2]
conv = sptk.InterSO3PoseConv(c, 8, 1, 1, r, s, NN, kanchor=NA, permute_modes=1).to(dev)
mult, ident = L._group_tables(conv.anchors)
rng = np.random.default_rng(0)
q = rng.standard_normal((B, P, 4)); q /= np.linalg.norm(q, axis=-1, keepdims=True)
w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
              2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(B, P, 3, 3).astype(np.float32)
rot = torch.from_numpy(R).to(dev)
idx = G.ball_query(xyz, xyz, r, NN)
rk = L.rotated_kernels(conv.anchors, conv.kernels)
gy = torch.randn(B, o, P, NA, device=dev)


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


fl = 2.0 * B * o * KS * P * NN * NA
gx0, non0 = _hip.so3_prep(xyz, xyz, idx, None, None, conv.anchors, ident)
rows, off, cnt, ent_p, ent_gx, rcap, _ = L._inverse_lists(idx, gx0, P, ident, non0)
v = sorted(timed(lambda: _hip.so3_inter_group_inv(gy, rows, off, cnt, ent_p, ent_gx, rk, None, s, NN)) for _ in range(6))[:5]
print(f'identity poses, two-tile kernel: median {v[2]:7.2f} ms = {fl / v[2] / 1e9 / 157.3:.3f} of peak', flush=True)

gx1, non1 = _hip.so3_prep(xyz, xyz, idx, rot, rot, conv.anchors, ident)
rows, off, cnt, ent_p, ent_gx, rcap, _ = L._inverse_lists(idx, gx1, P, ident, non1)
multinv = L._group_tables_inverse(mult)
coset = L._coset_tables(multinv, ident)
t_re = sorted(timed(lambda: _hip.anchor_reorder(gy, coset[0])) for _ in range(5))[2]
t_en = sorted(timed(lambda: _hip.so3_perm_entries(ent_p, ent_gx, coset[1], conv.anchors, ident, NA, P)) for _ in range(5))[2]
print(f'anchor re-order of gy: {t_re:.2f} ms; per-entry words: {t_en:.2f} ms', flush=True)
gyc = _hip.anchor_reorder(gy, coset[0])
ent_pc, ent_gx2 = _hip.so3_perm_entries(ent_p, ent_gx, coset[1], conv.anchors, ident, NA, P)


def perm2():
    return _hip.so3_inter_group_inv_perm2(gyc, rows, off, cnt, ent_pc, ent_gx2, rk, coset[0], s, NN)


def whole_row():
    return _hip.so3_inter_group_inv(gy, rows, off, cnt, ent_p, ent_gx, rk, multinv, s, NN, ident, conv.anchors, coset)


ABLATION = bool(os.environ.get('EAP_PERM_ABLATION'))         # library built with `make ABLATION=1`
CASES = [('full kernel', 0)] + ([('no block move (own block by DMA)', 32), ('no in-block XOR', 64), ('neither', 96), ('no row-end stores', 4),
                                 ('no feature DMA', 1), ('no LDS operand reads', 16), ('MFMAs only', 31 + 96)] if ABLATION else [])
res = {k: [] for k, _ in CASES}
for _ in range(6):
    for k, bits in CASES:
        os.environ['EAP_LISTS2_DEBUG'] = str(bits)
        res[k].append(timed(perm2))
os.environ['EAP_LISTS2_DEBUG'] = '0'
for k, bits in CASES:
    v = sorted(res[k][1:])
    print(f'random poses, two-tile PERM kernel alone: {k:36s} (bits {bits:3d}): median {v[2]:7.2f} ms = {fl / v[2] / 1e9 / 157.3:.3f} of peak', flush=True)
v = sorted(timed(whole_row) for _ in range(6))[:5]
print(f'random poses, whole-row kernel (incl. the {t_re:.1f} ms re-order): median {v[2]:7.2f} ms', flush=True)
