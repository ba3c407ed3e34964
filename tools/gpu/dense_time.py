"""Times of the dense product (csrc/so3_dense.hip) at the bench shape: 8 x 4096 points, deepest layer (128 -> 512), both
directions, both weight forms; the stored-operand split and the product separately (HIP events, medians)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import numpy as np
import torch
import synth_clouds
import vgtk.so3conv.functional as L
import vgtk.cuda.grouping as cuda_nn
from vgtk import _hip

dev = torch.device('cuda:0')
B, P, NN, NA, KS = int(os.environ.get('B', 8)), int(os.environ.get('P', 4096)), 64, 60, 24
layer = int(os.environ.get('LAYER', 2))
c, o, radius, sigma = synth_clouds.backbone_layers(P)[layer]
o = int(os.environ.get('O', o))
xyz = torch.from_numpy(synth_clouds.laptop_batch(0, B, P)[0]).to(dev).contiguous()
anchors = torch.from_numpy(np.asarray(L.get_anchors(NA), dtype=np.float32)).to(dev)
kernels = torch.from_numpy(L.get_sphereical_kernel_points_from_ply(0.7 * radius, 1)).to(dev)
rk = L.rotated_kernels(anchors, kernels)
idx = cuda_nn.ball_query(xyz, xyz, radius, NN)
head = L._ListHead(idx, P, None, None, dense_probe=(None, None))
rcap, _ = head.decide()
rp = L._dense_rows(rcap, P)
head.wait()
print('rows per cloud', head.n_rows.tolist(), 'rp', rp, 'dense possible', head.dense_possible())


def timed(fn, n=int(os.environ.get('REPS', 5))):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), r


gen = torch.Generator(device=dev).manual_seed(1)
TRIM = os.environ.get('TRIM', '1') != '0'
_hip.lib.eap_so3_dense_block_rows(int(os.environ.get('ROWS', 0)))
gy = torch.randn(B, o, P, NA, device=dev, generator=gen)
g = torch.randn(B, o, KS, rp * NA, device=dev, generator=gen)
SORT = os.environ.get('SORT', '1') != '0'       # occupancy-sorted query points + k-step lists (round 6); 0: index order, every k-step
for form in [int(f) for f in os.environ.get('FORMS', '1').split(',')]:
    _hip.lib.eap_so3_dense_form(form)
    mk = lambda: _hip.DenseGeometry(xyz, xyz, head.memb, head.rows, rp, rk, sigma, NN, head.n_rows if TRIM else None, sort=SORT)
    geo = mk()
    t_tab, _ = timed(mk)
    geo.mask(0); geo.mask(1)
    st0, st1 = (geo.steps(0), geo.steps(1)) if SORT else (None, None)
    if SORT:
        t_steps, _ = timed(lambda: (geo._masks.clear(), geo._steps.clear(), geo.mask(0), geo.steps(0), geo.mask(1), geo.steps(1)))
        k0, k1 = st0[:, :, 0].float(), st1[:, :, 0].float()
        print(f'masks + step lists of both directions {t_steps:.2f} ms; k-steps kept: backward {float(k0.mean()) / (st0.shape[2] - 1):.3f}, forward {float(k1.mean()) / (st1.shape[2] - 1):.3f} of all')
    cm = geo.columns()
    t_split, (scale, planes) = timed(lambda: _hip.so3_dense_split(gy, colmap=cm))
    z = torch.empty(B, o, KS, NA, rp, device=dev)
    t_prod, _ = timed(lambda: _hip.call('eap_so3_dense_product_steps_f32', gy, 0, B, o, P, NA, KS, rp, _hip._I64(NA * rp), _hip._F32(sigma), _hip._ptr(geo.n_rows), _hip._ptr(planes), _hip._ptr(scale),
                                        _hip._ptr(geo.pt), _hip._ptr(geo.kr), _hip._ptr(geo.mask(0)), _hip._ptr(st0), _hip._ptr(z)))
    fl = 6.0 * B * o * P * NA * KS * rp
    print(f'form {form} sort {int(SORT)} backward: tables {t_tab:.2f} ms, split {t_split:.2f} ms, product {t_prod:.2f} ms, algorithmic {2.0 * B * o * P * NA * KS * NN / t_prod / 1e9:.0f} TFLOP/s '
          f'(x3 / fp16 peak {3 * 2.0 * B * o * P * NA * KS * NN / t_prod / 1e9 / 2500:.3f})')
    del planes, scale, z
    t_split, (scale, planes) = timed(lambda: _hip.so3_dense_split(g, seg=rp, seg_pitch=rp * NA, shape=(B, o, KS * rp, NA), mapped=True, n_rows=geo.n_rows))
    yt = torch.empty(B, NA, o, P, device=dev)
    t_prod, _ = timed(lambda: _hip.call('eap_so3_dense_product_steps_f32', g, 1, B, o, P, NA, KS, rp, _hip._I64(0), _hip._F32(sigma), _hip._ptr(geo.n_rows), _hip._ptr(planes), _hip._ptr(scale),
                                        _hip._ptr(geo.pt), _hip._ptr(geo.kr), _hip._ptr(geo.mask(1)), _hip._ptr(st1), _hip._ptr(yt)))
    y = torch.empty(B, o, P, NA, device=dev)
    ps = torch.empty(o, B * ((P + 63) // 64), device=dev); pq = torch.empty_like(ps)
    if SORT:
        t_un, _ = timed(lambda: _hip.call('eap_so3_dense_untranspose_map_stats_f32', g, B, o, P, NA, P, _hip._ptr(geo.order), _hip._ptr(geo.pivot_pos), _hip._ptr(yt), _hip._ptr(y), _hip._ptr(ps), _hip._ptr(pq)))
    else:
        t_un, _ = timed(lambda: _hip.call('eap_so3_dense_untranspose_f32', g, B, o, P, NA, _hip._ptr(yt), _hip._ptr(y), _hip._ptr(ps), _hip._ptr(pq)))
    print(f'form {form} sort {int(SORT)} forward:  split {t_split:.2f} ms, product {t_prod:.2f} ms, untranspose (+ moments) {t_un:.2f} ms')
    del planes, scale, yt, y
_hip.lib.eap_so3_dense_form(1)
