"""tools/inv_pitch_experiment.py [B] -- the backward grouping launch (layer-3 shapes: O = 512 channels of dY,
4096 points, ~136 referenced rows) with dY rows 240 bytes apart (the boundary layout [B,O,P,60]) against dY rows
padded to 256 bytes ([B,O,P,64]: every 128-byte DMA piece is one cache line instead of straddling two).
Interleaved rounds; also checks that both give the same Z."""
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
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz = torch.from_numpy(xyz).to(dev)
for layer in (2, 1):
    c, o, r, s = synth_clouds.backbone_layers(P)[layer]
    conv = sptk.InterSO3PoseConv(c, 8, 1, 1, r, s, NN, kanchor=NA, permute_modes=1).to(dev)
    idx = G.ball_query(xyz, xyz, r, NN)
    gx, nonident = _hip.so3_prep(xyz, xyz, idx, None, None, conv.anchors, 29)
    rk = L.rotated_kernels(conv.anchors, conv.kernels)
    rows, off, cnt, ent_p, ent_gx, rcap, _ = L._inverse_lists(idx, gx, P, 29, nonident)
    gy = torch.randn(B, o, P, NA, device=dev)
    gy64 = torch.zeros(B, o, P, 64, device=dev)
    gy64[..., :NA] = gy
    z = torch.empty(B, o, KS, rcap, NA, device=dev)
    z64 = torch.empty_like(z)

    def run(src, pitch, out):
        _hip.call('eap_so3_inter_group_inv_pitch_f32', out, B, o, P, NN, NA, pitch, KS, rcap, _hip._F32(s), _hip._ptr(src), _hip._ptr(rows),
                  _hip._ptr(off), _hip._ptr(cnt), _hip._ptr(ent_p), _hip._ptr(ent_gx), _hip._ptr(rk), _hip._ptr(out))
    run(gy, NA, z); run(gy64, 64, z64)
    torch.cuda.synchronize()
    if not os.environ.get("EAP_LISTS_DEBUG"):
        assert torch.equal(z, z64)
    res = {60: [], 64: []}
    for _ in range(5):
        for pitch, src, out in ((NA, gy, z), (64, gy64, z64)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(src, pitch, out); e1.record(); torch.cuda.synchronize()
            res[pitch].append(e0.elapsed_time(e1))
    fl = 2.0 * B * o * KS * P * NN * NA
    for pitch, v in res.items():
        v.sort()
        print(f'layer {layer} O={o} rcap={rcap}: dY row pitch {pitch}: median {v[2]:.2f} ms  min {v[0]:.2f} ms  {fl / v[2] / 1e9:.1f} TFLOP/s (algorithmic)', flush=True)
