"""tools/inv_locality_experiment.py [B] -- how much of the backward grouping launch's time is memory: the same launch
(layer-3 / layer-2 shapes, real inverse lists of a synthetic batch) with the entries' dY rows
  real      : as the lists say (working set of a (cloud, 32-channel slice): 31 MB against 4 MB of L2 per XCD)
  fold 256  : row index folded to p % 256 (2 MB per slice: L2-resident)
  fold 16   : p % 16 (123 KB per slice)
The arithmetic, the list lengths, the DMA instruction count and the LDS traffic are identical; only where the rows come
from changes.  Interleaved rounds, median of 5."""
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
    z = torch.empty(B, o, KS, rcap, NA, device=dev)
    variants = {'real': ent_p, 'fold 256': ent_p % 256, 'fold 16': ent_p % 16}
    variants = {k: v.contiguous() for k, v in variants.items()}

    def run(ep):
        _hip.call('eap_so3_inter_group_inv_pitch_f32', z, B, o, P, NN, NA, NA, KS, rcap, _hip._F32(s), _hip._ptr(gy), _hip._ptr(rows),
                  _hip._ptr(off), _hip._ptr(cnt), _hip._ptr(ep), _hip._ptr(ent_gx), _hip._ptr(rk), _hip._ptr(z))
    fl = 2.0 * B * o * KS * P * NN * NA
    cn = cnt.flatten().float()
    print(f'layer {layer} O={o} rcap={rcap}: list lengths mean {cn[cn > 0].mean().item():.0f} max {cn.max().item():.0f} min {cn[cn > 0].min().item():.0f}', flush=True)
    zref = None
    for tiles, xmap in ((2, 1), (3, 1), (4, 1)):   # 3, 4: `make EXPERIMENTS=1` builds (3 x bf16 planes / 2 x fp16 planes on the 16-bit matrix cores)
        if _hip.lib.eap_so3_group_lists_tiles(tiles) != tiles:
            continue
        _hip.lib.eap_so3_group_lists_xcd_map(1, xmap)
        run(variants['real'])
        zd = z.double()
        if zref is None:
            zref = zd.clone()
        else:
            print(f'   tiles {tiles}: max |z - z(tiles 2)| / max |z| = {(zd - zref).abs().max().item() / zref.abs().max().item():.3e}   rms {((zd - zref).pow(2).mean().sqrt() / zref.pow(2).mean().sqrt()).item():.3e}', flush=True)
        del zd
        for ep in variants.values():
            run(ep)
        torch.cuda.synchronize()
        res = {k: [] for k in variants}
        for _ in range(5):
            for k, ep in variants.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(ep); e1.record(); torch.cuda.synchronize()
                res[k].append(e0.elapsed_time(e1))
        for k, v in res.items():
            v.sort()
            print(f'layer {layer} O={o} tiles/wave {tiles} xcd map {xmap}: rows {k:9s}: median {v[2]:.2f} ms  min {v[0]:.2f} ms  {fl / v[2] / 1e9:.1f} TFLOP/s algorithmic = {fl / v[2] / 1e9 / 157.3:.3f} of peak', flush=True)
    _hip.lib.eap_so3_group_lists_tiles(2)
    _hip.lib.eap_so3_group_lists_xcd_map(1, 1)
