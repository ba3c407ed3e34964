"""tools/zpconv_fwd_hot_ab.py: the shelved on-chip zpconv forward (tools/experiments/kernels/zpconv_fwd_hot.hip, library built with
`make EXPERIMENTS=1`: bash tools/gpu/experiments_build_run.sh zpconv_fwd_hot_ab.py) against the production entry and float64 on six
shapes, then both timed at the bench shape.  With `make EXPERIMENTS=1 ABLATION=1`, EAP_ZPFHOT_DEBUG takes the kernel apart
(profiles/r05_zpconv_fwd_hot_experiment.txt)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench  # noqa: F401  (puts the package on the path)
import vgtk.cuda.zpconv as Z
import vgtk.cuda.grouping as G
from vgtk import _hip
import synth_clouds
dev = torch.device('cuda:0')
NA, KS, NN = 60, 24, 64
_hip.lib.eap_inter_zpconv_fwd_hot_workspace.restype = ctypes.c_int64


def hot_forward(idx, w, feats):
    b, np_, na, ks, ann = idx.shape
    c, nq = feats.shape[1], feats.shape[2]
    out = torch.empty(b, c, ks, np_, na, dtype=feats.dtype, device=feats.device)
    nbytes = int(_hip.lib.eap_inter_zpconv_fwd_hot_workspace(b, np_, nq, na, ks, ann, c))
    assert nbytes > 0, 'shape not taken'
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=feats.device)
    _hip.call('eap_inter_zpconv_fwd_hot_f32', out, b, np_, nq, na, ks, ann, c, _hip._ptr(idx), _hip._ptr(w), _hip._ptr(feats), _hip._ptr(out),
              _hip._ptr(ws))
    return out


def operands(clouds, points, channels, ks=KS, na=NA, seed=0):
    xyz = torch.from_numpy(synth_clouds.laptop_batch(seed, clouds, points)[0]).to(dev)
    radius = synth_clouds.backbone_layers(4096)[1][2]
    ball = G.ball_query(xyz, xyz, radius, NN)
    idx = ball[:, :, None, None, :].expand(clouds, points, na, ks, NN).contiguous()
    g = torch.Generator(device=dev).manual_seed(seed)
    w = torch.rand(clouds, points, na, ks, NN, device=dev, generator=g)
    feats = torch.randn(clouds, channels, points, na, device=dev, generator=g)
    return ball, idx, w, feats


def case(clouds, points, channels, ks=KS, na=NA):
    ball, idx, w, feats = operands(clouds, points, channels, ks, na)
    rows = [int(torch.unique(ball[b]).numel()) for b in range(clouds)]
    hot, mat = hot_forward(idx, w, feats), Z.inter_zpconv_forward(idx, w, feats)
    ref = torch.einsum('pakn,cpna->ckpa', w[0].double(), feats[0].double()[:, ball[0].long(), :])
    scale = float(ref.abs().max())
    print(f'{clouds} x {points}, C={channels}, ks={ks}, na={na}: rows {min(rows)}..{max(rows)}; on-chip vs float64 (cloud 0) '
          f'{float((hot[0].double() - ref).abs().max()) / scale:.2e}, production vs float64 {float((mat[0].double() - ref).abs().max()) / scale:.2e}, '
          f'on-chip vs production (all clouds) {float((hot - mat).abs().max()) / scale:.2e}', flush=True)


def timed(fn, warm=2, reps=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


if not os.environ.get('EAP_ZPFHOT_DEBUG'):
    case(2, 512, 32)
    case(3, 1000, 64)
    case(1, 777, 96, ks=13, na=12)
    case(2, 600, 32, ks=32, na=8)
    case(2, 600, 32, ks=27, na=8)
    case(8, 4096, 64)
ball, idx, w, feats = operands(8, 4096, 64)
byts = 4.0 * 8 * (2.0 * 4096 * NA * KS * NN + 64 * 4096 * NA + 64 * KS * 4096 * NA)
for name, fn in (('on-chip rows (experiment)', lambda: hot_forward(idx, w, feats)), ('production', lambda: Z.inter_zpconv_forward(idx, w, feats))):
    ms = timed(fn)
    print(f'EAP_ZPFHOT_DEBUG={os.environ.get("EAP_ZPFHOT_DEBUG", "0")} {name}: forward {ms:.2f} ms = {byts / ms / 1e6 / 8000:.3f} of the HBM roofline', flush=True)
