"""tools/split_planes_probe.py [clouds]: the forward contraction of the two deep layers with three bf16 planes (six products),
two fp16 planes (three products) and on the fp32 matrix pipe -- error against fp64 on a slab of the output, and time.
Operands like the path's: weights ~ N(0, 0.05), grouped features non-negative with a wide dynamic range."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'equi-articulated-pose_amd'))
import torch
from vgtk import _hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
PA = 4096 * 60


def timed(fn, rounds=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def run(mode, fn):
    planes, split = {'bf16x3': (3, True), 'f16x2': (2, True), 'fp32': (3, False)}[mode]
    _hip.SPLIT_PLANES, _hip.SPLIT_BF16_CONTRACTION = planes, split
    try:
        return fn()
    finally:
        _hip.SPLIT_PLANES, _hip.SPLIT_BF16_CONTRACTION = 2, True


for (O, CK) in ((512, 3072), (128, 1536)):
    gen = torch.Generator().manual_seed(O)
    W = (torch.randn(O, CK, generator=gen) * 0.05).to(dev)
    XT = torch.randn(B, PA, CK, device=dev).abs_() * torch.exp(torch.randn(B, PA, 1, device=dev) * 1.5)
    Y = torch.empty(B, O, PA, device=dev)
    bound = (_hip.absmax_rows(XT, B, PA, CK, CK, PA * CK), 1, 1.0)
    rows = slice(0, 4096)
    ref = torch.matmul(W.double(), XT[0, rows].double().t())                # [O, 4096]
    den = torch.matmul(W.abs().double(), XT[0, rows].abs().double().t())
    print(f'Y[{O} x {PA}] = W[{O} x {CK}] . XT^T, {B} clouds; flops {2.0 * O * PA * CK * B:.3e}')
    for mode in ('bf16x3', 'f16x2', 'fp32'):
        call = lambda: _hip.gemm(0, 1, O, PA, CK, W, CK, 0, XT, CK, CK * PA, Y, PA, O * PA, B, b_bound=bound)
        ms = run(mode, lambda: timed(call))
        got = Y[0, :, rows].double()
        err = (got - ref).abs()
        print(f'   {mode:7s} {ms:8.3f} ms  {2.0 * O * PA * CK * B / ms / 1e9:7.1f} TFLOP/s-equivalent   max err / max|C| {err.max().item() / ref.abs().max().item():.3e}'
              f'   rms err / rms C {err.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item():.3e}   max err / sum|a||b| {(err / den).max().item():.3e}', flush=True)
    # the same with the scale taken from a loose bound (64 x the maximum: six binades of headroom unused)
    loose = (bound[0], 1, 64.0)
    ms = run('f16x2', lambda: timed(lambda: _hip.gemm(0, 1, O, PA, CK, W, CK, 0, XT, CK, CK * PA, Y, PA, O * PA, B, b_bound=loose)))
    err = (Y[0, :, rows].double() - ref).abs()
    print(f'   f16x2, bound 64 x max: max err / max|C| {err.max().item() / ref.abs().max().item():.3e}   rms {err.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item():.3e}')
    # pass over B included (no bound known)
    ms = run('f16x2', lambda: timed(lambda: _hip.gemm(0, 1, O, PA, CK, W, CK, 0, XT, CK, CK * PA, Y, PA, O * PA, B)))
    print(f'   f16x2 with its own pass over B (or the three-plane kernel below 384 rows): {ms:8.3f} ms')
    del XT, Y
