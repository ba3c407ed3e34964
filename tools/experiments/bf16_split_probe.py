"""Probe: rate of library bf16 GEMMs (torch -> hipBLASLt/rocBLAS) at the contraction's shape, and the
accuracy of a 6-product bf16 split against fp64."""
import torch, time
dev = torch.device('cuda:0')
M, N, K, B = 512, 245760, 3072, 2
torch.manual_seed(0)
W = torch.randn(M, K, device=dev) * 0.05
X = torch.randn(B, K, N, device=dev).abs_()          # grouped features are sums of non-negative weights x features
def split3(t):
    h = t.bfloat16(); r = t - h.float(); m = r.bfloat16(); l = (r - m.float()).bfloat16()
    return h, m, l
Wh, Wm, Wl = split3(W)
Xh, Xm, Xl = split3(X)
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
flops = 2.0 * M * N * K * B
try:
    ms = t_ms(lambda: torch.matmul(Wh, Xh))
    print(f'bf16 matmul -> bf16 out: {ms:.2f} ms, {flops / ms / 1e9:.0f} TFLOP/s')
except Exception as e: print('bf16 matmul failed', e)
for name, fn in (('mm out_dtype', lambda: torch.mm(Wh, Xh[0], out_dtype=torch.float32)),
                 ('bmm out_dtype', lambda: torch.bmm(Wh[None].expand(B, M, K), Xh, out_dtype=torch.float32))):
    try:
        ms = t_ms(fn)
        f = flops if 'bmm' in name else flops / B
        print(f'{name}: {ms:.2f} ms, {f / ms / 1e9:.0f} TFLOP/s, dtype {fn().dtype}')
    except Exception as e: print(name, 'failed:', str(e)[:200])
ms = t_ms(lambda: torch.matmul(W, X))
print(f'fp32 torch matmul: {ms:.2f} ms, {flops / ms / 1e9:.0f} TFLOP/s')
# accuracy of the 6-product split (fp32 accumulate emulated with float products of bf16 values) on a slice
Xs = X[0, :, :2048]
ref = W.double() @ Xs.double()
def prod(a, b): return (a.float().double() @ b[0, :, :2048].float().double())
six = prod(Wh, Xh) + prod(Wh, Xm) + prod(Wm, Xh) + prod(Wh, Xl) + prod(Wl, Xh) + prod(Wm, Xm)
three = prod(Wh, Xh) + prod(Wh, Xm) + prod(Wm, Xh)
fp32 = (W @ Xs).double()
for n, v in (('fp32 matmul', fp32), ('3-product split (exact accumulation)', three), ('6-product split (exact accumulation)', six)):
    print(f'{n}: max err / max |ref| = {float((v - ref).abs().max() / ref.abs().max()):.3e}')
