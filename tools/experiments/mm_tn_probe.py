"""Probe: library fp32 GEMM rate when the streamed operand is stored [N, K] (k contiguous)."""
import torch
dev = torch.device('cuda:0')
def t_ms(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (M, K, N, B) in ((512, 3072, 245760, 8), (128, 1536, 245760, 8)):
    W = torch.randn(M, K, device=dev)
    Xt = torch.randn(B, N, K, device=dev)        # [N, K]
    flops = 2.0 * M * N * K * B
    ms = t_ms(lambda: torch.matmul(W, Xt.transpose(1, 2)))          # [M,K] x [K,N] with B^T storage
    print(f'M={M} K={K}: W @ Xt^T  (out [B,M,N]): {ms:.2f} ms  {flops / ms / 1e9:.0f} TFLOP/s')
    ms = t_ms(lambda: torch.matmul(Xt, W.t()))                      # out [B,N,M] (transposed output)
    print(f'M={M} K={K}: Xt @ W^T  (out [B,N,M]): {ms:.2f} ms  {flops / ms / 1e9:.0f} TFLOP/s')
    del Xt
    X = torch.randn(B, K, N, device=dev)
    ms = t_ms(lambda: torch.matmul(W, X))
    print(f'M={M} K={K}: W @ X     (X [K,N])    : {ms:.2f} ms  {flops / ms / 1e9:.0f} TFLOP/s')
    del X
