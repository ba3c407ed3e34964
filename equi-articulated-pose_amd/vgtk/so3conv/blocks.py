"""Block-layer epilogue of the SO(3) conv blocks: BatchNorm2d + leaky_relu in one pass.

Reference: `feat = self.norm(x.feats); feat = self.relu(feat)` in
SPConvNets/utils/base_so3poseconv.py:L214-221 (`norm = nn.BatchNorm2d(dim_out)`, `relu =
F.leaky_relu` for the shipped 'leaky_relu' activation, L196-201).  SURVEY.md 8(f) row 1.

`BatchNormLeakyReLU` has nn.BatchNorm2d's parameters and buffers under the same names (weight,
bias, running_mean, running_var, num_batches_tracked), so a reference checkpoint's `norm.*`
entries load into it; its forward equals `F.leaky_relu(bn(x), negative_slope)`.
The arithmetic runs in csrc/bn_act.hip (no CPU / eager fallback).
"""
import torch
from torch import nn

from .. import _hip


class _BNAct(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, slope):
        x = x.contiguous()
        b, c = x.shape[0], x.shape[1]
        n = x.numel() // (b * c)
        count = b * n
        if training:
            pivot = x.reshape(b, c, n)[0, :, 0].double()
            s1, s2 = _hip.bn_stats(x, b, c, n)
            m = s1 / count
            mean = pivot + m
            var = (s2 / count - m * m).clamp_(min=0.0)                     # biased, as BatchNorm normalises with
            if running_mean is not None:
                with torch.no_grad():
                    running_mean.mul_(1.0 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
                    running_var.mul_(1.0 - momentum).add_((var * (count / max(count - 1, 1))).to(running_var.dtype), alpha=momentum)
        else:
            mean, var = running_mean.double(), running_var.double()
        invstd = torch.rsqrt(var + eps)
        scale64 = weight.double() * invstd
        scale = scale64.float()
        shift = (bias.double() - mean * scale64).float()
        y = _hip.bn_act_fwd(x, b, c, n, scale, shift, slope)
        ctx.save_for_backward(x, scale, shift, mean.float(), invstd.float())
        ctx.training, ctx.slope, ctx.dims = training, slope, (b, c, n)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, scale, shift, mean, invstd = ctx.saved_tensors
        b, c, n = ctx.dims
        gy = gy.contiguous()
        sg, sgx = _hip.bn_act_bwd_reduce(gy, x, b, c, n, scale, shift, mean, invstd, ctx.slope)
        g_x = None
        if ctx.needs_input_grad[0]:
            if ctx.training:
                k2 = (scale.double() * sg / (b * n)).float()
                k3 = (scale.double() * sgx / (b * n)).float()
            else:
                k2 = torch.zeros_like(scale)
                k3 = torch.zeros_like(scale)
            g_x = _hip.bn_act_bwd_apply(gy, x, b, c, n, scale, shift, mean, invstd, k2, k3, ctx.slope)
        return g_x, sgx.float(), sg.float(), None, None, None, None, None, None


class BatchNormLeakyReLU(nn.BatchNorm2d):
    """nn.BatchNorm2d followed by leaky_relu, fused.  Input [B, C, P, A] (any trailing shape whose
    product is a multiple of 4)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, negative_slope=0.01):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=True, track_running_stats=True)
        self.negative_slope = negative_slope

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('BatchNormLeakyReLU: tensor must be a CUDA(HIP) tensor (no CPU fallback)')
        if self.training:
            self.num_batches_tracked.add_(1)
        return _BNAct.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                            self.momentum, self.eps, self.negative_slope)
