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
import torch.distributed as dist
from torch import nn


def _syncing(sync):
    return bool(sync) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def batch_moments(s1, s2, pivot, count, sync=False):
    """Per-channel mean and biased variance from pivoted sums  s1 = sum(x - pivot), s2 = sum((x - pivot)^2)
    over `count` elements (float64 tensors [C]).  With `sync` (the reference wraps its models in
    nn.SyncBatchNorm for multi-GPU training, trainer_unsup_arti_align.py:L430) the raw moments of
    all ranks are summed by ONE all-reduce of 2C+1 doubles, so every rank normalises with the
    statistics of the whole batch.  Returns (mean, var, total_count)."""
    if not _syncing(sync):
        m = s1 / count
        return pivot + m, (s2 / count - m * m).clamp_(min=0.0), count
    raw = torch.cat([s1 + count * pivot, s2 + 2.0 * pivot * s1 + count * pivot * pivot,
                     torch.full((1,), float(count), dtype=torch.float64, device=s1.device)])
    dist.all_reduce(raw)
    c = s1.numel()
    total = raw[2 * c]
    mean = raw[:c] / total
    var = (raw[c:2 * c] / total - mean * mean).clamp_(min=0.0)
    return mean, var, total


def all_reduce_sums(*tensors, sync=False):
    """Sum float64 per-channel reductions over the ranks (one all-reduce); identity without sync."""
    if not _syncing(sync):
        return tensors
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat)
    out, o = [], 0
    for t in tensors:
        out.append(flat[o:o + t.numel()].view_as(t))
        o += t.numel()
    return tuple(out)


from .. import _hip  # noqa: E402  (after the pure-torch helpers: they are unit-tested without the HIP library)


class _BNAct(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, slope, sync, residual=None, pre_bias=None):
        """pre_bias [C]: act(bn(x + pre_bias)) without the pass that would add it -- a per-channel constant in front of a
        BatchNorm moves the batch mean and nothing else (training: it only enters the running mean, its gradient is
        exactly zero; eval: it shifts the pre-activation like beta)."""
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
            if residual.shape != x.shape:
                raise RuntimeError('BatchNormLeakyReLU: residual must have the shape of the input')
        b, c = x.shape[0], x.shape[1]
        n = x.numel() // (b * c)
        count = b * n
        if training:
            pivot = x.reshape(b, c, n)[0, :, 0].double()
            s1, s2 = _hip.bn_stats(x, b, c, n)
            mean, var, count = batch_moments(s1, s2, pivot, count, sync)    # biased variance, as BatchNorm normalises with
            if running_mean is not None:
                with torch.no_grad():
                    full = mean if pre_bias is None else mean + pre_bias.double()
                    running_mean.mul_(1.0 - momentum).add_(full.to(running_mean.dtype), alpha=momentum)
                    running_var.mul_(1.0 - momentum).add_((var * (count / (count - 1))).to(running_var.dtype), alpha=momentum)
        else:
            mean, var = running_mean.double(), running_var.double()
            if pre_bias is not None:
                mean = mean - pre_bias.double()
        invstd = torch.rsqrt(var + eps)
        scale64 = weight.double() * invstd
        scale = scale64.float()
        shift = (bias.double() - mean * scale64).float()
        y = _hip.bn_act_fwd(x, b, c, n, scale, shift, slope, residual)
        ctx.has_res = residual is not None
        ctx.has_pre_bias = pre_bias is not None
        ctx.save_for_backward(x, scale, shift, mean.float(), invstd.float())
        ctx.training, ctx.slope, ctx.dims, ctx.sync, ctx.count = training, slope, (b, c, n), sync, count
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
                tg, tgx = all_reduce_sums(sg, sgx, sync=ctx.sync)          # whole-batch means (SyncBatchNorm backward)
                k2 = (scale.double() * tg / ctx.count).float()
                k3 = (scale.double() * tgx / ctx.count).float()
            else:
                k2 = torch.zeros_like(scale)
                k3 = torch.zeros_like(scale)
            g_x = _hip.bn_act_bwd_apply(gy, x, b, c, n, scale, shift, mean, invstd, k2, k3, ctx.slope)
        g_pre = None
        if ctx.has_pre_bias:
            g_pre = torch.zeros_like(scale) if ctx.training else (sg * scale.double()).float()
        return g_x, sgx.float(), sg.float(), None, None, None, None, None, None, None, (gy if ctx.has_res else None), g_pre


class BatchNormLeakyReLU(nn.BatchNorm2d):
    """nn.BatchNorm2d followed by leaky_relu, fused.  Input [B, C, P, A] (any trailing shape whose
    product is a multiple of 4)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, negative_slope=0.01, sync=False):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=True, track_running_stats=True)
        self.negative_slope = negative_slope
        self.sync = sync          # batch statistics over all ranks (what nn.SyncBatchNorm does for the reference)

    def forward(self, x, residual=None, pre_bias=None):
        """leaky_relu(BatchNorm(x + pre_bias)) (+ residual: the separable block's `x.feats + skip_feature`,
        base_so3poseconv.py:L319-328, added in the same pass).  pre_bias [C]: the bias of the 1x1 conv in front, folded into
        the mean instead of a pass of its own.  In eval mode the normalisation is already folded into one per-channel
        scale / shift (no statistics pass): a single read + write of the tensor."""
        if not x.is_cuda:
            raise RuntimeError('BatchNormLeakyReLU: tensor must be a CUDA(HIP) tensor (no CPU fallback)')
        if self.training:
            self.num_batches_tracked.add_(1)
        return _BNAct.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                            self.momentum, self.eps, self.negative_slope, self.sync, residual, pre_bias)

    def folded(self, pre_bias=None, residual=None):
        """The layer in inference mode as one per-channel affine map + activation ("inference-mode BN
        folds into W"): leaky_relu(scale * x + shift) (+ residual), scale = gamma / sqrt(running_var + eps),
        shift = beta - (running_mean - pre_bias) * scale, computed in float64 exactly as the eval-mode forward does."""
        from .functional import FoldedEpilogue
        if self.training:
            raise RuntimeError('BatchNormLeakyReLU.folded: training-mode statistics depend on the data; call .eval() first')
        scale64 = self.weight.detach().double() * torch.rsqrt(self.running_var.double() + self.eps)
        mean = self.running_mean.double() if pre_bias is None else self.running_mean.double() - pre_bias.detach().double()
        return FoldedEpilogue(scale64.float(), (self.bias.detach().double() - mean * scale64).float(), self.negative_slope, residual)


def conv_norm_act(conv, norm, x, **conv_kwargs):
    """The block layer's `x = conv(x); feat = relu(norm(x.feats))` (SPConvNets/utils/base_so3poseconv.py:L205-222) for an
    InterSO3Conv / InterSO3PoseConv and a BatchNormLeakyReLU -> (inter_idx, inter_w, sample_idx, cloud with the activated
    features).  In inference (norm in eval mode, gradients off) the normalisation and the activation ride in the
    contraction's epilogue -- the feature map is written once, no pass of its own; in training the norm joins the conv's autograd
    node where the conv runs the dense product; otherwise conv and norm run as usual."""
    from vgtk.spconv import SphericalPointCloud, SphericalPointCloudPose
    from .functional import TrainEpilogue
    fold = not norm.training and not torch.is_grad_enabled()
    ep = norm.folded() if fold else None
    if ep is None and norm.training and isinstance(norm, BatchNormLeakyReLU) and norm.negative_slope > 0:
        # training: where the conv runs the dense product the normalisation joins its autograd node (no pass of its own in either
        # direction, the conv output is never written: vgtk/so3conv/functional.py TrainEpilogue)
        ep = TrainEpilogue(norm)
    if ep is not None:
        conv_kwargs = dict(conv_kwargs, epilogue=ep)
    inter_idx, inter_w, sample_idx, y = conv(x, **conv_kwargs)
    if ep is not None and ep.applied:
        feats = y.feats
        if norm.training:
            norm.num_batches_tracked.add_(1)
    else:
        feats = norm(y.feats)
    pose = getattr(y, 'pose', None)
    out = SphericalPointCloudPose(y.xyz, feats, y.anchors, pose) if pose is not None else SphericalPointCloud(y.xyz, feats, y.anchors)
    return inter_idx, inter_w, sample_idx, out


def pointwise_norm_act(conv1x1, norm, x, residual=None):
    """relu(norm(conv1x1(x))) (+ residual): the separable block's skip branch and sum (base_so3poseconv.py:L319-328) for an
    nn.Conv2d(c, o, 1) and a BatchNormLeakyReLU.  The conv bias always rides in the norm's mean; in inference the whole
    norm + activation + sum ride in the contraction's epilogue."""
    from .functional import so3_contract
    b, c, n, a = x.shape
    W = conv1x1.weight.view(conv1x1.out_channels, c)
    if not norm.training and not torch.is_grad_enabled():
        ep = norm.folded(pre_bias=conv1x1.bias, residual=residual)
        y = so3_contract(W, x.reshape(b, c, n * a), ep).view(b, conv1x1.out_channels, n, a)
        if ep.applied:
            return y
        return norm(y, residual=residual, pre_bias=conv1x1.bias)
    y = so3_contract(W, x.reshape(b, c, n * a)).view(b, conv1x1.out_channels, n, a)
    return norm(y, residual=residual, pre_bias=conv1x1.bias)


class InstanceNormLeakyReLU(nn.Module):
    """nn.InstanceNorm2d(affine=False) followed by leaky_relu, fused: the norm of the intra blocks
    (`self.norm = nn.InstanceNorm2d(dim_out, affine=False)`, base_so3poseconv.py:L88).  Instance
    statistics are batch statistics of a [1, B*C, P, A] view, so the same kernels serve it."""

    def __init__(self, num_features, eps=1e-5, negative_slope=0.01):
        super().__init__()
        self.num_features, self.eps, self.negative_slope = num_features, eps, negative_slope

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('InstanceNormLeakyReLU: tensor must be a CUDA(HIP) tensor (no CPU fallback)')
        b, c = x.shape[0], x.shape[1]
        ones = torch.ones(b * c, dtype=torch.float32, device=x.device)
        y = _BNAct.apply(x.contiguous().view(1, b * c, *x.shape[2:]), ones, torch.zeros_like(ones), None, None, True,
                         0.0, self.eps, self.negative_slope, False)
        return y.view_as(x)
