"""vgtk.so3conv.heads -- the heads that read the backbone's [B,C,N,A] feature map (SURVEY.md section 8(f) rows 2
and 3), mirroring the reference classes so a maintainer can swap the import:

  InvPPOutBlockOurs        SPConvNets/utils/base_so3conv.py:L842-917 (same constructor parameters, sub-module
                           names `linear.i`, `norm.i`, `attention_layer`, same outputs); the attention pooling
                           over the 60 anchors runs in one HIP pass (csrc/heads.hip)
  orbit_selection          ...pn_38_multi_stage.py:L1381-1399: distance -> arg-min orbit per slot
  slot_masked_mean         the masked per-slot point averages of SO3OutBlockRTWithMaskSep
                           (SPConvNets/models/model_utils.py:L470-484, L549-552) for ALL slots in one pass
  rotation_from_angle_axis model_utils.py angle -> R (Rodrigues), batched

The 1x1 convolutions / BatchNorms are dense torch layers (rocBLAS / MIOpen plumbing)."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _hip

_F32 = ctypes.c_float


class _AnchorAttnPool(torch.autograd.Function):
    """out[b,c,n] = sum_a x[b,c,n,a] softmax_a(logits[b,n,a] * T)  (base_so3conv.py:L905-912)."""

    @staticmethod
    def forward(ctx, x, logits, temperature):
        x, logits = x.contiguous(), logits.contiguous()
        b, c, n, na = x.shape
        out = torch.empty(b, c, n, dtype=torch.float32, device=x.device)
        conf = torch.empty(b, n, na, dtype=torch.float32, device=x.device)
        _hip.call('eap_anchor_attn_pool_fwd_f32', x, b, c, n, na, _F32(temperature), _hip._ptr(x), _hip._ptr(logits), _hip._ptr(out), _hip._ptr(conf))
        ctx.save_for_backward(x, logits)
        ctx.temperature = temperature
        ctx.mark_non_differentiable(conf)
        return out, conf

    @staticmethod
    def backward(ctx, g, _gconf):
        x, logits = ctx.saved_tensors
        b, c, n, na = x.shape
        dx = torch.empty_like(x)
        dl = torch.empty_like(logits)
        _hip.call('eap_anchor_attn_pool_bwd_f32', x, b, c, n, na, _F32(ctx.temperature), _hip._ptr(x), _hip._ptr(logits),
                  _hip._ptr(g.contiguous()), _hip._ptr(dx), _hip._ptr(dl))
        return dx, dl, None


def anchor_attention_pool(x, logits, temperature):
    """x [b,c,n,a], logits [b,n,a] (or [b,1,n,a]) -> pooled [b,c,n], confidence [b,n,a].
    The confidence is returned as data (the reference's callers use it for selection, not for a loss)."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError('anchor_attention_pool: float32 device tensors only')
    if logits.dim() == 4:
        logits = logits.squeeze(1)
    return _AnchorAttnPool.apply(x, logits, float(temperature))


class InvPPOutBlockOurs(nn.Module):
    """Per-point invariant head: 1x1-conv MLP over [B,C,N,A], then pooling over the anchors
    (base_so3conv.py:L842-917)."""

    def __init__(self, params, norm=None, pooling_method='max', sel_mode=None):
        super(InvPPOutBlockOurs, self).__init__()
        c_in = params['dim_in']
        mlp = params['mlp']
        self.outDim = params['k']
        self.linear = nn.ModuleList()
        self.norm = nn.ModuleList()
        self.sel_mode = sel_mode
        for c in mlp:
            self.linear.append(nn.Conv2d(c_in, c, 1))
            self.norm.append(nn.BatchNorm2d(c))
            c_in = c
        self.pooling_method = pooling_method if self.sel_mode is None else 'sel_mode'
        if self.pooling_method == 'attention' and self.sel_mode is None:
            self.temperature = params['temperature']
            self.attention_layer = nn.Conv2d(c_in, 1, 1)

    def forward(self, x, label=None, sel_mode_new=None):
        x_out = x.feats
        for lid, linear in enumerate(self.linear):
            x_out = F.relu(self.norm[lid](linear(x_out)))
        if self.pooling_method == 'mean':
            return x_out.mean(dim=-1)
        if self.pooling_method == 'debug':
            return x_out[..., 0].mean(2)
        if self.pooling_method == 'max':
            return x_out.max(-1)[0]
        if self.sel_mode is not None:
            if sel_mode_new is None:
                return x_out[..., self.sel_mode]
            idx = sel_mode_new.view(-1, 1, 1, 1).expand(-1, x_out.shape[1], x_out.shape[2], 1)
            return torch.gather(x_out, 3, idx).squeeze(-1)
        if self.pooling_method.startswith('attention'):
            out_feat = self.attention_layer(x_out)                       # [b,1,n,a]
            return anchor_attention_pool(x_out, out_feat, self.temperature)
        raise NotImplementedError(f"Pooling mode {self.pooling_method} is not implemented!")


def orbit_selection(minn_dist_ori_to_recon, minn_dist_recon_to_ori, slot_single_cd=0, slot_single_mode=0):
    """...pn_38_multi_stage.py:L1381-1399 on the [B,S,A] distances of
    extensions.chamfer_dist.orbit_reconstruction_distances: -> (distance [B,S] or [B], orbit index [B,S])."""
    d = minn_dist_ori_to_recon if slot_single_cd == 1 else minn_dist_ori_to_recon + minn_dist_recon_to_ori
    if slot_single_mode == 1:
        dist, orbit = torch.min(d.sum(1), dim=-1)
        return dist, orbit.unsqueeze(-1).repeat(1, d.shape[1]).contiguous()
    return torch.min(d, dim=-1)


class _SlotMaskedMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        x, mask = x.contiguous(), mask.contiguous()
        b, c, n, na = x.shape
        ns = mask.shape[1]
        inv_den = (1.0 / mask.sum(-1).clamp(min=1e-8)).contiguous()
        out = torch.empty(b, ns, c, na, dtype=torch.float32, device=x.device)
        _hip.call('eap_slot_masked_mean_fwd_f32', x, b, ns, c, n, na, _hip._ptr(x), _hip._ptr(mask), _hip._ptr(inv_den), _hip._ptr(out))
        ctx.save_for_backward(mask, inv_den)
        ctx.dims = (b, c, n, na, ns)
        return out

    @staticmethod
    def backward(ctx, g):
        mask, inv_den = ctx.saved_tensors
        b, c, n, na, ns = ctx.dims
        dx = torch.empty(b, c, n, na, dtype=torch.float32, device=g.device)
        _hip.call('eap_slot_masked_mean_bwd_f32', g, b, ns, c, n, na, _hip._ptr(g.contiguous()), _hip._ptr(mask), _hip._ptr(inv_den), _hip._ptr(dx))
        return dx, None


def slot_masked_mean(x, mask):
    """x [b,c,n,a], mask [b,s,n] (hard or soft slot weights, treated as data) ->
    [b,s,c,a] = sum_n mask x / clamp(sum_n mask, 1e-8): the masked point averages the pose head takes slot by slot
    (model_utils.py:L470-484 `x_out * mask` then `.mean(2)`-style pooling, L549-552 for the translations), for all
    slots (at most 8) in one pass over x."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError('slot_masked_mean: float32 device tensors only')
    return _SlotMaskedMean.apply(x, mask.to(torch.float32))


def rotation_from_angle_axis(angle, axis):
    """angle [...], unit axis [..., 3] -> R [..., 3, 3] (Rodrigues; the reference's compute_rotation_matrix_from_angle
    family in SPConvNets/models/model_utils.py), batched over every leading dimension."""
    c, s = torch.cos(angle), torch.sin(angle)
    x, y, z = axis.unbind(-1)
    C = 1.0 - c
    R = torch.stack([c + x * x * C, x * y * C - z * s, x * z * C + y * s,
                     y * x * C + z * s, c + y * y * C, y * z * C - x * s,
                     z * x * C - y * s, z * y * C + x * s, c + z * z * C], -1)
    return R.view(*angle.shape, 3, 3)
