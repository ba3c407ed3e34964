"""vgtk.so3conv.heads -- the heads that read the backbone's [B,C,N,A] feature map (SURVEY.md section 8(f) rows 2
and 3), mirroring the reference classes so a maintainer can swap the import:

  InvPPOutBlockOurs        SPConvNets/utils/base_so3conv.py:L842-917 (same constructor parameters, sub-module
                           names `linear.i`, `norm.i`, `attention_layer`, same outputs); the attention pooling
                           over the 60 anchors runs in one HIP pass (csrc/heads.hip)
  orbit_selection          ...pn_38_multi_stage.py:L1381-1399: distance -> arg-min orbit per slot
  slot_masked_mean         the masked per-slot point averages of SO3OutBlockRTWithMaskSep
                           (SPConvNets/models/model_utils.py:L470-484, L549-552) for ALL slots in one pass
  rotation_from_angle_axis model_utils.py angle -> R (Rodrigues), batched
  SO3OutBlockRTWithMaskSep SPConvNets/models/model_utils.py:L363-677, the pose head (rotation / translation / axis /
                           pivot regressors over the equivariant feature map): same constructor parameters,
                           sub-module and state_dict names, same output dictionary.  Its 1x1-convolution stacks run on
                           the path's own kernels (contraction GEMM + fused BatchNorm/ReLU epilogue); the small
                           regressors and the masked translation average stay torch ops.

  pose_head_over_subsets   the model's per-cloud loop around that head (...pn_38_multi_stage.py:L706-830) as one call per slot

Every 1x1 convolution over the [B,C,N,A] map (InvPPOutBlockOurs' included) is the path's contraction, every BatchNorm +
activation after one the fused epilogue; only the regressors on pooled [B,c,A] tensors are torch layers."""
import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _hip
from . import functional as L
from .blocks import _BNAct

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
        g = g.contiguous()            # bound to a local: the buffer must outlive the enqueue
        _hip.call('eap_anchor_attn_pool_bwd_f32', x, b, c, n, na, _F32(ctx.temperature), _hip._ptr(x), _hip._ptr(logits),
                  _hip._ptr(g), _hip._ptr(dx), _hip._ptr(dl))
        return dx, dl, None


def anchor_attention_pool(x, logits, temperature):
    """x [b,c,n,a], logits [b,n,a] (or [b,1,n,a]) -> pooled [b,c,n], confidence [b,n,a].
    The fused pass hands the confidence out as data; when the logits carry a gradient the returned confidence is the
    (small, [b,n,a]) torch softmax instead, so a caller that feeds it to a loss (the reference's `orbit_attn == 1`
    concatenation, ...pn_38_multi_stage.py:L612-613) gets the gradient the reference's softmax gives."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError('anchor_attention_pool: float32 device tensors only')
    if logits.dim() == 4:
        logits = logits.squeeze(1)
    out, conf = _AnchorAttnPool.apply(x, logits, float(temperature))
    if logits.requires_grad and torch.is_grad_enabled():
        conf = torch.softmax(logits * float(temperature), dim=-1)
    return out, conf


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
        if not x.feats.is_cuda:
            raise RuntimeError('InvPPOutBlockOurs: tensor must be a CUDA(HIP) tensor (no CPU fallback)')
        x_out = _unary_stack(x.feats, self.linear, self.norm)      # contraction GEMM + fused BatchNorm / ReLU epilogue
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
            out_feat = _conv1x1(self.attention_layer, x_out)             # [b,1,n,a]
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


def orbit_slot_distances(minn_dist_ori_to_recon_all_pts, minn_dist_ori_to_recon, hard_one_hot_labels, attn_ori):
    """Point-to-reconstruction distances [B,S,A,N] (the first and last outputs of
    extensions.chamfer_dist.orbit_reconstruction_distances) -> per-(slot, orbit) scalars [B,S,A], weighted over the
    points as ...pn_38_multi_stage.py:L1363-1380 does:
      hard    weights = the slot's 0/1 membership                 (minn_dist_ori_to_recon_hard)
      soft    weights = membership * attention                    (minn_dist_ori_to_recon, what orbit selection reads)
      all     weights = attention, on the unmasked distances      (minn_dist_ori_to_recon_all_pts)
    each a weighted sum over N divided by max(sum of weights, 1e-8).  hard_one_hot_labels [B,N,S], attn_ori [B,S,N]."""
    hard = hard_one_hot_labels.transpose(1, 2).to(minn_dist_ori_to_recon.dtype)         # [B,S,N]

    def wmean(d, w):
        w = w.unsqueeze(2)                                                              # [B,S,1,N]
        return (d * w).sum(-1) / w.sum(-1).clamp(min=1e-8)

    return (wmean(minn_dist_ori_to_recon, hard), wmean(minn_dist_ori_to_recon, hard * attn_ori),
            wmean(minn_dist_ori_to_recon_all_pts, attn_ori))


def _anchor_rows_vectorise(na):
    """The head kernels of csrc/heads.hip and the per-cloud epilogue of csrc/bn_act.hip move 16-byte words along the anchor
    axis and keep a point's anchors in one wavefront: anchor counts that are a multiple of 4, at most 64 (12, 20, 60 ...).
    Other counts (kanchor = 1, 3) take the same expressions as device torch ops."""
    return na % 4 == 0 and 0 < na <= 64


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
        g = g.contiguous()            # bound to a local: the buffer must outlive the enqueue
        _hip.call('eap_slot_masked_mean_bwd_f32', g, b, ns, c, n, na, _hip._ptr(g), _hip._ptr(mask), _hip._ptr(inv_den), _hip._ptr(dx))
        return dx, None


class _MaskedMax(torch.autograd.Function):
    """(x * mask).max(2) with its arg-max in one pass over x (csrc/heads.hip): x [b,c,n,a], mask [b,n] 0/1 -> [b,c,a]."""

    @staticmethod
    def forward(ctx, x, mask):
        x, mask = L._aligned16(x), mask.contiguous()                 # 16-byte loads along the anchor axis
        b, c, n, na = x.shape
        out = torch.empty(b, c, na, dtype=torch.float32, device=x.device)
        arg = torch.empty(b, c, na, dtype=torch.int32, device=x.device)
        _hip.call('eap_masked_max_fwd_f32', x, b, c, n, na, _hip._ptr(x), _hip._ptr(mask), _hip._ptr(out), _hip._ptr(arg))
        ctx.save_for_backward(arg, mask)
        ctx.dims = (b, c, n, na)
        return out

    @staticmethod
    def backward(ctx, g):
        arg, mask = ctx.saved_tensors
        b, c, n, na = ctx.dims
        g = g.contiguous()
        dx = torch.empty(b, c, n, na, dtype=torch.float32, device=g.device)
        _hip.call('eap_masked_max_bwd_f32', g, b, c, n, na, _hip._ptr(g), _hip._ptr(arg), _hip._ptr(mask), _hip._ptr(dx))
        return dx, None


def masked_max(x, mask):
    """x [b,c,n,a], mask [b,n] (0/1, treated as data) -> [b,c,a] = (x * mask[:, None, :, None]).max(2)[0]: the pose heads' 'max'
    pooling over a point subset without the masked copy of x."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError('masked_max: float32 device tensors only')
    if not _anchor_rows_vectorise(x.shape[3]):
        return (x * mask.to(x.dtype).view(x.shape[0], 1, x.shape[2], 1)).max(2)[0]
    return _MaskedMax.apply(x, mask.to(torch.float32))


def slot_masked_mean(x, mask):
    """x [b,c,n,a], mask [b,s,n] (hard or soft slot weights, treated as data) ->
    [b,s,c,a] = sum_n mask x / clamp(sum_n mask, 1e-8): the masked point averages the pose head takes slot by slot
    (model_utils.py:L470-484 `x_out * mask` then `.mean(2)`-style pooling, L549-552 for the translations), for all
    slots (at most 8) in one pass over x."""
    if x.dtype != torch.float32 or not x.is_cuda:
        raise RuntimeError('slot_masked_mean: float32 device tensors only')
    if mask.requires_grad:
        raise RuntimeError('slot_masked_mean: the slot weights are treated as data (no gradient reaches them); detach the mask or '
                           'use torch ops for a differentiable soft mask')
    if not _anchor_rows_vectorise(x.shape[3]) or mask.shape[1] > 8:
        m = mask.to(torch.float32)
        return torch.einsum('bcna,bsn->bsca', x, m) / m.sum(-1).clamp(min=1e-8)[:, :, None, None]
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


def rotation_axes(rots):
    """Unit rotation axis of every matrix of rots [..., 3, 3] -> [..., 3], batched; the three regimes of the reference's
    per-matrix Python loop (model_utils.py:L954-997): angle 0 -> e_x; angle pi -> from the symmetric part
    (R + R^T entries over 4 times the first axis component that is non-zero); otherwise the skew part over 2 sin(angle)."""
    R = rots.reshape(-1, 3, 3)
    t = ((R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2] - 1.0) * 0.5).clamp(-1.0, 1.0)
    sine = torch.sin(torch.acos(t))
    skew = torch.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], -1)
    generic = skew / (2.0 * sine).clamp_min(1e-30).unsqueeze(-1)
    # half-turns: the diagonal gives the squared components
    ax = torch.sqrt(((R[:, 0, 0] + 1.0) * 0.5).clamp_min(0.0))
    ay = torch.sqrt(((R[:, 1, 1] + 1.0) * 0.5).clamp_min(0.0))
    az = torch.sqrt(((R[:, 2, 2] + 1.0) * 0.5).clamp_min(0.0))
    zero = torch.zeros_like(ax)
    from_x = torch.stack([ax, (R[:, 0, 1] + R[:, 1, 0]) / (4.0 * ax).clamp_min(1e-30), (R[:, 0, 2] + R[:, 2, 0]) / (4.0 * ax).clamp_min(1e-30)], -1)
    from_y = torch.stack([zero, ay, (R[:, 1, 2] + R[:, 2, 1]) / (4.0 * ay).clamp_min(1e-30)], -1)
    from_z = torch.stack([zero, zero, az], -1)
    half = torch.where((ax > 1e-8).unsqueeze(-1), from_x, torch.where((ay > 1e-8).unsqueeze(-1), from_y, from_z))
    e_x = torch.tensor([1.0, 0.0, 0.0], dtype=R.dtype, device=R.device).expand_as(generic)
    out = torch.where(((t - 1.0).abs() < 1e-8).unsqueeze(-1), e_x, torch.where(((t + 1.0).abs() < 1e-8).unsqueeze(-1), half, generic))
    return out.reshape(*rots.shape[:-2], 3)


def compute_rotation_matrix_from_angle(anchors, angles, defined_axis=None):
    """anchors [na,3,3], angles [n,na,1] (or [n,na]) -> [n,na,3,3]: rotation by `angles` about `defined_axis` ([1,3] /
    [3] shared, or [n,na,3]) or, by default, about every anchor's own rotation axis.  Same signature and values as
    the reference's function (SPConvNets/models/model_utils.py:L1000-1043), which keeps the axis-norm-dependent form
    m_00 = u^2 + (v^2 + w^2) cos: equal to Rodrigues' formula for a unit axis, and followed term by term here so that a
    predicted axis that is unit only to rounding gives the reference's matrix."""
    ang = angles.squeeze(-1) if angles.dim() == 3 else angles
    axes = rotation_axes(anchors) if defined_axis is None else defined_axis
    if axes.dim() == 1:
        axes = axes.unsqueeze(0)
    if axes.dim() == 2:
        u, v, w = (axes[:, i].unsqueeze(0) for i in range(3))
    else:
        u, v, w = axes.unbind(-1)
    c, s = torch.cos(ang), torch.sin(ang)
    k = 1.0 - c
    rows = [u * u + (v * v + w * w) * c, u * v * k - w * s, u * w * k + v * s,
            u * v * k + w * s, v * v + (u * u + w * w) * c, v * w * k - u * s,
            u * w * k - v * s, v * w * k + u * s, w * w + (u * u + v * v) * c]
    rows = [r.expand_as(ang) if r.shape != ang.shape else r for r in rows]
    return torch.stack(rows, -1).reshape(*ang.shape, 3, 3)


def pointwise_conv(conv, x, add_bias=True):
    """nn.Conv2d(c, o, 1) on x [b,c,n,a] as the path's contraction (no vendor convolution, no layout transposes; narrow
    outputs -- 3 translation components, 1 attention logit -- are a single streaming read of x).  add_bias=False: the
    caller folds the bias into the BatchNorm that follows (BatchNormLeakyReLU.forward(..., pre_bias=conv.bias))."""
    b, c, n, a = x.shape
    y = L.so3_contract(conv.weight.view(conv.out_channels, c), x.reshape(b, c, n * a)).view(b, conv.out_channels, n, a)
    return y if (conv.bias is None or not add_bias) else y + conv.bias.view(1, -1, 1, 1)


_conv1x1 = pointwise_conv


def _dense_on_pooled_and_shared(d0, pooled, shared):
    """conv1x1(cat([pooled broadcast over the points, shared], dim=1)) of the dense translation branch
    (model_utils.py:L537-548) without the concatenated tensor: the pooled half of the input is constant over the points, so
    its product with the first half of the weights is a [b, o, a] term added to the contraction of the other half
    (half the GEMM, no 2 GB cat).  pooled [b,c,a], shared [b,c,n,a] -> [b,o,n,a] (bias not added)."""
    b, c, n, a = shared.shape
    W = d0.weight.view(d0.out_channels, 2 * c)
    y = L.so3_contract(W[:, c:], shared.reshape(b, c, n * a)).view(b, d0.out_channels, n, a)
    return y + torch.einsum('oc,bca->boa', W[:, :c], pooled).unsqueeze(2)


def _unary_stack(x, linears, norms):
    """relu(norm_i(conv1x1_i(x))) for every layer of a ModuleList pair (model_utils.py:L478-485): the 1x1 convolution
    is the path's contraction GEMM (csrc/gemm_dma_f32.hip through so3_contract), BatchNorm + ReLU the fused block
    epilogue (csrc/bn_act.hip, leaky slope 0) on the nn.BatchNorm2d's own parameters and running statistics."""
    for lid, linear in enumerate(linears):
        b, c, n, a = x.shape
        y = L.so3_contract(linear.weight.view(linear.out_channels, c), x.reshape(b, c, n * a)).view(b, linear.out_channels, n, a)
        fused = norms is not None and (n * a) % 4 == 0 and norms[lid].momentum is not None and norms[lid].track_running_stats
        if linear.bias is not None and not fused:
            y = y + linear.bias.view(1, -1, 1, 1)
        if norms is not None:
            bn = norms[lid]
            if not fused:                 # the fused epilogue moves 16-byte words; odd row lengths take the torch modules
                x = F.relu(bn(y))
                continue
            if bn.training:
                bn.num_batches_tracked.add_(1)
            # the conv bias rides into the BatchNorm as a shift of its mean (no pass of its own over the feature map)
            x = _BNAct.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps, 0.0, False,
                             None, linear.bias)
        else:
            x = F.relu(y)
    return x


class SO3OutBlockRTWithMaskSep(nn.Module):
    """The pose head of the articulated models (SPConvNets/models/model_utils.py:L363-677): two 1x1-conv stacks over
    the equivariant feature map (rotation branch on `x.feats`, translation branch on `trans_feats`), point pooling, a
    per-anchor rotation regressor (quaternion or angle), a dense per-point translation regressor rotated by the
    anchors and averaged over the (masked) points, optional axis / pivot-point / central-point regressors.
    Constructor parameters, sub-module names (state_dict keys) and the returned dictionary are the reference's."""

    def __init__(self, params, norm=None, pooling_method='mean', global_scalar=False, use_anchors=False,
                 feat_mode_num=60, num_heads=1, pred_R=True, representation='quat', num_in_channels=None, c_in_rot=None,
                 c_in_trans=None, pred_axis=False, pred_pv_points=False, pv_points_in_dim=None, pred_central_points=False,
                 central_points_in_dim=None, mtx_based_axis_regression=False):
        super(SO3OutBlockRTWithMaskSep, self).__init__()
        from .modules import PointnetSO3Conv
        c_in = params['dim_in'] if num_in_channels is None else num_in_channels
        mlp = params['mlp']
        na = params['kanchor']
        self.linear = nn.ModuleList()
        self.trans_linear = nn.ModuleList()
        self.temperature = params['temperature']
        self.global_scalar = global_scalar
        self.use_anchors = use_anchors
        self.feat_mode_num = feat_mode_num
        self.num_heads = num_heads
        self.representation = representation
        self.pred_axis = pred_axis
        self.pred_pv_points = pred_pv_points
        self.pred_central_points = pred_central_points
        self.pv_points_in_dim = mlp[-1] if pv_points_in_dim is None else pv_points_in_dim
        self.central_points_in_dim = mlp[-1] if central_points_in_dim is None else central_points_in_dim
        self.mtx_based_axis_regression = mtx_based_axis_regression
        self.pred_R = pred_R
        self.norm = nn.ModuleList() if norm is not None else None
        self.trans_norm = nn.ModuleList() if norm is not None else None
        self.pooling_method = pooling_method
        if self.pooling_method == 'pointnet':
            self.pointnet = PointnetSO3Conv(mlp[-1], mlp[-1], na)
        heads = 1 if self.feat_mode_num < 2 else num_heads
        if self.pred_R:
            self.regressor_layer = nn.Conv1d(mlp[-1], (4 if self.representation == 'quat' else 1) * heads, 1)
        if self.pred_axis:
            self.axis_regressor_layer = nn.Conv1d(mlp[-1], (4 if self.mtx_based_axis_regression else 3) * heads, 1)
        if self.pred_pv_points:
            self.pvp_regressor_layer = nn.Sequential(nn.Conv1d(self.pv_points_in_dim, 3 * heads, 1), nn.Sigmoid())
        if self.pred_central_points:
            self.central_point_regressor_layer = nn.Sequential(nn.Conv1d(self.central_points_in_dim, 3 * heads, 1), nn.Sigmoid())
        if self.global_scalar:
            self.regressor_scalar_layer = nn.Conv1d(mlp[-1], 1 * num_heads, 1)
        c_in = c_in_rot if c_in_rot is not None else c_in
        for c in mlp:
            self.linear.append(nn.Conv2d(c_in, c, 1))
            if norm is not None:
                self.norm.append(nn.BatchNorm2d(c))
            c_in = c
        c_in = c_in_trans if c_in_trans is not None else params['dim_in'] if num_in_channels is None else num_in_channels
        for c in mlp:
            self.trans_linear.append(nn.Conv2d(c_in, c, 1))
            if norm is not None:
                self.trans_norm.append(nn.BatchNorm2d(c))
            c_in = c
        self.regressor_dense_layer = nn.Sequential(nn.Conv2d(2 * mlp[-1], mlp[-1], 1), nn.BatchNorm2d(mlp[-1]),
                                                   nn.LeakyReLU(inplace=True), nn.Conv2d(c, 3 * num_heads, 1))

    def _pool(self, feats, xyz, mask):
        """-> (pooled [b,c,a], the per-point features the later layers see).  With 'max' pooling and a mask the
        reference zeroes the masked points IN PLACE (model_utils.py:L530-533), so its dense translation branch reads the
        zeroed map too; reproduced here."""
        if self.pooling_method == 'mean':
            return feats.mean(2), feats
        if self.pooling_method == 'max':
            if mask is not None:
                feats = feats * (mask >= 0.5).to(feats.dtype).unsqueeze(1).unsqueeze(-1)
            return feats.max(2)[0], feats
        if self.pooling_method == 'pointnet':
            from ..spconv import SphericalPointCloud
            return self.pointnet(SphericalPointCloud(xyz, feats, None)), feats
        raise NotImplementedError(f"Pooling mode {self.pooling_method} is not implemented!")

    def forward(self, x, mask, trans_feats, trans_xyz=None, anchors=None, soft_mask=None, pre_feats=None, use_offset=True,
                pred_pv_poitns_in_feats=None, pred_central_points_in_feats=None, pred_axis_in_feats=None):
        import math
        x_out = x.feats
        if not x_out.is_cuda:
            raise RuntimeError('SO3OutBlockRTWithMaskSep: tensors must be CUDA(HIP) tensors (no CPU fallback)')
        if mask is not None:
            x_out = x_out * mask.unsqueeze(1).unsqueeze(-1)
        nb, _, _, na = x_out.shape
        x_out, _ = self._pool(_unary_stack(x_out, self.linear, self.norm), x.xyz, mask)               # [b, c, a]
        trans_x_xyz = x.xyz if trans_xyz is None else trans_xyz
        trans_shared_feat = _unary_stack(trans_feats, self.trans_linear, self.trans_norm)            # [b, c, n, a]
        trans_x_out, trans_shared_feat = self._pool(trans_shared_feat, trans_x_xyz, mask)

        # dense branch: per point and anchor, 3 * num_heads translation components.  Layers 0-2 of the Sequential
        # (conv, BatchNorm, LeakyReLU(0.01)) go through the fused epilogue
        d0, dbn, dact, d1 = self.regressor_dense_layer
        y = _dense_on_pooled_and_shared(d0, trans_x_out, trans_shared_feat.contiguous())
        if dbn.training:
            dbn.num_batches_tracked.add_(1)
        y = _BNAct.apply(y, dbn.weight, dbn.bias, dbn.running_mean, dbn.running_var, dbn.training, dbn.momentum, dbn.eps,
                         dact.negative_slope, False, None, d0.bias)
        t_out = _conv1x1(d1, y)                                                                       # [b, 3 h, n, a]: 3 h output channels
        t_out = t_out.reshape((nb, self.num_heads, 3) + t_out.shape[-2:])                             # [b, h, 3, n, a]
        if self.global_scalar:
            y_t = self.regressor_scalar_layer(trans_shared_feat.max(dim=-1)[0]).reshape(nb, self.num_heads, -1)
            y_t = F.normalize(t_out, p=2, dim=2) * y_t.unsqueeze(2).unsqueeze(-1)
            if self.use_anchors:
                y_t = torch.matmul(anchors.unsqueeze(1), y_t.permute(0, 1, 4, 2, 3).contiguous())
            else:
                y_t = y_t.permute(0, 1, 4, 2, 3).contiguous()
        else:
            y_t = torch.matmul(anchors.unsqueeze(1), t_out.permute(0, 1, 4, 2, 3).contiguous())       # [b, h, a, 3, n]
        if use_offset:
            y_t = y_t + trans_x_xyz.unsqueeze(1).unsqueeze(1)
        wts = mask if mask is not None else soft_mask
        if wts is not None:
            wts = wts.unsqueeze(1).unsqueeze(1).unsqueeze(1)
            y_t = torch.sum(y_t * wts, dim=-1) / torch.clamp(torch.sum(wts, dim=-1), min=1e-8)
            y_t = y_t.contiguous().permute(0, 1, 3, 2).contiguous()
        else:
            y_t = y_t.mean(dim=-1).permute(0, 1, 3, 2).contiguous()                                   # [b, h, 3, a]

        output = {}
        if self.pred_R:
            if pre_feats is not None:
                pre_feats = pre_feats.contiguous().unsqueeze(-1).repeat(1, 1, x_out.size(-1)).contiguous()
                y = self.regressor_layer(pre_feats)
            else:
                y = self.regressor_layer(x_out)
        else:
            y = None
        output['R'] = y
        output['T'] = y_t
        if self.pred_axis:
            pred_axis_in_feats = x_out if pred_axis_in_feats is None else pred_axis_in_feats
            y_axis = self.axis_regressor_layer(pred_axis_in_feats)
            if self.mtx_based_axis_regression:
                e = torch.sigmoid(y_axis.contiguous().view(y_axis.size(0), self.num_heads, 4, na))
                alpha, beta = e[:, :, 0, :].unsqueeze(-2), e[:, :, 1, :].unsqueeze(-2)
                maxx_angle = 45.0
                y_angle = (maxx_angle / 180.) * beta * math.pi + ((90.0 - maxx_angle) / 180.0) * math.pi
                xz_len = torch.cos(y_angle)
                y_axis = torch.cat([torch.cos(alpha * 2.0 * math.pi) * xz_len, torch.sin(y_angle),
                                    torch.sin(alpha * 2.0 * math.pi) * xz_len], dim=-2)
            else:
                y_axis = y_axis / torch.clamp(torch.norm(y_axis, dim=1, keepdim=True, p=2), min=1e-6)
            output['axis'] = y_axis
        else:
            output['axis'] = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float32, device=x_out.device).unsqueeze(0).unsqueeze(
                -1).contiguous().repeat(x_out.size(0), 1, y.size(-1))
        if self.pred_pv_points:
            pred_pv_poitns_in_feats = x_out if pred_pv_poitns_in_feats is None else pred_pv_poitns_in_feats
            output['pv_points'] = self.pvp_regressor_layer(pred_pv_poitns_in_feats)
        if self.pred_central_points:
            pred_central_points_in_feats = x_out if pred_central_points_in_feats is None else pred_central_points_in_feats
            output['central_points'] = self.central_point_regressor_layer(pred_central_points_in_feats)
        if self.num_heads == 1:
            for key, value in output.items():
                if value is not None:
                    output[key] = value.squeeze(1)
        return output


class _SubsetBNAct(torch.autograd.Function):
    """leaky_relu(BatchNorm2d(y + bias)) with the statistics of every cloud's member points alone -- what B separate
    batch-1 calls on the gathered subsets compute -- as two passes over y forward (masked statistics, apply) and two
    backward (reduce, apply): csrc/bn_act.hip, the eap_bn_act_cloud_* entries.  Every point is normalised; the head's
    later masked poolings drop the non-members.  The running statistics receive the B momentum updates in cloud
    order, as the loop would apply them."""

    @staticmethod
    def forward(ctx, y, bias, mask, weight, beta, running_mean, running_var, training, momentum, eps, slope):
        y = y.contiguous()
        b, c, p, na = y.shape
        n = p * na
        if training:
            cnt = (mask.sum(1, dtype=torch.float64) * na).view(b, 1)                # member points x anchors
            pivot = y.view(b, c, n)[0, :, 0].double().view(1, c)
            s1, s2 = _hip.bn_stats_masked(y, b, c, n, na, mask)                     # of y - pivot, [b, c] float64
            d = s1 / cnt
            var = (s2 / cnt - d * d).clamp_min(0.0)                                 # biased, as BatchNorm normalises with
            mean = d + pivot                                                        # of y (without the bias)
            with torch.no_grad():
                full = (mean + bias.double().view(1, c)) if bias is not None else mean
                unb = var * (cnt / (cnt - 1.0).clamp_min(1.0))
                # B sequential updates r <- (1 - mom) r + mom v_i in closed form
                keep = (1.0 - momentum) ** torch.arange(b - 1, -1, -1, device=y.device, dtype=torch.float64).view(b, 1)
                running_mean.mul_((1.0 - momentum) ** b).add_((momentum * (keep * full).sum(0)).to(running_mean.dtype))
                running_var.mul_((1.0 - momentum) ** b).add_((momentum * (keep * unb).sum(0)).to(running_var.dtype))
        else:
            mean = (running_mean.double() - (bias.double() if bias is not None else 0.0)).view(1, c).expand(b, c)
            var = running_var.double().view(1, c).expand(b, c)
        invstd = torch.rsqrt(var + eps)
        scale64 = weight.double().view(1, c) * invstd
        scale = scale64.float().contiguous()
        shift = (beta.double().view(1, c) - mean * scale64).float().contiguous()
        out = _hip.bn_act_cloud_fwd(y, b, c, n, scale, shift, slope)
        ctx.save_for_backward(y, scale, shift, mean.float().contiguous(), invstd.float().contiguous(), mask)
        ctx.training, ctx.slope, ctx.dims, ctx.has_bias = training, slope, (b, c, n, na), bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        y, scale, shift, mean, invstd, mask = ctx.saved_tensors
        b, c, n, na = ctx.dims
        g = g.contiguous()
        sg, sgx = _hip.bn_act_cloud_bwd_reduce(g, y, b, c, n, scale, shift, mean, invstd, ctx.slope)     # [b, c] float64
        g_y = None
        if ctx.needs_input_grad[0]:
            if ctx.training:
                cnt = (mask.sum(1, dtype=torch.float64) * na).view(b, 1)
                k2 = (scale.double() * sg / cnt).float().contiguous()
                k3 = (scale.double() * sgx / cnt).float().contiguous()
            else:
                k2 = torch.zeros_like(scale)
                k3 = torch.zeros_like(scale)
            g_y = _hip.bn_act_cloud_bwd_apply(g, y, b, c, n, na, scale, shift, mean, invstd, k2, k3, mask, ctx.slope)
        # the bias is absorbed by the batch mean in training mode (its gradient is exactly zero); in eval mode it shifts the
        # pre-activation like beta does
        g_bias = None
        if ctx.has_bias:
            g_bias = torch.zeros(c, dtype=torch.float32, device=g.device) if ctx.training else (sg * scale.double()).sum(0).float()
        return g_y, g_bias, None, sgx.sum(0).float(), sg.sum(0).float(), None, None, None, None, None, None


def _subset_batchnorm_act_torch(y, bias, mask, bn, slope):
    """_SubsetBNAct as device torch ops, for anchor counts the 16-byte kernels do not take (see _anchor_rows_vectorise)."""
    b, c, p, na = y.shape
    if bias is not None:
        y = y + bias.view(1, c, 1, 1)
    if bn.training:
        m = mask.view(b, 1, p, 1)
        cnt = (mask.sum(1) * na).view(b, 1)
        ym = y * m
        mean = ym.sum((2, 3)) / cnt
        var = ((ym * ym).sum((2, 3)) / cnt - mean * mean).clamp_min(0.0)
        with torch.no_grad():
            unb = var * (cnt / (cnt - 1.0).clamp_min(1.0))
            for i in range(b):                                              # the loop's B momentum updates, in cloud order
                bn.running_mean.mul_(1.0 - bn.momentum).add_(mean[i], alpha=bn.momentum)
                bn.running_var.mul_(1.0 - bn.momentum).add_(unb[i], alpha=bn.momentum)
        mean, var = mean[:, :, None, None], var[:, :, None, None]
    else:
        mean, var = bn.running_mean.view(1, c, 1, 1), bn.running_var.view(1, c, 1, 1)
    z = (y - mean) * torch.rsqrt(var + bn.eps) * bn.weight.view(1, c, 1, 1) + bn.bias.view(1, c, 1, 1)
    return F.leaky_relu(z, slope) if slope else F.relu(z)


def _subset_batchnorm_act(y, bias, mask, bn, slope):
    """act(bn(y + bias)) per cloud subset; y [B,C,P,A] raw contraction output, mask [B,P] float 0/1, slope 0 = relu."""
    if bn.training:
        with torch.no_grad():
            bn.num_batches_tracked.add_(y.shape[0])
    if bn.momentum is None:
        raise NotImplementedError('pose_head_over_subsets: cumulative-average BatchNorm (momentum=None) is not batched')
    if not _anchor_rows_vectorise(y.shape[3]):
        return _subset_batchnorm_act_torch(y, bias, mask, bn, slope)
    return _SubsetBNAct.apply(y, bias, mask, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps, slope)


def _masked_unary_stack(x, mask, linears, norms):
    """_unary_stack for B independent batch-1 calls on point SUBSETS at once: x [B,C,P,A] full clouds, mask [B,P] the
    0/1 membership.  Training-mode BatchNorm statistics are those of each cloud's subset alone (what a batch-1 call on
    the gathered subset computes), and the running statistics receive the B momentum updates in cloud order."""
    for lid, linear in enumerate(linears):
        b, c, n, a = x.shape
        y = L.so3_contract(linear.weight.view(linear.out_channels, c), x.reshape(b, c, n * a)).view(b, linear.out_channels, n, a)
        if norms is not None:
            x = _subset_batchnorm_act(y, linear.bias, mask, norms[lid], 0.0)
        else:
            x = F.relu(y + linear.bias.view(1, -1, 1, 1) if linear.bias is not None else y)
    return x


def pose_head_over_subsets(head, feats, xyz, member, anchors, use_offset=True):
    """SO3OutBlockRTWithMaskSep on one point subset per cloud, all clouds at once -- the batched form of the model's
    inner loop `for i_bz in range(bz): head(SphericalPointCloud(xyz[i_bz, subset], feats[i_bz, :, subset]), mask=None,
    trans_feats=..., trans_xyz=..., anchors=...)` (...pn_38_multi_stage.py:L706-830; one call of this function per
    slot replaces the loop over the batch; every slot has its own head module, as in the reference).
    feats [B,C,P,A], xyz [B,3,P], member [B,P] in {0,1} (at least one point per cloud), anchors [A,3,3] or [B,A,3,3]
    -> the head's output dictionary with batch dimension B; identical (to rounding) to B separate batch-1 calls on the
    gathered subsets, BatchNorm statistics and running-statistic updates included (tests/test_gpu_lists_and_modules.py).
    Supported configuration: pooling 'mean' or 'max', no global_scalar."""
    if head.pooling_method not in ('mean', 'max') or head.global_scalar:
        raise NotImplementedError('pose_head_over_subsets: mean / max pooling without the global scalar only')
    import math
    b, _, n, na = feats.shape
    mask = member.to(feats.dtype).contiguous()                              # [B, P]
    m = mask.view(b, 1, n, 1)
    npts = mask.sum(1, keepdim=True)                                        # [B, 1]

    def pool(f):
        if head.pooling_method == 'max':
            return masked_max(f, mask)                                     # [B, c, A]: one pass over f (csrc/heads.hip)
        return slot_masked_mean(f, mask.view(b, 1, n))[:, 0]

    x_out = pool(_masked_unary_stack(feats, mask, head.linear, head.norm))                       # [B, c, A]
    shared = _masked_unary_stack(feats, mask, head.trans_linear, head.trans_norm)              # [B, c, P, A]
    trans_x_out = pool(shared)
    d0, dbn, dact, d1 = head.regressor_dense_layer
    y = _dense_on_pooled_and_shared(d0, trans_x_out, shared)
    y = _subset_batchnorm_act(y, d0.bias, mask, dbn, dact.negative_slope)
    t_out = _conv1x1(d1, y).reshape(b, head.num_heads, 3, n, na)
    A = anchors if anchors.dim() == 4 else anchors.unsqueeze(0)
    y_t = torch.matmul(A.unsqueeze(1), t_out.permute(0, 1, 4, 2, 3).contiguous())                # [B, h, A, 3, P]
    if use_offset:
        y_t = y_t + xyz.unsqueeze(1).unsqueeze(1)
    w = m.view(b, 1, 1, 1, n)
    y_t = ((y_t * w).sum(-1) / npts.view(b, 1, 1, 1)).permute(0, 1, 3, 2).contiguous()            # [B, h, 3, A]
    output = {'R': head.regressor_layer(x_out) if head.pred_R else None, 'T': y_t}
    if head.pred_axis:
        y_axis = head.axis_regressor_layer(x_out)
        if head.mtx_based_axis_regression:
            e = torch.sigmoid(y_axis.contiguous().view(b, head.num_heads, 4, na))
            alpha, beta = e[:, :, 0, :].unsqueeze(-2), e[:, :, 1, :].unsqueeze(-2)
            y_angle = (45.0 / 180.) * beta * math.pi + (45.0 / 180.0) * math.pi
            xz_len = torch.cos(y_angle)
            y_axis = torch.cat([torch.cos(alpha * 2.0 * math.pi) * xz_len, torch.sin(y_angle), torch.sin(alpha * 2.0 * math.pi) * xz_len], dim=-2)
        else:
            y_axis = y_axis / torch.clamp(torch.norm(y_axis, dim=1, keepdim=True, p=2), min=1e-6)
        output['axis'] = y_axis
    else:
        output['axis'] = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float32, device=feats.device).view(1, 3, 1).repeat(b, 1, na)
    if head.pred_pv_points:
        output['pv_points'] = head.pvp_regressor_layer(x_out)
    if head.pred_central_points:
        output['central_points'] = head.central_point_regressor_layer(x_out)
    if head.num_heads == 1:
        output = {k: (v.squeeze(1) if v is not None else None) for k, v in output.items()}
    return output


class _GatherPoints(torch.autograd.Function):
    """feats [b,c,n,a], rows int32 [b,m] (distinct point indices per cloud, -1 = padding) -> [b,c,m,a]; the backward scatters
    the gradient back to the named points (csrc/inv_lists.hip: eap_rows_gather_f32 / eap_rows_scatter_f32)."""

    @staticmethod
    def forward(ctx, feats, rows):
        feats = feats.contiguous()
        ctx.save_for_backward(rows)
        ctx.n = feats.shape[2]
        return _hip.rows_gather(feats, rows, rows.shape[1])

    @staticmethod
    def backward(ctx, g):
        rows, = ctx.saved_tensors
        g = g.contiguous()
        return _hip.rows_scatter(g, rows, ctx.n), None


def slot_point_groups(labels, n_slots, multiple=32):
    """labels [B,P] (slot of every point) -> per slot the member points of every cloud, compacted: a list of int32 [B, cap_s]
    (ascending point indices, -1 padded), cap_s = the largest membership of the slot over the clouds rounded up to `multiple`
    (32 points x 60 anchors = whole 128-column tiles for the contraction kernels).  A slot nobody chose in a cloud falls back to
    the whole cloud, as the model does.  ONE device -> host read (the capacities) for all slots."""
    b, p = labels.shape
    member = labels.unsqueeze(0) == torch.arange(n_slots, device=labels.device).view(n_slots, 1, 1)        # [S,B,P]
    member = member | (member.sum(2, keepdim=True) == 0)
    counts = member.sum(2)                                                                                  # [S,B]
    caps = counts.max(1).values.tolist()
    order = torch.sort(member.to(torch.int8), dim=2, descending=True, stable=True).indices                 # members first, ascending
    groups = []
    for s_, cap in enumerate(caps):
        cap = min(-(-int(cap) // multiple) * multiple, -(-p // multiple) * multiple)
        rows = order[s_, :, :min(cap, p)].to(torch.int32)
        if cap > p:
            rows = torch.cat([rows, rows.new_full((b, cap - p), -1)], 1)
        live = torch.arange(cap, device=labels.device).view(1, cap) < counts[s_].view(b, 1)
        groups.append(torch.where(live, rows, rows.new_full((), -1)).contiguous())
    return groups


def pose_head_over_slot_groups(heads, feats, xyz, labels, anchors, use_offset=True):
    """The model's loop `for slot: for cloud: head_slot(points of the slot in the cloud)` (...pn_38_multi_stage.py:L706-830) with
    the work of P points per cloud, not slots x P: every slot's member points are compacted into their own [B,C,cap_s,A]
    tensor (slot_point_groups + one gather pass) and run through pose_head_over_subsets, whose per-cloud statistics and pooling
    already ignore the padding.  heads: one module per slot; -> list of the heads' output dictionaries."""
    outs = []
    p = feats.shape[2]
    for s_, (head, rows) in enumerate(zip(heads, slot_point_groups(labels, len(heads)))):
        if rows.shape[1] * 10 >= p * 9:
            # some cloud gives (nearly) all of its points to this slot -- or none, which falls back to the whole cloud: nothing to
            # save, so no compacted copy of the feature map either: the masked form on the full clouds
            member = labels == s_
            outs.append(pose_head_over_subsets(head, feats, xyz, member | (member.sum(1, keepdim=True) == 0), anchors, use_offset))
            continue
        member = rows >= 0
        sel = rows.clamp(min=0).long()
        sub_feats = _GatherPoints.apply(feats, rows)
        sub_xyz = torch.gather(xyz, 2, sel.unsqueeze(1).expand(-1, xyz.shape[1], -1)).contiguous()
        outs.append(pose_head_over_subsets(head, sub_feats, sub_xyz, member, anchors, use_offset))
    return outs
