"""vgtk.spconv.modules -- the S^2 ("ZP") convolution layers (reference: vgtk/vgtk/spconv/modules.py:L17-161; used by
SPConvNets/models/unsup_seg_basicconv_pn.py:L38 `zptk.InterZPConv`).  Class names, constructor arguments, parameter and
buffer names (`basic_conv.W`, `basic_conv.bias`, `anchors`, `anchor_out`, `kernels`, `intra_idx`, `intra_w`, `idx`, `w`)
and return structures are the reference's, so its checkpoints and callers fit; the grouping contractions run in the HIP
zpconv kernels (csrc/zpconv*.hip) and the dense layer on the contraction GEMM (csrc/gemm_*.hip)."""
import torch
import torch.nn as nn

import vgtk.so3conv.functional as LL
from . import functional as L
from .base import SphericalPointCloud


class BasicZPConv(nn.Module):
    """Dense layer over (channel, kernel) pairs: [b, c1, k, p, a] -> [b, c2, p, a], y = W x + bias with W [c2, c1 * k]
    (Xavier-normal over [c2, c1, k], relu gain) and bias [1, c2, 1] = 1e-3.  debug=True: W = ones as a buffer, no bias."""

    def __init__(self, dim_in, dim_out, kernel_size, debug=False):
        super().__init__()
        self.dim_in, self.dim_out, self.kernel_size = dim_in, dim_out, kernel_size
        fan = dim_in * kernel_size
        if debug:
            self.register_buffer('W', torch.ones(dim_out, fan))
            return
        weight = nn.init.xavier_normal_(torch.empty(dim_out, dim_in, kernel_size), gain=nn.init.calculate_gain('relu'))
        self.W = nn.Parameter(weight.reshape(dim_out, fan))
        self.bias = nn.Parameter(torch.full((1, dim_out, 1), 1e-3))

    def forward(self, x):
        b, _, _, p, a = x.shape
        y = LL.so3_contract(self.W, x.reshape(b, self.dim_in * self.kernel_size, p * a))
        bias = getattr(self, 'bias', None)
        return (y if bias is None else y + bias).view(b, self.dim_out, p, a)


class IntraZPConv(nn.Module):
    """Convolution over the anchor sphere at every point: [b, c1, p, a_in] -> [b, c1, k, p, a_out] -> [b, c2, p, a_out];
    each output anchor mixes its `anchor_nn` nearest input anchors through `kernel_size` angular bins."""

    def __init__(self, dim_in, dim_out, kernel_size, aperture, sigma, anchor_nn, anchor_in, anchor_out=None):
        super().__init__()
        src = L.get_anchors(anchor_in)
        dst = src if anchor_out is None else L.get_anchors(anchor_out)
        bins = L.get_intra_kernels(aperture, kernel_size)
        self.dim_in, self.dim_out, self.kernel_size = dim_in, dim_out, bins.shape[0]
        self.aperture, self.sigma, self.anchor_nn = aperture, sigma, anchor_nn
        self.basic_conv = BasicZPConv(dim_in, dim_out, self.kernel_size)
        idx, w = L.get_intra_kernel_weights(src, dst, bins, anchor_nn, aperture, sigma)
        for name, table in (('anchor_out', dst), ('kernels', bins), ('intra_idx', idx), ('intra_w', w)):
            self.register_buffer(name, table)

    def forward(self, x):
        grouped = L.intra_zpconv_grouping_naive(self.intra_idx, self.intra_w, x.feats)
        return SphericalPointCloud(x.xyz, self.basic_conv(grouped), self.anchor_out)


class InterZPConv(nn.Module):
    """Convolution over the spatial neighbourhood: [b, c1, p1, a] -> [b, c1, k, p2, a] -> [b, c2, p2, a].  The dense
    layer is sized by the ANCHOR count, as in the reference (modules.py:L121): the grouping hands over its weights with
    the anchor and kernel axes exchanged (vgtk.spconv.functional.inter_zpconv_grouping)."""

    def __init__(self, dim_in, dim_out, kernel_size, stride, radius, aperture, sigma, anchors_dim, n_neighbor, anchor_nn,
                 multiplier=3, lazy_sample=True):
        super().__init__()
        rings = L.get_kernel_rings_np(radius, aperture, kernel_size, multiplier=multiplier)
        self.dim_in, self.dim_out, self.kernel_size, self.stride = dim_in, dim_out, rings.shape[0], stride
        self.radius, self.aperture, self.sigma = radius, aperture, sigma
        self.n_neighbor, self.anchor_nn, self.lazy_sample = n_neighbor, anchor_nn, lazy_sample
        self.basic_conv = BasicZPConv(dim_in, dim_out, anchors_dim)
        self.register_buffer('anchors', L.get_anchors(anchors_dim))
        self.register_buffer('kernels', torch.from_numpy(rings))

    def forward(self, x, inter_idx=None, inter_w=None):
        inter_idx, inter_w, centres, grouped = L.inter_zpconv_grouping(
            x.xyz, x.feats, self.stride, self.n_neighbor, self.anchors, self.kernels, self.anchor_nn, self.radius,
            self.aperture, self.sigma, inter_idx, inter_w, self.lazy_sample)
        return inter_idx, inter_w, SphericalPointCloud(centres, self.basic_conv(grouped), self.anchors)


class AnchorProp(nn.Module):
    """Features carried from one anchor set to another by k-nearest-anchor interpolation: [b, c, p, a1] -> [b, c, p, a2]."""

    def __init__(self, anchor_in, anchor_out, sigma, k=6):
        super().__init__()
        self.sigma = sigma
        dst = L.get_anchors(anchor_out)
        idx, w = L.compute_anchor_weights(L.get_anchors(anchor_in), dst, k=k, sigma=sigma)
        self.register_buffer('anchor_out', dst)
        self.register_buffer('idx', idx)
        self.register_buffer('w', w)

    def forward(self, x):
        return SphericalPointCloud(x.xyz, L.anchor_prop(x.feats, self.idx, self.w), self.anchor_out)
