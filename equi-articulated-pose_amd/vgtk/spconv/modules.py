"""vgtk.spconv.modules -- nn.Modules of the S^2 ("ZP") convolution (reference:
vgtk/vgtk/spconv/modules.py:L17-146).  Same class names, constructor arguments, parameter / buffer names
(`basic_conv.W`, `basic_conv.bias`, `anchor_out`, `kernels`, `intra_idx`, `intra_w`, `anchors`, `idx`, `w`)
and return structures; the grouping contractions run in the HIP zpconv kernels (csrc/zpconv*.hip) and the
dense contraction on the fp32 matrix cores (csrc/gemm_f32.hip).  None of the shipped models instantiates
these; they complete boundary B1."""
import torch
import torch.nn as nn

from vgtk.spconv import SphericalPointCloud
from . import functional as L
import vgtk.so3conv.functional as LL


class BasicZPConv(nn.Module):
    """[b, c1, k, p, a] -> [b, c2, p, a]: W [c2, c1*k] + bias [1, c2, 1] (modules.py:L17-59)."""

    def __init__(self, dim_in, dim_out, kernel_size, debug=False):
        super(BasicZPConv, self).__init__()
        self.dim_in = dim_in
        self.dim_out = dim_out
        self.kernel_size = kernel_size
        if debug:
            self.register_buffer('W', torch.ones(dim_out, dim_in * kernel_size))
        else:
            W = torch.empty(dim_out, dim_in, kernel_size)
            nn.init.xavier_normal_(W, gain=nn.init.calculate_gain('relu'))
            self.register_parameter('W', nn.Parameter(W.view(dim_out, dim_in * kernel_size)))
            self.register_parameter('bias', nn.Parameter((torch.zeros(dim_out) + 1e-3).view(1, dim_out, 1)))

    def forward(self, x):
        bs, np_, na = x.shape[0], x.shape[3], x.shape[4]
        y = LL.so3_contract(self.W, x.reshape(bs, self.dim_in * self.kernel_size, np_ * na))
        if hasattr(self, 'bias'):
            y = y + self.bias
        return y.view(bs, self.dim_out, np_, na)


class IntraZPConv(nn.Module):
    """[b, c1, p, a_in] -> [b, c1, k, p, a_out] -> [b, c2, p, a_out] (modules.py:L62-99)."""

    def __init__(self, dim_in, dim_out, kernel_size, aperture, sigma, anchor_nn, anchor_in, anchor_out=None):
        super(IntraZPConv, self).__init__()
        if anchor_out is None:
            anchor_out = anchor_in
        anchor_in = L.get_anchors(anchor_in)
        anchor_out = L.get_anchors(anchor_out)
        kernels = L.get_intra_kernels(aperture, kernel_size)
        self.dim_in = dim_in
        self.dim_out = dim_out
        self.kernel_size = kernels.shape[0]
        self.basic_conv = BasicZPConv(dim_in, dim_out, self.kernel_size)
        self.aperture = aperture
        self.sigma = sigma
        self.anchor_nn = anchor_nn
        intra_idx, intra_w = L.get_intra_kernel_weights(anchor_in, anchor_out, kernels, self.anchor_nn,
                                                        self.aperture, self.sigma)
        self.register_buffer('anchor_out', anchor_out)
        self.register_buffer('kernels', kernels)
        self.register_buffer('intra_idx', intra_idx)
        self.register_buffer('intra_w', intra_w)

    def forward(self, x):
        feats = L.intra_zpconv_grouping_naive(self.intra_idx, self.intra_w, x.feats)
        feats = self.basic_conv(feats)
        return SphericalPointCloud(x.xyz, feats, self.anchor_out)


class InterZPConv(nn.Module):
    """[b, c1, p1, a] -> [b, c1, k, p2, a] -> [b, c2, p2, a] (modules.py:L102-140).  As in the reference the
    dense layer is sized by the anchor count (`BasicZPConv(dim_in, dim_out, anchors_dim)`, L121)."""

    def __init__(self, dim_in, dim_out, kernel_size, stride, radius, aperture, sigma, anchors_dim, n_neighbor,
                 anchor_nn, multiplier=3, lazy_sample=True):
        super(InterZPConv, self).__init__()
        anchors = L.get_anchors(anchors_dim)
        kernels = L.get_kernel_rings_np(radius, aperture, kernel_size, multiplier=multiplier)
        self.dim_in = dim_in
        self.dim_out = dim_out
        self.kernel_size = kernels.shape[0]
        self.stride = stride
        self.basic_conv = BasicZPConv(dim_in, dim_out, anchors_dim)
        self.radius = radius
        self.aperture = aperture
        self.sigma = sigma
        self.n_neighbor = n_neighbor
        self.anchor_nn = anchor_nn
        self.lazy_sample = lazy_sample
        self.register_buffer('anchors', anchors)
        self.register_buffer('kernels', torch.from_numpy(kernels))

    def forward(self, x, inter_idx=None, inter_w=None):
        inter_idx, inter_w, xyz, feats = \
            L.inter_zpconv_grouping(x.xyz, x.feats, self.stride, self.n_neighbor, self.anchors, self.kernels,
                                    self.anchor_nn, self.radius, self.aperture, self.sigma, inter_idx, inter_w,
                                    self.lazy_sample)
        feats = self.basic_conv(feats)
        return inter_idx, inter_w, SphericalPointCloud(xyz, feats, self.anchors)


class AnchorProp(nn.Module):
    """[b, c, p, a1] -> [b, c, p, a2] k-NN interpolation between anchor sets (modules.py:L143-158)."""

    def __init__(self, anchor_in, anchor_out, sigma, k=6):
        super(AnchorProp, self).__init__()
        anchor_in = L.get_anchors(anchor_in)
        anchor_out = L.get_anchors(anchor_out)
        idx, w = L.compute_anchor_weights(anchor_in, anchor_out, k=k, sigma=sigma)
        self.sigma = sigma
        self.register_buffer('anchor_out', anchor_out)
        self.register_buffer('idx', idx)
        self.register_buffer('w', w)

    def forward(self, x):
        feats = L.anchor_prop(x.feats, self.idx, self.w)
        return SphericalPointCloud(x.xyz, feats, self.anchor_out)
