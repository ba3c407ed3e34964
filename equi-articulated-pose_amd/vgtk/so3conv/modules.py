"""vgtk.so3conv.modules -- nn.Modules of the SO(3) convolution (reference:
vgtk/vgtk/so3conv/modules.py).  Parameter / buffer names and shapes match the reference
(`basic_conv.W` [O, C*K], `anchors`, `kernels`, `intra_idx`) so its checkpoints load."""
import numpy as np
import torch
from .. import _hip
import torch.nn as nn

from vgtk.spconv import SphericalPointCloud, SphericalPointCloudPose
import vgtk.pc as pctk
from . import functional as L

KERNEL_CONDENSE_RATIO = 0.7


def _kernel_points(radius, kernel_size):
    """The spherical kernel-point set of a layer, scaled to KERNEL_CONDENSE_RATIO of its ball radius."""
    return L.get_sphereical_kernel_points_from_ply(KERNEL_CONDENSE_RATIO * radius, kernel_size)


def _const(module, **arrays):
    """numpy constants as (contiguous) buffers of `module` under the reference's names."""
    for name, value in arrays.items():
        module.register_buffer(name, torch.from_numpy(np.ascontiguousarray(value)))


def _attrs(module, **values):
    """Plain attributes the reference classes expose (read by model code and by repr / checkpoints tooling)."""
    for name, value in values.items():
        setattr(module, name, value)


class BasicSO3Conv(nn.Module):
    """[b, c1, k, p, a] -> [b, c2, p, a]  (modules.py:L21-55): W [c2, c1*k], no bias,
    xavier-normal with relu gain."""

    def __init__(self, dim_in, dim_out, kernel_size, debug=False):
        super().__init__()
        _attrs(self, dim_in=dim_in, dim_out=dim_out, kernel_size=kernel_size)
        if debug:
            self.register_buffer('W', torch.ones(dim_out, dim_in * kernel_size))
        else:
            init = nn.init.xavier_normal_(torch.empty(dim_out, dim_in, kernel_size), gain=nn.init.calculate_gain('relu'))
            self.W = nn.Parameter(init.view(dim_out, dim_in * kernel_size))

    def forward(self, x):
        bs, np_, na = x.shape[0], x.shape[3], x.shape[4]
        y = L.so3_contract(self.W, x.reshape(bs, self.dim_in * self.kernel_size, np_ * na))
        return y.view(bs, self.dim_out, np_, na)


class KernelPropagation(nn.Module):
    """Initial features from a point fragment (modules.py:L57-119): kernel-weight query around `n_center`
    centres (native initial_anchor_query, csrc/grouping.hip) -> BasicSO3Conv over the kernel axis."""

    def __init__(self, dim_in, dim_out, n_center, kernel_size, radius, sigma, kanchor=60):
        super().__init__()
        anchors = L.get_anchors(kanchor)
        rotated = np.transpose(anchors @ _kernel_points(radius, kernel_size).T, (2, 0, 1))          # [ks, na, 3]
        _attrs(self, radius=radius, sigma=sigma, n_center=n_center)
        _const(self, anchors=anchors, kernels=rotated)
        self.basic_conv = BasicSO3Conv(dim_in, dim_out, rotated.shape[0])

    def _subsample(self, clouds):
        return pctk.furthest_sample(clouds, self.n_center, False)[1]

    def forward(self, frag, clouds):
        """frag [m,3], clouds [b,3,n] -> SphericalPointCloud(centers [b,3,nc], feats [b,c_out,nc,na])."""
        centers = clouds if clouds.shape[2] == self.n_center else self._subsample(clouds)
        weights, hits = L.initial_anchor_query(frag, centers, self.kernels, self.radius, self.sigma)
        feats = self.basic_conv((weights / (hits + 1.0)).unsqueeze(1))
        return SphericalPointCloud(centers, feats, self.anchors)


class InterSO3Conv(nn.Module):
    """Pose-free inter conv (modules.py:L125-174)."""

    def __init__(self, dim_in, dim_out, kernel_size, stride, radius, sigma, n_neighbor,
                 lazy_sample=True, pooling=None, kanchor=60):
        super().__init__()
        points = _kernel_points(radius, kernel_size)
        _attrs(self, dim_in=dim_in, dim_out=dim_out, kernel_size=points.shape[0], stride=stride, radius=radius, sigma=sigma,
               n_neighbor=n_neighbor, lazy_sample=lazy_sample, pooling=pooling)
        self.basic_conv = BasicSO3Conv(dim_in, dim_out, points.shape[0])
        _const(self, anchors=L.get_anchors(kanchor), kernels=points)

    def forward(self, x, inter_idx=None, inter_w=None, epilogue=None):
        """epilogue (not in the reference): a vgtk.so3conv.functional.FoldedEpilogue the contraction may apply in
        inference (vgtk.so3conv.blocks.conv_norm_act)."""
        if inter_idx is None and self.stride == 1:
            # fused path: grouping + contraction in one autograd node (re-associated backward)
            xyz = x.xyz
            inter_idx, inter_w, feats = L.inter_so3conv_fused(xyz, None, x.feats, self.basic_conv.W,
                                                              self.n_neighbor, self.anchors, self.kernels,
                                                              self.radius, self.sigma, False, epilogue=epilogue)
            sample_idx = torch.arange(xyz.shape[2], dtype=torch.long, device=xyz.device).unsqueeze(0).repeat(xyz.shape[0], 1)
            return inter_idx, inter_w, sample_idx, SphericalPointCloud(xyz, feats, self.anchors)
        inter_idx, inter_w, xyz, grouped, sample_idx = L.inter_so3conv_grouping(
            x.xyz, x.feats, self.stride, self.n_neighbor, self.anchors, self.kernels, self.radius, self.sigma, inter_idx, inter_w,
            self.lazy_sample, pooling=self.pooling)
        return inter_idx, inter_w, sample_idx, SphericalPointCloud(xyz, self.basic_conv(grouped), self.anchors)


class InterSO3PoseConv(InterSO3Conv):
    """Pose-aware inter conv (modules.py:L177-322); forward returns
    (inter_idx, inter_w, sample_idx, SphericalPointCloudPose).  Same parameters / buffers as InterSO3Conv
    (`basic_conv.W`, `anchors`, `kernels`) plus the pose-handling switches."""

    def __init__(self, dim_in, dim_out, kernel_size, stride, radius, sigma, n_neighbor,
                 lazy_sample=True, pooling=None, kanchor=60, permute_modes=0, use_2d=False,
                 use_art_mode=False):
        super().__init__(dim_in, dim_out, kernel_size, stride, radius, sigma, n_neighbor, lazy_sample=lazy_sample, pooling=pooling,
                         kanchor=kanchor)
        self.permute_modes, self.use_2d, self.use_art_mode = permute_modes, use_2d, use_art_mode

    def forward(self, x, inter_idx=None, inter_w=None, seg=None, epilogue=None):
        if self.pooling is not None and self.stride > 1 and x.feats.shape[1] > 1:
            raise ValueError('xyz_pooling is not None?!!')             # functional.py:L913
        if self.use_2d:
            # anchor axis (60, 4) = anchors x residual rotations about y (functional.py:L1718-2130; scripts/train/eyeglasses.sh):
            # four ordinary convolutions on the fused kernels while the residual index is unpermuted (identity poses), the
            # reference's expression in device tensor ops otherwise
            if inter_idx is None and self.stride > 1:
                raise NotImplementedError('use_2d with stride > 1 (the reference forces stride 1 for this configuration)')
            w, feats = L.inter_so3conv_fused_2d(x.xyz, x.pose, x.feats, self.basic_conv.W, self.n_neighbor, self.anchors, self.kernels,
                                                self.radius, self.sigma, self.permute_modes != 0)
            return inter_idx, w, None, SphericalPointCloudPose(x.xyz, feats, self.anchors, x.pose)
        if self.use_art_mode and not (inter_idx is None and self.stride > 1):
            # articulation-state mode, stride-1 branch (functional.py:L1420-1520): x.xyz is [b, n_states, 3, p], `seg` picks
            # every point's state; the strided branch of that mode is the regular one (L1326-1331)
            w, feats = L.inter_so3conv_fused_art_mode(x.xyz, x.pose, x.feats, self.basic_conv.W, seg, self.n_neighbor, self.anchors,
                                                      self.kernels, self.radius, self.sigma, self.permute_modes != 0)
            return inter_idx, w, None, SphericalPointCloudPose(x.xyz, feats, self.anchors, x.pose)
        if inter_idx is None and self.stride > 1:
            # strided branch (functional.py:L931-1013): furthest-point sampled centres (native FPS kernel unless
            # lazy_sample), their poses, centres -> all points ball query; returns inter_idx = None (L1013)
            sample_idx, q_xyz, q_pose = L._strided_centres(x.xyz, x.pose, self.stride, self.lazy_sample)
            _, w, feats = L.inter_so3conv_fused(x.xyz, x.pose, x.feats, self.basic_conv.W, self.n_neighbor,
                                                self.anchors, self.kernels, self.radius, self.sigma,
                                                self.permute_modes != 0, q_xyz=q_xyz, q_pose=q_pose)
            return None, w, sample_idx.long(), SphericalPointCloudPose(q_xyz, feats, self.anchors, q_pose)      # spconv/functional.py:L476
        # stride-1 branch of the reference (functional.py:L1025-1286): the neighbourhood is recomputed
        # on every call and the passed-in inter_idx is handed back unchanged; grouping + contraction
        # run as one autograd node (csrc/so3_inter_*.hip + gemm_dma_f32.hip)
        _, w, feats = L.inter_so3conv_fused(x.xyz, x.pose, x.feats, self.basic_conv.W, self.n_neighbor,
                                            self.anchors, self.kernels, self.radius, self.sigma,
                                            self.permute_modes != 0, epilogue=epilogue)
        return inter_idx, w, None, SphericalPointCloudPose(x.xyz, feats, self.anchors, x.pose)


class IntraSO3Conv(nn.Module):
    """Group conv over the 12 icosahedral neighbours of each anchor (modules.py:L325-347);
    only valid for kanchor = 60."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        taps = L.get_intra_idx()                                              # [60, 12] neighbouring anchors of every anchor
        _attrs(self, dim_in=dim_in, dim_out=dim_out, kernel_size=taps.shape[1])
        self.basic_conv = BasicSO3Conv(dim_in, dim_out, taps.shape[1])
        _const(self, anchors=L.get_anchors())
        self.register_buffer('intra_idx', torch.from_numpy(taps).long())

    def forward(self, x):
        feats = x.feats
        if feats.is_cuda and feats.dtype == torch.float32 and feats.shape[3] % 4 == 0 and feats.shape[3] <= 64:
            # implicit GEMM in both directions: the 12-tap gather is folded into the contraction's operand load, the
            # [B,C,12,P,A] tensor of the reference (48 GB at C = 512) is never written (L._IntraConv)
            out = L.intra_so3conv(feats, self.basic_conv.W, self.intra_idx)
        else:
            out = self.basic_conv(L.intra_so3conv_grouping(self.intra_idx, feats))
        return SphericalPointCloud(x.xyz, out, self.anchors)


class IntraSO3Conv2D(IntraSO3Conv):
    """IntraSO3Conv on an anchor axis of (60, 4) in-plane residual rotations (modules.py:L350-373): same parameters and
    buffers, the 12-tap gather acts on the 60-axis."""

    def forward(self, x):
        out = self.basic_conv(L.intra_so3conv_grouping_2D(self.intra_idx, x.feats))
        return SphericalPointCloud(x.xyz, out, self.anchors)


class PointnetSO3Conv(nn.Module):
    """Equivariant pointnet aggregation (modules.py:L376-412): 1x1 conv over [feats ; A^T xyz]
    then max over points.  Dense torch layers (out of the HIP scope)."""

    def __init__(self, dim_in, dim_out, kanchor=60, return_raw=False):
        super().__init__()
        _attrs(self, dim_in=dim_in + 3, dim_out=dim_out, return_raw=return_raw)      # + 3: the coordinates in the anchor's frame
        self.embed = nn.Conv2d(dim_in + 3, dim_out, 1)
        _const(self, anchors=L.get_anchors(kanchor))

    def forward(self, x):
        """x.xyz [b,3,n], x.feats [b,c,n,na] -> [b,dim_out,na] (or the per-point map with return_raw): every anchor sees
        the centred coordinates in ITS frame, A_a^T (x - mean), stacked under the features; a shared 1x1 layer; max over
        the points."""
        centred = x.xyz - x.xyz.mean(dim=2, keepdim=True)                               # [b,3,n]
        if x.feats.shape[3] == 1:
            coords = centred.unsqueeze(-1)                                              # one frame: the identity
        else:
            coords = torch.matmul(self.anchors.transpose(1, 2).unsqueeze(0), centred.unsqueeze(1)).permute(0, 2, 3, 1)   # [b,3,n,na]
        embedded = self.embed(torch.cat([x.feats, coords], dim=1))
        return embedded if self.return_raw else embedded.max(dim=2)[0]


class PointnetSO3PoseConv(PointnetSO3Conv):
    """modules.py:L415-449 (same computation, always max-pooled)."""

    def __init__(self, dim_in, dim_out, kanchor=60):
        super().__init__(dim_in, dim_out, kanchor, return_raw=False)
