"""extensions.chamfer_dist -- ChamferFunction / ChamferDistance with the reference's interface
(extensions/chamfer_dist/__init__.py:L13-45), computed by the HIP kernels in csrc/chamfer.hip."""
import torch

import chamfer


class ChamferFunction(torch.autograd.Function):
    """(a [B,N,3], b [B,M,3]) -> (squared distance of every a-point to its nearest b-point [B,N], and the converse [B,M]);
    the nearest-neighbour indices are kept for the backward (csrc/chamfer.hip)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        a, b = xyz1.contiguous(), xyz2.contiguous()
        d_ab, d_ba, nearest_ab, nearest_ba = chamfer.forward(a, b)
        ctx.save_for_backward(a, b, nearest_ab, nearest_ba)
        return d_ab, d_ba

    @staticmethod
    def backward(ctx, g_ab, g_ba):
        a, b, nearest_ab, nearest_ba = ctx.saved_tensors
        return tuple(chamfer.backward(a, b, nearest_ab, nearest_ba, g_ab, g_ba))


class ChamferDistance(torch.nn.Module):
    """mean(d_ab) + mean(d_ba) (or the two raw distance maps).  ignore_zeros: for a single pair of clouds, points whose
    coordinates sum to zero (padding) are dropped first, as the reference does."""

    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    @staticmethod
    def _without_padding(cloud):
        keep = cloud.sum(dim=2).ne(0)                  # [1, n]
        return cloud[keep].unsqueeze(0)

    def forward(self, xyz1, xyz2, return_raw=False):
        if self.ignore_zeros and xyz1.size(0) == 1:
            xyz1, xyz2 = self._without_padding(xyz1), self._without_padding(xyz2)
        d_ab, d_ba = ChamferFunction.apply(xyz1, xyz2)
        return (d_ab, d_ba) if return_raw else d_ab.mean() + d_ba.mean()


MASKED_DISTANCE = 99999.0     # the constant the reference writes into masked entries of the distance tensor


def orbit_reconstruction_distances(transformed_pts, ori_pts, hard_one_hot_labels):
    """The slot/orbit reconstruction distances of the reference's orbit selection
    (SPConvNets/models/unsup_seg_so3_pose_conv_pn_38_multi_stage.py:L1341-1361) WITHOUT its
    `dist_recon_ori [B, S, A, M, N]` tensor (4 GB at B=8, S=2, A=60, M=256, N=4096): a 60-way batched
    chamfer on the chamfer kernels (csrc/chamfer.hip), SURVEY.md 8(f) row 2.

    transformed_pts [B,S,A,M,3]  per-slot, per-anchor reconstructions
    ori_pts         [B,3,N]      the input cloud
    hard_one_hot_labels [B,N,S]  0/1 point-to-slot assignment
    ->  (minn_dist_ori_to_recon_all_pts [B,S,A,N],  min over M of the unmasked distances
         minn_dist_recon_to_ori_all_pts [B,S,A],    mean over M of the min over N, unmasked
         minn_dist_recon_to_ori         [B,S,A],    the same with points outside the slot masked
         minn_dist_ori_to_recon         [B,S,A,N])  min over M with masked points set to 99999
    Gradients flow to transformed_pts exactly as through the reference's min / mean."""
    b, s, a, m, _ = transformed_pts.shape
    n = ori_pts.shape[2]
    recon = transformed_pts.reshape(b * s * a, m, 3)
    ori = ori_pts.transpose(1, 2)                                                   # [B,N,3]
    ori_all = ori[:, None, None].expand(b, s, a, n, 3).reshape(b * s * a, n, 3)
    r2o_all, o2r_all = ChamferFunction.apply(recon, ori_all)                         # [B',M], [B',N]
    in_slot = hard_one_hot_labels.transpose(1, 2) >= 0.5                             # [B,S,N]
    # masked: points outside the slot are moved out of reach; the reference's 99999 entries are what
    # an empty slot's minimum sees, hence the clamp
    far = torch.full_like(ori, 1e4)
    ori_masked = torch.where(in_slot[..., None], ori[:, None].expand(b, s, n, 3), far[:, None].expand(b, s, n, 3))
    ori_masked = ori_masked[:, :, None].expand(b, s, a, n, 3).reshape(b * s * a, n, 3)
    r2o_masked, _ = ChamferFunction.apply(recon, ori_masked)
    r2o_masked = r2o_masked.clamp(max=MASKED_DISTANCE)
    o2r_all = o2r_all.view(b, s, a, n)
    o2r_masked = torch.where(in_slot[:, :, None].expand(b, s, a, n), o2r_all, o2r_all.new_full((), MASKED_DISTANCE))
    return (o2r_all, r2o_all.view(b, s, a, m).mean(-1), r2o_masked.view(b, s, a, m).mean(-1), o2r_masked)
