"""oracle/orbit_ref.py -- TEST INFRASTRUCTURE ONLY.  The reference's orbit-selection distance block,
restated op for op on torch-CPU (SPConvNets/models/unsup_seg_so3_pose_conv_pn_38_multi_stage.py:
L1341-1361; `safe_transpose` = transpose().contiguous(), model_util.py).  It materialises the
`[B,S,A,M,N]` distance tensor exactly as the reference does."""
import torch


def orbit_reconstruction_distances(transformed_pts, ori_pts, hard_one_hot_labels):
    k = transformed_pts.shape[2]
    recon_part_M = transformed_pts.shape[3]
    # L1341-1342
    dist_recon_ori = torch.sum((transformed_pts.unsqueeze(-2) - ori_pts.transpose(-1, -2).contiguous().unsqueeze(
        1).unsqueeze(1).unsqueeze(1)) ** 2, dim=-1)
    # L1343-1344
    expanded = hard_one_hot_labels.transpose(-1, -2).contiguous().unsqueeze(2).unsqueeze(2).repeat(1, 1, k, recon_part_M, 1)
    # L1346-1349
    minn_dist_ori_to_recon_all_pts, _ = torch.min(dist_recon_ori, dim=-2)
    minn_dist_recon_to_ori_all_pts, _ = torch.min(dist_recon_ori, dim=-1)
    minn_dist_recon_to_ori_all_pts = minn_dist_recon_to_ori_all_pts.mean(dim=-1)
    # L1351-1354
    dist_recon_ori = dist_recon_ori.clone()
    dist_recon_ori[expanded < 0.5] = 99999.0
    minn_dist_recon_to_ori, _ = torch.min(dist_recon_ori, dim=-1)
    minn_dist_recon_to_ori = minn_dist_recon_to_ori.mean(-1)
    # L1358
    minn_dist_ori_to_recon, _ = torch.min(dist_recon_ori, dim=-2)
    return minn_dist_ori_to_recon_all_pts, minn_dist_recon_to_ori_all_pts, minn_dist_recon_to_ori, minn_dist_ori_to_recon
