"""config3_step.py -- BASELINE.json configuration 3 as ONE unit: "Full SPConvNets unsup-arti-align training step (fwd+bwd
incl. chamfer) batch=16 on 1 x MI355X", composed from the hot-path operators exactly as SURVEY.md section 8(d) counts
the reference's step (stage 1, num_iters = 1, scripts/train/laptop_syn.sh):

  1. frozen stage-0 model under no_grad: `glb_backbone` = 3 separable blocks (inter + intra + 1x1 skip), forward only
     (SPConvNets/trainer_unsup_arti_align.py:L594-597, ...pn_38_multi_stage.py:L369-374)
  2. trained model: `backbone` and `backbone_sec`, 3 inter blocks each, forward + backward (L505-508, L515-518)
  3. per-point invariant features from backbone_sec (InvPPOutBlockOurs, attention pooling, L608-611) -> slot scores ->
     arg-max labels (L626); per-slot pose heads on the backbone features, batched over the clouds
     (SO3OutBlockRTWithMaskSep through pose_head_over_subsets, L706-830) -> angle -> R (L1103-1112), T
  4. slot shapes posed by (R, T) of the selected orbit -> one chamfer pair [B,N,3] <-> [B,N,3], forward + backward
     (extensions.chamfer_dist, L1744-1746)
  5. Adam on every trained parameter

Stand-ins for the reference's control plane (out of scope): the slot-attention module is a linear
scorer on the invariant features, the shape decoders are one learnable canonical point set per slot, the orbit is the
arg-min of the head's translation norm instead of the full reconstruction search.  Everything that costs time in the
reference's step -- nine conv layers, heads, chamfer, optimiser -- is the real operator.  bench.py times it
(`config3_step` in the JSON line); tests/test_gpu_config3.py runs it at a reduced size."""
import numpy as np
import torch
import torch.nn as nn

NN, NA, SLOTS = 64, 60, 2
ROT_ANGLE_FACTOR = 0.5


class InterBackbone(nn.Module):
    """3 x (InterSO3PoseConv -> BatchNorm2d + leaky_relu): `backbone` / `backbone_sec`."""

    def __init__(self, plan):
        super().__init__()
        import vgtk.so3conv as sptk
        self.convs, self.norms = nn.ModuleList(), nn.ModuleList()
        for (c, o, r, s) in plan:
            self.convs.append(sptk.InterSO3PoseConv(c, o, 1, 1, r, s, NN, kanchor=NA, permute_modes=1))
            self.norms.append(sptk.BatchNormLeakyReLU(o, negative_slope=0.01))

    def forward(self, xyz, pose):
        import vgtk.so3conv as sptk
        import vgtk.spconv as zptk
        x = zptk.SphericalPointCloudPose(xyz, sptk.get_occupancy_features(xyz.transpose(1, 2), NA, False), None, pose)
        for conv, norm in zip(self.convs, self.norms):
            # `x = conv(x); feat = relu(norm(x.feats))` (SPConvNets/utils/base_so3poseconv.py:L205-222) through the block-layer helper
            _, _, _, x = sptk.conv_norm_act(conv, norm, x)
        return x.feats


class SeparableBackbone(nn.Module):
    """3 x SeparableSO3PoseConvBlock (SPConvNets/utils/base_so3poseconv.py:L270-328): the frozen `glb_backbone`."""

    def __init__(self, plan):
        super().__init__()
        import vgtk.so3conv as sptk
        self.inter, self.inter_norm = nn.ModuleList(), nn.ModuleList()
        self.intra, self.intra_norm = nn.ModuleList(), nn.ModuleList()
        self.skip, self.skip_norm = nn.ModuleList(), nn.ModuleList()
        for (c, o, r, s) in plan:
            self.inter.append(sptk.InterSO3PoseConv(c, o, 1, 1, r, s, NN, kanchor=NA, permute_modes=1))
            self.inter_norm.append(sptk.BatchNormLeakyReLU(o, negative_slope=0.01))
            self.intra.append(sptk.IntraSO3Conv(o, o))
            self.intra_norm.append(sptk.InstanceNormLeakyReLU(o, negative_slope=0.01))
            self.skip.append(nn.Conv2d(c, o, 1))
            self.skip_norm.append(sptk.BatchNormLeakyReLU(o, negative_slope=0.01))

    def forward(self, xyz, pose):
        import vgtk.so3conv as sptk
        import vgtk.spconv as zptk
        x = zptk.SphericalPointCloudPose(xyz, sptk.get_occupancy_features(xyz.transpose(1, 2), NA, False), None, pose)
        for i in range(len(self.inter)):
            skip = x.feats
            # frozen stage (eval mode, no_grad): the inter block's BatchNorm + leaky_relu and the skip branch's BatchNorm +
            # leaky_relu + sum ride in the epilogues of their contractions (vgtk/so3conv/blocks.py)
            _, _, _, y = sptk.conv_norm_act(self.inter[i], self.inter_norm[i], x)
            y = self.intra[i](zptk.SphericalPointCloud(y.xyz, y.feats, y.anchors))
            f = sptk.pointwise_norm_act(self.skip[i], self.skip_norm[i], skip, residual=self.intra_norm[i](y.feats))
            x = zptk.SphericalPointCloudPose(x.xyz, f, y.anchors, x.pose)
        return x.feats


class Config3Model(nn.Module):
    def __init__(self, points, plan=None, head_width=256, recon_points=None):
        super().__init__()
        import synth_clouds
        import vgtk.so3conv as sptk
        plan = plan or synth_clouds.backbone_layers(points)
        feat = plan[-1][1]
        self.glb_backbone = SeparableBackbone(plan)
        self.backbone = InterBackbone(plan)
        self.backbone_sec = InterBackbone(plan)
        outblock = {'dim_in': feat, 'mlp': [head_width], 'fc': [64], 'k': SLOTS, 'pooling': 'attention', 'temperature': 3.0, 'kanchor': NA}
        self.ppint_outblk = sptk.InvPPOutBlockOurs(outblock, norm=1, pooling_method='attention')
        self.slot_scorer = nn.Linear(head_width, SLOTS)                       # stand-in for slot attention
        self.slot_heads = nn.ModuleList([sptk.SO3OutBlockRTWithMaskSep(
            outblock, norm=1, pooling_method='max', global_scalar=False, use_anchors=False, feat_mode_num=NA, num_heads=1,
            representation='angle', c_in_rot=feat, c_in_trans=feat, pred_axis=True, pred_central_points=True,
            central_points_in_dim=head_width) for _ in range(SLOTS)])
        m = (recon_points or points) // SLOTS
        self.slot_shapes = nn.Parameter(torch.randn(SLOTS, m, 3) * 0.1)      # stand-in for the per-slot shape decoders
        for p in self.glb_backbone.parameters():
            p.requires_grad_(False)
        self.glb_backbone.eval()

    def trained_parameters(self):
        return [p for p in self.parameters() if p.requires_grad]

    def forward(self, xyz, pose):
        """xyz [B,3,N], pose [B,N,4,4] -> (loss, dict of intermediate results)."""
        import vgtk.so3conv as sptk
        import vgtk.spconv as zptk
        from extensions.chamfer_dist import ChamferDistance
        b, _, n = xyz.shape
        with torch.no_grad():                                                   # 1. frozen stage-0 forward
            glb_feats = self.glb_backbone(xyz, pose)
            glb_orbit = glb_feats.mean((1, 2)).argmax(-1)                       # [B] (the global orbit the stage hands on)
            del glb_feats
        feats = self.backbone(xyz, pose)                                        # 2. [B,512,N,A]
        feats_sec = self.backbone_sec(xyz, pose)
        ppinv, conf = self.ppint_outblk(zptk.SphericalPointCloud(xyz, feats_sec, None))      # 3. [B,256,N], [B,N,A]
        scores = self.slot_scorer(ppinv.transpose(1, 2))                        # [B,N,S]
        labels = scores.argmax(-1)
        anchors = self.backbone.convs[0].anchors
        recon, slot_R, slot_T, centre_reg = [], [], [], 0.0
        # every slot's member points compacted (an empty slot falls back to the whole cloud): the heads work on P points per
        # cloud in total, not on slots x P
        for s_, out in enumerate(sptk.pose_head_over_slot_groups(self.slot_heads, feats, xyz, labels, anchors)):
            ang = (torch.sigmoid(out['R']) * np.pi * ROT_ANGLE_FACTOR).reshape(b, NA)
            Rm = sptk.compute_rotation_matrix_from_angle(anchors, ang, defined_axis=out['axis'].transpose(1, 2))   # [B,A,3,3]
            T = out['T'].transpose(1, 2)                                        # [B,A,3]
            orbit = T.norm(dim=-1).argmin(-1)                                   # stand-in for the reconstruction-distance search
            pick = orbit.view(b, 1, 1, 1)
            R_sel = Rm.gather(1, pick.expand(b, 1, 3, 3)).squeeze(1)            # [B,3,3]
            T_sel = T.gather(1, orbit.view(b, 1, 1).expand(b, 1, 3)).squeeze(1) # [B,3]
            recon.append(torch.matmul(self.slot_shapes[s_].unsqueeze(0), R_sel.transpose(1, 2)) + T_sel.unsqueeze(1))
            slot_R.append(Rm)
            slot_T.append(T)
            centre_reg = centre_reg + (out['central_points'] - 0.5).square().mean()      # keeps the centre regressor in the graph
        recon = torch.cat(recon, dim=1)                                         # 4. [B, N, 3]
        d1, d2 = ChamferDistance()(recon, xyz.transpose(1, 2).contiguous(), return_raw=True)
        entropy = -(torch.softmax(scores, -1) * torch.log_softmax(scores, -1)).sum(-1).mean()
        loss = d1.mean() + d2.mean() + 0.01 * entropy + 1e-3 * centre_reg
        return loss, {'labels': labels, 'slot_R': torch.stack(slot_R, 1), 'slot_T': torch.stack(slot_T, 1), 'glb_orbit': glb_orbit,
                      'recon': recon, 'conf': conf, 'scores': scores}


def algorithmic_flops(batch, points, plan, head_width=256):
    """Algorithmic flops of one step (MFMA-shaped work only; 24 kernel points, 60 anchors, 64 neighbours):
    grouping 2 C K P NN A + contraction 2 O C K P A per inter layer, intra 2 O O 12 P A; forward x1, backward x2."""
    pa, pakn = points * NA, points * NA * 24 * NN
    inter = sum((2 * c * pakn if c >= 16 else (2 * c + 11) * pakn) + 2 * o * c * 24 * pa for c, o, _, _ in plan)
    intra = sum(2 * o * o * 12 * pa for _, o, _, _ in plan)
    skip = sum(2 * o * c * pa for c, o, _, _ in plan)
    feat = plan[-1][1]
    heads = SLOTS * (2 * 2 * head_width * feat * pa + 2 * head_width * 2 * head_width * pa) + 2 * head_width * feat * pa
    return batch * ((inter + intra + skip) + 3 * 2 * inter + 3 * heads)


def stage_fingerprints(model, xyz, pose):
    """The composite forward stage by stage under no_grad -> {stage: (float64 sum, xor-sum of the bit patterns)} in program order: what
    tests/test_gpu_config3.py prints when two forwards differ, and what tools/gpu/config3_flake_hunt.py compares over many forwards."""
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk

    def fp(t):
        t = t.detach().contiguous()
        x = (t.view(torch.int32) if t.dtype == torch.float32 else t).reshape(-1).to(torch.int64)
        n2 = x.numel() // 2 * 2
        return (float(t.double().sum()) if t.is_floating_point() else int(x.sum()), int(torch.bitwise_xor(x[:n2:2], x[1:n2:2]).sum()))

    out = {}
    with torch.no_grad():
        g = model.glb_backbone(xyz, pose); out['glb_backbone'] = fp(g); del g
        feats = model.backbone(xyz, pose); out['backbone'] = fp(feats)
        feats_sec = model.backbone_sec(xyz, pose); out['backbone_sec'] = fp(feats_sec)
        ppinv, conf = model.ppint_outblk(zptk.SphericalPointCloud(xyz, feats_sec, None))
        out['inv_head.ppinv'], out['inv_head.conf'] = fp(ppinv), fp(conf)
        scores = model.slot_scorer(ppinv.transpose(1, 2)); out['slot_scorer'] = fp(scores)
        labels = scores.argmax(-1); out['labels'] = fp(labels)
        anchors = model.backbone.convs[0].anchors
        for s_, o in enumerate(sptk.pose_head_over_slot_groups(model.slot_heads, feats, xyz, labels, anchors)):
            for k in ('R', 'T', 'axis', 'central_points'):
                out[f'slot{s_}.{k}'] = fp(o[k])
        loss, res = model(xyz, pose)
        out['recon'], out['loss'] = fp(res['recon']), (float(loss), 0)
    return out
