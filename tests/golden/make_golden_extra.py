"""tests/golden/make_golden_extra.py -- golden vectors for SURVEY.md section 8(f) rows 2 and 4, produced by RUNNING THE
REFERENCE on CPU in the build container (through tests/golden/ref_import.py).  Data only.

  strided_pose.npz   InterSO3PoseConv with stride 2 (so3conv/modules.py:L222-322 -> so3conv/functional.py:L931-1013,
                     `inter_idx is None and stride > 1`): lazy centres (the first ceil(P/2) points) and furthest-point
                     sampled centres, random per-point poses, permute_modes = 1; outputs + autograd gradients
  inv_head.npz       InvPPOutBlockOurs (SPConvNets/utils/base_so3conv.py:L842-917) with attention / max / mean pooling,
                     training and eval mode, outputs + autograd gradients of the attention variant
  orbit.npz          the orbit-selection distance block, ...pn_38_multi_stage.py:L1341-1399.  That block is inline in a
                     1700-line model method that cannot run here (hard .cuda() calls, datasets, compiled chamfer), so
                     THOSE LINES are read from the reference file at generation time, dedented and executed on seeded
                     inputs with a stub `self` carrying the four attributes they read -- the reference's own statements
                     produce the expected values; no reference text is stored

Re-run:  python tests/golden/make_golden_extra.py"""
import importlib.util
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (imports the reference through ref_import)
from make_golden import poses, save  # noqa: E402

vgtk, sptk, L, zptk = MG.vgtk, MG.sptk, MG.L, MG.zptk
for _name, _dir in (('SPConvNets', 'SPConvNets'), ('SPConvNets.utils', 'SPConvNets/utils'), ('SPConvNets.models', 'SPConvNets/models')):
    _m = types.ModuleType(_name)
    _m.__path__ = [os.path.join('/root/reference', _dir)]
    sys.modules[_name] = _m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


synth_clouds = _load('synth_clouds', os.path.join(REPO, 'equi-articulated-pose_amd', 'synth_clouds.py'))


def strided():
    gen = torch.Generator().manual_seed(414)
    xyz = torch.from_numpy(synth_clouds.laptop_batch(3, 2, 96)[0])
    out = {}
    for tag, lazy in (('lazy', True), ('fps', False)):
        torch.manual_seed(2913)
        conv = sptk.InterSO3PoseConv(6, 8, 1, 2, 0.2, 0.02, 16, lazy_sample=lazy, kanchor=60, permute_modes=1)
        pose = poses(gen, 2, 96, 'random')
        feats = torch.randn(2, 6, 96, 60, generator=gen).requires_grad_(True)
        inter_idx, inter_w, sample_idx, y = conv(zptk.SphericalPointCloudPose(xyz, feats, None, pose))
        assert inter_idx is None
        gy = torch.randn(y.feats.shape, generator=gen)
        gfe, gW = torch.autograd.grad(y.feats, [feats, conv.basic_conv.W], gy)
        out.update({f'{tag}_pose': pose, f'{tag}_feats': feats, f'{tag}_W': conv.basic_conv.W, f'{tag}_sample_idx': sample_idx,
                    f'{tag}_new_xyz': y.xyz, f'{tag}_new_pose': y.pose, f'{tag}_inter_w_head': inter_w[:, :4], f'{tag}_out': y.feats,
                    f'{tag}_grad_out': gy, f'{tag}_grad_feats': gfe, f'{tag}_grad_W': gW})
    save('strided_pose.npz', xyz=xyz, anchors=conv.anchors, kernels=conv.kernels, radius=np.float32(0.2), sigma=np.float32(0.02),
         nn=np.int32(16), stride=np.int32(2), **out)


def inv_head():
    INV = _load('ref_base_so3conv', '/root/reference/SPConvNets/utils/base_so3conv.py').InvPPOutBlockOurs
    params = {'dim_in': 24, 'mlp': [16, 12], 'fc': [12], 'k': 12, 'kanchor': 60, 'temperature': 3.0}

    def build(mode):
        torch.manual_seed(12)
        head = INV(params, norm=1, pooling_method=mode)
        with torch.no_grad():
            for m in head.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5)
                    m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 2.0)
        return head

    def relu_margin(x):
        """Smallest |pre-activation| any ReLU of the attention head sees (training and eval mode): an input within fp32
        rounding of zero would let rounding pick the ReLU's side, and the gradients of the two sides differ by O(1)."""
        head = build('attention')
        state = {k: v.clone() for k, v in head.state_dict().items()}
        seen = []
        hooks = [m.register_forward_hook(lambda mod, inp, outp: seen.append(float(outp.detach().abs().min()))) for m in head.norm]
        for training in (True, False):
            head.load_state_dict(state)
            head.train(training)
            with torch.no_grad():
                head(zptk.SphericalPointCloud(None, x, None))
        for h in hooks:
            h.remove()
        return min(seen)

    seed = 99
    while True:                                  # the first seed whose ReLU inputs all stay 2e-5 away from zero
        gen = torch.Generator().manual_seed(seed)
        x0 = torch.randn(2, 24, 37, 60, generator=gen)
        margin = relu_margin(x0)
        if margin > 2e-5:
            break
        seed += 1
    print(f'inv_head: input seed {seed}, smallest |ReLU input| {margin:.2e}')
    out = {'x': x0, 'x_seed': np.int32(seed), 'relu_margin': np.float32(margin)}
    for mode in ('attention', 'max', 'mean'):
        head = build(mode)
        state = {k: v.clone() for k, v in head.state_dict().items()}
        out.update({f'{mode}_state_{k}': v for k, v in state.items()})
        for phase in ('train', 'eval'):
            head.load_state_dict(state)
            head.train(phase == 'train')
            x = x0.clone().requires_grad_(True)
            res = head(zptk.SphericalPointCloud(None, x, None))
            if mode == 'attention':
                y, conf = res
                out[f'{mode}_{phase}_conf'] = conf.detach()
            else:
                y = res
            out[f'{mode}_{phase}_out'] = y.detach()
            if mode == 'attention':
                g = torch.randn(y.shape, generator=gen)
                names = [n for n, _ in head.named_parameters()]
                grads = torch.autograd.grad(y, [x] + list(head.parameters()), g)
                out[f'{mode}_{phase}_grad_out'] = g
                out[f'{mode}_{phase}_grad_x'] = grads[0]
                for n, gr in zip(names, grads[1:]):
                    out[f'{mode}_{phase}_grad_{n}'] = gr
            if phase == 'train':
                for k, v in head.state_dict().items():
                    if 'running' in k:
                        out[f'{mode}_after_{k}'] = v.clone()
    save('inv_head.npz', **out)


def orbit():
    path = '/root/reference/SPConvNets/models/unsup_seg_so3_pose_conv_pn_38_multi_stage.py'
    lines = open(path).read().split('\n')
    first, last = 1341, 1399                       # `dist_recon_ori = torch.sum(...` ... `slot_dist_ori_recon, slot_orbits = torch.min(...)`
    assert lines[first - 1].lstrip().startswith('dist_recon_ori = torch.sum('), lines[first - 1]
    assert 'slot_orbits = torch.min(orbit_slot_dist_ori_recon' in lines[last - 1], lines[last - 1]
    block = textwrap.dedent('\n'.join(lines[first - 1:last]))
    safe_transpose = _load('ref_common_utils_min', '/root/reference/SPConvNets/models/common_utils.py').safe_transpose
    gen = torch.Generator().manual_seed(5)
    B, S, A, M, N = 2, 3, 60, 17, 83
    out = {}
    for tag, (cd, single) in (('cd0_multi', (0, 0)), ('cd1_single', (1, 1))):
        transformed_pts = torch.randn(B, S, A, M, 3, generator=gen) * 0.3
        ori_pts = torch.randn(B, 3, N, generator=gen) * 0.3
        lab = torch.randint(0, S, (B, N), generator=gen)
        lab[1][lab[1] == 2] = 0                                                        # an empty slot in cloud 1
        hard_one_hot_labels = torch.eye(S)[lab]                                        # [B,N,S]
        attn_ori = torch.softmax(torch.randn(B, S, N, generator=gen), dim=1)
        ns = {'torch': torch, 'safe_transpose': safe_transpose, 'transformed_pts': transformed_pts.clone(), 'ori_pts': ori_pts,
              'hard_one_hot_labels': hard_one_hot_labels, 'attn_ori': attn_ori, 'k': A,
              'self': types.SimpleNamespace(recon_part_M=M, slot_single_cd=cd, slot_single_mode=single, num_slots=S)}
        exec(compile(block, path + ':L%d-%d' % (first, last), 'exec'), ns)
        out.update({f'{tag}_transformed_pts': transformed_pts, f'{tag}_ori_pts': ori_pts, f'{tag}_hard_one_hot_labels': hard_one_hot_labels,
                    f'{tag}_attn_ori': attn_ori})
        for k in ('minn_dist_ori_to_recon_all_pts', 'minn_dist_recon_to_ori_all_pts', 'minn_dist_recon_to_ori', 'minn_dist_ori_to_recon',
                  'minn_dist_ori_to_recon_hard', 'orbit_slot_dist_ori_recon', 'slot_orbits'):
            out[f'{tag}_{k}'] = ns[k]
        out[f'{tag}_slot_dist'] = ns['slot_dist_ori_recon_all_slots' if single else 'slot_dist_ori_recon']
    save('orbit.npz', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['strided', 'inv_head', 'orbit']
    for w in which:
        globals()[w]()
