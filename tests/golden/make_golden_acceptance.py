"""tests/golden/make_golden_acceptance.py -- golden vectors of the north-star acceptance chain

    cloud -> 3-block inter backbone (1 -> 64 -> 128 -> 512, BatchNorm2d + leaky_relu) -> InvPPOutBlockOurs (attention
    pooling over the 60 anchors) -> per-point invariant features -> seeded linear slot scorer -> arg-max labels
          -> SO3OutBlockRTWithMaskSep per slot (mask = the slot's points) -> angle -> rotation matrices R, translations T

produced by RUNNING THE REFERENCE CLASSES on CPU in the build container (through tests/golden/ref_import.py):

    backbone   SPConvNets/utils/base_so3poseconv.py  BasicSO3PoseConvBlock of 'inter_block's with build_model's
               hyper-parameters (...pn_38_multi_stage.py:L505-508, L2146-2248), training-mode BatchNorm
    inv head   SPConvNets/utils/base_so3conv.py:L842-917  InvPPOutBlockOurs(pooling 'attention')  (...pn_38...:L608-611)
    labels     torch.argmax over slot scores (...pn_38...:L626); the slot-attention module between the two is the
               reference's control plane (out of scope), a seeded nn.Linear stands in for it on both sides
    pose head  SPConvNets/models/model_utils.py:L363-677  SO3OutBlockRTWithMaskSep with the model's constructor
               arguments (...pn_38...:L300-316: max pooling, global_scalar, angle representation, axis / pivot / centre)
    R          model_utils.py:L1000-1043 compute_rotation_matrix_from_angle on sigmoid(angle) * pi * rot_angle_factor
               (...pn_38...:L1103-1112)

Four cases:
  cfg1_512   BASELINE config 1: one 512-point cloud, everything above from the reference classes.
  parts_512  the same chain on a 512-point cloud whose two rigid parts carry different rotations (per-point pose = the
             part's rotation): neighbours across the hinge have a non-identity relative rotation, so the reference's 60x60
             anchor-permutation search (so3conv/functional.py:L1199-1204, 4.2 GB here) returns real permutations --
             everything from the reference classes.
  random_512 one Haar rotation per POINT: every neighbour pair is permuted.
  p4096      one 4096-point cloud at full widths.  The reference's grouping materialises a [P,64,60,60,3,3]
             intermediate (34 GB at P = 4096), so the three conv layers of THIS case come from oracle/so3_ref.py
             (the slab-wise restatement, itself pinned against the reference by tests/test_oracle_golden.py) with the
             reference's own nn.BatchNorm2d / leaky_relu between them; both heads are the reference classes.

All module parameters are drawn from seeded CPU generators; the fixture stores their float64 checksums (the test
re-draws them with the same seeds and refuses to run on a mismatch), the inputs, labels, slot scores, R, T and a
strided sample of the invariant features.  Data only.  Re-run:  python tests/golden/make_golden_acceptance.py [cfg1_512|p4096]
"""
import importlib.util
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, '..', '..'))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import acceptance_common as AC  # noqa: E402
import ref_import  # noqa: E402

vgtk, sptk, L, zptk = ref_import.import_reference()
# SPConvNets/__init__.py, utils/__init__.py and models/__init__.py import the whole model zoo (compiled chamfer module,
# datasets, ...): register the three packages as bare path holders so that only the files named below are executed
import types  # noqa: E402
for _name, _dir in (('SPConvNets', 'SPConvNets'), ('SPConvNets.utils', 'SPConvNets/utils'), ('SPConvNets.models', 'SPConvNets/models')):
    _m = types.ModuleType(_name)
    _m.__path__ = [os.path.join('/root/reference', _dir)]
    sys.modules[_name] = _m


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join('/root/reference', relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


BLK = _load('ref_base_so3poseconv', 'SPConvNets/utils/base_so3poseconv.py')
MU = _load('ref_model_utils', 'SPConvNets/models/model_utils.py')
try:
    INV = _load('ref_base_so3conv', 'SPConvNets/utils/base_so3conv.py').InvPPOutBlockOurs
except Exception as e:          # the file's star-imports pull in the whole model zoo; the class itself is self-contained
    print('base_so3conv.py does not import here (%s: %s); extracting the class by name is not possible -- abort' % (type(e).__name__, e))
    raise


synth_clouds = _load('synth_clouds', os.path.join(REPO, 'equi-articulated-pose_amd', 'synth_clouds.py'))


def run_case(name, P, plan_points):
    from oracle import so3_ref
    t0 = time.time()
    xyz, part, pose = synth_clouds.laptop_batch(AC.CLOUD_SEED[name], 1, P)
    pose = AC.case_poses(name, part).numpy()
    xyz_t, pose_t = torch.from_numpy(xyz), torch.from_numpy(pose)
    layers = synth_clouds.backbone_layers(plan_points)

    # ---- modules (reference classes), parameters from seeded generators
    params = [{'type': 'inter_block', 'args': {'dim_in': c, 'dim_out': o, 'kernel_size': 1, 'stride': 1, 'radius': r, 'sigma': s,
                                                'n_neighbor': AC.NN, 'lazy_sample': True, 'dropout_rate': 0.0, 'multiplier': 2,
                                                'activation': 'leaky_relu', 'pooling': 'none', 'kanchor': 60, 'norm': 'BatchNorm2d',
                                                'permute_modes': 1, 'use_art_mode': False}} for (c, o, r, s) in layers]
    backbone = BLK.BasicSO3PoseConvBlock(params)
    inv = INV(AC.OUTBLOCK, norm=1, pooling_method='attention')
    heads = [MU.SO3OutBlockRTWithMaskSep(AC.OUTBLOCK, **AC.POSE_HEAD_KW) for _ in range(AC.SLOTS)]
    scorer = torch.nn.Linear(AC.OUTBLOCK['mlp'][-1], AC.SLOTS)
    sums = AC.seed_parameters(backbone, inv, heads)
    backbone.train(); inv.train()
    for h in heads:
        h.train()

    with torch.no_grad():
        if P <= 512:
            x = BLK.preprocess_input(xyz_t, 60, pose_t, False)
            feats = backbone(x).feats
        else:
            f = torch.ones(1, 1, P, 60)
            for blk, (c, o, r, s) in zip(backbone.blocks, layers):
                y = so3_ref.inter_so3poseconv_layer(xyz_t, pose_t, f, blk.conv.basic_conv.W, blk.conv.anchors, blk.conv.kernels,
                                                    r, s, AC.NN, permute_modes=1, chunk=128, skip_perm_search=True)
                f = blk.relu(blk.norm(y))
                print(f'  layer {c}->{o} done at {time.time() - t0:.0f} s', flush=True)
            feats = f
        x = zptk.SphericalPointCloud(xyz_t, feats, None)
        ppinv, conf = inv(x)                                            # [1,256,P], [1,P,60]
        for scorer_seed in range(AC.PARAM_SEED, AC.PARAM_SEED + 5000):
            scorer_sums = AC.seed_scorer(scorer, scorer_seed)
            scores = scorer(ppinv.transpose(1, 2))                      # [1,P,S]
            labels = torch.argmax(scores, dim=-1)                       # [1,P]   (...pn_38_multi_stage.py:L626)
            top2 = scores.topk(2, dim=-1).values
            margin = (top2[..., 0] - top2[..., 1]).min().item()
            counts = np.bincount(labels.numpy().ravel(), minlength=AC.SLOTS)
            if margin >= AC.MIN_MARGIN[name] and counts.min() >= P // 8:       # well-posed arg-max, every slot populated
                break
        else:
            raise RuntimeError('no scorer seed with a margin >= %g' % AC.MIN_MARGIN[name])
        sums = sums + scorer_sums
        anchors = torch.from_numpy(np.ascontiguousarray(L.get_anchors(60))).float()
        out = {}
        for s_, head in enumerate(heads):
            mask = (labels == s_).float()
            res = head(zptk.SphericalPointCloud(xyz_t, feats.clone(), None), mask, feats.clone(), trans_xyz=xyz_t,
                       anchors=anchors.unsqueeze(0))
            ang = torch.sigmoid(res['R']) * np.pi * AC.ROT_ANGLE_FACTOR                               # [1,1,60]
            axis = res['axis'][:, :, 0]                                                                 # the slot's defined axis [1,3]
            Rm = MU.compute_rotation_matrix_from_angle(anchors, ang.transpose(-1, -2).reshape(1, 60, 1), defined_axis=axis)
            out[f'slot{s_}_R'] = Rm.numpy().copy()                                                      # [1,60,3,3]
            out[f'slot{s_}_T'] = res['T'].numpy().copy()                                                # [1,3,60]
            out[f'slot{s_}_angle_logit'] = res['R'].numpy().copy()
            out[f'slot{s_}_axis'] = res['axis'].numpy().copy()
            out[f'slot{s_}_central_points'] = res['central_points'].numpy().copy()
    print(f'{name}: labels {np.bincount(labels.numpy().ravel(), minlength=AC.SLOTS).tolist()}, min score margin {margin:.3e}, '
          f'|feats| max {feats.abs().max().item():.3f}, {time.time() - t0:.0f} s')
    arrs = {'xyz': xyz, 'pose_is_identity': np.array(int(AC.POSE_KIND[name] == 'identity')), 'pose_rotations': pose[:, :, :3, :3].copy(), 'labels': labels.numpy().astype(np.int64), 'scores': scores.numpy(),
            'min_margin': np.array(margin), 'scorer_seed': np.array(scorer_seed), 'ppinv_sample': ppinv[:, :, ::AC.SAMPLE_STRIDE[name]].numpy().copy(),
            'conf_sample': conf[:, ::AC.SAMPLE_STRIDE[name]].numpy().copy(),
            'feats_sample': feats[:, ::AC.CH_STRIDE, ::AC.SAMPLE_STRIDE[name]].numpy().copy(),
            'checksums': np.array(sums, dtype=np.float64)}
    arrs.update(out)
    path = os.path.join(HERE, f'acceptance_{name}.npz')
    np.savez_compressed(path, **arrs)
    print(f'{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB')


if __name__ == '__main__':
    which = sys.argv[1:] or ['cfg1_512', 'p4096', 'parts_512', 'random_512']
    for n in which:
        run_case(n, AC.POINTS[n], AC.POINTS[n])
