"""BASELINE.json configuration 3 ("Full SPConvNets unsup-arti-align training step (fwd+bwd incl. chamfer) batch=16") as a
unit: config3_step.Config3Model -- frozen separable glb_backbone forward, backbone + backbone_sec forward/backward,
invariant head -> labels, batched per-slot pose heads -> R, T, one chamfer pair forward/backward, Adam -- at a reduced
size (the bench leg `config3_step` runs it at 16 x 4096).  Checked here: every trained parameter receives a finite
gradient, the frozen stage does not, the loss goes down over a few optimiser steps, the forward is bit-reproducible, and the
piece config 3 adds over the other configurations -- the separable block with its intra conv at C = 512 on a 4096-point
cloud -- agrees with the oracle on a slab of points."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_config3_composite_step_trains():
    import synth_clouds
    import config3_step as C3
    dev = torch.device('cuda:0')
    P, B = 512, 3
    plan = [(c, min(o, 64), r, s) for (c, o, r, s) in synth_clouds.backbone_layers(P)]
    plan = [(1, 16, plan[0][2], plan[0][3]), (16, 32, plan[1][2], plan[1][3]), (32, 64, plan[2][2], plan[2][3])]
    xyz, _, pose = synth_clouds.laptop_batch(60, B, P)
    xyz, pose = T(xyz).to(dev), T(pose).to(dev)
    losses = []
    for trial in range(2):
        torch.manual_seed(2913)
        model = C3.Config3Model(P, plan=plan, head_width=32).to(dev)
        params = model.trained_parameters()
        opt = torch.optim.Adam(params, lr=2e-3)
        hist = []
        for it in range(4):
            opt.zero_grad(set_to_none=True)
            loss, out = model(xyz, pose)
            loss.backward()
            if it == 0:
                missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
                # the orbit / label selections are arg-max's: everything else must be reached by the chamfer + entropy loss
                assert not missing, missing
                assert all(torch.isfinite(p.grad).all() for p in params)
                assert all(p.grad is None for p in model.glb_backbone.parameters())
                assert out['labels'].shape == (B, P) and out['slot_R'].shape == (B, C3.SLOTS, 60, 3, 3) and out['slot_T'].shape == (B, C3.SLOTS, 60, 3)
                RtR = torch.matmul(out['slot_R'].transpose(-1, -2), out['slot_R'])
                assert (RtR - torch.eye(3, device=dev)).abs().max().item() < 1e-4
            opt.step()
            hist.append(loss.item())
        losses.append(hist)
    assert losses[0][-1] < losses[0][0], losses[0]
    # the hot-path kernels are atomics-free; torch's own gather / index backward kernels in the stand-in layers use
    # atomics, so only the first step (pure forward) is bit-reproducible, later ones to rounding
    assert losses[0][0] == losses[1][0], 'the forward of the composite step is not reproducible run to run'
    # later steps: torch's atomics-based index kernels in the stand-in layers perturb the update at 1e-7, and an arg-min / arg-max
    # selection that flips on it moves a loss by ~1 % (observed in one run of six at step 4)
    np.testing.assert_allclose(losses[0], losses[1], rtol=5e-2)


def test_config3_composite_step_at_full_size():
    """BASELINE config 3 at ITS size -- 16 x 4096-point clouds, full widths (1 -> 64 -> 128 -> 512, head width 256): the step
    bench.py times as `config3_step` (~0.9 s, ~92 GB).  Every trained parameter receives a finite gradient, the frozen
    stage none, the forward is bit-reproducible down to the pre-arg-max slot scores and the pose hypotheses, the rotations are
    orthonormal, and the loss falls over three optimiser steps."""
    import synth_clouds
    import config3_step as C3
    dev = torch.device('cuda:0')
    P, B = 4096, 16
    xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
    xyz, pose = T(xyz).to(dev), T(pose).to(dev)
    torch.manual_seed(2913)
    model = C3.Config3Model(P).to(dev)
    params = model.trained_parameters()
    opt = torch.optim.Adam(params, lr=1e-4)        # bench.py's rate; 2e-3 (the reduced-size test's) overshoots at full width
    # Run-to-run reproducibility of the whole composite forward, STRICT (round-5 advisor finding: this assert had been relaxed to "two of
    # three" after one mismatch in eight runs of the whole suite, never reproduced in isolation -- tools/gpu/config3_flake_hunt.py: 36
    # forwards with the allocator's cache perturbed and NaN-filled between them, every stage bit-equal, profiles/r06_config3_flake_hunt.txt;
    # tools/gpu/uninit_check.py finds no read of unwritten memory).  The comparison runs with torch's own kernels pinned to their
    # deterministic algorithms (rocBLAS atomics off: the layers behind the backbones are torch modules), so that whatever still differs
    # is this package's; a mismatch fails and names the first stage that differs.
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        with torch.no_grad():
            first = model(xyz, pose)
            l0, o0 = float(first[0]), {k: first[1][k].clone() for k in ('scores', 'slot_R', 'slot_T', 'labels', 'recon')}
            del first
            # (training-mode BatchNorms only move their running statistics between the two calls; the batch statistics they
            # normalise with are the same)
            again = model(xyz, pose)
            same = float(again[0]) == l0 and all(torch.equal(again[1][k], v) for k, v in o0.items())
            if not same:
                diffs = {k: float((again[1][k].double() - v.double()).abs().max()) for k, v in o0.items()}
                f1, f2 = C3.stage_fingerprints(model, xyz, pose), C3.stage_fingerprints(model, xyz, pose)
                first_bad = next((k for k in f1 if f1[k] != f2[k]), None)
                raise AssertionError(f'config-3 composite forward differed between two calls: loss {l0} vs {float(again[0])}, max differences {diffs}; '
                                     f'first differing stage of two more forwards: {first_bad}')
            del again
    finally:
        torch.use_deterministic_algorithms(False)
    with torch.no_grad():
        # the hot path itself: the three backbones twice, bit for bit
        for bb in (model.glb_backbone, model.backbone, model.backbone_sec):
            f1 = bb(xyz, pose)
            f2 = bb(xyz, pose)
            assert torch.equal(f1, f2)
            del f1, f2
    RtR = torch.matmul(o0['slot_R'].transpose(-1, -2), o0['slot_R'])
    assert (RtR - torch.eye(3, device=dev)).abs().max().item() < 1e-4
    assert o0['labels'].shape == (B, P) and o0['slot_T'].shape == (B, C3.SLOTS, 60, 3)
    hist = []
    for it in range(4):
        opt.zero_grad(set_to_none=True)
        loss, out = model(xyz, pose)
        loss.backward()
        if it == 0:
            missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
            assert not missing, missing
            assert all(torch.isfinite(p.grad).all() for p in params)
            assert all(p.grad is None for p in model.glb_backbone.parameters())
        opt.step()
        hist.append(loss.item())
        del loss, out
    assert all(np.isfinite(hist)) and hist[-1] < hist[0], hist
    assert torch.cuda.max_memory_allocated(dev) < 200 * 2 ** 30


def test_separable_block_at_4096_points_full_width_vs_oracle():
    """The deepest separable block (128 -> 512 inter conv, IntraSO3Conv at C = 512, InstanceNorm + leaky_relu, 1x1 skip
    + BatchNorm + leaky_relu, sum) on ONE 4096-point cloud: the inter and intra convolutions against the oracle on a slab
    of query points (both are independent across query points), the two fused epilogues against torch on the full maps."""
    import synth_clouds
    import vgtk.so3conv as sptk
    import vgtk.spconv as zptk
    from oracle import so3_ref
    dev = torch.device('cuda:0')
    P = 4096
    c, o, r, s = synth_clouds.backbone_layers(P)[2]
    xyz, _, pose = synth_clouds.laptop_batch(61, 1, P)
    torch.manual_seed(9)
    inter = sptk.InterSO3PoseConv(c, o, 1, 1, r, s, 64, kanchor=60, permute_modes=1)
    intra = sptk.IntraSO3Conv(o, o)
    skip = torch.nn.Conv2d(c, o, 1)
    gen = torch.Generator().manual_seed(10)
    f = torch.randn(1, c, P, 60, generator=gen)
    # CPU copies for the oracle (Module.to moves the module in place)
    anchors_c, kernels_c, W_inter_c = inter.anchors.clone(), inter.kernels.clone(), inter.basic_conv.W.detach().clone()
    W_intra_c, intra_idx_c = intra.basic_conv.W.detach().clone(), intra.intra_idx.clone()
    inter_d, intra_d, skip_d = inter.to(dev), intra.to(dev), skip.to(dev)
    n1 = sptk.BatchNormLeakyReLU(o, negative_slope=0.01).to(dev)
    n2 = sptk.InstanceNormLeakyReLU(o, negative_slope=0.01).to(dev)
    n3 = sptk.BatchNormLeakyReLU(o, negative_slope=0.01).to(dev)
    with torch.no_grad():
        x = zptk.SphericalPointCloudPose(T(xyz).to(dev), f.to(dev), None, T(pose).to(dev))
        _, _, _, y = inter_d(x)
        a1 = n1(y.feats)
        z = intra_d(zptk.SphericalPointCloud(y.xyz, a1, y.anchors)).feats
        a2 = n2(z)
        sk = skip_d(f.to(dev))
        out = n3(sk, residual=a2)
    q0, q1 = 1500, 1564
    # inter conv, slab of 64 query points
    res = so3_ref._poseconv_slab(T(xyz)[:, :, q0:q1].contiguous(), T(pose)[:, q0:q1].contiguous(), T(xyz), T(pose),
                                 so3_ref.add_shadow_feature(f), 64, anchors_c, kernels_c, r, s, 1, True)
    y_ref = so3_ref.basic_so3conv(W_inter_c, res[3])
    assert rel_err(y.feats[:, :, q0:q1].cpu().numpy(), y_ref.numpy()) < 2e-5
    # intra conv at C = 512 on the slab, from the GPU's own (already checked) input
    z_ref = so3_ref.intra_so3conv_layer(a1[:, :, q0:q1].cpu(), W_intra_c, intra_idx_c)
    assert rel_err(z[:, :, q0:q1].cpu().numpy(), z_ref.numpy()) < 2e-5
    # epilogues on the full maps (torch modules on the CPU)
    lr = torch.nn.functional.leaky_relu
    assert rel_err(a1.cpu().numpy(), lr(torch.nn.BatchNorm2d(o)(y.feats.cpu()), 0.01).detach().numpy()) < 1e-5
    assert rel_err(a2.cpu().numpy(), lr(torch.nn.InstanceNorm2d(o, affine=False)(z.cpu()), 0.01).numpy()) < 1e-5
    want = lr(torch.nn.BatchNorm2d(o)(sk.cpu()), 0.01).detach() + a2.cpu()
    assert rel_err(out.cpu().numpy(), want.numpy()) < 1e-5
