"""tools/gpu/uninit_check.py: does any result depend on memory nobody wrote?  torch.empty / empty_like / new_empty are patched to hand out
float buffers filled with NaN; the headline step (fwd + bwd) and the config-3 composite forward must come out finite and bit-equal to the
unpatched run."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench
import synth_clouds
dev = torch.device('cuda:0')
_empty, _empty_like = torch.empty, torch.empty_like
POISON = [False]


def empty(*a, **k):
    t = _empty(*a, **k)
    if POISON[0] and t.is_floating_point() and t.is_cuda:
        t.fill_(float('nan'))
    return t


def empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if POISON[0] and t.is_floating_point() and t.is_cuda:
        t.fill_(float('nan'))
    return t


torch.empty, torch.empty_like = empty, empty_like


def headline(poses):
    torch.manual_seed(2913)
    model = bench.Backbone(4096, None).to(dev)
    xyz_np, lab, pose_np = synth_clouds.laptop_batch(0, 4, 4096)
    if poses:
        import numpy as np
        rng = np.random.default_rng(1)
        q = rng.standard_normal((4, 2, 4)); q /= np.linalg.norm(q, axis=-1, keepdims=True)
        w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        rot = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(4, 2, 3, 3)
        pose_np = pose_np.copy()
        for b in range(4):
            pose_np[b, :, :3, :3] = rot[b][lab[b]]
    xyz, pose = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(pose_np.astype('float32')).to(dev)
    feats = model(xyz, pose)
    loss = bench.StandInLoss.apply(feats, model.pose_head.weight, model.pose_head.bias)
    loss.backward()
    return [float(loss)] + [float(p.grad.double().abs().sum()) for p in model.parameters() if p.grad is not None]


def config3():
    import config3_step as C3
    torch.manual_seed(2913)
    xyz, _, pose = synth_clouds.laptop_batch(0, 4, 4096)
    model = C3.Config3Model(4096).to(dev)
    with torch.no_grad():
        out = model(torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev))
    return [float(out[0]), float(out[1]['scores'].double().sum()), float(out[1]['recon'].double().sum())]


for name, fn in (('headline step, identity poses', lambda: headline(False)), ('headline step, one rotation per rigid part', lambda: headline(True)),
                 ('config-3 composite forward', config3)):
    POISON[0] = False
    a = fn()
    POISON[0] = True
    b = fn()
    POISON[0] = False
    import math
    ok = all(math.isfinite(v) for v in b)
    print(f'{name}: finite {ok}, bit-equal {a == b}', flush=True)
    if a != b:
        print('   plain   ', a[:6]); print('   poisoned', b[:6])
