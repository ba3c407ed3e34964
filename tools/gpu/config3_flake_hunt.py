"""tools/gpu/config3_flake_hunt.py: which stage of the config-3 composite forward is not run-to-run reproducible, if any?  (Round-5 advisor
finding: tests/test_gpu_config3.py accepted "two of three forwards bit-equal" after one mismatch in eight suite runs, blamed on torch's BLAS
without evidence.)  N forwards of the full-size composite (16 x 4096) in one process with a fingerprint (float64 sum + xor of the bit patterns)
after every stage -- frozen backbone, the two trained backbones, invariant head, slot scorer, every slot's pose head outputs, chamfer -- and,
between the forwards, unrelated allocations of changing sizes so that the caching allocator hands the stages different blocks (the mismatch
was never seen in isolation).  DET=1: under torch.use_deterministic_algorithms(True, warn_only=True)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
import bench  # noqa: F401
import synth_clouds
import config3_step as C3
import vgtk.so3conv as sptk
import vgtk.spconv as zptk

dev = torch.device('cuda:0')
N = int(os.environ.get('N', 24))
if os.environ.get('DET') == '1':
    torch.use_deterministic_algorithms(True, warn_only=True)
P, B = 4096, 16
xyz, _, pose = synth_clouds.laptop_batch(0, B, P)
xyz, pose = torch.from_numpy(xyz).to(dev), torch.from_numpy(pose).to(dev)
torch.manual_seed(2913)
model = C3.Config3Model(P).to(dev)


staged = lambda: C3.stage_fingerprints(model, xyz, pose)


ref = staged()
bad = {}
gen = torch.Generator(device=dev).manual_seed(1)
for i in range(N):
    # perturb the allocator's cache between forwards
    junk = [torch.empty(int(torch.randint(1, 4096, (1,)).item()) * 65536 + 17 * i, device=dev).fill_(float('nan')) for _ in range(3)]
    del junk
    if i % 4 == 3:
        torch.cuda.empty_cache()
    cur = staged()
    for k, v in cur.items():
        if v != ref[k]:
            bad.setdefault(k, []).append((i, ref[k], v))
print(f'{N} forwards after the reference one; stages that differed: {list(bad) or "none"}')
order = list(ref)
for k in order:
    if k in bad:
        print(f'  first differing stage in program order: {k}: {len(bad[k])} of {N} runs, e.g. {bad[k][0]}')
        break
for k, v in bad.items():
    print(f'  {k}: {len(v)} of {N}')
