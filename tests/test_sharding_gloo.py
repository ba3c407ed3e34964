"""CPU, world_size 2 over gloo: the N>1 path of the hot path -- clouds sharded across ranks
with no data-path collective, one pose-hypothesis all-gather, one bucketed gradient all-reduce."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, os.path.join(root, 'equi-articulated-pose_amd'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vgtk import sharding
    import synth_clouds

    # 1. contiguous cloud shards cover the batch exactly once
    start, stop = sharding.shard_range(7)
    # 2. pose-hypothesis all-gather: every rank ends with all clouds, rank-major
    b_loc, S, A = 3, 2, 60
    gen = torch.Generator().manual_seed(100 + rank)
    R = torch.randn(b_loc, S, A, 3, 3, generator=gen)
    Tt = torch.randn(b_loc, S, A, 3, generator=gen)
    allR, allT = sharding.all_gather_pose_hypotheses(R, Tt)
    # 3. bucketed gradient all-reduce (average)
    params = [torch.nn.Parameter(torch.zeros(4, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
    params[0].grad = torch.full((4, 5), float(rank + 1))
    params[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
    sharding.all_reduce_gradients(params)   # params[2] has no grad: skipped
    # 4. sharded synthetic clouds are the global batch, split
    xyz = synth_clouds.laptop_batch(rank * 2, 2, 64)[0]
    np.savez(os.path.join(out_dir, f'r{rank}.npz'), start=start, stop=stop, allR=allR.numpy(), allT=allT.numpy(),
             R=R.numpy(), T=Tt.numpy(), g0=params[0].grad.numpy(), g1=params[1].grad.numpy(), xyz=xyz)
    dist.destroy_process_group()


def test_world_size_2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(tmp_path / f'r{i}.npz')) for i in range(world)]
    assert (int(r[0]['start']), int(r[0]['stop'])) == (0, 4) and (int(r[1]['start']), int(r[1]['stop'])) == (4, 7)
    for i in range(world):
        np.testing.assert_array_equal(r[i]['allR'], np.concatenate([r[0]['R'], r[1]['R']]))
        np.testing.assert_array_equal(r[i]['allT'], np.concatenate([r[0]['T'], r[1]['T']]))
        np.testing.assert_allclose(r[i]['g0'], np.full((4, 5), 1.5))
        np.testing.assert_allclose(r[i]['g1'], np.arange(7) * 1.5)
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'equi-articulated-pose_amd'))
    import synth_clouds
    full = synth_clouds.laptop_batch(0, 4, 64)[0]
    np.testing.assert_array_equal(np.concatenate([r[0]['xyz'], r[1]['xyz']]), full)


def test_single_process_is_identity():
    from vgtk import sharding
    R, Tt = torch.randn(2, 2, 60, 3, 3), torch.randn(2, 2, 60, 3)
    a, b = sharding.all_gather_pose_hypotheses(R, Tt)
    assert a is R and b is Tt
    assert sharding.shard_range(10) == (0, 10)
    assert sharding.shard_range(10, rank=1, world=4) == (3, 6)


def _bn_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, os.path.join(root, 'equi-articulated-pose_amd'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vgtk.so3conv.blocks import batch_moments, all_reduce_sums
    gen = torch.Generator().manual_seed(7)
    x = (torch.randn(5, 6, 40, generator=gen, dtype=torch.float64) * 3 + 100.0)   # global batch [B, C, N], large mean
    mine = x[:3] if rank == 0 else x[3:]                                            # uneven shards
    pivot = mine[0, :, 0]
    d = mine - pivot[None, :, None]
    mean, var, total = batch_moments(d.sum((0, 2)), (d * d).sum((0, 2)), pivot, mine.shape[0] * mine.shape[2], sync=True)
    g = torch.arange(6, dtype=torch.float64) * (rank + 1)
    sg, = all_reduce_sums(g, sync=True)
    np.savez(os.path.join(out_dir, f'bn{rank}.npz'), mean=mean.numpy(), var=var.numpy(), total=float(total), sg=sg.numpy(),
             ref_mean=x.mean((0, 2)).numpy(), ref_var=x.var((0, 2), unbiased=False).numpy())
    dist.destroy_process_group()


def test_batchnorm_statistics_exchange_world_size_2(tmp_path):
    """The block epilogue's SyncBatchNorm-style exchange (vgtk/so3conv/blocks.py): every rank ends
    with the statistics of the whole batch from one all-reduce of raw moments."""
    world = 2
    mp.spawn(_bn_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for i in range(world):
        r = dict(np.load(tmp_path / f'bn{i}.npz'))
        np.testing.assert_allclose(r['mean'], r['ref_mean'], rtol=1e-13)
        np.testing.assert_allclose(r['var'], r['ref_var'], rtol=1e-9)
        assert r['total'] == 5 * 40
        np.testing.assert_allclose(r['sg'], np.arange(6) * 3.0)


def _uneven_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, os.path.join(root, 'equi-articulated-pose_amd'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vgtk import sharding
    n_items, S, A = 7, 2, 60                                        # 7 clouds over 2 ranks: shards of 4 and 3
    start, stop = sharding.shard_range(n_items)
    gen = torch.Generator().manual_seed(5)
    R_all = torch.randn(n_items, S, A, 3, 3, generator=gen)
    T_all = torch.randn(n_items, S, A, 3, generator=gen)
    a1, b1 = sharding.all_gather_pose_hypotheses(R_all[start:stop], T_all[start:stop], n_items=n_items)    # sizes from the split
    a2, b2 = sharding.all_gather_pose_hypotheses(R_all[start:stop], T_all[start:stop])                       # sizes exchanged
    ok = all(torch.equal(x, y) for x, y in ((a1, R_all), (b1, T_all), (a2, R_all), (b2, T_all)))
    # overlapped gradient reduction: three "layers", a bucket size that splits them 2 + 1, one parameter unused this step
    torch.manual_seed(3)
    layers = [torch.nn.Linear(6, 5), torch.nn.Linear(5, 4), torch.nn.Linear(4, 3)]
    unused = torch.nn.Parameter(torch.ones(3))
    params = [p for l in layers for p in l.parameters()] + [unused]
    reducer = sharding.GradientReducer(params, bucket_bytes=4 * (4 * 3 + 3 + 5 * 4 + 4))
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(10 + rank))
    y = x
    for l in layers:
        y = torch.tanh(l(y))
    y.square().sum().backward()
    local = [p.grad.clone() for p in params[:-1]]
    reducer.finish()
    grads = [p.grad.clone() for p in params[:-1]]
    # second step on the same reducer (hooks re-arm)
    for p in params:
        p.grad = None
    y = x * 2.0
    for l in layers:
        y = torch.tanh(l(y))
    y.sum().backward()
    local2 = [p.grad.clone() for p in params[:-1]]
    reducer.finish()
    grads2 = [p.grad.clone() for p in params[:-1]]
    np.savez(os.path.join(out_dir, f'u{rank}.npz'), ok=ok, n_buckets=len(reducer.buckets), unused_grad_is_none=unused.grad is None,
             **{f'l{i}': g.numpy() for i, g in enumerate(local)}, **{f'g{i}': g.numpy() for i, g in enumerate(grads)},
             **{f'm{i}': g.numpy() for i, g in enumerate(local2)}, **{f'h{i}': g.numpy() for i, g in enumerate(grads2)})
    dist.destroy_process_group()


def test_uneven_shards_and_overlapped_gradient_reduction(tmp_path):
    """7 clouds over 2 ranks (shards of 4 and 3): the pose all-gather returns all 7, rank-major, whether the shard sizes
    come from the split or are exchanged; GradientReducer (all-reduce launched from autograd hooks while the backward
    is still running) ends with the rank-averaged gradients, two steps in a row."""
    world = 2
    mp.spawn(_uneven_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(tmp_path / f'u{i}.npz')) for i in range(world)]
    for i in range(world):
        assert bool(r[i]['ok']) and int(r[i]['n_buckets']) >= 2 and bool(r[i]['unused_grad_is_none'])
        for k in range(6):
            np.testing.assert_allclose(r[i][f'g{k}'], 0.5 * (r[0][f'l{k}'] + r[1][f'l{k}']), rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(r[i][f'h{k}'], 0.5 * (r[0][f'm{k}'] + r[1][f'm{k}']), rtol=1e-6, atol=1e-7)


def _divergent_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    sys.path.insert(0, os.path.join(root, 'equi-articulated-pose_amd'))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from vgtk import sharding
    torch.manual_seed(3)
    # registration order never, a, b, c, d  ->  buckets (deepest first) [d], [c], [b], [a], [never]: one parameter per bucket
    a, b, c, d = (torch.nn.Parameter(torch.randn(5)) for _ in range(4))
    never = torch.nn.Parameter(torch.randn(5))
    params = [never, a, b, c, d]           # `never` registers first = the LAST bucket (an unused head bucket would hold every launch until finish())
    reducer = sharding.GradientReducer(params, bucket_bytes=4 * 5)
    w = torch.arange(1.0, 6.0) * (rank + 1)

    def loss(use_c_on, use_b_on):
        y = (a * w).sum() + (d * w * 2).sum()
        if rank in use_c_on:
            y = y + (c * w * 3).sum()      # data-dependent branch: c is used on SOME ranks only
        if rank in use_b_on:
            y = y + (b * w * 4).sum()
        return y

    # step 1: c only on rank 0, b on both.  Rank 1's bucket [c] never completes, bucket [b] does: launching from the completing
    # hook alone would pair rank 0's [c] with rank 1's [b].
    loss({0}, {0, 1}).backward()
    reducer.finish()
    s1 = {n: (None if p.grad is None else p.grad.clone()) for n, p in zip('nabcd', params)}
    # step 2 with zero_grad(set_to_none=False): b unused everywhere this step -> its stale step-1 average must not be reduced again
    for p in params:
        if p.grad is not None:
            p.grad.zero_()
    loss({0, 1}, set()).backward()
    reducer.finish()
    s2 = {n: (None if p.grad is None else p.grad.clone()) for n, p in zip('nabcd', params)}
    # step 3 WITHOUT zero_grad, c on rank 0 only: rank 1 still holds step 2's average in c.grad and does not fire -- it contributes
    # zeros and its stale tensor is REPLACED, so both ranks end with the same (documented) value: avg over ranks of what the hooks
    # saw = (rank 0's accumulated c.grad + 0) / 2
    c_before = c.grad.clone()
    reads = []
    orig_tolist = torch.Tensor.tolist
    torch.Tensor.tolist = lambda self: (reads.append(1), orig_tolist(self))[1]
    loss({0}, set()).backward()
    reducer.finish()
    torch.Tensor.tolist = orig_tolist
    s3_c, s3_expect = c.grad.clone(), 0.5 * (c_before + 3 * torch.arange(1.0, 6.0))     # rank 0: stale + new (torch accumulates), rank 1: 0
    # a second backward while a bucket is in flight raises
    for p in params:
        p.grad = None
    loss({0, 1}, {0, 1}).backward()
    try:
        loss({0, 1}, {0, 1}).backward()
        raised = False
    except RuntimeError:
        raised = True
    reducer.finish()
    out = {'raised': raised, 'never_none': s1['n'] is None and s2['n'] is None, 's3_c': s3_c.numpy(), 's3_expect': s3_expect.numpy(), 'host_reads': len(reads)}
    for tag, s in (('s1', s1), ('s2', s2)):
        for n in 'abcd':
            out[f'{tag}_{n}'] = s[n].numpy()
    np.savez(os.path.join(out_dir, f'd{rank}.npz'), **out)
    dist.destroy_process_group()


def test_gradient_reducer_with_rank_dependent_graphs(tmp_path):
    """Ranks whose autograd graphs differ (a parameter used on one rank only) still issue the same collectives in the same
    order, end with identical averaged gradients (materialised where p.grad was None), a parameter unused in a step adds
    nothing even when zero_grad keeps the tensors, one that no rank touches keeps grad None, and a backward that meets an
    in-flight bucket raises."""
    world = 2
    mp.spawn(_divergent_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [dict(np.load(tmp_path / f'd{i}.npz')) for i in range(world)]
    w = np.arange(1.0, 6.0)
    for i in range(world):
        assert bool(r[i]['raised']) and bool(r[i]['never_none'])
        np.testing.assert_allclose(r[i]['s1_a'], 1.5 * w)
        np.testing.assert_allclose(r[i]['s1_d'], 3.0 * w)
        np.testing.assert_allclose(r[i]['s1_c'], 0.5 * 3 * w)          # rank 0 only: (3w + 0) / 2, on BOTH ranks
        np.testing.assert_allclose(r[i]['s1_b'], 1.5 * 4 * w)
        np.testing.assert_allclose(r[i]['s2_c'], 1.5 * 3 * w)
        np.testing.assert_allclose(r[i]['s2_b'], 0 * w)                 # zeroed by zero_grad, untouched by the reduction
        np.testing.assert_allclose(r[i]['s2_a'], 1.5 * w)
        np.testing.assert_allclose(r[i]['s3_c'], r[0]['s3_expect'])     # identical on both ranks
        assert int(r[i]['host_reads']) == 1                              # one stacked read per finish(), not one per bucket (5 here)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts 2 ranks itself (the form the driver uses for N = 1) and refuses a
    world size that is not the request; --check-launch runs the path's two exchanges on small tensors (gloo here, no GPU)."""
    import json
    import subprocess
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--check-launch'], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['launch_check'] is True and line['n_gpus'] == 2 and line['requested_gpus'] == 2
    env.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    bad = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--check-launch'], env=env,
                         capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and 'refusing' in bad.stderr


def test_bench_rooflines_price_the_split_contractions_on_their_pipe():
    """bench.roofline_object: the two-plane contraction executes 3 fp16 MFMA flops per algorithmic flop, the three-plane one 6
    bf16 flops, both against the 2.5 PFLOP/s dense 16-bit peak; every other kernel against the fp32-MFMA peak (no GPU needed)."""
    import bench
    k = {'flops': 2.0e12, 'ms': 10.0, 'launches': 2, 'entries': {'x'}}
    two = bench.roofline_object('gemm_f16x2_kernel<4, 4>', k, False)
    three = bench.roofline_object('gemm_bf16x3_kernel<4, 4>', k, False)
    plain = bench.roofline_object('so3_group_lists2_kernel<true, 0>', k, False)
    assert abs(two['achieved'] - 600.0) < 1e-6 and abs(two['frac'] - 600.0 / bench.PEAK_BF16_MFMA_TFLOPS) < 1e-12 and 'fp16' in two['pipe']
    assert abs(three['achieved'] - 1200.0) < 1e-6 and abs(three['fp32_equivalent_TFLOPs'] - 200.0) < 1e-9
    assert abs(plain['achieved'] - 200.0) < 1e-9 and plain['peak'] == bench.PEAK_F32_MFMA_TFLOPS and plain['avg_launch_ms'] == 5.0
