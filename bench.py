"""bench.py -- headline benchmark of the SE(3)-equivariant point-convolution hot path.

Metric (BASELINE.json): point-clouds/sec (4096 pts, 60 anchors) fwd+bwd.
One "step" = one batch of synthetic 4096-point 'laptop' clouds through the reference's 3-block
inter backbone (channels 1->64->128->512, NN=64, K=24, A=60; each block = InterSO3PoseConv ->
BatchNorm2d -> leaky_relu as in SPConvNets/utils/base_so3poseconv.py:L205-222), forward +
backward + Adam step on the conv weights, with identity per-point poses and anchor permutation
enabled (what the shipped model runs).  Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--points P] [--fwd-only]
For N > 1 launch through torch.distributed.run (one rank per GPU, RCCL): every rank processes
its own B clouds (weak scaling, no data-path collective), then one pose-hypothesis all-gather
and one bucketed gradient all-reduce per step.

Rank 0 prints ONE JSON line with the metric, a `roofline` object for the dominant HIP KERNEL (launch
times are attributed to the kernel, template arguments included, that each C-ABI entry reports through
eap_last_kernel() -- the same names rocprofv3 prints; achieved = algorithmic flops / measured launch
time from HIP events recorded on the launch stream inside the timed region), `kernel_rooflines` for the
grouping and contraction kernels, `whole_step` (algorithmic flops of the step / step time / peak),
`config3_step` (BASELINE config 3 as one unit, config3_step.py) and a `cpu_baseline` object (the CPU
oracle -- an op-for-op restatement of the reference's torch path -- timed on the host cores on a
bounded sample; checker/baseline only, never part of the measured path).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'equi-articulated-pose_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import re  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (same table); a 3 x bf16 split kernel spends 6 bf16 MFMA flops per fp32-equivalent flop
PEAK_HBM_GBS = 8000.0
NN, KS, NA, SLOTS = 64, 24, 60, 2


class Backbone(nn.Module):
    """3 x (InterSO3PoseConv -> BatchNorm2d -> leaky_relu; the last two fused): the `backbone` of
    SPConvNets/models/unsup_seg_so3_pose_conv_pn_38_multi_stage.py:L505-508 with build_model's
    hyper-parameters (L2089-2225)."""

    def __init__(self, input_num, plan_points=None):
        super().__init__()
        import synth_clouds
        import vgtk.so3conv as sptk
        self.convs = nn.ModuleList()
        self.norms = nn.ModuleList()
        # plan_points: take the radii / sigmas build_model derives for THAT input size (e.g. the 512-point plan on
        # 4096-point clouds: small balls, most support rows referenced -- the textbook-backward regime)
        for (c, o, r, s) in synth_clouds.backbone_layers(plan_points or input_num):
            self.convs.append(sptk.InterSO3PoseConv(c, o, 1, 1, r, s, NN, kanchor=NA, permute_modes=1))
            self.norms.append(sptk.BatchNormLeakyReLU(o, negative_slope=0.01))   # = nn.BatchNorm2d + F.leaky_relu, fused (csrc/bn_act.hip)
        # stand-in for the pose head's output layer: pooled features -> per-slot, per-anchor
        # (R as 9 numbers, T as 3) hypotheses -- only so the all-gather moves real data
        self.pose_head = nn.Linear(512, SLOTS * 12)

    def forward(self, xyz, pose):
        import vgtk.so3conv as sptk
        import vgtk.spconv as zptk
        feats = sptk.get_occupancy_features(xyz.transpose(1, 2), NA, False)
        x = zptk.SphericalPointCloudPose(xyz, feats, None, pose)
        for conv, norm in zip(self.convs, self.norms):
            # `x = conv(x); feat = relu(norm(x.feats))` (SPConvNets/utils/base_so3poseconv.py:L205-222) through the block-layer helper
            _, _, _, x = sptk.conv_norm_act(conv, norm, x)
        return x.feats

    def hypotheses(self, feats, pooled=None):
        pooled = (feats.mean(2) if pooled is None else pooled).transpose(1, 2)                       # [B, A, 512]
        h = self.pose_head(pooled).view(feats.shape[0], NA, SLOTS, 12).transpose(1, 2)
        return h[..., :9].reshape(feats.shape[0], SLOTS, NA, 3, 3).contiguous(), h[..., 9:].contiguous()


class SeparableBackbone(nn.Module):
    """The frozen stage-0 `glb_backbone`: 3 x SeparableSO3PoseConvBlock = inter conv block -> intra
    conv block -> + relu(norm(1x1 skip conv)) (SPConvNets/utils/base_so3poseconv.py:L270-328,
    ...pn_38_multi_stage.py:L369-374).  The reference only ever runs it forward, under no_grad
    (trainer_unsup_arti_align.py:L594-597); `bench.py --separable` reports it separately
    (SURVEY.md 8(d)).  The 1x1 skip conv is the pointwise contraction (vgtk.so3conv.pointwise_conv)."""

    def __init__(self, input_num):
        super().__init__()
        import synth_clouds
        import vgtk.so3conv as sptk
        self.inter, self.inter_norm = nn.ModuleList(), nn.ModuleList()
        self.intra, self.intra_norm = nn.ModuleList(), nn.ModuleList()
        self.skip, self.skip_norm = nn.ModuleList(), nn.ModuleList()
        for (c, o, r, s) in synth_clouds.backbone_layers(input_num):
            self.inter.append(sptk.InterSO3PoseConv(c, o, 1, 1, r, s, NN, kanchor=NA, permute_modes=1))
            self.inter_norm.append(sptk.BatchNormLeakyReLU(o, negative_slope=0.01))
            self.intra.append(sptk.IntraSO3Conv(o, o))
            self.intra_norm.append(sptk.InstanceNormLeakyReLU(o, negative_slope=0.01))   # base_so3poseconv.py:L88
            self.skip.append(nn.Conv2d(c, o, 1))
            self.skip_norm.append(sptk.BatchNormLeakyReLU(o, negative_slope=0.01))
        self.pose_head = nn.Linear(512, SLOTS * 12)

    def forward(self, xyz, pose):
        import vgtk.so3conv as sptk
        import vgtk.spconv as zptk
        feats = sptk.get_occupancy_features(xyz.transpose(1, 2), NA, False)
        x = zptk.SphericalPointCloudPose(xyz, feats, None, pose)
        for i in range(len(self.inter)):
            skip = x.feats
            # (frozen stage: eval mode under no_grad -- the BatchNorms + activations (+ the skip sum) ride in the epilogues
            # of their contractions; in training mode the same calls run conv and fused norm passes)
            _, _, _, y = sptk.conv_norm_act(self.inter[i], self.inter_norm[i], x)
            y = self.intra[i](zptk.SphericalPointCloud(y.xyz, y.feats, y.anchors))
            f = sptk.pointwise_norm_act(self.skip[i], self.skip_norm[i], skip, residual=self.intra_norm[i](y.feats))
            x = zptk.SphericalPointCloudPose(x.xyz, f, y.anchors, x.pose)
        return x.feats

    hypotheses = Backbone.hypotheses


class StandInLoss(torch.autograd.Function):
    """loss = mean(feats^2) + mean(h^2), h = pose_head(mean over points of feats): what the bench
    back-propagates into the backbone (the reference's heads and losses are out of scope).  Written
    as one Function so the backbone gradient 2 f / N + broadcast(pooled gradient) is produced in
    ONE pass over the 4 GB feature tensor instead of pow-backward + expand + add_ (3 passes, 16 ms
    of torch glue per step that has nothing to do with the path being measured)."""

    @staticmethod
    def forward(ctx, feats, weight, bias, pooled=None):
        b, c, p, a = feats.shape
        if pooled is None:
            pooled = feats.mean(2)                                        # [B, C, A]; the step hands over the one its hypotheses used
        h = torch.addmm(bias, pooled.transpose(1, 2).reshape(b * a, c), weight.t())
        loss = torch.linalg.vector_norm(feats).square() / feats.numel() + h.square().mean()
        ctx.save_for_backward(feats, weight, pooled, h)
        return loss

    @staticmethod
    def backward(ctx, g):
        feats, weight, pooled, h = ctx.saved_tensors
        b, c, p, a = feats.shape
        gh = h * (g * (2.0 / h.numel()))
        g_w = gh.t() @ pooled.transpose(1, 2).reshape(b * a, c)
        g_b = gh.sum(0)
        g_pool = (gh @ weight).view(b, a, c).transpose(1, 2) / p          # [B, C, A]
        g_f = torch.addcmul(g_pool.unsqueeze(2), feats, g * (2.0 / feats.numel()))
        return g_f, g_w, g_b, None


CPU_BASELINE_THREADS = (16, 32, 64, 128)   # torch CPU ops of this path slow down beyond ~16-32 threads (measured on the 256-thread GPU host)


def cpu_probe(points, threads, skip, slab=32, runs=3, deadline=60.0):
    """One thread count of the CPU baseline, in THIS process: 1 warm-up + up to `runs` timed runs (as many as end before
    `deadline` seconds, at least one) of the oracle (oracle/so3_ref.py) fwd+bwd of the 3 layers on the first `slab` query
    points of one `points`-point cloud (slab == points: the whole cloud, nothing scaled) -> dict."""
    import synth_clouds
    from oracle import so3_ref
    consts = np.load(os.path.join(PKG, 'vgtk', 'data', 'anchors', 'constants.npz'))
    import vgtk.so3conv.functional as L
    anchors = torch.from_numpy(np.ascontiguousarray(L.get_anchors()))
    xyz, _, pose = synth_clouds.laptop_batch(0, 1, points)
    xyz, pose = torch.from_numpy(xyz), torch.from_numpy(pose)
    # away from the cores the bench process launches its GPU work from (the probes run beside the bench's side legs): the upper
    # half of the host's logical CPUs when the thread count fits there
    ncpu = os.cpu_count() or 1
    if hasattr(os, 'sched_setaffinity') and 4 * threads <= ncpu:      # (never fewer than two logical CPUs per thread: 128 threads pinned
        try:                                                             #  to 128 logical CPUs took 456 s for what 16 threads do in 1.4 s)
            os.sched_setaffinity(0, set(range(ncpu - 2 * threads, ncpu)))
        except OSError:
            pass
    torch.set_num_threads(threads)
    slab = min(slab, points)
    t_start = time.perf_counter()

    def one_run():
        total = 0.0
        gen = torch.Generator().manual_seed(2913)
        for (c, o, r, s) in synth_clouds.backbone_layers(points):
            kern = torch.from_numpy(so3_ref.kernel_points(consts['kpsphere24'], 0.7 * r))
            feats = (torch.ones(1, 1, points, NA) if c == 1 else torch.randn(1, c, points, NA, generator=gen)).requires_grad_(c > 1)
            W = torch.randn(o, c * KS, generator=gen).requires_grad_(True)
            t0 = time.perf_counter()
            fs = so3_ref.add_shadow_feature(feats)
            res = so3_ref._poseconv_slab(xyz[:, :, :slab].contiguous(), pose[:, :slab].contiguous(), xyz, pose,
                                         fs, NN, anchors, kern, r, s, 1, skip)
            y = so3_ref.basic_so3conv(W, res[3])
            y.square().mean().backward()
            total += time.perf_counter() - t0
        return total

    warm = one_run()                                    # warm-up
    times = []
    while len(times) < runs and (not times or time.perf_counter() - t_start + times[-1] < deadline):
        times.append(one_run())
    times.sort()
    med = times[len(times) // 2]
    return {'threads': threads, 'clouds_per_sec': 1.0 / (med * points / slab), 'query_points': slab, 'points': points,
            'warmup_s': round(warm, 3), 'runs_s': [round(t, 3) for t in times]}


def cpu_baseline(points, slab=256):
    """The CPU oracle beside the GPU number, by BASELINE.md section 2's protocol (1 warm-up + 3 timed runs, median), in child
    processes without a GPU, each under a time budget (a run that would overrun it is not started: fewer than 3 timed runs are
    reported as such, never a missing entry):
      1. thread sweep {16, 32, 64, 128} (those the host has) on a 32-point slab of the `points`-point cloud -> the fastest count
         (the smallest count first, the wider ones after the runs below: see the comment in the body);
      2. at that count: a slab of `slab` (256) query points, scaled by points / slab -> `value`;
      3. at that count: config 1 of BASELINE.json DIRECTLY -- one whole 512-point cloud, nothing scaled -> `config1`;
      4. at that count: 2. with the reference's 60x60 anchor-permutation search short-circuited (identity poses)."""
    import subprocess

    def start(threads, skip, pts, slab_, deadline):
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-probe', str(threads), '--points', str(pts), '--probe-slab', str(slab_),
               '--probe-deadline', str(deadline)] + (['--probe-skip-search'] if skip else [])
        env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
        return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)

    def finish(proc, limit=None, what=None):
        t0 = time.perf_counter()
        try:
            out, err = proc.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.communicate()
            dt = time.perf_counter() - t0
            # (a thread count at which not even the warm-up run ends within the limit: reported as the bound that follows from it)
            return dict(what, clouds_per_sec=0.0, clouds_per_sec_upper_bound=1.0 / (dt * points / 32), note=f'stopped after {dt:.0f} s inside its warm-up run')
        lines = [ln for ln in out.splitlines() if ln.startswith('{')]
        if proc.returncode != 0 or not lines:
            raise RuntimeError('cpu baseline probe failed: ' + err[-400:])
        return json.loads(lines[-1])

    ncpu = os.cpu_count() or 1
    counts = [t for t in CPU_BASELINE_THREADS if t <= ncpu] or [ncpu]
    probe = lambda t: finish(start(t, False, points, 32, 8.0), limit=30.0, what={'threads': t, 'query_points': 32, 'points': points})
    # Order: the smallest count of the sweep, the three protocol runs at it, THEN the wider counts of the sweep.  The wide probes (64
    # threads and up cannot be pinned away from the bench's launch thread) disturb the launch-heavy GPU side legs this job runs beside
    # (articulated input: 93 -> 77 clouds/s, profiles/r05_o_bench.json) -- at the end of the job those legs are over.  Should a wider
    # count turn out faster, the three runs are repeated at it.
    sweep = [probe(counts[0])]
    best = counts[0]

    def protocol_runs(threads):
        # one after the other (side by side they slow each other down by 1.8 x on the 256-thread host: memory bandwidth,
        # profiles/r05_h_bench.json)
        return (finish(start(threads, False, points, slab, 75.0)), finish(start(threads, False, 512, 512, 80.0)), finish(start(threads, True, points, slab, 30.0)))

    main, config1, short = protocol_runs(best)
    sweep += [probe(t) for t in counts[1:]]
    fastest = max(sweep, key=lambda d: d['clouds_per_sec'])['threads']
    if fastest != best:
        best = fastest
        main, config1, short = protocol_runs(best)
    side_by_side = False
    model = 'unknown'
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                model = ln.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'value': main['clouds_per_sec'], 'unit': 'point-clouds/sec', 'cores': best, 'kind': 'port',
            'host_logical_cpus': ncpu, 'host_cpu_model': model,
            'protocol': 'BASELINE.md section 2: 1 warm-up + 3 timed runs, median (fewer timed runs where three would overrun the '
                        'probe\'s time budget: see runs_s); torch.set_num_threads at each listed count; every probe in a child process without a GPU',
            'thread_sweep': {'sample': f'32 of {points} query points', 'by_threads': sweep, 'fastest': best},
            'probes_side_by_side': side_by_side,
            'runs_s': main['runs_s'], 'warmup_s': main['warmup_s'],
            'sample': f'oracle fwd+bwd of the 3 backbone layers on {main["query_points"]} of {points} query points of 1 cloud (every op of the path is '
                      f'independent across query points; the support is the whole cloud), time x {points}/{main["query_points"]}; includes the '
                      f'reference\'s 60x60 anchor-permutation search',
            'config1': {'value': config1['clouds_per_sec'], 'unit': 'point-clouds/sec', 'cores': best, 'runs_s': config1['runs_s'], 'warmup_s': config1['warmup_s'],
                        'sample': 'BASELINE.json config 1 directly: ONE whole 512-point cloud (512-point radii), fwd+bwd of the 3 backbone layers, nothing scaled'},
            'value_perm_search_short_circuited': short['clouds_per_sec'], 'short_circuit_runs_s': short['runs_s']}


def expected_scaling(world, ms_per_step, n_params):
    """What the N-rank line should look like, written down BEFORE anyone has run this path on RCCL: per step
    and rank the compute is the single-GPU step (weak scaling: the same clouds per GPU) plus
      * the gradient all-reduce (4 n_params bytes, ring over xGMI: 2 (N - 1) / N of the bytes per link at ~40 GB/s effective for
        a few-MB message + 2 (N - 1) hops of ~10 us), launched from autograd hooks: the deepest layer's bucket (89 % of the
        bytes) travels under the ~10 ms of the two shallower layers' backward -- only the last, small bucket is exposed;
      * one all-gather of 5.76 kB per cloud of pose hypotheses (latency-bound, ~(N - 1) x 10 us by the default ring);
      * 6 SyncBatchNorm moment exchanges (3 forward, 3 backward, <= 8 kB each): blocking, ~(N - 1) x 10 us each by the ring;
      * the max over ranks of box-to-box speed differences (+-3 % between boxes measured this round) and barrier skew.
    -> estimate of the exposed communication per step and the efficiency band the first SCALE record should fall in."""
    if world <= 1:
        return None
    hop_us = 10.0
    last_bucket_bytes = 4.0 * 64 * 24                           # the first layer's weights (+ the stand-in head): what cannot overlap
    allreduce_exposed = 2 * (world - 1) * hop_us * 1e-3 + 2.0 * (world - 1) / world * last_bucket_bytes / 40e9 * 1e3
    allgather = (world - 1) * hop_us * 1e-3
    syncbn = 6 * (world - 1) * hop_us * 1e-3
    exposed = allreduce_exposed + allgather + syncbn
    return {'exposed_comm_ms_per_step_estimate': exposed,
            'gradient_allreduce_bytes': 4.0 * n_params, 'pose_allgather_bytes_per_rank': 5760.0 * 16,
            'efficiency_from_comm_alone': ms_per_step / (ms_per_step + exposed),
            'expected_efficiency_band': [0.94, 0.995],
            'note': 'weak scaling, no data-path collective; the band is the comm estimate widened by the +-3 % box-to-box spread (the line is the MAX '
                    'over ranks) -- an efficiency below 0.94 means an exchange is NOT overlapped / is slower than a latency-bound ring and should be '
                    'looked at first: SyncBatchNorm moments (blocking), then the hook-launched buckets'}


def distributed_report(world, backend, dev):
    """rank, device and the communication-library settings of every rank, gathered on rank 0 (the first RCCL run must be readable)."""
    info = {'rank': int(os.environ.get('RANK', 0)), 'local_rank': int(os.environ.get('LOCAL_RANK', 0)), 'device': str(dev),
            'device_name': torch.cuda.get_device_name(dev), 'pid': os.getpid(),
            'env': {k: v for k, v in sorted(os.environ.items()) if k.startswith(('NCCL_', 'RCCL_', 'HSA_', 'HIP_VISIBLE', 'ROCR_VISIBLE', 'TORCH_NCCL', 'MASTER_'))}}
    if world == 1:
        return [info]
    out = [None] * world
    dist.all_gather_object(out, info)
    return out


def zpconv_roofline(dev, points, clouds=8, channels=64):
    """The standalone native zpconv ops (vgtk.cuda.zpconv.inter_zpconv_forward / _backward, the op the north
    star puts an HBM-roofline target on): algorithmic bytes = idx + w read once, feats / grad
    read once, out / gfeats written once; time = HIP events around groups of 3 launches after 3 warm-up launches, median of 5 groups.  Separate from
    the timed steps (the shipped models never call this op: they use the fused grouping)."""
    import synth_clouds
    import vgtk.cuda.zpconv as Z
    import vgtk.cuda.grouping as G
    torch.cuda.synchronize()
    torch.cuda.empty_cache()            # (the 12 GB operands of this leg from fresh blocks, whatever the legs before it left cached)
    xyz = torch.from_numpy(synth_clouds.laptop_batch(0, clouds, points)[0]).to(dev)
    radius = synth_clouds.backbone_layers(points)[1][2]
    ball = G.ball_query(xyz, xyz, radius, NN)
    idx = ball[:, :, None, None, :].expand(clouds, points, NA, KS, NN).contiguous()
    w = torch.rand(clouds, points, NA, KS, NN, device=dev)
    feats = torch.randn(clouds, channels, points, NA, device=dev)
    byts = 4.0 * clouds * (2.0 * points * NA * KS * NN + channels * points * NA + channels * KS * points * NA)

    def timed(fn, warm=3, reps=3, groups=5):
        # median over `groups` event-timed groups of `reps` launches (one group of 5 swung by 10 % between runs of the same build on the
        # same box -- 8.96 / 9.76 ms for the backward -- depending on what ran just before)
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(groups):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / reps)
        ms.sort()
        return ms[len(ms) // 2]

    ms = timed(lambda: Z.inter_zpconv_forward(idx, w, feats))
    grad = torch.randn(clouds, channels, KS, points, NA, device=dev)
    ms_b = timed(lambda: Z.inter_zpconv_backward(idx, w, grad, points))
    gbs = byts / ms / 1e6
    # fabric-side bytes of the kernels behind the two entries, from the committed counter passes (profiles/rNN_pmc_traffic.json,
    # per launch; NOT measured in this run -- the object names its source file): the index check reads the 5-D index once per
    # call; the backward keeps its scatter target on chip (csrc/zpconv_bwd_hot.hip), so its kernel moves the operands only
    traffic = None
    pmc = {k: pmc_of_kernel(k) for k in ('zpconv_index_check_kernel', 'zpconv_mfma_kernel', 'zp_hot_kernel')}
    if all(pmc.values()) and clouds == 8 and points == 4096 and channels == 64:
        tot = lambda k: pmc[k]['fetch'] + pmc[k]['write']
        idx_bytes = 4.0 * clouds * points * NA * KS * NN
        fwd_b = idx_bytes + tot('zpconv_mfma_kernel')
        bwd_b = idx_bytes + tot('zp_hot_kernel')
        traffic = {'forward_bytes': fwd_b, 'forward_GBps': fwd_b / ms / 1e6, 'forward_frac_of_peak': fwd_b / ms / 1e6 / 8000.0,
                   'backward_bytes': bwd_b, 'backward_GBps': bwd_b / ms_b / 1e6, 'backward_frac_of_peak': bwd_b / ms_b / 1e6 / 8000.0,
                   'source': pmc['zp_hot_kernel'].get('source'),
                   'per_launch': {k: {'fetch': v['fetch'], 'write': v['write'], 'l2_hit': v.get('l2_hit')} for k, v in pmc.items()},
                   'note': 'FETCH_SIZE x 2 + WRITE_SIZE of separate rocprofv3 --pmc passes (committed file, not this run)'}
    return {'bound': 'hbm', 'traffic': traffic, 'kernel': 'zpconv_index_check_kernel + zpconv_mfma_kernel (v_mfma_f32_32x32x2_f32, streamed weights)',
            'entry': 'eap_inter_zpconv_fwd_ws_f32', 'achieved': gbs,
            'peak': 8000.0, 'unit': 'GB/s', 'frac': gbs / 8000.0, 'ms': ms, 'bytes': byts,
            'backward': {'entry': 'eap_inter_zpconv_bwd_hot_f32',
                         'kernels': 'zp_hot_rows_kernel + zp_hot_kernel with zpconv_index_check_kernel streaming beside it (v_mfma_f32_32x32x2_f32, scatter target in LDS: '
                                    'no per-(point, neighbour) intermediate; the weights are read by the two channel halves of an anchor quad)',
                         'ms': ms_b, 'achieved': byts / ms_b / 1e6, 'frac': byts / ms_b / 1e6 / 8000.0,
                         'bytes': byts, 'intermediate_bytes': 0.0,
                         'note': 'algorithmic bytes as the forward (idx + w + grad read, gfeats written); nothing else is written '
                                 '(batches below 8 clouds add one accumulator image per point range, ~36 MB at 2 clouds)'},
            'workload': f'{clouds} x {points} points, C={channels}, A={NA}, K={KS}, NN={NN}, one neighbour list per point '
                        f'broadcast over (a,k) as the Python layer builds it, radius {radius}'}


KERNEL_OF_ENTRY = {   # C-ABI entry -> the HIP kernel that dominates it (fallback when the entry reports no kernel name)
    'eap_gemm_dma_f32': 'gemm_dma_f32_kernel (v_mfma_f32_32x32x2_f32, operands staged global -> VGPR -> LDS, 3-stage ring)',
    'eap_gemm_dma_f32_reduce': 'gemm_dma_f32_kernel (v_mfma_f32_32x32x2_f32, operands staged global -> VGPR -> LDS, 3-stage ring), split-K',
    'eap_gemm_f32': 'gemm_f32_kernel (v_mfma_f32_32x32x2_f32)',
    'eap_gemm_f32_reduce': 'gemm_f32_kernel (v_mfma_f32_32x32x2_f32), split-K',
    'eap_so3_inter_group_fwd_f32': 'so3_group_lists_kernel<false, 0> (v_mfma_f32_32x32x2_f32)',
    'eap_so3_inter_group_fwd_xb_f32': 'so3_group_lists_kernel<false, 1> (v_mfma_f32_32x32x2_f32), blocked output',
    'eap_gemm_f32_xb': 'gemm_f32_kernel (v_mfma_f32_32x32x2_f32), blocked B operand',
    'eap_gemm_f32_reduce_xb': 'gemm_f32_kernel (v_mfma_f32_32x32x2_f32), split-K, blocked B operand',
    'eap_so3_inter_group_fwd_t_f32': 'so3_group_lists_kernel<false, 2> (v_mfma_f32_32x32x2_f32), transposed output',
    'eap_so3_intra_conv_f32': 'gemm_f32_kernel<GATHER> (v_mfma_f32_32x32x2_f32), implicit intra conv',
    'eap_so3_inter_group_inv_f32': 'so3_group_lists_kernel<true, 0> (v_mfma_f32_32x32x2_f32)',
}


def summarize_by_kernel(records):
    """(entry, tag, e0, e1) records -> per HIP kernel (name with template arguments, as reported by the entry through
    eap_last_kernel(); entries that report none are listed under their own name): total ms, launches, algorithmic flops."""
    by_kernel = {}
    for name, tag, e0, e1 in records:
        kname = (tag or {}).get('kernel') or name
        k = by_kernel.setdefault(kname, {'ms': 0.0, 'launches': 0, 'flops': 0.0, 'executed_f16_flops': 0.0, 'entries': set()})
        k['ms'] += e0.elapsed_time(e1)
        k['launches'] += 1
        k['flops'] += (tag or {}).get('flops', 0.0)
        k['executed_f16_flops'] += (tag or {}).get('executed_f16_flops', 0.0)
        k['entries'].add(name)
    return by_kernel


def pmc_file():
    """The newest committed counter summary profiles/rNN_pmc_traffic.json (tools/pmc_traffic.py) -> (path, dict) or (None, {})."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_traffic.json')))
    if not files:
        return None, {}
    return files[-1], json.load(open(files[-1]))


def pmc_of_kernel(kname):
    """Fabric-side bytes per launch + matrix-pipe utilisation of a kernel from the committed rocprofv3 --pmc passes
    (profiles/rNN_pmc_traffic.json, tools/pmc_traffic.py; collected at the default workload, NOT in this run: the entry
    names its source file)."""
    path, d = pmc_file()
    for pat, v in d.get('per_kernel', {}).items():
        # (the profiler prints defaulted template arguments, eap_last_kernel() does not: '<true, 0, false>' is '<true, 0>')
        if kname.startswith(pat) or kname.startswith(re.sub(r',\s*false>$', '>', pat)):
            return dict(v, source=os.path.relpath(path, ROOT), collected=d.get('collected'))
    return None


def roofline_object(kname, k, default_cfg):
    ach = k['flops'] / (k['ms'] * 1e-3) / 1e12 if k['ms'] > 0 else 0.0
    pmc = pmc_of_kernel(kname) if default_cfg else None
    if kname.startswith('kc_gemm_kernel'):
        # the deepest inter conv re-associated over its referenced rows (csrc/so3_dense.hip): per (cloud, anchor) a dense GEMM
        # over ALL referenced rows (rows / nsample times the list kernels' flops: every point meets every referenced row, a 0/1
        # mask keeps its own neighbours) with fp32-accurate products from two fp16 planes per operand (3 fp16 MFMAs per
        # product).  `achieved` / `frac` = the fp16 flops the kernel EXECUTES on the fp16 matrix pipe; `algorithmic` = the flops
        # of the operator as SURVEY.md section 8(d) counts them (2 O P A K nsample per cloud), i.e. what the list kernel was priced on
        ex = k['executed_f16_flops'] / (k['ms'] * 1e-3) / 1e12 if k['ms'] > 0 else 0.0
        return {'bound': 'mfma', 'pipe': 'fp16 (dense product over the referenced rows; 2 x fp16 split operands: 3 fp16 MFMAs per fp32-equivalent product, fp32 accumulate)',
                'kernel': kname, 'entries': sorted(k['entries']), 'achieved': ex, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': ex / PEAK_BF16_MFMA_TFLOPS,
                'algorithmic': {'TFLOPs': ach, 'over_fp32_mfma_peak': ach / PEAK_F32_MFMA_TFLOPS, 'x3_over_fp16_peak': 3.0 * ach / PEAK_BF16_MFMA_TFLOPS,
                                'executed_over_algorithmic_x3': (ex / (3.0 * ach)) if ach > 0 else None},
                'traffic': (pmc['fetch'] + pmc['write']) if pmc else None, 'traffic_detail': pmc, 'launches': k['launches'],
                'avg_launch_ms': k['ms'] / max(k['launches'], 1)}
    if 'f16x2' in kname:
        # fp32 operands, fp32 accumulation, every product as three fp16 MFMA products of two-plane splits (csrc/gemm_bf16x3.hip,
        # gemm_f16x2_kernel): the roofline is the fp16 matrix pipe (same dense peak as bf16), 3 x the algorithmic flops on it
        peak = PEAK_BF16_MFMA_TFLOPS
        return {'bound': 'mfma', 'pipe': 'fp16 (2 x fp16 split of scaled fp32 operands: 3 fp16 MFMAs per fp32-equivalent product, fp32 accumulate)',
                'kernel': kname, 'entries': sorted(k['entries']), 'achieved': 3.0 * ach, 'peak': peak, 'unit': 'TFLOP/s',
                'frac': 3.0 * ach / peak, 'fp32_equivalent_TFLOPs': ach, 'fp32_equivalent_over_fp32_mfma_peak': ach / PEAK_F32_MFMA_TFLOPS,
                'traffic': (pmc['fetch'] + pmc['write']) if pmc else None, 'traffic_detail': pmc, 'launches': k['launches'],
                'avg_launch_ms': k['ms'] / max(k['launches'], 1)}
    if 'bf16x3' in kname:
        # fp32 operands, fp32 accumulation, every product as six bf16 MFMA products (csrc/gemm_bf16x3.hip): the roofline of
        # this kernel is the bf16 matrix pipe, and it executes 6 x the algorithmic flops on it
        peak = PEAK_BF16_MFMA_TFLOPS
        return {'bound': 'mfma', 'pipe': 'bf16 (3 x bf16 split of fp32 operands: 6 bf16 MFMAs per fp32-equivalent product, fp32 accumulate)',
                'kernel': kname, 'entries': sorted(k['entries']), 'achieved': 6.0 * ach, 'peak': peak, 'unit': 'TFLOP/s',
                'frac': 6.0 * ach / peak, 'fp32_equivalent_TFLOPs': ach, 'fp32_equivalent_over_fp32_mfma_peak': ach / PEAK_F32_MFMA_TFLOPS,
                'traffic': (pmc['fetch'] + pmc['write']) if pmc else None, 'traffic_detail': pmc, 'launches': k['launches'],
                'avg_launch_ms': k['ms'] / max(k['launches'], 1)}
    return {'bound': 'mfma', 'kernel': kname, 'entries': sorted(k['entries']), 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS,
            'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS,
            'traffic': (pmc['fetch'] + pmc['write']) if pmc else None,
            'traffic_detail': pmc, 'launches': k['launches'], 'avg_launch_ms': k['ms'] / max(k['launches'], 1)}


def summarize_kernels(records):
    """(name, tag, e0, e1) records of the timed steps -> per C-ABI entry: total ms, launches,
    algorithmic flops; and per launch shape."""
    by_name, by_shape = {}, {}
    for name, tag, e0, e1 in records:
        ms = e0.elapsed_time(e1)
        b = by_name.setdefault(name, {'ms': 0.0, 'launches': 0, 'flops': 0.0})
        b['ms'] += ms
        b['launches'] += 1
        if tag is not None:
            b['flops'] += tag['flops']
            s = by_shape.setdefault((name, tag['shape']), {'entry': name, 'shape': list(tag['shape']), 'ms': 0.0,
                                                           'launches': 0, 'flops': 0.0})
            s['ms'] += ms
            s['launches'] += 1
            s['flops'] += tag['flops']
    shapes = sorted(by_shape.values(), key=lambda d: -d['ms'])
    for s in shapes:
        s['avg_ms'] = s['ms'] / s['launches']
        s['tflops'] = s['flops'] / (s['ms'] * 1e-3) / 1e12
        del s['ms'], s['flops']
    return by_name, shapes


def quick_run(dev, batch, points, fwd_only=False, plan_points=None, steps=3, warmup=2, partial=False, part_poses=False):
    """One more configuration of the same step on this GPU, a few steps: -> dict(value, ms_per_step, ...).
    part_poses: every rigid part of a cloud carries its own rotation (articulated input: the relative rotations between
    neighbours across the joint select real anchor permutations, vgtk/so3conv/functional.py:L1199-1204)."""
    import synth_clouds
    from vgtk import _hip
    torch.manual_seed(2913)
    model = Backbone(points, plan_points).to(dev)
    params = [p for p in model.parameters()]
    opt = torch.optim.Adam(params, lr=1e-4)
    xyz_np, lab_np, pose_np = synth_clouds.laptop_batch(0, batch, points, partial=partial)
    if part_poses:
        rng = np.random.default_rng(2913)
        q = rng.standard_normal((batch, 2, 4))
        q /= np.linalg.norm(q, axis=-1, keepdims=True)
        w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
        rot = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                        2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(batch, 2, 3, 3)
        pose_np = pose_np.copy()
        for bi in range(batch):
            pose_np[bi, :, :3, :3] = rot[bi][lab_np[bi]]
    xyz, pose = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(pose_np.astype(np.float32)).to(dev)

    def step():
        if fwd_only:
            with torch.no_grad():
                model.hypotheses(model(xyz, pose))
            return
        opt.zero_grad(set_to_none=True)
        feats = model(xyz, pose)
        StandInLoss.apply(feats, model.pose_head.weight, model.pose_head.bias).backward()
        opt.step()

    import vgtk.so3conv.functional as L
    for _ in range(warmup - 1):
        step()
    L.BACKWARD_LOG = []
    step()
    regimes, L.BACKWARD_LOG = L.BACKWARD_LOG, None
    torch.cuda.synchronize()
    _hip.KERNEL_TIMES = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    records, _hip.KERNEL_TIMES = _hip.KERNEL_TIMES, None
    kern, _ = summarize_kernels(records)
    top = sorted(kern.items(), key=lambda kv: -kv[1]['ms'])[:4]
    out = {'clouds_per_gpu': batch, 'points': points, 'clouds': 'partial (depth-buffer visible)' if partial else 'complete',
           'poses': 'one rotation per rigid part' if part_poses else 'identity', 'backward_regimes': regimes, 'pass': 'fwd' if fwd_only else 'fwd+bwd+Adam',
           'radii_of_input_size': plan_points or points, 'value': batch * steps / dt, 'unit': 'point-clouds/sec',
           'ms_per_step': dt / steps * 1e3, 'steps': steps,
           'top_kernels_ms_per_step': {n: round(k['ms'] / steps, 2) for n, k in top}}
    del model, opt, xyz, pose
    torch.cuda.empty_cache()
    return out


def other_configs(dev):
    """Short runs of the other BASELINE.json shapes on one GPU (each a few steps, same code path as the timed
    run): config 2 (forward only), configs 3 / 4 per-GPU shape (16 clouds), config 5 per-GPU shape (8 clouds of
    8192 points), and the small-radius regime in which more than a quarter of the support rows are referenced and
    the backward takes the textbook dX route."""
    return [
        dict(name='config 2: 8 x 4096, forward only', short='cfg2_fwd_8x4096', **quick_run(dev, 8, 4096, fwd_only=True)),
        dict(name='configs 3/4 per-GPU shape: 16 x 4096', short='cfg4_shard_16x4096', **quick_run(dev, 16, 4096)),
        dict(name='config 5 per-GPU shape: 8 x 8192 partial (depth-buffer visible) clouds', short='cfg5_shard_8x8192_partial', **quick_run(dev, 8, 8192, partial=True)),
        dict(name='8 x 8192 complete clouds', short='8x8192_complete', **quick_run(dev, 8, 8192)),
        dict(name='8 x 4096 with the 512-point radii (textbook-backward regime)', short='8x4096_radii512', **quick_run(dev, 8, 4096, plan_points=512)),
        dict(name='8 x 4096, articulated input: one rotation per rigid part (anchor permutations on)', short='8x4096_articulated', **quick_run(dev, 8, 4096, part_poses=True)),
        # the reference's own operating point (scripts/train/laptop_syn.sh:L9, L23: 8 processes x batch 1 x 380-512 points; SURVEY.md
        # 8(d) asks for [16, 3, 512] beside config 3): 16 clouds of 512 points with the radii build_model derives for 512 points
        dict(name='reference operating point: 16 x 512-pt clouds, 512-point radii', short='ref_point_16x512', **quick_run(dev, 16, 512, steps=10, warmup=3)),
    ]


def config3_step(dev, batch=16, points=4096, steps=2, warmup=1):
    """BASELINE config 3 as one unit (config3_step.py): frozen separable glb_backbone forward + backbone and backbone_sec
    forward/backward + invariant head + batched per-slot pose heads + one chamfer pair forward/backward + Adam."""
    import synth_clouds
    import config3_step as C3
    from vgtk import _hip
    torch.manual_seed(2913)
    model = C3.Config3Model(points).to(dev)
    opt = torch.optim.Adam(model.trained_parameters(), lr=1e-4)
    xyz_np, _, pose_np = synth_clouds.laptop_batch(0, batch, points)
    xyz, pose = torch.from_numpy(xyz_np).to(dev), torch.from_numpy(pose_np).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _ = model(xyz, pose)
        loss.backward()
        opt.step()
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(dev)
    _hip.KERNEL_TIMES = []
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    records, _hip.KERNEL_TIMES = _hip.KERNEL_TIMES, None
    kern = summarize_by_kernel(records)
    # flops of the launches actually made (each C-ABI launch carries its algorithmic count; the re-associated backward
    # does less than the textbook 2 x forward, and the torch layers of the stand-ins are not counted): a LOWER bound
    flops = sum(k['flops'] for k in kern.values()) / steps
    top = sorted(kern.items(), key=lambda kv: -kv[1]['ms'])[:6]
    out = {'workload': f'{batch} x {points}-pt clouds: 3 separable blocks forward (frozen, no_grad) + 2 x 3 inter blocks forward+backward + '
                       f'InvPPOutBlockOurs + {C3.SLOTS} batched SO3OutBlockRTWithMaskSep heads + chamfer [{batch},{points},3] forward+backward + Adam',
           'value': batch * steps / dt, 'unit': 'point-clouds/sec', 'ms_per_step': dt / steps * 1e3, 'steps': steps, 'warmup': warmup,
           'loss': float(loss.detach()), 'peak_memory_GB': torch.cuda.max_memory_allocated(dev) / 2 ** 30,
           'whole_step': {'algorithmic_flops_of_the_hot_path_launches': flops, 'textbook_flops': C3.algorithmic_flops(batch, points, synth_clouds.backbone_layers(points)), 'achieved_TFLOPs': flops / (dt / steps) / 1e12,
                          'frac_of_fp32_mfma_peak': flops / (dt / steps) / 1e12 / PEAK_F32_MFMA_TFLOPS},
           'kernel_time_share': sum(k['ms'] for k in kern.values()) / (dt * 1e3),
           'top_kernels_ms_per_step': {n: round(k['ms'] / steps, 2) for n, k in top}}
    del model, opt, xyz, pose
    torch.cuda.empty_cache()
    return out


LINE_BUDGET = 6000          # bytes of the LAST stdout line (the driver's record keeps a bounded tail: round 5's 20 KB line did not parse)


def _clean(obj):
    """NaN / inf -> None, numpy scalars -> Python, floats to 6 significant digits (strict JSON, short)."""
    if isinstance(obj, dict):
        return {str(k): _clean(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple, set)):
        return [_clean(v) for v in (sorted(obj) if isinstance(obj, set) else obj)]
    if isinstance(obj, (np.floating, float)):
        f = float(obj)
        return float(f'{f:.6g}') if np.isfinite(f) else None
    if isinstance(obj, (np.integer,)):
        return int(obj)
    if isinstance(obj, (np.bool_,)):
        return bool(obj)
    return obj


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _short_roofline(r):
    """The judged fields of a roofline object (bench contract) without the prose."""
    out = _pick(r, 'kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'launches', 'share_of_kernel_time')
    if 'pipe' in r:
        out['pipe'] = r['pipe'].split(' ')[0]                       # 'fp16' / 'bf16'
    if 'algorithmic' in r:
        out['algorithmic'] = r['algorithmic']
    if 'fp32_equivalent_TFLOPs' in r:
        out['fp32_equivalent_TFLOPs'] = r['fp32_equivalent_TFLOPs']
    td = r.get('traffic_detail') or {}
    if td:
        out['traffic_source'] = td.get('source')
        out['mfma_util'] = td.get('mfma_util')
    return out


def compact_line(full, detail_file='bench_detail.json'):
    """The ONE line the driver parses: the contract's keys, `roofline`, `cpu_baseline`, and one number (or a handful) per side leg --
    everything else (`kernels`, `launch_shapes`, sweeps, notes, per-kernel counter detail) stays in `detail_file`, written next to it.
    Strict JSON (no NaN), a few KB: tests/test_bench_line.py builds it from a canned record and checks both."""
    line = _pick(full, 'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                 'dtype', 'data', 'rccl_ranks', 'config')
    if 'roofline' in full:
        line['roofline'] = _short_roofline(full['roofline'])
    cb = full.get('cpu_baseline')
    if isinstance(cb, dict):
        c = _pick(cb, 'value', 'unit', 'cores', 'kind', 'host_logical_cpus', 'host_cpu_model', 'runs_s', 'warmup_s', 'error', 'value_perm_search_short_circuited')
        if 'sample' in cb:
            c['sample'] = cb['sample'][:160]
        if isinstance(cb.get('config1'), dict):
            c['config1'] = _pick(cb['config1'], 'value', 'runs_s')
        if isinstance(cb.get('thread_sweep'), dict):
            c['thread_sweep'] = {str(p['threads']): p.get('clouds_per_sec') for p in cb['thread_sweep'].get('by_threads', [])}
        line['cpu_baseline'] = c
    if 'speedup_vs_cpu_baseline' in full:
        line['speedup_vs_cpu_baseline'] = full['speedup_vs_cpu_baseline']
    zr = full.get('zpconv_roofline')
    if isinstance(zr, dict):
        z = _pick(zr, 'bound', 'achieved', 'peak', 'unit', 'frac', 'ms', 'bytes')
        if isinstance(zr.get('backward'), dict):
            z['backward'] = _pick(zr['backward'], 'frac', 'ms')
        tr = zr.get('traffic') or {}
        if tr.get('forward_bytes') and zr.get('bytes'):
            z['traffic_ratio'] = tr['forward_bytes'] / zr['bytes']
            z['traffic_source'] = tr.get('source')
        line['zpconv_roofline'] = z
    if full.get('kernel_rooflines'):
        line['kernel_rooflines'] = [dict(_pick(r, 'kernel', 'frac', 'avg_launch_ms', 'share_of_kernel_time'),
                                         **({'executed_over_algorithmic_x3': r['algorithmic'].get('executed_over_algorithmic_x3')} if 'algorithmic' in r else {}))
                                    for r in full['kernel_rooflines'][:6]]
    if 'whole_step' in full:
        line['whole_step'] = _pick(full['whole_step'], 'achieved_TFLOPs_per_gpu', 'frac_of_fp32_mfma_peak', 'kernel_time_share_of_step')
    if full.get('other_configs'):
        line['other_configs'] = {c.get('short', c['name']): c['value'] for c in full['other_configs']}
        line['other_configs_regimes'] = {c.get('short', c['name']): '/'.join(r.get('regime', '?') for r in (c.get('backward_regimes') or []))
                                         for c in full['other_configs'] if c.get('backward_regimes')}
    if isinstance(full.get('config3_step'), dict):
        line['config3_step'] = _pick(full['config3_step'], 'value', 'unit', 'ms_per_step', 'peak_memory_GB', 'steps')
    for k in ('fp32_mfma_contraction', 'bf16x3_contraction'):
        if isinstance(full.get(k), dict):
            line[k] = _pick(full[k], 'value')
    for k in ('backend', 'functional_check_only'):
        if k in full:
            line[k] = full[k]
    if isinstance(full.get('expected_scaling'), dict):
        line['expected_scaling'] = _pick(full['expected_scaling'], 'exposed_comm_ms_per_step_estimate', 'efficiency_from_comm_alone', 'expected_efficiency_band')
    if full.get('ranks'):
        line['ranks'] = [_pick(r, 'rank', 'device', 'device_name') for r in full['ranks']]
    line['detail'] = detail_file
    line = _clean(line)
    # the budget is a hard one: drop the least judged objects first rather than print a line the record cannot hold
    for k in ('ranks', 'other_configs_regimes', 'kernel_rooflines', 'whole_step', 'bf16x3_contraction', 'fp32_mfma_contraction', 'expected_scaling'):
        if len(json.dumps(line, allow_nan=False)) <= LINE_BUDGET:
            break
        line.pop(k, None)
    return line


def emit(full):
    """Detail record -> bench_detail.json (repo root; a copy under gpurun_out/ when that directory exists, so a gpurun call brings it
    back) and the compact line as the LAST line of stdout."""
    detail = _clean(full)
    text = json.dumps(detail, allow_nan=False, indent=1)
    for d in (ROOT, os.path.join(ROOT, 'gpurun_out')):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, 'bench_detail.json'), 'w') as f:
                    f.write(text)
            except OSError:
                pass
    sys.stdout.flush()
    print(json.dumps(compact_line(detail), allow_nan=False), flush=True)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv=None):
    """`python bench.py --gpus N` started WITHOUT a launcher: start N ranks of this file through torch.distributed.run, one per
    GPU over RCCL (the reference starts its 8 processes the same way, scripts/train/laptop_syn.sh:L23
    `python -m torch.distributed.launch --nproc_per_node=8`).  On a box with fewer than N devices the ranks share the
    devices round-robin and talk over gloo -- a functional check of the N > 1 path, flagged as such in the line
    (`rccl_ranks: 0`, `functional_check_only`); it is not a scaling measurement.  -> exit code of the job."""
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    n_dev = torch.cuda.device_count()
    if 'EAP_DIST_BACKEND' not in env:
        env['EAP_DIST_BACKEND'] = 'nccl' if n_dev >= n else 'gloo'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def dist_setup(args):
    """Rank / world / device of this process from the launcher's environment; refuses a world size that is not --gpus
    (a line whose n_gpus differs from the request would be a wrong scaling record)."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to print a line '
                         f'whose n_gpus is not the request')
    n_dev = torch.cuda.device_count()
    backend = os.environ.get('EAP_DIST_BACKEND', 'nccl')
    if n_dev == 0:
        dev = torch.device('cpu')                     # --check-launch on a box without a GPU (tests/test_sharding_gloo.py)
        backend = 'gloo'
    else:
        if backend == 'nccl' and world > n_dev:
            raise SystemExit(f'bench.py: {world} RCCL ranks need {world} devices, {n_dev} visible (EAP_DIST_BACKEND=gloo runs the '
                             f'functional check with shared devices)')
        # (gloo + several ranks on one device: a functional check of the N > 1 path on a 1-GPU box only)
        local = local % n_dev if backend != 'nccl' else local
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus
    return rank, world, dev, backend, n_dev


def check_launch(args):
    """`--check-launch`: the ranks rendezvous, run the path's two exchanges (pose-hypothesis all-gather over uneven shards,
    hook-driven gradient all-reduce) on small tensors and rank 0 prints what was started.  Needs no GPU."""
    from vgtk import sharding
    rank, world, dev, backend, n_dev = dist_setup(args)
    n_items = 2 * world + 1
    start, stop = sharding.shard_range(n_items)
    gen = torch.Generator().manual_seed(5)
    R_all, T_all = torch.randn(n_items, SLOTS, NA, 3, 3, generator=gen), torch.randn(n_items, SLOTS, NA, 3, generator=gen)
    R, Tt = sharding.all_gather_pose_hypotheses(R_all[start:stop].to(dev), T_all[start:stop].to(dev), n_items=n_items)
    ok = torch.equal(R.cpu(), R_all) and torch.equal(Tt.cpu(), T_all)
    torch.manual_seed(3)
    lin = nn.Linear(6, 4).to(dev)
    reducer = sharding.GradientReducer(list(lin.parameters()))
    x = torch.randn(8, 6, generator=torch.Generator().manual_seed(10 + rank)).to(dev)
    lin(x).square().sum().backward()
    reducer.finish()
    g = lin.weight.grad.detach().clone()
    if world > 1:
        g0 = g.clone()
        dist.broadcast(g0, 0)
        ok = ok and torch.equal(g0, g)
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    if rank == 0:
        print(json.dumps({'launch_check': ok, 'n_gpus': world, 'requested_gpus': args.gpus, 'backend': backend,
                          'rccl_ranks': world if backend == 'nccl' and world > 1 else 0, 'devices_visible': n_dev}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8, help='clouds per GPU per step')
    ap.add_argument('--points', type=int, default=4096)
    ap.add_argument('--fwd-only', action='store_true', help='BASELINE config 2 (forward only)')
    ap.add_argument('--separable', action='store_true', help='the separable (inter + intra + skip) glb_backbone instead of the inter backbone; implies --fwd-only as in the reference')
    ap.add_argument('--partial', action='store_true', help='partial (depth-buffer visible) clouds: BASELINE config 5 with --points 8192')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--plan-points', type=int, default=None, help='radii / sigmas of the backbone built for this input size (default: --points)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the short runs of the other BASELINE configurations (config 3 composite included)')
    ap.add_argument('--plain', action='store_true', help='warm-up + timed steps only (no attribution loop, no A/B leg, no extras): what the counter passes of tools/gpu/profile_round.sh run')
    ap.add_argument('--cpu-baseline-probe', type=int, default=0, help=argparse.SUPPRESS)      # child of cpu_baseline(): one thread count, no GPU
    ap.add_argument('--probe-skip-search', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--probe-slab', type=int, default=32, help=argparse.SUPPRESS)
    ap.add_argument('--probe-deadline', type=float, default=60.0, help=argparse.SUPPRESS)
    ap.add_argument('--check-launch', action='store_true', help='start the ranks, run the two exchanges on small tensors, print what was started (no GPU needed)')
    args = ap.parse_args(argv)

    if args.cpu_baseline_probe > 0:
        print(json.dumps(cpu_probe(args.points, args.cpu_baseline_probe, args.probe_skip_search, slab=args.probe_slab, deadline=args.probe_deadline)))
        return
    if args.gpus > 1 and 'RANK' not in os.environ:
        # no launcher around us: start the N ranks ourselves (the driver's N = 1 command form with --gpus N)
        sys.exit(launch_ranks(args.gpus, argv))
    if args.check_launch:
        sys.exit(check_launch(args))

    t_start = time.perf_counter()
    rank, world, dev, backend, n_dev = dist_setup(args)
    if n_dev == 0:
        raise SystemExit('bench.py: no GPU visible (this package has no CPU path)')

    import synth_clouds
    from vgtk import _hip, sharding

    torch.manual_seed(2913)
    if args.separable:
        args.fwd_only = True
    model = (SeparableBackbone(args.points) if args.separable else Backbone(args.points, args.plan_points)).to(dev)
    if args.separable:
        model.eval()                    # the reference runs this stage frozen, in eval mode (trainer_unsup_arti_align.py:L594-597)
    for m in model.modules():           # the reference converts its BatchNorms to SyncBatchNorm for multi-GPU runs
        if hasattr(m, 'sync') and hasattr(m, 'negative_slope'):
            m.sync = world > 1
    conv_params = [p for p in model.parameters()]
    opt = torch.optim.Adam(conv_params, lr=1e-4)
    # gradient all-reduce launched from autograd hooks: the deepest layer's bucket travels while the shallower layers'
    # backward kernels still run (vgtk/sharding.py); a no-op on one GPU
    reducer = sharding.GradientReducer(conv_params)
    n_items = args.batch * world
    xyz_np, _, pose_np = synth_clouds.laptop_batch(rank * args.batch, args.batch, args.points, partial=args.partial)
    xyz = torch.from_numpy(xyz_np).to(dev)
    pose = torch.from_numpy(pose_np).to(dev)

    def step():
        if args.fwd_only:
            with torch.no_grad():
                feats = model(xyz, pose)
                R, T = model.hypotheses(feats)
                sharding.all_gather_pose_hypotheses(R, T, n_items=n_items)
            return
        opt.zero_grad(set_to_none=True)
        feats = model(xyz, pose)
        with torch.no_grad():
            pooled = feats.mean(2)                                        # one pass over the 4 GB feature map for both consumers
            R, T = model.hypotheses(feats, pooled)
        allR, allT = sharding.all_gather_pose_hypotheses(R, T, n_items=n_items)
        loss = StandInLoss.apply(feats, model.pose_head.weight, model.pose_head.bias, pooled)
        loss.backward()
        reducer.finish()
        opt.step()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n):
        """n steps between two barrier + synchronize pairs -> seconds, MAX over ranks."""
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt

    ranks_report = distributed_report(world, backend, dev)          # (collective: every rank)
    for _ in range(args.warmup):
        step()
    # the headline loop: EXACTLY --steps steps, no per-launch events
    dt = timed(args.steps)
    # kernel attribution in a second, short loop: every C-ABI launch bracketed by two HIP events on the launch stream
    if args.plain:
        if rank == 0:
            print(json.dumps({'metric': 'point-clouds/sec (4096 pts, 60 anchors) ' + ('fwd' if args.fwd_only else 'fwd+bwd'), 'value': args.batch * world * args.steps / dt,
                              'unit': 'point-clouds/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
                              'plain': True}))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    attr_steps = min(args.steps, 3)
    _hip.KERNEL_TIMES = []
    dt_attr = timed(attr_steps)
    records, _hip.KERNEL_TIMES = _hip.KERNEL_TIMES, None
    # same-run A/B of the forward contraction: the fp32-MFMA kernel, and the three-bf16-plane split, instead of the two-fp16-plane split
    ab = ab3 = None
    if not args.fwd_only and not args.separable and _hip.SPLIT_BF16_CONTRACTION:
        _hip.SPLIT_BF16_CONTRACTION = False
        step()
        dt_ab = timed(attr_steps)
        _hip.SPLIT_BF16_CONTRACTION = True
        ab = {'value': args.batch * world * attr_steps / dt_ab, 'ms_per_step': dt_ab / attr_steps * 1e3, 'steps': attr_steps,
              'note': 'same run, vgtk._hip.SPLIT_BF16_CONTRACTION = False: the forward contraction on the fp32 matrix pipe (csrc/gemm_dma_f32.hip)'}
        if _hip.SPLIT_PLANES == 2:
            _hip.SPLIT_PLANES = 3
            step()
            dt_ab3 = timed(attr_steps)
            _hip.SPLIT_PLANES = 2
            ab3 = {'value': args.batch * world * attr_steps / dt_ab3, 'ms_per_step': dt_ab3 / attr_steps * 1e3, 'steps': attr_steps,
                   'note': 'same run, vgtk._hip.SPLIT_PLANES = 3: the forward contraction with three bf16 planes per operand (six products, exact splits: round 3)'}

    if rank == 0:
        kern, shapes = summarize_kernels(records)
        by_kernel = summarize_by_kernel(records)
        default_cfg = args.points == 4096 and args.batch == 8 and not args.fwd_only and not args.separable and args.plan_points is None and not args.partial
        # the dominant KERNEL (template instantiation, as rocprofv3 lists them) among those doing matrix work
        dom_name = max((k for k in by_kernel if by_kernel[k]['flops'] > 0), key=lambda k: by_kernel[k]['ms'])
        total_kernel_ms = max(sum(k['ms'] for k in by_kernel.values()), 1e-9)
        roof = roofline_object(dom_name, by_kernel[dom_name], default_cfg)
        roof['share_of_kernel_time'] = by_kernel[dom_name]['ms'] / total_kernel_ms
        roof['flops_definition'] = ('algorithmic: 2*M*N*K*batch for GEMMs; 2*channels*K(24)*P*NN*A*B for the grouping kernels '
                                    '(the MFMA tiles pad K 24->32 and the anchors 60->64, not counted)')
        step_flops = sum(k['flops'] for k in by_kernel.values()) / attr_steps
        clouds = args.batch * world * args.steps
        line = {
            'metric': 'point-clouds/sec (4096 pts, 60 anchors) ' + ('fwd' if args.fwd_only else 'fwd+bwd'),
            'value': clouds / dt, 'unit': 'point-clouds/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'rccl_ranks': world if backend == 'nccl' and world > 1 else 0,
            'dtype_note': 'fp32 tensors, fp32 accumulation everywhere.  The dense product of the deep layers (csrc/so3_dense.hip, BOTH directions), the '
                          'contractions and the small GEMMs around them form their fp32 products on the fp16 matrix cores from two-plane splits '
                          'x = h + l of the stored operand after a power-of-two scale per row (three partial products h h + h l + l h, fp32 accumulate; '
                          '|err| <= 2^-21 sum|a||b|); tests/test_gpu_split_planes.py, tests/test_gpu_dense.py against float64.  Same-run A/B legs: '
                          'bf16x3_contraction (three bf16 planes, six products) and fp32_mfma_contraction (vgtk._hip.SPLIT_PLANES = 3 / '
                          'SPLIT_BF16_CONTRACTION = False)',
            'config': {'workload': f'{args.batch} x {args.points}-pt synthetic ' + ('partial (depth-buffer visible) ' if args.partial else '') + 'laptop clouds per GPU, 3-block '
                                   + ('separable (inter+intra+skip) glb_backbone' if args.separable else 'inter backbone')
                                   + f' 1->64->128->512 (NN=64,K=24,A=60), '
                                   + ('forward' if args.fwd_only else 'forward+backward+Adam'),
                       'clouds_per_gpu': args.batch, 'points': args.points, 'anchors': NA,
                       'sharding': f'clouds x{world}, pose all-gather + gradient all-reduce overlapped with the backward' if world > 1 else 'single GPU'},
            'timing': {'headline_loop': 'no per-launch events', 'ms_per_step_with_launch_events': dt_attr / attr_steps * 1e3,
                       'attribution_steps': attr_steps,
                       'note': 'roofline / kernels / whole_step come from the second loop (two HIP events per C-ABI launch on the launch stream)'},
            'roofline': roof,
            # the same object for every kernel that takes more than 5 % of the kernel time and does matrix work
            'kernel_rooflines': [dict(roofline_object(n, k, default_cfg), share_of_kernel_time=k['ms'] / total_kernel_ms)
                                 for n, k in sorted(by_kernel.items(), key=lambda kv: -kv[1]['ms'])
                                 if k['flops'] > 0 and k['ms'] / total_kernel_ms > 0.05],
            'whole_step': {'algorithmic_flops_per_gpu': step_flops, 'achieved_TFLOPs_per_gpu': step_flops / (dt / args.steps) / 1e12,
                           'frac_of_fp32_mfma_peak': step_flops / (dt / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                           'note': 'algorithmic fp32 flops over the fp32-MFMA peak; part of them (the forward contraction) run as split products on the fp16 / bf16 pipe',
                           'kernel_time_share_of_step': total_kernel_ms / (dt_attr * 1e3)},
            'kernels': {n: {'ms_per_step': k['ms'] / attr_steps, 'launches_per_step': k['launches'] / attr_steps,
                            'tflops': (k['flops'] / (k['ms'] * 1e-3) / 1e12) if k['flops'] > 0 else None}
                        for n, k in sorted(kern.items(), key=lambda kv: -kv[1]['ms'])},
            'dominant_kernel': dom_name,
            'launch_shapes': shapes[:8],
        }
        def progress(what):
            print(f'[bench {time.perf_counter() - t_start:7.1f} s] {what}', file=sys.stderr, flush=True)

        progress('timed steps done')
        if ab is not None:
            line['fp32_mfma_contraction'] = ab
        if ab3 is not None:
            line['bf16x3_contraction'] = ab3
        if world > 1:
            line['ranks'] = ranks_report
            line['backend'] = backend + (' (RCCL)' if backend == 'nccl' else '')
            line['expected_scaling'] = expected_scaling(world, line['ms_per_step'], sum(p.numel() for p in conv_params))
        if world > 1 and backend != 'nccl':
            line['functional_check_only'] = (f'{world} ranks over {backend} on {n_dev} device(s): the N > 1 code path runs, '
                                             f'this is NOT a scaling measurement')
        if world == 1 and default_cfg and not args.no_other_configs:
            del model, opt, xyz, pose
            torch.cuda.empty_cache()
            line['config3_step'] = config3_step(dev)       # (before the CPU probes start: its many small launches feel busy host cores)
            progress('config-3 composite done')
        if world == 1 and default_cfg and not args.no_other_configs:
            line['other_configs'] = other_configs(dev)
            progress('other configurations done')
        if world == 1 and not args.fwd_only:
            line['zpconv_roofline'] = zpconv_roofline(dev, args.points)
            progress('native zpconv done')
        if world == 1 and not args.no_cpu_baseline:
            # child processes on the host cores, AFTER every GPU leg (beside them the wide probes disturbed the launch-heavy side
            # legs: articulated input 93 -> 77 clouds/s in round 5); a failing probe is reported in the line, the GPU numbers stand
            try:
                line['cpu_baseline'] = cpu_baseline(args.points)
                line['speedup_vs_cpu_baseline'] = line['value'] / line['cpu_baseline']['value']
            except Exception as e:          # noqa: BLE001
                line['cpu_baseline'] = {'value': None, 'unit': 'point-clouds/sec', 'cores': 0, 'kind': 'port', 'error': repr(e)[:300]}
            progress('cpu baseline done')
        emit(line)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
