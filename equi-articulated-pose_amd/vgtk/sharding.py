"""Multi-GPU layer of the hot path: one process per GPU, clouds sharded across ranks, RCCL
(torch.distributed backend "nccl" on ROCm) over xGMI for the only two exchanges the path has.

The reference is plain DDP with batch 1 per GPU (SPConvNets/trainer_unsup_arti_align.py:L52,
L430-440); clouds are independent units in the forward, so the data path
needs NO collective.  What crosses GPUs:
  * all-gather of per-part pose hypotheses -- slot_R [B_loc,S,A,3,3] and slot_T [B_loc,S,A,3]
    (cf. ...pn_38_multi_stage.py:L1122-1123), 5.76 kB per cloud at S=2: latency-bound, so both
    tensors travel in ONE flat buffer and one all_gather_into_tensor call;
  * (training) the gradient all-reduce of the conv weights, flattened into one bucket
    (~1.8 M floats for the 3-layer backbone): a single all-reduce instead of one per tensor.
On CPU (tests) the same code runs over gloo.
"""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(n_items, rank=None, world=None):
    """Contiguous split of `n_items` clouds over the ranks -> (start, stop) of this rank."""
    if rank is None:
        rank = dist.get_rank() if is_distributed() else 0
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(n_items, world=None):
    """Number of clouds every rank holds under shard_range's contiguous split."""
    if world is None:
        world = dist.get_world_size() if is_distributed() else 1
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def all_gather_pose_hypotheses(slot_R, slot_T, n_items=None, group=None):
    """slot_R [B_loc,S,A,3,3], slot_T [B_loc,S,A,3] -> the same for all clouds of the job, rank-major
    ([sum of B_loc, ...]).  Ranks may hold DIFFERENT numbers of clouds (shard_range hands out uneven shards when the
    batch does not divide): every rank's block is padded to the largest shard for the one all_gather_into_tensor and the
    padding is dropped afterwards.  The shard sizes follow from `n_items` (the job's cloud count) without communication;
    without it they are exchanged first (one int per rank)."""
    if not is_distributed():
        return slot_R, slot_T
    world = dist.get_world_size(group)
    b = slot_R.shape[0]
    if n_items is not None:
        sizes = shard_sizes(n_items, world)
        if sizes[dist.get_rank(group)] != b:
            raise ValueError(f'all_gather_pose_hypotheses: this rank holds {b} clouds, shard_range({n_items}) says {sizes[dist.get_rank(group)]}')
    else:
        mine = torch.tensor([b], dtype=torch.int64, device=slot_R.device)
        every = torch.empty(world, dtype=torch.int64, device=slot_R.device)
        dist.all_gather_into_tensor(every, mine, group=group)
        sizes = every.tolist()
    bmax = max(sizes)
    flat = torch.cat([slot_R.reshape(b, -1), slot_T.reshape(b, -1)], dim=1)
    if b < bmax:
        flat = torch.cat([flat, flat.new_zeros(bmax - b, flat.shape[1])], dim=0)
    flat = flat.contiguous()
    out = torch.empty(world * bmax, flat.shape[1], dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    if min(sizes) != bmax:
        keep = torch.cat([torch.arange(r * bmax, r * bmax + n, device=out.device) for r, n in enumerate(sizes)])
        out = out.index_select(0, keep)
    nR = slot_R[0].numel() if b else slot_R.new_empty((1,) + tuple(slot_R.shape[1:]))[0].numel()
    total = sum(sizes)
    R = out[:, :nR].reshape(total, *slot_R.shape[1:])
    Tt = out[:, nR:].reshape(total, *slot_T.shape[1:])
    return R, Tt


def all_reduce_gradients(params, average=True, group=None):
    """One bucketed all-reduce over the gradients of `params` (in place)."""
    if not is_distributed():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class GradientReducer:
    """Gradient all-reduce overlapped with the backward pass: the parameters are packed, in REVERSE registration order
    (the order their gradients become ready: the deepest layer first), into buckets of at most `bucket_bytes`; a
    bucket's all-reduce is launched asynchronously once its last gradient has been accumulated
    (Tensor.register_post_accumulate_grad_hook) AND every bucket before it has been launched, so it travels over xGMI
    while the earlier layers' backward kernels still run; finish() waits for the buckets and writes the averaged
    gradients back.  With the 3-block backbone the 6 MB deepest-layer bucket is in flight during the ~15 ms of the two
    shallower layers' backward.

        reducer = GradientReducer(params)        # once
        loss.backward(); reducer.finish(); optimizer.step()

    Every rank issues the SAME collectives in the SAME order whatever its autograd graph looked like this step (the
    reference gets this from DistributedDataParallel, trainer_unsup_arti_align.py:L432-440): buckets launch strictly in
    index order -- a complete bucket waits for its predecessors, which finish() launches if a data-dependent branch
    left them incomplete on this rank.  A parameter without a gradient on this rank contributes zeros (whatever an
    earlier step left in the buffer is cleared); each bucket carries one "fired" count per parameter, so a parameter
    that received a gradient on ANY rank ends with the averaged gradient on EVERY rank (p.grad is materialised where it
    was None -- the replicas stay identical), and one that no rank touched keeps p.grad = None.  A backward that reaches
    a bucket whose all-reduce is still in flight (a second backward before finish()) raises.

    Single-process runs register nothing and finish() is a no-op."""

    def __init__(self, params, bucket_bytes=8 << 20, average=True, group=None):
        self.group, self.average = group, average
        self.buckets, self._handles = [], []
        self._next = 0                               # index of the first bucket not launched yet in this step
        if not is_distributed():
            return
        params = [p for p in params if p.requires_grad]
        cur, cur_bytes = [], 0
        for p in reversed(params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._add_bucket(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._add_bucket(cur)

    def _add_bucket(self, params):
        n = sum(p.numel() for p in params)
        # [gradients of the bucket's parameters | one "fired" flag per parameter]
        flat = torch.zeros(n + len(params), dtype=params[0].dtype, device=params[0].device)
        bucket = {'params': params, 'flat': flat, 'n': n, 'fired': [False] * len(params), 'work': None, 'offsets': []}
        off = 0
        for p in params:
            bucket['offsets'].append(off)
            off += p.numel()
        for i, p in enumerate(params):
            self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(len(self.buckets), i)))
        self.buckets.append(bucket)

    def _launch(self, bucket):
        for i, (p, o) in enumerate(zip(bucket['params'], bucket['offsets'])):
            if not bucket['fired'][i]:
                bucket['flat'][o:o + p.numel()].zero_()
                bucket['flat'][bucket['n'] + i].zero_()
        bucket['work'] = dist.all_reduce(bucket['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _launch_complete_prefix(self):
        while self._next < len(self.buckets) and all(self.buckets[self._next]['fired']):
            self._launch(self.buckets[self._next])
            self._next += 1

    def _make_hook(self, bucket_index, slot):
        def hook(p):
            bucket = self.buckets[bucket_index]
            if bucket['work'] is not None:
                raise RuntimeError('GradientReducer: a gradient arrived for a bucket whose all-reduce is in flight -- finish() must follow '
                                   'EVERY backward (accumulating several backward passes into one reduction is not supported)')
            o = bucket['offsets'][slot]
            bucket['flat'][o:o + p.numel()].copy_(p.grad.reshape(-1))      # p.grad is the accumulated gradient: a second visit overwrites
            bucket['flat'][bucket['n'] + slot].fill_(1.0)                  # device-side fill: no host -> device copy inside the backward
            bucket['fired'][slot] = True
            self._launch_complete_prefix()
        return hook

    def finish(self):
        """Launch what the hooks could not (in bucket order), wait for every bucket and write the reduced gradients back.

        After finish(), p.grad of every parameter that fired on ANY rank since the previous finish() is the average over the ranks
        of what the hooks saw (a rank where it did not fire contributes zeros and has whatever p.grad it still held REPLACED --
        the replicas end identical); call it after EVERY backward, with zero_grad() between steps as usual.  Accumulating over
        several backward passes before one finish() is not supported (the hook raises as soon as a launched bucket is reached
        again)."""
        if not self.buckets:
            return
        world = dist.get_world_size(self.group)
        for bucket in self.buckets[self._next:]:
            self._launch(bucket)
        self._next = len(self.buckets)
        for bucket in self.buckets:
            bucket['work'].wait()
        # which parameters fired anywhere: ONE device -> host read for all buckets (a read per bucket stalls the host once per
        # bucket and step on every rank)
        touched_all = torch.cat([bucket['flat'][bucket['n']:].float() for bucket in self.buckets]).tolist()
        k = 0
        for bucket in self.buckets:
            touched = touched_all[k:k + len(bucket['params'])]
            k += len(bucket['params'])
            if self.average:
                bucket['flat'][:bucket['n']] /= world
            for i, (p, o) in enumerate(zip(bucket['params'], bucket['offsets'])):
                if touched[i] == 0:
                    continue                                                # no rank produced a gradient: leave p.grad as it is
                g = bucket['flat'][o:o + p.numel()].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
            bucket['fired'], bucket['work'] = [False] * len(bucket['params']), None
        self._next = 0

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
