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


def all_gather_pose_hypotheses(slot_R, slot_T, group=None):
    """slot_R [B_loc,S,A,3,3], slot_T [B_loc,S,A,3] -> the same for all clouds of the job
    ([B_loc*world, ...], rank-major).  Every rank must pass the same B_loc."""
    if not is_distributed():
        return slot_R, slot_T
    world = dist.get_world_size(group)
    b = slot_R.shape[0]
    flat = torch.cat([slot_R.reshape(b, -1), slot_T.reshape(b, -1)], dim=1).contiguous()
    out = torch.empty(world * b, flat.shape[1], dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    nR = slot_R[0].numel()
    R = out[:, :nR].reshape(world * b, *slot_R.shape[1:])
    Tt = out[:, nR:].reshape(world * b, *slot_T.shape[1:])
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
