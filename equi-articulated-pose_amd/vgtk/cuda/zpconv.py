"""vgtk.cuda.zpconv -- replaces the pybind module of vgtk/vgtk/cuda/zpconv_cuda.cpp."""
import torch

from .. import _hip


# upper bound of the backward's scratch buffer (the batch is processed in slices that fit).  The reference op needs no
# scratch at all, so the bound is modest (two 4096-point clouds at 64 channels) and a cloud that does not fit (or an allocation that fails) takes the
# scratch-free scatter kernel instead of raising
BWD_WORKSPACE_BYTES = 8 << 30


def _check(idx, w, x):
    _hip.check_input(idx, w, x)
    if idx.dtype != torch.int32:
        raise RuntimeError('zpconv: neighbour index must be int32')
    if w.dtype != x.dtype:
        raise RuntimeError('zpconv: weights and features must have the same dtype')


def inter_zpconv_forward(idx, w, feats):
    """(idx int32 [B,P,A,K,ANN], w T [B,P,A,K,ANN], feats T [B,C,Q,A]) -> T [B,C,K,P,A];
    zpconv_cuda.cpp:L41-56."""
    _check(idx, w, feats)
    b, np_, na, ks, ann = idx.shape
    c, nq = feats.shape[1], feats.shape[2]
    out = torch.empty(b, c, ks, np_, na, dtype=feats.dtype, device=feats.device)
    if feats.dtype == torch.float32:
        # scratch for the device-side index check + the per-point neighbour lists (csrc/zpconv_mfma.hip)
        nbytes = int(_hip.lib.eap_inter_zpconv_fwd_workspace(b, np_, ann))
        ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=feats.device)
        _hip.call('eap_inter_zpconv_fwd_ws_f32', out, b, np_, nq, na, ks, ann, c,
                  _hip._ptr(idx), _hip._ptr(w), _hip._ptr(feats), _hip._ptr(out), _hip._ptr(ws))
        return out
    _hip.call('eap_inter_zpconv_fwd_' + _hip.suffix(feats), out, b, np_, nq, na, ks, ann, c,
              _hip._ptr(idx), _hip._ptr(w), _hip._ptr(feats), _hip._ptr(out))
    return out


def inter_zpconv_backward(idx, w, grad, npoint):
    """(idx, w, grad T [B,C,K,P,A], int npoint) -> T [B,C,npoint,A]; zpconv_cuda.cpp:L58-75."""
    _check(idx, w, grad)
    b, np_, na, ks, ann = idx.shape
    c = grad.shape[1]
    out = torch.empty(b, c, int(npoint), na, dtype=grad.dtype, device=grad.device)
    if grad.dtype == torch.float32 and b > 0:
        # 1. scatter target on chip (csrc/zpconv_bwd_hot.hip): clouds whose lists reference few support rows, whole batch
        #    in one launch, nothing but the operands moves; the per-cloud status comes back in one small read
        todo = _backward_on_chip(idx, w, grad, out)
        if not todo:
            return out
        if len(todo) < b:
            # the clouds it left, in runs of consecutive clouds (slices, no copies of the 1.5 GB-per-cloud operands): one call per run
            run = [todo[0]]
            for i in todo[1:] + [None]:
                if i is not None and i == run[-1] + 1:
                    run.append(i)
                    continue
                b0, b1 = run[0], run[-1] + 1
                _backward_with_products(idx[b0:b1], w[b0:b1], grad[b0:b1], int(npoint), out[b0:b1])
                run = [i]
            return out
        return _backward_with_products(idx, w, grad, int(npoint), out)
    _hip.call('eap_inter_zpconv_bwd_' + _hip.suffix(grad), out, b, np_, int(npoint), na, ks, ann, c,
              _hip._ptr(idx), _hip._ptr(w), _hip._ptr(grad), _hip._ptr(out))
    return out


ON_CHIP_BACKWARD = True     # False: always the product pipeline of csrc/zpconv_bwd.hip (A/B runs, tests)


_HOT_VERDICTS = {}          # (storage pointer, version, shape) of an index tensor -> (weakref, clouds the on-chip kernel left)
_HOT_PENDING = []           # [event, pinned status copy, expected clouds, key]: verdict re-checks whose device result has not been looked at yet
VERIFY_REMEMBERED = True    # False: trust the remembered verdict without the asynchronous re-check (A/B timing only)


def _verdict_key(idx):
    return (idx.data_ptr(), idx._version, tuple(idx.shape))


def _check_pending_verdicts(block=False):
    """The remembered verdict of an index tensor is trusted WITHOUT a host read (no stall in a training loop), but every later call
    still gets the kernel's status back asynchronously and it is compared here, at the next call into this module (block=True:
    now).  A mismatch means the index CONTENTS changed under an unchanged version counter (`idx.data.copy_`, memory shared with
    numpy / dlpack, a raw-pointer kernel writing into the buffer): the clouds the on-chip kernel then rejected were not redone, so the
    previous result is wrong -- raise, and forget the verdict.  Contract: an index tensor handed to inter_zpconv_backward is either
    left alone or modified through torch in-place ops (which bump `_version`)."""
    for item in list(_HOT_PENDING):
        ev, host, known, key = item
        if not (block or ev.query()):
            continue
        if block:
            ev.synchronize()
        _HOT_PENDING.remove(item)
        seen = tuple(i for i, s_ in enumerate(host.tolist()) if s_ != 0)
        if seen != tuple(known):
            _HOT_VERDICTS.pop(key, None)
            raise RuntimeError('inter_zpconv_backward: the index tensor\'s contents changed without a version bump (clouds left by the on-chip kernel '
                               f'were {list(known)}, now {list(seen)}); the result of the previous call with it is invalid -- modify index tensors '
                               'through torch in-place ops or pass a new tensor')


def _backward_on_chip(idx, w, grad, out):
    """-> clouds still to do (all of them when the shape is not taken).  The verdict depends on the INDEX alone (which clouds
    name a row twice / reference too many rows), so it is remembered per index tensor: a training loop that keeps its
    neighbourhood pays the status read once, and a batch the kernel cannot take at all (padded lists: small radii, sparse or
    partial input) skips its prelude from the second call on.  Re-checked on the device every call: _check_pending_verdicts."""
    _check_pending_verdicts()
    b, np_, na, ks, ann = idx.shape
    c, nq = grad.shape[1], out.shape[2]
    nbytes = int(_hip.lib.eap_inter_zpconv_bwd_hot_workspace(b, np_, nq, na, ks, ann, c)) if ON_CHIP_BACKWARD else 0
    if nbytes <= 0 or any(t.data_ptr() % 16 for t in (idx, w, grad, out)):
        return list(range(b))
    key = (_verdict_key(idx), c, nq)
    hit = _HOT_VERDICTS.get(key)
    known = hit[1] if (hit is not None and hit[0]() is idx) else None
    if known is not None and len(known) == b:
        return list(known)                                   # nothing for the on-chip kernel here (every cloud goes to the product pipeline: correct whatever the index holds)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=grad.device)
    status = torch.empty(b, dtype=torch.int32, device=grad.device)
    _hip.call('eap_inter_zpconv_bwd_hot_f32', out, b, np_, nq, na, ks, ann, c, _hip._ptr(idx), _hip._ptr(w), _hip._ptr(grad),
              _hip._ptr(out), _hip._ptr(ws), _hip._ptr(status))
    if known is not None:
        # same index, same shapes: the same clouds as last time, no host read now -- the status travels to pinned memory behind the
        # kernel and is compared at the next call
        if VERIFY_REMEMBERED and len(_HOT_PENDING) < 8:
            host = torch.empty(b, dtype=torch.int32, pin_memory=True)
            host.copy_(status, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            _HOT_PENDING.append([ev, host, tuple(known), key])
        return list(known)
    todo = [i for i, s in enumerate(status.tolist()) if s != 0]
    if len(_HOT_VERDICTS) > 64:
        _HOT_VERDICTS.clear()
    import weakref
    _HOT_VERDICTS[key] = (weakref.ref(idx), tuple(todo))
    return todo


def _backward_with_products(idx, w, grad, npoint, out=None):
    """csrc/zpconv_bwd.hip: products in forward order + sorted sums (any number of referenced rows), scratch-free scatter
    kernel when no scratch can be had."""
    b, np_, na, ks, ann = idx.shape
    c = grad.shape[1]
    if out is None:
        out = torch.empty(b, c, npoint, na, dtype=grad.dtype, device=grad.device)
    # atomics-free path (csrc/zpconv_bwd.hip): scratch for the per-(point, neighbour) products, a few clouds at a time
    per_cloud = int(_hip.lib.eap_inter_zpconv_bwd_workspace(1, np_, int(npoint), na, ann, c))
    step = min(b, BWD_WORKSPACE_BYTES // max(per_cloud, 1))
    ws = None
    if step >= 1:
        try:
            ws = torch.empty((int(_hip.lib.eap_inter_zpconv_bwd_workspace(step, np_, int(npoint), na, ann, c)) + 3) // 4,
                             dtype=torch.int32, device=grad.device)
        except torch.cuda.OutOfMemoryError:
            ws = None
    if ws is None:
        _hip.call('eap_inter_zpconv_bwd_f32', out, b, np_, int(npoint), na, ks, ann, c,
                  _hip._ptr(idx), _hip._ptr(w), _hip._ptr(grad), _hip._ptr(out))
        return out
    for b0 in range(0, b, step):
        nb = min(step, b - b0)
        _hip.call('eap_inter_zpconv_bwd_ws_f32', out, nb, np_, int(npoint), na, ks, ann, c, _hip._ptr(idx[b0:b0 + nb]),
                  _hip._ptr(w[b0:b0 + nb]), _hip._ptr(grad[b0:b0 + nb]), _hip._ptr(out[b0:b0 + nb]), _hip._ptr(ws))
    return out


def intra_zpconv_forward(idx, w, feats):
    """(idx int32 [A_out,ANN], w T [A_out,K,ANN], feats T [B,C,P,A_in]) -> T [B,C,K,P,A_out];
    zpconv_cuda.cpp:L77-92."""
    _check(idx, w, feats)
    na_out, ann = idx.shape
    ks = w.shape[1]
    b, c, np_, na_in = feats.shape
    out = torch.empty(b, c, ks, np_, na_out, dtype=feats.dtype, device=feats.device)
    _hip.call('eap_intra_zpconv_fwd_' + _hip.suffix(feats), out, b, np_, na_in, na_out, ks, ann, c,
              _hip._ptr(idx), _hip._ptr(w), _hip._ptr(feats), _hip._ptr(out))
    return out


def intra_zpconv_backward(idx, w, grad, anchor_in):
    """(idx, w, grad T [B,C,K,P,A_out], int anchor_in) -> T [B,C,P,anchor_in]; zpconv_cuda.cpp:L94-110."""
    _check(idx, w, grad)
    na_out, ann = idx.shape
    ks = w.shape[1]
    b, c, _, np_, _ = grad.shape
    out = torch.empty(b, c, np_, int(anchor_in), dtype=grad.dtype, device=grad.device)
    _hip.call('eap_intra_zpconv_bwd_' + _hip.suffix(grad), out, b, np_, int(anchor_in), na_out, ks, ann, c,
              _hip._ptr(idx), _hip._ptr(w), _hip._ptr(grad), _hip._ptr(out))
    return out
