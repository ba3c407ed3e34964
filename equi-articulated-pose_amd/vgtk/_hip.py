"""ctypes binding of libeap_hip.so -- the C ABI declared in include/eap_hip.h.

PyTorch is used for device memory and streams only: every call takes torch tensors, checks
them the way the reference's C++ wrappers do (device + contiguous, e.g.
vgtk/vgtk/cuda/zpconv_cuda.cpp:L37-39 CHECK_INPUT -> RuntimeError), and launches on the
CURRENT torch stream of the tensor's device (the reference launches on the legacy default
stream with no device guard, zpconv_cuda_kernel.cu:L219).

There is NO CPU or eager fallback: if the shared library is missing this module raises at
import, and non-device tensors are rejected.
"""
import ctypes
import os

import torch

_PKG_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
LIB_PATH = os.path.join(_PKG_ROOT, 'libeap_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f'{LIB_PATH} not found: build the HIP extension first '
        '(python -c "import __graft_entry__ as g; g.build()" or make -C equi-articulated-pose_amd/csrc). '
        'This package has no CPU fallback.')

lib = ctypes.CDLL(LIB_PATH)
lib.eap_last_error.restype = ctypes.c_char_p
lib.eap_last_kernel.restype = ctypes.c_char_p
lib.eap_gemm_f32_reduce_workspace.restype = ctypes.c_int64
lib.eap_so3_inter_group_bwd_workspace.restype = ctypes.c_int64
lib.eap_inter_zpconv_fwd_workspace.restype = ctypes.c_int64
lib.eap_inter_zpconv_bwd_workspace.restype = ctypes.c_int64
lib.eap_inter_zpconv_bwd_hot_workspace.restype = ctypes.c_int64
lib.eap_bn_act_segments.restype = ctypes.c_int

_I64 = ctypes.c_int64
_F32 = ctypes.c_float


def _ptr(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def check_input(*tensors):
    """CHECK_INPUT of the reference C++ wrappers: device tensor + contiguous."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('tensor must be a CUDA(HIP) tensor')
        if not t.is_contiguous():
            raise RuntimeError('tensor must be contiguous')


def stream_of(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


# Optional per-launch timing (bench.py): when KERNEL_TIMES is a list, every launch is bracketed
# by two HIP events recorded on the launch stream and (name, tag, start, stop) is appended.
KERNEL_TIMES = None


def call(name, ref_tensor, *args, tag=None):
    """Invoke lib.<name>(*args, stream) under the device of `ref_tensor`; raise on error."""
    fn = getattr(lib, name)
    with torch.cuda.device(ref_tensor.device):
        if KERNEL_TIMES is not None:
            stream = torch.cuda.current_stream(ref_tensor.device)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            lib.eap_last_kernel()        # (clear)
            e0.record(stream)
            rc = fn(*args, stream_of(ref_tensor))
            e1.record(stream)
            if tag is not None:          # the HIP kernel (with its template arguments) this entry just launched
                tag = dict(tag, kernel=lib.eap_last_kernel().decode() or name)
            KERNEL_TIMES.append((name, tag, e0, e1))
        else:
            rc = fn(*args, stream_of(ref_tensor))
    if rc != 0:
        raise RuntimeError(f'{name} failed (hip error {rc}): {lib.eap_last_error().decode()}')


def suffix(t):
    if t.dtype == torch.float32:
        return 'f32'
    if t.dtype == torch.float64:
        return 'f64'
    raise RuntimeError(f'unsupported dtype {t.dtype} (float32 / float64 only)')


# ---- raw launchers used by the operator layer ---------------------------------------------------

# csrc/gemm_dma_f32.hip (DMA-fed three-stage ring) serves every GEMM whose operands it can take (K a multiple of 16,
# 16-byte aligned rows); csrc/gemm_f32.hip the rest (first layer K = 24, blocked intermediates, odd shapes).
# EAP_DMA_GEMM=0 forces the older kernel everywhere (A/B runs).
USE_DMA_GEMM = os.environ.get('EAP_DMA_GEMM', '1') != '0'
lib.eap_gemm_dma_f32_reduce_workspace.restype = ctypes.c_int64
lib.eap_gemm_skinny_reduce_workspace.restype = ctypes.c_int64
lib.eap_gemm_bf16x3_reduce_workspace.restype = ctypes.c_int64


def _dma_ok(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB):
    return USE_DMA_GEMM and bool(lib.eap_gemm_dma_f32_supported(int(transA), int(transB), M, N, K, _ptr(A), _I64(lda), _I64(strideA),
                                                              _ptr(B), _I64(ldb), _I64(strideB)))


# Contractions whose A operand is shared by the batch (the inter conv's forward contraction on the transposed
# intermediate, the pointwise contraction so3_contract, the implicit intra conv) on the 16-bit matrix cores with split
# operands, fp32 in, fp32 accumulate (csrc/gemm_bf16x3.hip).  False: the fp32-MFMA kernels everywhere.
SPLIT_BF16_CONTRACTION = True
# Planes per operand on those kernels: 3 = three bf16 planes, six products (exact splits); 2 = two fp16 planes after a
# power-of-two scale per tensor, three products (include/eap_hip.h, eap_gemm_f16x2_f32: representation error <= 2^-23 per
# element, bounded against fp64 by the fp32-MFMA kernel's error in tests/test_gpu_split_planes.py).
# Accuracy bar of the default (2): every product carries a representation error <= 2^-21 |a||b| (the l l' term is dropped: NOT
# bit-for-bit an fp32 product), and an operand row scaled from a BOUND on its magnitude that is L times too large adds an absolute
# 2^-39 L max|row| per element; against float64 the result stays within 1.25 x (max) / 1.1 x (rms) of the fp32-MFMA kernel's
# error on the same operands, with bounds up to 2^20 too large (README.md "Numerics", tests/test_gpu_split_planes.py).  3 planes
# are exact splits (dropped terms <= 2^-23 |a||b|) and need no magnitudes.
def _split_planes_from_env():
    v = os.environ.get('EAP_SPLIT_PLANES', '2')
    if v not in ('2', '3'):
        raise ImportError(f'EAP_SPLIT_PLANES={v!r}: 2 (two fp16 planes, three products) or 3 (three bf16 planes, six products)')
    return int(v)


SPLIT_PLANES = _split_planes_from_env()      # (the environment switch is for A/B runs)


SPLIT_PLANES_SCAN_ROWS = 384     # see _planes2 (tests set it to 0 to reach the two-plane kernel at every shape)


def _pieces_ok(t, *dims):
    return not any(int(d) & 3 for d in dims) and t.data_ptr() % 16 == 0


def absmax_rows(t, batch, rows, cols, ld, stride):
    """Largest magnitude of every row of a [batch][rows][cols] view (row pitch ld, item stride) -> int32 [batch, rows] holding
    float bit patterns, or None when the view is not made of whole, aligned 16-byte pieces."""
    if not _pieces_ok(t, cols, ld, stride):
        return None
    out = torch.empty(batch, rows, dtype=torch.int32, device=t.device)
    call('eap_absmax_rows_f32', t, _ptr(t), batch, rows, cols, _I64(ld), _I64(stride), _ptr(out))
    return out


def absmax_colgroups(t, batch, rows, cols, ld, stride, grp):
    """Largest magnitude over the rows and over each group of grp consecutive columns -> int32 [batch, cols // grp], or None."""
    if not _pieces_ok(t, cols, ld, stride, grp) or cols % grp:
        return None
    out = torch.empty(batch, cols // grp, dtype=torch.int32, device=t.device)
    call('eap_absmax_colgroups_f32', t, _ptr(t), batch, rows, cols, _I64(ld), _I64(stride), grp, _ptr(out))
    return out


def so3_grouped_bound(feats, idx):
    """A bound per point on the inter conv's grouped tensor, from the features' per-point maxima (include/eap_hip.h,
    eap_so3_grouped_bound_f32): feats [b,c,n,na], idx int32 [b,p,nn] -> int32 [b, p] (float bit patterns), or None."""
    b, c, n, na = feats.shape
    pm = absmax_colgroups(feats, b, c, n * na, n * na, c * n * na, na) if na % 4 == 0 else None
    if pm is None:
        return None
    out = torch.empty(b, idx.shape[1], dtype=torch.int32, device=feats.device)
    call('eap_so3_grouped_bound_f32', feats, b, idx.shape[1], idx.shape[2], n, _ptr(pm), _ptr(idx), _ptr(out))
    return out


def _planes2(transB, M, N, K, A, lda, batch, B, ldb, strideB, b_bound):
    """The magnitude words of eap_gemm_f16x2_f32 -> (abs_a, abs_b, grp_b, mult_b), or None: take the three-plane kernel.
    b_bound = (words [batch, N // grp], grp, factor) when the caller knows a bound on B's columns."""
    if SPLIT_PLANES != 2:
        return None
    if b_bound is None and M < SPLIT_PLANES_SCAN_ROWS:
        # without a bound the kernel needs a pass over B first: 4 bytes per element at ~5 TB/s against the M / 2 fp32-equivalent
        # products per element it saves at ~245 TFLOP/s -- pays from M = 384 rows upwards
        return None
    abs_a = absmax_rows(A, 1, M, K, lda, 0)
    if abs_a is None:
        return None
    if b_bound is not None:
        return abs_a, b_bound[0], int(b_bound[1]), float(b_bound[2])
    abs_b = absmax_rows(B, batch, N, K, ldb, strideB) if transB else absmax_colgroups(B, batch, K, N, ldb, strideB, 4)
    return None if abs_b is None else (abs_a, abs_b, 1 if transB else 4, 1.0)


def gemm(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, batch, b_blocked=False, b_bound=None):
    """b_blocked: B is stored blocked by 4 (include/eap_hip.h); `ldb` is then ignored.  b_bound = (words int32 [batch, N // grp] holding
    non-negative float bit patterns, grp = consecutive columns per word, factor): words * factor bounds the magnitudes of B's columns,
    see _planes2 (saves the two-plane kernel its pass over B)."""
    tag = {'flops': 2.0 * M * N * K * batch, 'shape': ('gemm', int(transA), int(transB), M, N, K, batch)}
    if SPLIT_BF16_CONTRACTION and not b_blocked and not transA and transB and (strideA == 0 or batch == 1) and \
            lib.eap_gemm_bf16x3_f32_supported(M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(strideB)):
        two = _planes2(1, M, N, K, A, lda, batch, B, ldb, strideB, b_bound)
        if two is not None:
            call('eap_gemm_f16x2_f32', C, 1, M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(strideB), _ptr(C), _I64(ldc), _I64(strideC),
                 batch, _ptr(two[0]), _ptr(two[1]), two[2], _F32(two[3]), None, None, _F32(0.0), None, _I64(0), tag=tag)
            return
        call('eap_gemm_bf16x3_f32', C, M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(strideB), _ptr(C), _I64(ldc), _I64(strideC),
             batch, tag=tag)
        return
    if SPLIT_BF16_CONTRACTION and not b_blocked and not transA and not transB and (strideA == 0 or batch == 1) and \
            lib.eap_gemm_bf16x3_nn_f32_supported(M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(strideB)):
        two = _planes2(0, M, N, K, A, lda, batch, B, ldb, strideB, b_bound)
        if two is not None:
            call('eap_gemm_f16x2_f32', C, 0, M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(strideB), _ptr(C), _I64(ldc), _I64(strideC),
                 batch, _ptr(two[0]), _ptr(two[1]), two[2], _F32(two[3]), None, None, _F32(0.0), None, _I64(0), tag=tag)
            return
        call('eap_gemm_bf16x3_nn_f32', C, M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(strideB), _ptr(C), _I64(ldc), _I64(strideC),
             batch, tag=tag)
        return
    if not b_blocked and _dma_ok(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB):
        call('eap_gemm_dma_f32', C, int(transA), int(transB), M, N, K, _ptr(A), _I64(lda), _I64(strideA), _ptr(B), _I64(ldb),
             _I64(strideB), _ptr(C), _I64(ldc), _I64(strideC), batch, tag=tag)
        return
    call('eap_gemm_f32_xb' if b_blocked else 'eap_gemm_f32', C, int(transA), int(transB), M, N, K, _ptr(A), _I64(lda), _I64(strideA),
         _ptr(B), _I64((N if transB else K) if b_blocked else ldb), _I64(strideB), _ptr(C), _I64(ldc), _I64(strideC), batch, tag=tag)


def gemm_epilogue(transB, M, N, K, A, lda, B, ldb, strideB, C, ldc, strideC, batch, scale, shift, slope, residual=None, b_bound=None):
    """C_z = leaky_relu(scale[row] * (A B_z) + shift[row], slope) (+ residual_z, laid out like C) on the split kernel
    (eap_gemm_bf16x3_ep_f32): an inference-mode BatchNorm + activation folded into the contraction.  -> False when the
    operands do not qualify for that kernel (nothing launched: the caller runs product and epilogue separately)."""
    if not SPLIT_BF16_CONTRACTION:
        return False
    ok = (lib.eap_gemm_bf16x3_f32_supported if transB else lib.eap_gemm_bf16x3_nn_f32_supported)(
        M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(strideB))
    if not ok:
        return False
    tag = {'flops': 2.0 * M * N * K * batch, 'shape': ('gemm_epilogue', 0, int(transB), M, N, K, batch)}
    two = _planes2(int(transB), M, N, K, A, lda, batch, B, ldb, strideB, b_bound)
    if two is not None:
        call('eap_gemm_f16x2_f32', C, int(transB), M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(strideB), _ptr(C), _I64(ldc),
             _I64(strideC), batch, _ptr(two[0]), _ptr(two[1]), two[2], _F32(two[3]), _ptr(scale), _ptr(shift), _F32(slope), _ptr(residual),
             _I64(strideC), tag=tag)
        return True
    call('eap_gemm_bf16x3_ep_f32', C, int(transB), M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(strideB), _ptr(C), _I64(ldc),
         _I64(strideC), batch, _ptr(scale), _ptr(shift), _F32(slope), _ptr(residual), _I64(strideC), tag=tag)
    return True


def gemm_nn_takes_split(M, N, K, A, lda, B, ldb):
    """Would gemm(0, 0, ...) with these operands (one item) run on a split-operand kernel?"""
    return bool(SPLIT_BF16_CONTRACTION and lib.eap_gemm_bf16x3_nn_f32_supported(M, N, K, _ptr(A), _I64(lda), _ptr(B), _I64(ldb), _I64(0)))


def gemm_reduce_takes_split(M, N, K, A, lda, strideA, B, ldb, strideB, ldc):
    """Would gemm_reduce(0, 1, ...) with these operands run on the split-bf16 kernel?"""
    return bool(SPLIT_BF16_CONTRACTION and lib.eap_gemm_bf16x3_reduce_f32_supported(M, N, K, _ptr(A), _I64(lda), _I64(strideA), _ptr(B), _I64(ldb),
                                                                                  _I64(strideB), _I64(ldc)))


def gemm_reduce(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, batch, b_blocked=False):
    tag = {'flops': 2.0 * M * N * K * batch, 'shape': ('gemm_reduce', int(transA), int(transB), M, N, K, batch)}
    if SPLIT_BF16_CONTRACTION and not b_blocked and not transA and transB and \
            lib.eap_gemm_bf16x3_reduce_f32_supported(M, N, K, _ptr(A), _I64(lda), _I64(strideA), _ptr(B), _I64(ldb), _I64(strideB), _I64(ldc)):
        ws = torch.empty(int(lib.eap_gemm_bf16x3_reduce_workspace(M, N, K, batch)), dtype=torch.float32, device=C.device)
        call('eap_gemm_bf16x3_reduce_f32', C, M, N, K, _ptr(A), _I64(lda), _I64(strideA), _ptr(B), _I64(ldb), _I64(strideB), _ptr(C),
             _I64(ldc), batch, _ptr(ws), tag=tag)
        return
    if not b_blocked and not transA and transB and lib.eap_gemm_skinny_reduce_f32_supported(M, N, K, _ptr(A), _I64(lda), _I64(strideA), _ptr(B),
                                                                                         _I64(ldb), _I64(strideB)):
        # a small output over a long contraction (the first layer's weight gradient): streaming reduction, csrc/gemm_skinny.hip
        ws = torch.empty(max(int(lib.eap_gemm_skinny_reduce_workspace(M, N, K, batch)), 1), dtype=torch.float32, device=C.device)
        call('eap_gemm_skinny_reduce_f32', C, M, N, K, _ptr(A), _I64(lda), _I64(strideA), _ptr(B), _I64(ldb), _I64(strideB), _ptr(C), _I64(ldc),
             batch, _ptr(ws), tag=tag)
        return
    if not b_blocked and _dma_ok(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB):
        ws = torch.empty(max(int(lib.eap_gemm_dma_f32_reduce_workspace(M, N, K, batch)), 1), dtype=torch.float32, device=C.device)
        call('eap_gemm_dma_f32_reduce', C, int(transA), int(transB), M, N, K, _ptr(A), _I64(lda), _I64(strideA), _ptr(B), _I64(ldb),
             _I64(strideB), _ptr(C), _I64(ldc), batch, _ptr(ws), tag=tag)
        return
    n_ws = lib.eap_gemm_f32_reduce_workspace(M, N, K, batch)
    ws = torch.empty(max(int(n_ws), 1), dtype=torch.float32, device=C.device)
    call('eap_gemm_f32_reduce_xb' if b_blocked else 'eap_gemm_f32_reduce', C, int(transA), int(transB), M, N, K, _ptr(A), _I64(lda), _I64(strideA),
         _ptr(B), _I64((N if transB else K) if b_blocked else ldb), _I64(strideB), _ptr(C), _I64(ldc), batch, _ptr(ws), tag=tag)


def so3_prep(q_xyz, s_xyz, idx, q_pose, s_pose, anchors, identity_anchor):
    b, _, p = q_xyz.shape
    n = s_xyz.shape[2]
    nn = idx.shape[2]
    gx = torch.empty(b, p, nn, 4, dtype=torch.float32, device=q_xyz.device)
    nonident = torch.empty(b, dtype=torch.int32, device=q_xyz.device)
    call('eap_so3_prep_f32', gx, b, p, n, nn, anchors.shape[0], _ptr(q_xyz), _ptr(s_xyz), _ptr(idx),
         _ptr(q_pose), _ptr(s_pose), _ptr(anchors), int(identity_anchor), _ptr(gx), _ptr(nonident))
    return gx, nonident


def so3_inter_weights(gx, rk, sigma):
    b, p, nn, _ = gx.shape
    na, ks, _ = rk.shape
    w = torch.empty(b, p, na, ks, nn, dtype=torch.float32, device=gx.device)
    call('eap_so3_inter_weights_f32', w, b, p, nn, na, ks, _F32(sigma), _ptr(gx), _ptr(rk), _ptr(w))
    return w


def so3_anchor_perm(gx, mult):
    b, p, nn, _ = gx.shape
    na = mult.shape[0]
    perm = torch.empty(b, p, nn, na, dtype=torch.int64, device=gx.device)
    call('eap_so3_anchor_perm', gx, b, p, nn, na, _ptr(gx), _ptr(mult), _ptr(perm))
    return perm


def so3_inter_group_fwd_can_block(c, n, na, ks, has_mult, has_flag):
    return bool(lib.eap_so3_inter_group_fwd_can_block(c, n, na, ks, int(has_mult), int(has_flag)))


_TP_COLUMNS = {}


def so3_group_fwd_tp_takes(c, na, ks):
    return bool(lib.eap_so3_group_fwd_tp_takes(int(c), int(na), int(ks)))


def so3_group_fwd_tp_columns(c, ks, device):
    """int64 [c*ks] on `device`: position j of a row of the store-order transposed intermediate holds column perm[j] of the
    plain one (include/eap_hip.h: eap_so3_group_fwd_tp_columns)."""
    key = (int(c), int(ks), str(device))
    hit = _TP_COLUMNS.get(key)
    if hit is None:
        import numpy as np
        perm = np.empty(c * ks, np.int32)
        if lib.eap_so3_group_fwd_tp_columns(int(c), int(ks), perm.ctypes.data_as(ctypes.c_void_p)) != 0:
            raise RuntimeError(lib.eap_last_error().decode())
        hit = _TP_COLUMNS[key] = torch.from_numpy(perm.astype(np.int64)).to(device)
    return hit


def so3_inter_group_fwd(feats, idx, gx, rk, mult, sigma, nonident=None, blocked=False, coset=None, store_order=False):
    """-> X [b,c,ks,p,na]; with `blocked` the same numbers as [b,p,na/4,c,ks,4] (returned with the
    nominal shape: only eap_gemm_f32_xb may read it).  coset = the coset tables of `mult`
    (vgtk.so3conv.functional._coset_tables): permuted clouds of the transposed layout take the two-tile kernel.
    store_order: the transposed intermediate with its columns in the kernel's store order (so3_group_fwd_tp_columns) -- only a
    contraction with W[:, columns] may read it."""
    b, c, n, na = feats.shape
    p, nn = idx.shape[1], idx.shape[2]
    ks = rk.shape[1]
    out = torch.empty(b, c, ks, p, na, dtype=torch.float32, device=feats.device)
    if (coset is not None and int(blocked) == 2 and mult is not None and nonident is not None and nn > 0
            and so3_group_perm_lists2_takes(c, na, ks, n)):
        # clouds with anchor permutations on the two-tile kernel (csrc/so3_inter_lists2.hip, PERM): their features with a coset-major
        # anchor axis, the per-entry words of (idx, gx) prepared once; clouds whose flag is 0 cost three empty launches
        feats_c = torch.empty_like(feats)
        call('eap_anchor_reorder_clouds_f32', feats, b, _I64(c * n), na, _ptr(feats), _ptr(coset[0]), _ptr(nonident), _ptr(feats_c))
        ent_pc, ent_gx2 = so3_perm_entries(idx.view(b, p * nn), gx.view(b, p * nn, 4), coset[1], None, 0, na, n, nonident)
        call('eap_so3_inter_group_fwd_perm2_t_f32', out, b, c, p, n, nn, na, ks, _F32(sigma), _ptr(feats), _ptr(feats_c), _ptr(idx), _ptr(gx),
             _ptr(ent_pc), _ptr(ent_gx2), _ptr(rk), _ptr(coset[0]), _ptr(nonident), int(bool(store_order)), _ptr(out),
             tag={'flops': 2.0 * b * c * ks * p * nn * na, 'shape': ('group_fwd_perm2', b, c, p, nn, na, ks)})
        return out
    if store_order:
        if mult is not None or int(blocked) != 2:
            raise RuntimeError('store-order columns: transposed layout, clouds without permutation (or the coset tables for the flagged ones)')
        call('eap_so3_inter_group_fwd_tp_f32', out, b, c, p, n, nn, na, ks, _F32(sigma), _ptr(feats), _ptr(idx), _ptr(gx), _ptr(rk), _ptr(out),
             tag={'flops': 2.0 * b * c * ks * p * nn * na, 'shape': ('group_fwd_tp', b, c, p, nn, na, ks)})
        return out
    call({0: 'eap_so3_inter_group_fwd_f32', 1: 'eap_so3_inter_group_fwd_xb_f32', 2: 'eap_so3_inter_group_fwd_t_f32'}[int(blocked)], out, b, c, p, n, nn, na, ks, _F32(sigma), _ptr(feats), _ptr(idx),
         _ptr(gx), _ptr(rk), _ptr(mult), _ptr(nonident), _ptr(out),
         tag={'flops': 2.0 * b * c * ks * p * nn * na, 'shape': ('group_fwd', b, c, p, nn, na, ks)})
    return out


FORCE_ATOMIC_BWD = False


def so3_inter_group_bwd(gout, idx, gx, rk, mult, sigma, n, identity_anchor=0):
    b, c, ks, p, na = gout.shape
    nn = idx.shape[2]
    gfeats = torch.empty(b, c, n, na, dtype=torch.float32, device=gout.device)
    if ks <= 24 and na % 4 == 0 and not FORCE_ATOMIC_BWD:
        ws = torch.empty(int(lib.eap_so3_inter_group_bwd_workspace(b, c, p, n, na)), dtype=torch.float32,
                         device=gout.device)
        call('eap_so3_inter_group_bwd_slab_f32', gfeats, b, c, p, n, nn, na, ks, _F32(sigma), _ptr(gout), _ptr(idx),
             _ptr(gx), _ptr(rk), _ptr(mult), int(identity_anchor), _ptr(gfeats), _ptr(ws))
        return gfeats
    call('eap_so3_inter_group_bwd_f32', gfeats, b, c, p, n, nn, na, ks, _F32(sigma), _ptr(gout), _ptr(idx),
         _ptr(gx), _ptr(rk), _ptr(mult), _ptr(gfeats))
    return gfeats


def so3_intra_group_fwd(feats, intra_idx):
    b, c, p, na = feats.shape
    t = intra_idx.shape[1]
    out = torch.empty(b, c, t, p, na, dtype=torch.float32, device=feats.device)
    call('eap_so3_intra_group_fwd_f32', out, b, c, p, na, t, _ptr(feats), _ptr(intra_idx), _ptr(out))
    return out


def so3_intra_group_bwd(gout, intra_idx):
    b, c, t, p, na = gout.shape
    gfeats = torch.empty(b, c, p, na, dtype=torch.float32, device=gout.device)
    call('eap_so3_intra_group_bwd_f32', gfeats, b, c, p, na, t, _ptr(gout), _ptr(intra_idx), _ptr(gfeats))
    return gfeats


def so3_inter_group_inv(gy, rows, off, cnt, ent_p, ent_gx, rk, multinv, sigma, nn, identity_anchor=0, anchors=None, coset=None):
    """gy [b,o,p,na] + inverse neighbour lists -> z [b,o,ks,rcap,na] (see include/eap_hip.h).  coset = (order, code) of
    `multinv` (vgtk.so3conv.functional._coset_tables): the permuted clouds' operand is kept coset-major in LDS."""
    b, o, p, na = gy.shape
    rcap = rows.shape[1]
    ks = rk.shape[1]
    z = torch.empty(b, o, ks, rcap, na, dtype=torch.float32, device=gy.device)
    if multinv is not None and coset is not None:
        if na % 8 == 4:          # the kernel's DMA loader copies rows as they are: re-order the anchor axis once (csrc/so3_inter_inv.hip)
            gyc = torch.empty_like(gy)
            call('eap_anchor_reorder_f32', gy, _I64(b * o * p), na, _ptr(gy), _ptr(coset[0]), _ptr(gyc))
            gy = gyc
        call('eap_so3_inter_group_inv_coset_f32', z, b, o, p, nn, na, ks, rcap, _F32(sigma), _ptr(gy), _ptr(rows), _ptr(off),
             _ptr(cnt), _ptr(ent_p), _ptr(ent_gx), _ptr(rk), _ptr(multinv), _ptr(anchors), int(identity_anchor), _ptr(coset[0]), _ptr(coset[1]),
             _ptr(z), tag={'flops': 2.0 * b * o * ks * p * nn * na, 'shape': ('group_inv_coset', b, o, p, nn, na, ks, rcap)})
        return z
    call('eap_so3_inter_group_inv_f32', z, b, o, p, nn, na, ks, rcap, _F32(sigma), _ptr(gy), _ptr(rows), _ptr(off),
         _ptr(cnt), _ptr(ent_p), _ptr(ent_gx), _ptr(rk), _ptr(multinv), _ptr(anchors if multinv is not None else None), int(identity_anchor), _ptr(z),
         tag={'flops': 2.0 * b * o * ks * p * nn * na, 'shape': ('group_inv', b, o, p, nn, na, ks, rcap)})
    return z


def anchor_reorder(t, order):
    """t [..., na] -> the same with its last axis re-ordered: out[..., i] = t[..., order[i]] (order uint8 [>= na])."""
    t = t.contiguous()
    na = t.shape[-1]
    out = torch.empty_like(t)
    call('eap_anchor_reorder_f32', t, _I64(t.numel() // na), na, _ptr(t), _ptr(order), _ptr(out))
    return out


def so3_group_perm_lists2_takes(channels, na, ks, n_support):
    return bool(lib.eap_so3_group_perm_lists2_takes(int(channels), int(na), int(ks), int(n_support)))


def so3_perm_entries(ent_p, ent_gx, code, anchors, identity_anchor, na, n_support, nonident=None):
    """per-entry words of the permuted two-tile kernel (include/eap_hip.h): -> ent_pc int32 [b,per,4,4], ent_gx2 [b,per,4]."""
    b, per = ent_p.shape[0], ent_p[0].numel()
    ent_pc = torch.empty(b, per, 4, 4, dtype=torch.int32, device=ent_p.device)
    ent_gx2 = torch.empty(b, per, 4, dtype=torch.float32, device=ent_p.device)
    anchors = anchors.contiguous() if anchors is not None else None
    call('eap_so3_perm_entries_f32', ent_p, b, per, int(na), int(n_support), _ptr(ent_p), _ptr(ent_gx), _ptr(code),
         _ptr(anchors), int(identity_anchor), _ptr(nonident), _ptr(ent_pc), _ptr(ent_gx2))
    return ent_pc, ent_gx2


def so3_inter_group_inv_perm2(gy_c, rows, off, cnt, ent_pc, ent_gx2, rk, order, sigma, nn):
    """gy_c [b,o,p,na] (anchor axis coset-major) -> z [b,o,ks,rcap,na] (coset-major), see include/eap_hip.h."""
    b, o, p, na = gy_c.shape
    rcap, ks = rows.shape[1], rk.shape[1]
    z = torch.empty(b, o, ks, rcap, na, dtype=torch.float32, device=gy_c.device)
    call('eap_so3_inter_group_inv_perm2_f32', z, b, o, p, nn, na, ks, rcap, _F32(sigma), _ptr(gy_c), _ptr(rows), _ptr(off), _ptr(cnt),
         _ptr(ent_pc), _ptr(ent_gx2), _ptr(rk), _ptr(order), _ptr(z),
         tag={'flops': 2.0 * b * o * ks * p * nn * na, 'shape': ('group_inv_perm2', b, o, p, nn, na, ks, rcap)})
    return z


def inv_lists_rows(idx, n):
    """idx int32 [b,p,nn] -> rows, off, cnt [b,n] and n_rows [b] (csrc/inv_lists.hip); no host sync."""
    b, p, nn = idx.shape
    dev = idx.device
    counts = torch.empty(b, n, dtype=torch.int32, device=dev)
    rows, off, cnt = torch.empty_like(counts), torch.empty_like(counts), torch.empty_like(counts)
    n_rows = torch.empty(b, dtype=torch.int32, device=dev)
    call('eap_inv_lists_rows', idx, b, p, n, nn, _ptr(idx), _ptr(counts), _ptr(rows), _ptr(off), _ptr(cnt), _ptr(n_rows))
    return rows, off, cnt, n_rows


def inv_lists_fill(idx, gx, rows, off, rcap):
    """-> ent_p int32 [b,p*nn], ent_gx [b,p*nn,4] for the first rcap rows of every cloud."""
    b, p, nn = idx.shape
    n = rows.shape[1]
    ent_p = torch.empty(b, p * nn, dtype=torch.int32, device=idx.device)
    ent_gx = torch.empty(b, p * nn, 4, dtype=torch.float32, device=idx.device)
    call('eap_inv_lists_fill', idx, b, p, n, nn, int(rcap), _ptr(idx), _ptr(gx), _ptr(rows), _ptr(off), _ptr(ent_p), _ptr(ent_gx))
    return ent_p, ent_gx


def rows_gather(src, rows, rcap):
    """src [b,c,n,na], rows int32 [b,>=rcap] -> [b,c,rcap,na] (zeros for rows < 0)."""
    b, c, n, na = src.shape
    dst = torch.empty(b, c, rcap, na, dtype=torch.float32, device=src.device)
    call('eap_rows_gather_f32', src, b, c, n, na, int(rcap), rows.stride(0), _ptr(rows), _ptr(src), _ptr(dst))
    return dst


def rows_scatter(src, rows, n):
    """src [b,c,rcap,na] -> [b,c,n,na], zero except the rows named by `rows`."""
    b, c, rcap, na = src.shape
    dst = torch.empty(b, c, n, na, dtype=torch.float32, device=src.device)
    call('eap_rows_scatter_f32', src, b, c, n, na, int(rcap), rows.stride(0), _ptr(rows), _ptr(src), _ptr(dst))
    return dst


# ---- block-layer epilogue (csrc/bn_act.hip) -------------------------------------------------------

# ---- the inter conv re-associated over its referenced rows as a dense product (csrc/so3_dense.hip) ----------------------
lib.eap_so3_dense_mask_words.restype = ctypes.c_int64


def so3_dense_supported(p, na, ks, rp, o):
    return bool(lib.eap_so3_dense_supported(int(p), int(na), int(ks), int(rp), int(o)))


DENSE_MAX_ROWS = 512        # csrc/so3_dense.hip MEMB_WORDS * 32


def so3_dense_member(idx, rows, n_rows, n):
    """Which of a cloud's referenced rows (rows [b, >= 512], n_rows [b]; eap_inv_lists_rows) every neighbour list names:
    -> memb int32 [b,p,16] (bit masks), flags int32 [b] (non-zero: the cloud cannot take the dense product -- a list names a row
    twice, or more than 512 rows are referenced).  No host value is needed: runs before the row count is known."""
    b, p, nn = idx.shape
    dev = idx.device
    memb = torch.empty(b, p, 16, dtype=torch.int32, device=dev)
    flags = torch.empty(b, dtype=torch.int32, device=dev)
    slot_of = torch.empty(b, n, dtype=torch.int32, device=dev)
    call('eap_so3_dense_member', idx, b, p, n, nn, min(DENSE_MAX_ROWS, n), rows.stride(0), _ptr(idx), _ptr(rows), _ptr(n_rows), _ptr(slot_of),
         _ptr(memb), _ptr(flags))
    return memb, flags


# Occupancy-sorted query points (round 6): a cloud's query points are handed to the dense product sorted by WHICH 16-row groups of the
# referenced rows their lists touch (eap_so3_dense_point_keys), and every 256-column block of the product runs only the k-steps in
# which it generates a weight that is not masked out (eap_so3_dense_steps).  On the bench clouds 31 % (128 -> 512 layer) to 52 %
# (64 -> 128) of the k-steps drop out; skipped steps would have added exact zeros, so the result is bit-equal to running them all
# in the same point order.  The point order is internal: the backward reads dY through it as a column map, the forward's re-ordering
# pass writes Y through it.  SORT_DENSE_POINTS = False: index order and every k-step, as round 5 (A/B runs, tests).
SORT_DENSE_POINTS = os.environ.get('EAP_DENSE_SORT', '1') != '0'
SKIP_DENSE_STEPS = os.environ.get('EAP_DENSE_SKIP', '1') != '0'
lib.eap_so3_dense_steps_words.restype = ctypes.c_int64


class DenseGeometry:
    """What the dense product needs besides its stored operand, for one neighbourhood of a batch of clouds WITHOUT pose
    rotations: the lane masks of both directions (built on demand from the membership bits), the k-step lists of their column
    blocks, and the two float4 tables of the expanded weight, for the first rp referenced rows of every cloud."""

    def __init__(self, q_xyz, s_xyz, memb, rows, rp, rk, sigma, nn, n_rows=None, row_rot=None, sort=None):
        """row_rot float32 [b, rows.stride(0), 3, 3] (optional): one rotation per row slot applied to the kernel offsets -- the query
        points are ONE rigid part of a posed cloud (include/eap_hip.h: eap_so3_dense_tables_f32)."""
        b, p = memb.shape[:2]
        n = s_xyz.shape[2]
        na, ks, _ = rk.shape
        dev = memb.device
        self.order = self.pivot_pos = None
        if SORT_DENSE_POINTS if sort is None else sort:
            keys = torch.empty(b, p, dtype=torch.int32, device=dev)
            call('eap_so3_dense_point_keys', memb, b, p, _ptr(memb), _ptr(keys))
            order = torch.sort(keys, dim=1, stable=True).indices                               # int64 [b,p]: column j of the product is point order[b, j]
            q_xyz = q_xyz.gather(2, order[:, None, :].expand(b, 3, p)).contiguous()
            memb = memb.gather(1, order[:, :, None].expand(b, p, memb.shape[2])).contiguous()
            self.order = order.to(torch.int32).contiguous()
            self.pivot_pos = (order[0] == 0).to(torch.int32).argmax().to(torch.int32).view(1)   # the column of cloud 0 that is point 0
        self.b, self.p, self.na, self.ks, self.rp, self.nn, self.memb, self.sigma = b, p, na, ks, int(rp), int(nn), memb, float(sigma)
        self.n_rows = n_rows                      # int32 [b] or None: the products stop at every cloud's own rows (include/eap_hip.h)
        p_pad, kd_pad = (p + 31) // 32 * 32, (ks * self.rp + 31) // 32 * 32
        self.centre = torch.empty(b, 4, dtype=torch.float32, device=dev)
        self.pt = torch.empty(b, p_pad, 4, dtype=torch.float32, device=dev)
        self.kr = torch.empty(b, na, kd_pad, 4, dtype=torch.float32, device=dev)
        if row_rot is not None and (row_rot.dtype != torch.float32 or not row_rot.is_contiguous() or row_rot.numel() != b * rows.stride(0) * 9):
            raise RuntimeError('DenseGeometry: row_rot must be a contiguous float32 [b, rows.stride(0), 3, 3]')
        call('eap_so3_dense_tables_f32', memb, b, p, n, na, ks, self.rp, rows.stride(0), _F32(sigma), _ptr(q_xyz), _ptr(s_xyz), _ptr(rows),
             _ptr(rk), _ptr(row_rot), _ptr(self.centre), _ptr(self.pt), _ptr(self.kr))
        self._masks, self._steps, self._composed = {}, {}, {}

    def mask(self, direction):
        m = self._masks.get(direction)
        if m is None:
            words = int(lib.eap_so3_dense_mask_words(self.b, self.p, self.ks, self.rp, int(direction)))
            m = torch.empty(words, dtype=torch.int64, device=self.memb.device)
            call('eap_so3_dense_masks', m, self.b, self.p, self.ks, self.rp, int(direction), _ptr(self.memb), _ptr(m))
            self._masks[direction] = m
        return m

    def steps(self, direction):
        """int32 [b, column blocks, k-steps + 1] (count, then the k-steps a column block runs) or None (SKIP_DENSE_STEPS off: every k-step)."""
        if not SKIP_DENSE_STEPS:
            return None
        st = self._steps.get(direction)
        if st is None:
            kd = (self.ks * self.rp + 31) // 32 * 32
            blocks_n = ((self.p if direction else self.ks * self.rp) + 255) // 256
            k_steps = (kd if direction else self.p) // 32
            words = int(lib.eap_so3_dense_steps_words(self.b, self.p, self.ks, self.rp, int(direction)))
            assert words == self.b * blocks_n * (k_steps + 1)
            st = torch.empty(self.b, blocks_n, k_steps + 1, dtype=torch.int32, device=self.memb.device)
            call('eap_so3_dense_steps', st, self.b, self.p, self.ks, self.rp, int(direction), 1, _ptr(self.n_rows), _ptr(self.mask(direction)), _ptr(st))
            self._steps[direction] = st
        return self._steps[direction]

    def columns(self, col_map=None):
        """The column map of a launch: this geometry's point order composed with the caller's map (int32 [b,p]; negative: padding) -> int32
        [b,p] or None (index order, no map)."""
        if self.order is None:
            return col_map
        if col_map is None:
            return self.order
        hit = self._composed.get(id(col_map))
        if hit is None or hit[0] is not col_map:
            hit = (col_map, col_map.gather(1, self.order.long()).contiguous())
            self._composed = {id(col_map): hit}
        return hit[1]


def so3_dense_split(src, seg=0, seg_pitch=0, shape=None, mapped=False, n_rows=None, rowmax=None, colmap=None):
    """src [b,m,l,na] -> (scale [2,b,na,m], planes): the stored operand of the dense product (two fp16 planes of the scaled rows,
    fragment order).  seg > 0 (with shape = (b, m, l, na)): a row's l elements lie in l / seg segments of seg elements whose
    starts are seg_pitch floats apart (the rows of a GEMM output with padded columns).  mapped: the output's element l is the
    (k, r) pair with dense index l (the forward's operand), n_rows trims every cloud to its own rows.  rowmax int32 [b,m,na]: the
    rows' largest magnitudes (float bit patterns) when the producer already has them -- no pass over src for them (an upper bound will
    do).  colmap int32 [b,l']: the operand is the columns colmap[b, :] of src [b,m,l,na] (negative: a zero column)."""
    b, m, l, na = src.shape if shape is None else shape
    if colmap is not None:
        if colmap.dtype != torch.int32 or not colmap.is_contiguous() or colmap.shape[0] != b:
            raise RuntimeError('so3_dense_split: colmap must be a contiguous int32 [b, columns]')
        seg_pitch, l = l * na, colmap.shape[1]
    scale = torch.empty(2, b, na, m, dtype=torch.float32, device=src.device)      # [0]: [b,na,m]; [1]: the same numbers as [b,m,na]
    planes = torch.empty(b * na * m * ((l + 31) // 32 * 32), dtype=torch.int32, device=src.device)       # 4 bytes per element
    call('eap_so3_dense_split_f32', src, b, m, l, na, int(seg), _I64(seg_pitch), int(bool(mapped)), _ptr(n_rows), _ptr(rowmax), _ptr(colmap), _ptr(src), _ptr(scale),
         _ptr(planes))
    return scale, planes


def dense_pitch(n):
    """Row pitch for a matrix with n columns that the split contraction kernels take as an operand (whole 128-column tiles)."""
    return (n + 127) // 128 * 128


def _dense_executed_flops(geo, o, p, direction):
    """fp16 flops the product kernel executes for this geometry: 6 O A x 256 columns x 32 x (k-steps its column blocks run); without the
    step lists 6 O P A K x (every cloud's own rows rounded to 16).  The counts live on the device; they are read (a host wait) only while
    bench.py attributes launch times, else an upper bound nobody reads is returned."""
    rows = geo.rp * geo.b
    if KERNEL_TIMES is not None:
        st = geo.steps(direction)
        if st is not None and st.shape[2] - 1 <= 512:
            return 6.0 * o * geo.na * 256 * 32 * float(st[:, :, 0].sum().item())
        if geo.n_rows is not None:
            rows = sum(min((int(r) + 15) & ~15, geo.rp) for r in geo.n_rows.tolist())
    return 6.0 * o * p * geo.na * geo.ks * rows


def so3_dense_bwd(gy, geo, ldz=None, colmap=None, rowmax=False):
    """gy [b,o,p,na] -> Z [b,o,ks,ldz] whose rows hold [na,rp] (the inverse-list kernel's Z with the anchor axis in front of the row
    axis); ldz >= na*rp (default: equal) pads the rows for the GEMMs that follow -- the padding is NOT written.
    colmap int32 [b,p']: the product runs over the columns colmap[b, :] of gy (the query points of one rigid part: geo is that part's);
    rowmax: row maxima of gy somebody already took from the hint (None: none available; default: ask the hint)."""
    b, o, p, na = gy.shape
    ldz = na * geo.rp if ldz is None else int(ldz)
    colmap = geo.columns(colmap)                                  # (the geometry's own point order: occupancy-sorted)
    scale, planes = so3_dense_split(gy, rowmax=take_rowmax_hint(gy) if rowmax is False else rowmax, colmap=colmap)
    if colmap is not None:
        p = colmap.shape[1]
    z = torch.empty(b, o, geo.ks, ldz, dtype=torch.float32, device=gy.device)
    call('eap_so3_dense_product_steps_f32', gy, 0, b, o, p, na, geo.ks, geo.rp, _I64(ldz), _F32(geo.sigma), _ptr(geo.n_rows), _ptr(planes), _ptr(scale), _ptr(geo.pt),
         _ptr(geo.kr), _ptr(geo.mask(0)), _ptr(geo.steps(0)), _ptr(z),
         tag={'flops': 2.0 * b * o * p * na * geo.ks * geo.nn, 'executed_f16_flops': _dense_executed_flops(geo, o, p, 0),
              'shape': ('so3_dense', 0, b, o, p, na, geo.ks, geo.rp)})
    return z


def so3_dense_gplanes(fc4, W3, geo):
    """The dense forward's stored operand G = W F over the referenced rows, made and written as the product's planes by ONE kernel
    (eap_so3_dense_gplanes_f32): fc4 [b,c,rp,na] the referenced feature rows, W3 [o*ks, c] -> (scale [2,b,na,o], planes), or None when the
    shape is not taken (the caller then runs a GEMM + so3_dense_split).  The plane scale of an output row comes from a bound on its
    magnitude: too large a bound costs dynamic range only."""
    b, c, rp, na = fc4.shape
    o = W3.shape[0] // geo.ks
    if not GPLANES or not lib.eap_so3_dense_gplanes_supported(o, c, na, geo.ks, rp):
        return None
    ft = fc4.permute(0, 3, 2, 1).contiguous()                                       # [b,na,rp,c]
    # |G[o,(k,r),a]| <= max_k sum_c |W[o,c,k]| x max_{c,r} |F[c,r,a]|: two reductions over small tensors (the tighter sum_c |W| max_r |F_c|
    # is a matrix product whose launch path cost more host time than it saved range)
    wsum = W3.abs().sum(1).view(o, geo.ks).amax(1)                                  # [o]
    fm = fc4.abs().amax((1, 2))                                                     # [b,na]
    bound = (fm[:, :, None] * (wsum * 1.0001)[None, None, :]).contiguous()          # [b,na,o]
    scale = torch.empty(2, b, na, o, dtype=torch.float32, device=fc4.device)
    planes = torch.empty(b * na * o * ((geo.ks * rp + 31) // 32 * 32), dtype=torch.int32, device=fc4.device)
    call('eap_so3_dense_gplanes_f32', fc4, b, o, c, na, geo.ks, rp, _ptr(W3), _ptr(ft), _ptr(geo.n_rows), _ptr(bound), _ptr(scale), _ptr(planes),
         tag={'flops': 0.0, 'shape': ('so3_dense_gplanes', b, o, c, na, geo.ks, rp)})      # (write-bound operand preparation: not priced as matrix work)
    return scale, planes


GPLANES = os.environ.get('EAP_DENSE_GPLANES', '1') != '0'      # 0: G by a GEMM + the split pass, as round 5 (A/B runs, tests)


def so3_dense_fwd(g, geo, p, c=0, ldg=None, out=None, col_map=None, operand=None, o=None):
    """g [b,o,ks,ldg] (rows hold [rp,na]: W . F over the referenced rows, F with c channels; ldg >= rp*na, default equal) -> y [b,o,p,na];
    or operand = (scale, planes) of so3_dense_gplanes with g None (then o = the output width).
    (c only prices the launch for bench.py: the reference's grouping einsum + contraction, minus the small GEMM that made g.)
    out [b,o,p_dst,na] with col_map int32 [b,p]: the p columns are the points col_map[b, :] of `out` (negative: padding) -- the launch
    of one rigid part of posed clouds; returns out."""
    yt = _dense_fwd_yt(g, geo, p, c, ldg, operand, o)
    g = yt
    b, na, o = yt.shape[0], geo.na, yt.shape[2]
    if col_map is not None:
        if out is None or col_map.dtype != torch.int32 or tuple(col_map.shape) != (b, p) or not col_map.is_contiguous() or not out.is_contiguous():
            raise RuntimeError('so3_dense_fwd: col_map must be a contiguous int32 [b,p] and come with a contiguous out')
        call('eap_so3_dense_untranspose_map_f32', g, b, o, p, na, out.shape[2], _ptr(geo.columns(col_map)), _ptr(yt), _ptr(out))
        return out
    y = torch.empty(b, o, p, na, dtype=torch.float32, device=g.device)
    # the re-ordering pass also leaves the channel moments a BatchNorm right behind this layer starts with (its own pass otherwise)
    chunks = (p + 63) // 64
    ps = torch.empty(o, b * chunks, dtype=torch.float32, device=g.device)
    pq = torch.empty_like(ps)
    if geo.order is not None:                                     # columns in the geometry's point order: written through it
        call('eap_so3_dense_untranspose_map_stats_f32', g, b, o, p, na, p, _ptr(geo.order), _ptr(geo.pivot_pos), _ptr(yt), _ptr(y), _ptr(ps), _ptr(pq))
    else:
        call('eap_so3_dense_untranspose_f32', g, b, o, p, na, _ptr(yt), _ptr(y), _ptr(ps), _ptr(pq))
    leave_stats_hint(y, (ps, pq))
    return y


def _dense_fwd_yt(g, geo, p, c, ldg, operand, o):
    """the forward product alone -> Yt [b,na,o,p] (columns in the geometry's point order)"""
    na = geo.na
    if operand is None:
        b, o = g.shape[:2]
        ldg = geo.rp * na if ldg is None else int(ldg)
        scale, planes = so3_dense_split(g, seg=geo.rp, seg_pitch=ldg, shape=(b, o, geo.ks * geo.rp, na), mapped=True, n_rows=geo.n_rows)
    else:
        scale, planes = operand
        b = geo.b
    yt = torch.empty(b, na, o, p, dtype=torch.float32, device=planes.device)
    call('eap_so3_dense_product_steps_f32', yt, 1, b, o, p, na, geo.ks, geo.rp, _I64(0), _F32(geo.sigma), _ptr(geo.n_rows), _ptr(planes), _ptr(scale), _ptr(geo.pt),
         _ptr(geo.kr), _ptr(geo.mask(1)), _ptr(geo.steps(1)), _ptr(yt),
         tag={'flops': 2.0 * b * c * geo.ks * na * (p * geo.nn + o * p - o * geo.rp), 'executed_f16_flops': _dense_executed_flops(geo, o, p, 1),
              'shape': ('so3_dense', 1, b, o, p, na, geo.ks, geo.rp)})
    return yt


def so3_dense_fwd_bnact(g, geo, p, c, ldg, norm_moments, operand=None, o=None, affine=None):
    """The dense forward with the training-mode BatchNorm + leaky_relu behind it applied by the re-ordering pass (the conv + BatchNorm
    node of vgtk/so3conv/functional.py): product -> Yt; statistics pass over Yt; norm_moments(s1, s2, pivot, count) -> (scale, shift, slope)
    per channel (float32 [o]); y' = leaky(scale y + shift) written through the geometry's point order.  affine = (scale, shift, slope)
    instead of norm_moments: an inference-mode norm folded into one per-channel map (FoldedEpilogue), no statistics pass.  -> y' [b,o,p,na]"""
    yt = _dense_fwd_yt(g, geo, p, c, ldg, operand, o)
    b, na, o = yt.shape[0], geo.na, yt.shape[2]
    if affine is not None:
        bn_scale, bn_shift, slope = affine[0].contiguous(), affine[1].contiguous(), float(affine[2])
    else:
        # moments of every channel over (cloud, anchor, point): Yt is [b na][o][p] for the statistics kernel; the pivot is its own first element
        ps, pq = _partials(yt, b * na, o, p)
        call('eap_bn_stats_f32', yt, b * na, o, _I64(p), _ptr(yt), _ptr(ps), _ptr(pq))
        bn_scale, bn_shift, slope = norm_moments(ps.sum(1, dtype=torch.float64), pq.sum(1, dtype=torch.float64), yt[0, 0, :, 0].double(), b * na * p)
    y = torch.empty(b, o, p, na, dtype=torch.float32, device=yt.device)
    call('eap_so3_dense_untranspose_bnact_f32', yt, b, o, p, na, p, _ptr(geo.order), _ptr(yt), _ptr(bn_scale), _ptr(bn_shift), _F32(slope), _ptr(y))
    return y


def bn_act_bwd_reduce_fromy(gy, y, beta, inv_gamma, slope):
    """gy, y [b,c,p,na] (y = the layer's activated output) -> (sum g, sum g xhat) float64 [c] and the per-row maxima gmax, xmax float32 [b,c,na]
    (csrc/bn_act.hip bn_act_bwd_reduce_fromy_kernel)."""
    b, c, p, na = y.shape
    n = p * na
    nblk = int(lib.eap_bn_act_fromy_blocks(_I64(n), na))
    pg = torch.empty(c, b * nblk, dtype=torch.float32, device=y.device)
    pgx = torch.empty_like(pg)
    gmax = torch.empty(b, c, na, dtype=torch.int32, device=y.device)
    xmax = torch.empty_like(gmax)
    call('eap_bn_act_bwd_reduce_fromy_f32', y, b, c, _I64(n), na, _F32(slope), _ptr(gy), _ptr(y), _ptr(beta), _ptr(inv_gamma), _ptr(pg), _ptr(pgx), _ptr(gmax), _ptr(xmax))
    return pg.sum(1, dtype=torch.float64), pgx.sum(1, dtype=torch.float64), gmax.view(torch.float32), xmax.view(torch.float32)


def so3_dense_bwd_bn(gy, yact, geo, ldz, coef, rowbound, slope):
    """so3_dense_bwd for the gradient BEHIND a training-mode BatchNorm + leaky_relu, formed while it is split into the product's planes
    (eap_so3_dense_split_bn_f32): gy = dL/dy', yact = y' [b,o,p,na], coef float32 [5,o] = k1, k2, k3, beta, 1/gamma, rowbound float32 [b,o,na] >=
    max |gx| of every row.  -> Z as so3_dense_bwd."""
    b, o, p, na = gy.shape
    ldz = na * geo.rp if ldz is None else int(ldz)
    colmap = geo.columns(None)
    l = p if colmap is None else colmap.shape[1]
    scale = torch.empty(2, b, na, o, dtype=torch.float32, device=gy.device)
    planes = torch.empty(b * na * o * ((l + 31) // 32 * 32), dtype=torch.int32, device=gy.device)
    call('eap_so3_dense_split_bn_f32', gy, b, o, l, p, na, _ptr(rowbound.view(torch.int32)), _ptr(colmap), _ptr(gy), _ptr(yact), _ptr(coef), _F32(slope), _ptr(scale),
         _ptr(planes))
    z = torch.empty(b, o, geo.ks, ldz, dtype=torch.float32, device=gy.device)
    call('eap_so3_dense_product_steps_f32', gy, 0, b, o, l, na, geo.ks, geo.rp, _I64(ldz), _F32(geo.sigma), _ptr(geo.n_rows), _ptr(planes), _ptr(scale), _ptr(geo.pt),
         _ptr(geo.kr), _ptr(geo.mask(0)), _ptr(geo.steps(0)), _ptr(z),
         tag={'flops': 2.0 * b * o * l * na * geo.ks * geo.nn, 'executed_f16_flops': _dense_executed_flops(geo, o, l, 0),
              'shape': ('so3_dense', 0, b, o, l, na, geo.ks, geo.rp)})
    return z


def _partials(x, b, c, n):
    nseg = int(lib.eap_bn_act_segments(_I64(n)))
    return (torch.empty(c, b * nseg, dtype=torch.float32, device=x.device),
            torch.empty(c, b * nseg, dtype=torch.float32, device=x.device))


# Channel moments of a tensor, handed from the kernel that WROTE it (the dense forward's re-ordering pass) to the BatchNorm behind
# it: one entry, keyed like the row-maximum hint below (tensor object + version + shape)
_STATS_HINT = [None]
STATS_HINTS_TAKEN = 0


def leave_stats_hint(t, partials):
    import weakref
    _STATS_HINT[0] = (weakref.ref(t), t._version, tuple(t.shape), partials)


def take_stats_hint(t):
    global STATS_HINTS_TAKEN
    h, _STATS_HINT[0] = _STATS_HINT[0], None
    if not USE_ROWMAX_HINT or h is None or h[0]() is not t or h[1] != t._version or h[2] != tuple(t.shape):
        return None
    STATS_HINTS_TAKEN += 1
    return h[3]


def bn_stats(x, b, c, n):
    """-> (sum, sumsq) of x - pivot per channel, float64 [c] (pivot = x[0, c, 0])."""
    hint = take_stats_hint(x)
    if hint is not None and hint[0].shape[0] == c:
        return hint[0].sum(1, dtype=torch.float64), hint[1].sum(1, dtype=torch.float64)
    ps, pq = _partials(x, b, c, n)
    call('eap_bn_stats_f32', x, b, c, _I64(n), _ptr(x), _ptr(ps), _ptr(pq))
    return ps.sum(1, dtype=torch.float64), pq.sum(1, dtype=torch.float64)


def bn_act_fwd(x, b, c, n, scale, shift, slope, residual=None):
    y = torch.empty_like(x)
    if residual is None:
        call('eap_bn_act_fwd_f32', x, b, c, _I64(n), _F32(slope), _ptr(x), _ptr(scale), _ptr(shift), _ptr(y))
    else:
        call('eap_bn_act_add_fwd_f32', x, b, c, _I64(n), _F32(slope), _ptr(x), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(y))
    return y


def bn_act_bwd_reduce(gy, x, b, c, n, scale, shift, mean, invstd, slope):
    pg, pgx = _partials(x, b, c, n)
    call('eap_bn_act_bwd_reduce_f32', x, b, c, _I64(n), _F32(slope), _ptr(gy), _ptr(x), _ptr(scale), _ptr(shift),
         _ptr(mean), _ptr(invstd), _ptr(pg), _ptr(pgx))
    return pg.sum(1, dtype=torch.float64), pgx.sum(1, dtype=torch.float64)


# Row maxima of a gradient tensor, handed from the kernel that WROTE it (the BatchNorm backward) to the one that splits it into
# fp16 planes (the dense backward product's stored operand): one entry, keyed on the tensor OBJECT (a weak reference: autograd
# hands the same tensor to the next node when nothing is accumulated into it) and its version counter.  Anything else -- another
# tensor, a tensor modified in between, no entry -- and the split finds the maxima itself.
_ROWMAX_HINT = [None]
USE_ROWMAX_HINT = True        # False: the split always finds its maxima itself (A/B runs, tests)
ROWMAX_HINTS_TAKEN = 0        # how often a hint was accepted (tests)


def leave_rowmax_hint(t, rowmax):
    import weakref
    _ROWMAX_HINT[0] = (weakref.ref(t), t._version, tuple(t.shape), rowmax)


def take_rowmax_hint(t):
    global ROWMAX_HINTS_TAKEN
    h, _ROWMAX_HINT[0] = _ROWMAX_HINT[0], None
    if not USE_ROWMAX_HINT or h is None or h[0]() is not t or h[1] != t._version or h[2] != tuple(t.shape):
        return None
    ROWMAX_HINTS_TAKEN += 1
    return h[3]


def bn_act_bwd_apply(gy, x, b, c, n, scale, shift, mean, invstd, k2, k3, slope):
    gx = torch.empty_like(x)
    if x.dim() == 4 and x.shape[3] % 4 == 0 and x.shape[3] <= 64 and n == x.shape[2] * x.shape[3] and c % 128 == 0 and gx.data_ptr() % 16 == 0 \
            and gy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0:
        # [b, c, points, anchors] at a width the dense backward product takes: the same pass also leaves the row maxima its
        # stored-operand split needs (csrc/bn_act.hip bn_act_bwd_apply_rowmax_kernel)
        rowmax = torch.empty(b, c, x.shape[3], dtype=torch.int32, device=x.device)
        call('eap_bn_act_bwd_apply_rowmax_f32', x, b, c, _I64(n), int(x.shape[3]), _F32(slope), _ptr(gy), _ptr(x), _ptr(scale), _ptr(shift),
             _ptr(mean), _ptr(invstd), _ptr(k2), _ptr(k3), _ptr(gx), _ptr(rowmax))
        leave_rowmax_hint(gx, rowmax)
        return gx
    call('eap_bn_act_bwd_apply_f32', x, b, c, _I64(n), _F32(slope), _ptr(gy), _ptr(x), _ptr(scale), _ptr(shift),
         _ptr(mean), _ptr(invstd), _ptr(k2), _ptr(k3), _ptr(gx))
    return gx


# per-cloud statistics over a point subset (the pose heads' batched per-cloud calls); scale .. k3 are [b, c]

def bn_stats_masked(x, b, c, n, na, mask):
    """-> (sum, sumsq) of mask * (x - pivot) per (cloud, channel), float64 [b, c] (pivot = x[0, c, 0])."""
    ps, pq = _partials(x, b, c, n)
    call('eap_bn_stats_masked_f32', x, b, c, _I64(n), na, _ptr(x), _ptr(mask), _ptr(ps), _ptr(pq))
    return ps.view(c, b, -1).sum(2, dtype=torch.float64).t(), pq.view(c, b, -1).sum(2, dtype=torch.float64).t()


def bn_act_cloud_fwd(x, b, c, n, scale, shift, slope):
    y = torch.empty_like(x)
    call('eap_bn_act_cloud_fwd_f32', x, b, c, _I64(n), _F32(slope), _ptr(x), _ptr(scale), _ptr(shift), _ptr(y))
    return y


def bn_act_cloud_bwd_reduce(gy, x, b, c, n, scale, shift, mean, invstd, slope):
    pg, pgx = _partials(x, b, c, n)
    call('eap_bn_act_cloud_bwd_reduce_f32', x, b, c, _I64(n), _F32(slope), _ptr(gy), _ptr(x), _ptr(scale), _ptr(shift),
         _ptr(mean), _ptr(invstd), _ptr(pg), _ptr(pgx))
    return pg.view(c, b, -1).sum(2, dtype=torch.float64).t(), pgx.view(c, b, -1).sum(2, dtype=torch.float64).t()


def bn_act_cloud_bwd_apply(gy, x, b, c, n, na, scale, shift, mean, invstd, k2, k3, mask, slope):
    gx = torch.empty_like(x)
    call('eap_bn_act_cloud_bwd_apply_f32', x, b, c, _I64(n), na, _F32(slope), _ptr(gy), _ptr(x), _ptr(scale), _ptr(shift),
         _ptr(mean), _ptr(invstd), _ptr(k2), _ptr(k3), _ptr(mask), _ptr(gx))
    return gx


def so3_intra_conv(feats, W, intra_idx32):
    """Implicit-GEMM intra conv forward: feats [b,c,p,na], W [o, c*nt], intra_idx int32 [na,nt] -> [b,o,p,na]."""
    b, c, p, na = feats.shape
    o, nt = W.shape[0], intra_idx32.shape[1]
    out = torch.empty(b, o, p, na, dtype=torch.float32, device=feats.device)
    split = SPLIT_BF16_CONTRACTION and W.data_ptr() % 16 == 0 and lib.eap_so3_intra_conv_bf16x3_f32_supported(b, o, c, p, na, nt)
    tag = {'flops': 2.0 * b * o * c * nt * p * na, 'shape': ('intra_conv', b, o, c, p, na, nt)}
    if split and SPLIT_PLANES == 2 and feats.data_ptr() % 16 == 0:
        abs_w = absmax_rows(W, 1, o, c * nt, c * nt, 0)
        abs_f = absmax_colgroups(feats, b, c, p * na, p * na, c * p * na, na) if na % 4 == 0 else None
        if abs_w is not None and abs_f is not None:
            call('eap_so3_intra_conv_f16x2_f32', out, b, o, c, p, na, nt, _ptr(W), _ptr(feats), _ptr(intra_idx32), _ptr(out), _ptr(abs_w),
                 _ptr(abs_f), tag=tag)
            return out
    call('eap_so3_intra_conv_bf16x3_f32' if split else 'eap_so3_intra_conv_f32', out, b, o, c, p, na, nt, _ptr(W), _ptr(feats),
         _ptr(intra_idx32), _ptr(out), tag=tag)
    return out
