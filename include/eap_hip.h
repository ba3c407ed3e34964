/*
 * include/eap_hip.h -- C ABI of libeap_hip.so, the MI355X (gfx950) implementation of the
 * SE(3)-equivariant point-convolution hot path of Meowuu7/equi-articulated-pose.
 *
 * Drop-in boundary B2: every entry point below replaces one pybind11
 * function of the reference's CUDA extensions (the .cpp files of vgtk/vgtk/cuda and extensions/chamfer_dist)
 * or one stage of the Python "naive" operator path that the shipped models execute
 * (vgtk/vgtk/so3conv/functional.py).  The reference interface each one replaces is cited as
 * file:line (paths relative to /root/reference).
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers, tensors are
 *     contiguous in the layout written next to each argument;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); launches are
 *     asynchronous, nothing synchronises;
 *   - outputs are fully written by the call (no pre-zeroing needed) unless stated otherwise;
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch / argument
 *     check (eap_last_error() returns a message).  The reference only printf()s launch errors
 *     (vgtk/vgtk/cuda/zpconv_cuda_kernel.cu:L232-234); this ABI reports them.
 *   - *_f32 / *_f64: the reference dispatches AT_DISPATCH_FLOATING_TYPES for its native ops.
 */
#ifndef EAP_HIP_H
#define EAP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *eap_stream_t;

const char *eap_last_error(void);
/* Name, with its template arguments, of the dominant HIP kernel the calling thread's last entry launched ("" when the
 * entry does not record one); reading clears it.  Lets a profiler-less caller (bench.py) attribute launch times to kernels. */
const char *eap_last_kernel(void);
int eap_abi_version(void);

/* ---- grouping (vgtk/vgtk/cuda/grouping_cuda.cpp) ------------------------------------------ */

/* ball_query: grouping_cuda.cpp:L71-86, kernel grouping_cuda_kernel.cu:L68-113.
 * new_xyz [b,3,m], xyz [b,3,n] -> idx int32 [b,m,nsample]: first `nsample` support indices in
 * index order with d2 < radius^2; repeat-padding when fewer than nsample-1 hits. */
int eap_ball_query_f32(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                       const float *xyz, int32_t *idx, eap_stream_t stream);
int eap_ball_query_f64(int b, int n, int m, float radius, int nsample, const double *new_xyz,
                       const double *xyz, int32_t *idx, eap_stream_t stream);

/* furthest_point_sampling: grouping_cuda.cpp:L160-174, kernel .cu:L352-466.
 * xyz [b,3,n] -> idx int32 [b,m]; temp [b,n] is scratch (initialised by the call). */
int eap_furthest_point_sampling_f32(int b, int n, int m, const float *xyz, float *temp,
                                    int32_t *idx, eap_stream_t stream);

/* anchor_query (S^2 variant): grouping_cuda.cpp:L88-108, kernel .cu:L181-247.
 * grouped_xyz [b,3,np,nn], anchors [na,3], kernel_pts [ks,2] -> w [b,np,na,ks,nn]. */
int eap_anchor_query_f32(int b, int np, int nn, int na, int ks, const float *grouped_xyz,
                         const float *anchors, const float *kernel_pts, float *w,
                         eap_stream_t stream);

/* initial_anchor_query: grouping_cuda.cpp:L138-158, kernel .cu:L117-167.
 * centers [b,3,nc], xyz [m,3], kernel_pts [ks,na,3] -> w, cnt [b,ks,nc,na]. */
int eap_initial_anchor_query_f32(int b, int nc, int m, int na, int ks, float radius, float sigma,
                                 const float *centers, const float *xyz, const float *kernel_pts,
                                 float *w, float *cnt, eap_stream_t stream);

/* ---- gathering (vgtk/vgtk/cuda/gathering_cuda.cpp) ---------------------------------------- */

/* gather_points_forward: gathering_cuda.cpp:L29-43. pts [b,c,n], idx [b,m] -> out f32 [b,c,m]. */
int eap_gather_points_fwd_f32(int b, int c, int n, int m, const float *pts, const int32_t *idx,
                              float *out, eap_stream_t stream);
/* gather_points_backward: gathering_cuda.cpp:L45-60. grad_out [b,c,m] -> grad_pts [b,c,n]. */
int eap_gather_points_bwd_f32(int b, int c, int n, int m, const float *grad_out,
                              const int32_t *idx, float *grad_pts, eap_stream_t stream);
int eap_gather_points_bwd_f64(int b, int c, int n, int m, const double *grad_out,
                              const int32_t *idx, double *grad_pts, eap_stream_t stream);

/* ---- zpconv (vgtk/vgtk/cuda/zpconv_cuda.cpp) ---------------------------------------------- */

/* inter_zpconv_forward: zpconv_cuda.cpp:L41-56, kernel zpconv_cuda_kernel.cu:L33-73.
 * idx,w [b,np,na,ks,ann], feats [b,c,nq,na] -> out [b,c,ks,np,na]. */
int eap_inter_zpconv_fwd_f32(int b, int np, int nq, int na, int ks, int ann, int c,
                             const int32_t *idx, const float *w, const float *feats, float *out,
                             eap_stream_t stream);
int eap_inter_zpconv_fwd_f64(int b, int np, int nq, int na, int ks, int ann, int c,
                             const int32_t *idx, const double *w, const double *feats,
                             double *out, eap_stream_t stream);
/* The same op with a scratch buffer of eap_inter_zpconv_fwd_workspace(b, np, ann) BYTES (16-byte aligned): the index is
 * checked on device for the pattern every reference caller produces (one neighbour list per point broadcast over
 * (a,k): spconv/functional.py:L232-249) and those clouds run on the matrix cores (csrc/zpconv_mfma.hip); any other
 * cloud, size or alignment falls through to eap_inter_zpconv_fwd_f32's kernels.  Same results either way. */
int64_t eap_inter_zpconv_fwd_workspace(int b, int np, int ann);
int eap_inter_zpconv_fwd_ws_f32(int b, int np, int nq, int na, int ks, int ann, int c,
                                const int32_t *idx, const float *w, const float *feats, float *out,
                                void *workspace, eap_stream_t stream);
/* inter_zpconv_backward: zpconv_cuda.cpp:L58-75, kernel .cu:L77-116.
 * grad [b,c,ks,np,na] -> gfeats [b,c,nq,na]. */
int eap_inter_zpconv_bwd_f32(int b, int np, int nq, int na, int ks, int ann, int c,
                             const int32_t *idx, const float *w, const float *grad, float *gfeats,
                             eap_stream_t stream);
int eap_inter_zpconv_bwd_f64(int b, int np, int nq, int na, int ks, int ann, int c,
                             const int32_t *idx, const double *w, const double *grad,
                             double *gfeats, eap_stream_t stream);
/* The same op without atomics, with a scratch buffer of eap_inter_zpconv_bwd_workspace(b, np, nq, na, ann, c) BYTES
 * (256-byte aligned; dominated by 4*b*np*na*ann*c for the per-(point, neighbour) products -- split the batch to bound
 * it): for clouds whose index is one neighbour list per point (checked on device) the products are formed in forward
 * order on the matrix cores and summed per support point over device-built inverse lists, in a fixed order
 * (csrc/zpconv_bwd.hip); any other cloud, size or alignment goes through eap_inter_zpconv_bwd_f32's scatter kernel. */
int64_t eap_inter_zpconv_bwd_workspace(int b, int np, int nq, int na, int ann, int c);
int eap_inter_zpconv_bwd_ws_f32(int b, int np, int nq, int na, int ks, int ann, int c,
                                const int32_t *idx, const float *w, const float *grad, float *gfeats,
                                void *workspace, eap_stream_t stream);
/* The same op (zpconv_cuda.cpp:L58-75) with the scatter target held in LDS -- no per-(point, neighbour) intermediate
 * (csrc/zpconv_bwd_hot.hip): for clouds whose index is one neighbour list per point AND whose lists reference at most
 * eap_inter_zpconv_bwd_hot_rows() distinct support rows (large balls under the reference's first-nsample-in-index-order
 * ball query), every (cloud, anchor quad, 32 channels) accumulates its rows on chip and writes them once; sums in a fixed
 * order, no atomics on global memory.  status[i] (device, int32 [b]) = 0: cloud i done; 1: untouched apart from being
 * zeroed -- hand it to eap_inter_zpconv_bwd_ws_f32.  eap_inter_zpconv_bwd_hot_workspace returns 0 for shapes this
 * path does not take (ks != 24, ann != 64, na or c not multiples of 4 / 32); workspace 256-byte aligned. */
int eap_inter_zpconv_bwd_hot_rows(void);
int64_t eap_inter_zpconv_bwd_hot_workspace(int b, int np, int nq, int na, int ks, int ann, int c);
int eap_inter_zpconv_bwd_hot_f32(int b, int np, int nq, int na, int ks, int ann, int c,
                                 const int32_t *idx, const float *w, const float *grad, float *gfeats,
                                 void *workspace, int32_t *status, eap_stream_t stream);
/* intra_zpconv_forward: zpconv_cuda.cpp:L77-92, kernel .cu:L120-156.
 * idx [na_out,ann], w [na_out,ks,ann], feats [b,c,np,na_in] -> out [b,c,ks,np,na_out]. */
int eap_intra_zpconv_fwd_f32(int b, int np, int na_in, int na_out, int ks, int ann, int c,
                             const int32_t *idx, const float *w, const float *feats, float *out,
                             eap_stream_t stream);
int eap_intra_zpconv_fwd_f64(int b, int np, int na_in, int na_out, int ks, int ann, int c,
                             const int32_t *idx, const double *w, const double *feats,
                             double *out, eap_stream_t stream);
/* intra_zpconv_backward: zpconv_cuda.cpp:L94-110, kernel .cu:L160-195. */
int eap_intra_zpconv_bwd_f32(int b, int np, int na_in, int na_out, int ks, int ann, int c,
                             const int32_t *idx, const float *w, const float *grad, float *gfeats,
                             eap_stream_t stream);
int eap_intra_zpconv_bwd_f64(int b, int np, int na_in, int na_out, int ks, int ann, int c,
                             const int32_t *idx, const double *w, const double *grad,
                             double *gfeats, eap_stream_t stream);

/* ---- SO(3) inter convolution (vgtk/vgtk/so3conv/functional.py) ---------------------------- */

/* so3_prep: relative offsets and relative-rotation anchor of every (point, neighbour) pair --
 * so3conv/functional.py:L1061-1078 (gather + R_p R_n^T + rotate offsets) and the nearest-anchor
 * half of L1199-1204.
 * q_xyz [b,3,p], s_xyz [b,3,n], idx int32 [b,p,nn], pose [b,n,4,4] (q_pose [b,p,4,4]) or NULL
 * for identity poses, anchors [na,3,3]
 *  -> gx float4 [b,p,nn] = (gx,gy,gz, bit-cast int r) with g = R_rel (x_n - x_p) and
 *     r = argmax_g tr(R_rel A_g)  (r = index of the identity anchor when pose == NULL);
 *     nonident int32 [b] (optional): set to 1 for clouds where some r != identity_anchor. */
int eap_so3_prep_f32(int b, int p, int n, int nn, int na, const float *q_xyz, const float *s_xyz,
                     const int32_t *idx, const float *q_pose, const float *s_pose,
                     const float *anchors, int identity_anchor, float *gx, int32_t *nonident,
                     eap_stream_t stream);

/* so3_inter_weights: inter_so3conv_grouping_anchor, so3conv/functional.py:L2508-2549.
 * gx float4 [b,p,nn] (from so3_prep), rk [na,ks,3] = A_a kappa_k
 *  -> w [b,p,na,ks,nn] = relu(1 - |g - rk|^2 / sigma). */
int eap_so3_inter_weights_f32(int b, int p, int nn, int na, int ks, float sigma, const float *gx,
                              const float *rk, float *w, eap_stream_t stream);

/* so3_anchor_perm: the full index of so3conv/functional.py:L1199-1204,
 * perm[b,p,n,a] = mult[r[b,p,n]][a] as int64 (what torch.argmax returns). */
int eap_so3_anchor_perm(int b, int p, int nn, int na, const float *gx, const uint8_t *mult,
                        int64_t *perm, eap_stream_t stream);

/* so3_inter_group_fwd: fused kernel-weight + anchor-permutation + gather + weighted sum,
 * so3conv/functional.py:L1112-1261 (einsum 'bcpna,bpakn->bckpa' at L1261) without materialising
 * the [b,p,na,ks,nn] weights.
 * feats [b,c,n,na], idx [b,p,nn], gx float4 [b,p,nn], rk [na,ks,3], mult [na,na] (NULL = no
 * permutation, permute_modes == 0), nonident int32 [b] from so3_prep or NULL (clouds whose flag
 * is 0 skip the table) -> out [b,c,ks,p,na]. */
int eap_so3_inter_group_fwd_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma,
                                const float *feats, const int32_t *idx, const float *gx,
                                const float *rk, const uint8_t *mult, const int32_t *nonident,
                                float *out, eap_stream_t stream);
/* The two implementations behind so3_inter_group_fwd, exported for tests and profiling:
 * _mfma (c >= 16): v_mfma_f32_32x32x2_f32 with the kernel weights generated in registers as the
 * B operand; _valu: lanes = anchors, register tile of channels x kernel points. */
int eap_so3_inter_group_fwd_mfma_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma,
                                     const float *feats, const int32_t *idx, const float *gx,
                                     const float *rk, const uint8_t *mult, const int32_t *nonident,
                                     float *out, eap_stream_t stream);
int eap_so3_inter_group_fwd_valu_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma,
                                     const float *feats, const int32_t *idx, const float *gx,
                                     const float *rk, const uint8_t *mult, float *out,
                                     eap_stream_t stream);

/* so3_inter_group_bwd: transpose of the above w.r.t. feats.
 * gout [b,c,ks,p,na] -> gfeats [b,c,n,na] (zero-initialised by the call, fp32 atomics). */
int eap_so3_inter_group_bwd_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma,
                                const float *gout, const int32_t *idx, const float *gx,
                                const float *rk, const uint8_t *mult, float *gfeats,
                                eap_stream_t stream);

/* Atomics-free variant of so3_inter_group_bwd (ks <= 24, na % 4 == 0): every block owns a private slab of
 * partial sums in `workspace` (eap_so3_inter_group_bwd_workspace floats, zeroed by the call) and
 * a second kernel reduces the slabs -- deterministic, and ~4x faster than fp32 row atomics. */
int64_t eap_so3_inter_group_bwd_workspace(int b, int c, int p, int n, int na);
int eap_so3_inter_group_bwd_slab_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma,
                                     const float *gout, const int32_t *idx, const float *gx,
                                     const float *rk, const uint8_t *mult, int identity_anchor,
                                     float *gfeats, float *workspace, eap_stream_t stream);

/* Which entry-list grouping kernel serves eap_so3_inter_group_fwd*_f32 / eap_so3_inter_group_inv*_f32 when no anchor
 * permutation is in play: 2 (default) = the fp32-MFMA kernel with two channel tiles per wave where the channel count fills
 * 64-channel blocks (csrc/so3_inter_lists2.hip), 1 = always the one-tile fp32-MFMA kernel (csrc/so3_inter_lists.hip);
 * 0 = query.  Returns the value in force.  1 and 2 agree bit for bit (tests compare them).  (Measured-slower experiments --
 * a 3 x bf16 split grouping kernel, a re-cut zpconv forward -- live in tools/experiments/kernels/ and are compiled in by
 * `make EXPERIMENTS=1` only.) */
int eap_so3_group_lists_tiles(int tiles);
/* Row end of the two-tile forward kernel writing the transposed intermediate: 1 (default) = MFMA operands exchanged so that a
 * lane holds four consecutive kernel points of a channel, 16-byte stores; 0 = dword stores in 96-byte runs.  Bit-identical
 * results.  Returns the previous setting; other values only query. */
int eap_so3_group_lists_store16(int on);
/* forward grouping of clouds WITH anchor permutations: 1 (default) = the entry-list kernel of csrc/so3_inter_inv.hip in its
 * forward mode (global -> LDS DMA rows), 0 = the register-staged kernel of csrc/so3_inter_mfma.hip (round 1); returns the
 * previous setting, any other argument only queries.  Same results to rounding; for A/B runs and tests. */
int eap_so3_group_perm_fwd(int on);
/* Clouds WITH anchor permutations on the two-tile kernel (csrc/so3_inter_lists2.hip, PERM; round 4): the operand's anchor
 * axis is COSET-MAJOR (eap_anchor_reorder_f32 with the `order` of vgtk.so3conv.functional._coset_tables), the block move of an
 * entry's permutation rides on the DMA source addresses, the in-block XOR is 8 selects per operand read, and everything that
 * depends on the entry's rotation alone is prepared once per entry:
 *   eap_so3_group_perm_lists2(on)      1 (default) / 0: switch for A/B runs and tests; returns the previous setting
 *   eap_so3_group_perm_lists2_takes    1 if this shape goes there (else the whole-row kernel, eap_so3_inter_group_inv_coset_f32)
 *   eap_so3_perm_entries_f32           ent_p int32 [b*per_cloud], ent_gx [b*per_cloud,4] (w = bits of the rotation's anchor r),
 *                                      nonident int32 [b] or NULL (clouds with flag 0 are skipped),
 *                                      code uint8 [na,16] (coset code table of the permutation table in force), anchors [na,3,3]
 *                                      or NULL (backward: offset vectors rotated by A_r)
 *                                      -> ent_pc uint32 [b*per_cloud,4,4] (byte offset of each 16-byte piece's source in a channel
 *                                         row of the cloud, per anchor group and piece), ent_gx2 [b*per_cloud,4] (w = in-block
 *                                         XOR bits of the 15 blocks; shadow neighbours get a dead offset vector)
 *   eap_so3_inter_group_inv_perm2_f32  Z of the re-associated backward (replaces autograd of so3conv/functional.py:L1199-1261
 *                                      for permuted clouds): gy [b,o,p,na] coset-major -> z [b,o,ks,rcap,na] coset-major */
int eap_so3_group_perm_lists2(int on);
int eap_so3_group_perm_lists2_takes(int channels, int na, int ks, int n_support);
int eap_so3_perm_entries_f32(int b, int per_cloud, int na, int n_support, const int32_t *ent_p, const float *ent_gx, const uint8_t *code,
                             const float *anchors, int identity_anchor, const int32_t *nonident, int32_t *ent_pc, float *ent_gx2,
                             eap_stream_t stream);
/* the forward with the same kernel: eap_so3_inter_group_fwd_t_f32's output (X^T, memory order) where the clouds WITH
 * permutations (nonident[b] != 0) read feats_c = feats with a coset-major anchor axis (eap_anchor_reorder_clouds_f32) and the
 * per-entry words of (idx, gx) (eap_so3_perm_entries_f32 with the multiplication table's code, anchors = NULL, the flags); the
 * other clouds take the plain kernel.  Replaces so3conv/functional.py:L1199-1261 for those clouds. */
int eap_so3_inter_group_fwd_perm2_t_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                                        const float *feats_c, const int32_t *idx, const float *gx, const int32_t *ent_pc,
                                        const float *ent_gx2, const float *rk, const uint8_t *order, const int32_t *nonident,
                                        int store_order_columns, float *out, eap_stream_t stream);
/* eap_so3_inter_group_fwd_t_f32 (clouds without anchor permutations) with the COLUMNS of the transposed matrix in the order the
 * kernel's lanes hold them, so that every store instruction writes one contiguous 1 KB run (round 4; the intermediate is
 * scratch between the grouping and the contraction, and a contraction sums over the columns: it reads W[:, perm] beside it).
 * eap_so3_group_fwd_tp_takes: 1 if the shape is taken (whole 64-channel blocks, ks % 8 == 0); eap_so3_group_fwd_tp_columns
 * fills the HOST array perm [c*ks]: position j of a row holds column perm[j] = channel * ks + kernel point of the plain matrix. */
int eap_so3_group_fwd_tp_takes(int c, int na, int ks);
int eap_so3_group_fwd_tp_columns(int c, int ks, int32_t *perm);
int eap_so3_inter_group_fwd_tp_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                                   const int32_t *idx, const float *gx, const float *rk, float *out, eap_stream_t stream);
int eap_so3_inter_group_inv_perm2_f32(int b, int o, int p, int nn, int na, int ks, int rcap, float sigma, const float *gy,
                                      const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_pc,
                                      const float *ent_gx2, const float *rk, const uint8_t *order, float *z, eap_stream_t stream);
/* Block -> XCD map of the two-tile kernel: mode 1 = an XCD (one L2) owns whole (channel slice, cloud) pairs, 2 = whole
 * (channel slice, cloud, anchor group) triples; which = 0 forward, 1 backward; mode 0 = query.  Same results either way. */
int eap_so3_group_lists_xcd_map(int which, int mode);

/* so3_inter_group_inv: the feature gradient of the whole inter convolution by re-association,
 *   dF[c,q,a'] = sum_{o,k} W[o,(c,k)] Z[o,k,q,a'],
 *   Z[o,k,q,a'] = sum_{(p,n): idx[p,n]=q} dY[o,p,a] w(p,a,k,n),  a = perm_n^-1(a')
 * i.e. the forward grouping applied to dY over INVERSE neighbour lists (autograd of
 * so3conv/functional.py:L1221-1261 + modules.py:L48-55 without ever forming dX = W^T dY).
 * gy [b,o,p,na]; the referenced support rows of each batch item are compacted to `rcap` slots:
 * rows [b,rcap] (support index or -1), off/cnt [b,rcap] (range of the row's entries in the
 * per-item sorted entry list), ent_p [b,p*nn] (query point of each entry), ent_gx float4
 * [b,p*nn] (its so3_prep word), rk [na,ks,3], multinv [na,na] (multinv[r][a'] = a, NULL = no
 * permutation), anchors [na,3,3] (the rotations the table was built from; needed with multinv: an
 * entry's offset vector is rotated by A_r once so that its weights are those of the accumulator's own
 * anchor)  ->  z [b,o,ks,rcap,na].  The caller finishes with eap_gemm_dma_f32. */
int eap_so3_inter_group_inv_f32(int b, int o, int p, int nn, int na, int ks, int rcap, float sigma,
                                const float *gy, const int32_t *rows, const int32_t *off,
                                const int32_t *cnt, const int32_t *ent_p, const float *ent_gx,
                                const float *rk, const uint8_t *multinv, const float *anchors,
                                int identity_anchor, float *z, eap_stream_t stream);
/* The same for clouds with anchor permutations, given the coset tables of `multinv` (order uint8 [64]: anchor at every position of
 * a coset-major ordering, blocks of 4 = left cosets of a Klein four-group of the anchor group; code uint8 [na,16], 4-byte
 * aligned: per permutation row and block `sigma | x << 4`): the LDS operand is kept in that order, so the four permuted anchors
 * of a block are one 16-byte read of block sigma + an index-XOR shuffle.  Same results to the last bit of the products summed
 * (the anchors are visited in another order only across waves). */
/* dst [rows, na] = src with its anchor axis re-ordered, dst[., i] = src[., order[i]]: for na = 4 mod 8 the entry below takes gy's
 * rows by DMA and expects them in coset-major order already. */
int eap_anchor_reorder_f32(int64_t rows, int na, const float *src, const uint8_t *order, float *dst, eap_stream_t stream);
int eap_so3_inter_group_inv_coset_f32(int b, int o, int p, int nn, int na, int ks, int rcap, float sigma,
                                      const float *gy, const int32_t *rows, const int32_t *off, const int32_t *cnt,
                                      const int32_t *ent_p, const float *ent_gx, const float *rk, const uint8_t *multinv,
                                      const float *anchors, int identity_anchor, const uint8_t *coset_order,
                                      const uint8_t *coset_code, float *z, eap_stream_t stream);

/* so3_inter_group_inv without anchor permutation, gy stored with a row pitch: gy [b,o,p,gy_pitch], gy_pitch a
 * multiple of 4 and >= na (64 makes every 60-anchor row start on a 256-byte boundary). */
int eap_so3_inter_group_inv_pitch_f32(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap,
                                      float sigma, const float *gy, const int32_t *rows, const int32_t *off,
                                      const int32_t *cnt, const int32_t *ent_p, const float *ent_gx,
                                      const float *rk, float *z, eap_stream_t stream);

/* Inverse neighbour lists for so3_inter_group_inv, built on the device (the autograd transpose of the
 * gather at so3conv/functional.py:L1221-1252 needs, per referenced support row, the (point, slot) pairs
 * that reference it).  idx int32 [b,p,nn] with values in [0,n) (other values are ignored), n <= 16384.
 * eap_inv_lists_rows: counts [b,n] (scratch) -> rows [b,n] (referenced support rows, longest list first,
 *   ties by index; -1 past the end), cnt [b,n], off [b,n] (start of each row's entries in the per-cloud
 *   entry list), n_rows [b].
 * eap_inv_lists_fill: for the first `rcap` rows of every cloud (rows / off as written above, leading
 *   dimension n): ent_p int32 [b,p*nn] (query point of each entry, entries of a row in (p,slot) order),
 *   ent_gx float4 [b,p*nn] (its so3_prep word).  p*nn must be a multiple of 4. */
int eap_inv_lists_rows(int b, int p, int n, int nn, const int32_t *idx, int32_t *counts, int32_t *rows,
                       int32_t *off, int32_t *cnt, int32_t *n_rows, eap_stream_t stream);
int eap_inv_lists_fill(int b, int p, int n, int nn, int rcap, const int32_t *idx, const float *gx,
                       const int32_t *rows, const int32_t *off, int32_t *ent_p, float *ent_gx,
                       eap_stream_t stream);
/* Referenced rows of a feature tensor: dst [b,c,rcap,na] = src [b,c,n,na][:, :, rows[b,r], :] (zeros where
 * rows < 0), and the transpose dst [b,c,n,na] = 0; dst[:, :, rows[b,r], :] = src [b,c,rcap,na].
 * rows int32 with leading dimension rows_ld; na a multiple of 4. */
int eap_rows_gather_f32(int b, int c, int n, int na, int rcap, int rows_ld, const int32_t *rows,
                        const float *src, float *dst, eap_stream_t stream);
int eap_rows_scatter_f32(int b, int c, int n, int na, int rcap, int rows_ld, const int32_t *rows,
                         const float *src, float *dst, eap_stream_t stream);

/* ---- the fused inter conv re-associated over its referenced rows as a dense product (csrc/so3_dense.hip) ----
 * Replaces, for clouds WITHOUT pose rotations whose neighbour lists name few support rows, the grouping einsum +
 * contraction of so3conv/functional.py:L1112-1261 + so3conv/modules.py:L48-55 (forward) and their autograd transpose
 * (backward).  With rows[b, 0..rp) the referenced support rows of cloud b (eap_inv_lists_rows; rp a multiple of 4, empty
 * slots = -1) and m[p,r] = 1 iff row r is in idx[b,p,:]:
 *     Wd[p,(k,r),a] = m[p,r] relu(1 - |x_row(r) - x_p - rk[a,k]|^2 / sigma)
 *     dir 0 (backward)  Z [b,o,k,a,r]   = sum_p      dY[b,o,p,a]     Wd[p,(k,r),a]
 *     dir 1 (forward)   Yt[b,a,o,p]     = sum_(k,r)  G [b,o,(k,r),a] Wd[p,(k,r),a]      (G = W F over the referenced rows)
 * on the fp16 matrix cores with two planes per operand and fp32 accumulation (the arithmetic of eap_gemm_f16x2_f32).  Inside, the
 * (k, r) axis runs in the DENSE INDEX order d = (r / 16) 16 ks + 16 k + r % 16; a cloud with n_rows[b] < rp referenced rows uses a
 * prefix of it, and with n_rows given the products stop there (dir 0 zeroes the slots past ceil16(n_rows[b]) in Z).
 *   eap_so3_dense_form        how the weights are evaluated: 1 (default) from the squared distance, 0 from the expanded square
 *                             (fewer instructions, ~3 x the rounding error); tables and product under the same setting; -> old setting
 *   eap_so3_dense_supported   p % 32 == 0, na % 4 == 0, na <= 64, ks % 2 == 0, rp % 16 == 0, rp <= 512, o % 128 == 0 (256-row blocks when o % 256 == 0)
 *   eap_so3_dense_member      slot_of int32 [b,n] (scratch), memb uint32 [b,p,16] (bit r of point p = m[p,r]; rp <= 512),
 *                             flags int32 [b]: 1 = a list names a row twice (padded short lists, grouping_cuda_kernel.cu:L98-107:
 *                             not representable by a 0/1 mask), 2 = a list names a row outside rows[:, :rp] -- such clouds
 *                             must take the list kernels
 *   eap_so3_dense_mask_words  uint64 words of the mask table of one direction; eap_so3_dense_masks fills it:
 *                             [b][64-column wave tile][k-step of 32][64 lanes] one bit per weight a lane of the product kernel generates
 *   eap_so3_dense_tables_f32  centre [b,4] (centroid of the support points), pt float4 [b, ceil32(p)] and
 *                             kr float4 [b, na, ceil32(ks rp)]: the two sides of the weight (centred coordinates), evaluated in float64;
 *                             row_rot (may be null) float [b, rows_ld, 9]: a rotation per row slot applied to the kernel offsets --
 *                             the query points of ONE rigid part of a posed cloud (R_rel^T of so3conv/functional.py:L1112-1160)
 *   eap_so3_dense_split_f32   src [b,m,l,na] -> scale [2][b,na,m] (power of two per row; the second copy as [b,m,na]) and the two fp16 planes of the scaled
 *                             rows in the product kernel's fragment order: 4 b na m ceil32(l) bytes
 *                             (rowmax uint32 [b,m,na], may be null: the rows' largest magnitudes as float bit patterns when the producer
 *                             of src already has them, eap_bn_act_bwd_apply_rowmax_f32; seg > 0: a row's l elements come in segments of seg elements seg_pitch floats apart -- G as a GEMM with
 *                             padded columns leaves it; colmap int32 [b,l], may be null: element i of a row is element colmap[b,i] of a source row of
 *                             seg_pitch floats, negative = zero -- the columns of dY that are the query points of one rigid part)
 *   eap_so3_dense_product_f32 the product (planes / scale of dY [b,o,p,na] for dir 0, of G [b,o,ks rp,na] for dir 1); dir 0 writes
 *                             Z with ldz >= na rp floats between its (o, k) rows (padding for the GEMMs that follow, not written)
 *   eap_so3_dense_untranspose_f32   Yt [b,na,o,p] -> Y [b,o,p,na]; psum / psq (may be null) float [o, b ceil(p/64)]: partial sums of
 *                             y - y[0,o,0,0] and of its square, the moments of the BatchNorm that follows (eap_bn_stats_f32's pivot)
 *   eap_so3_dense_untranspose_map_f32   the same into a Y [b,o,p_dst,na] that several launches fill: column pp of cloud b is point
 *                             map[b,pp] (int32 [b,p]; negative: padding, dropped) -- the rigid parts of posed clouds, one launch each
 *   eap_so3_dense_untranspose_map_stats_f32   ... with the moments of eap_so3_dense_untranspose_f32 (columns whose map entry is >= 0);
 *                             pivot_pos int32 [1] on the device: the column of cloud 0 that is point 0 (the pivot stays Y[0,o,0,0])
 * Occupancy-sorted query points (round 6).  The reference's ball query keeps the first nsample hits in index order
 * (grouping_cuda_kernel.cu:L68-113), so a point's list names rows of only 0.45-0.7 of a cloud's 16-row groups; with a cloud's query
 * points sorted by WHICH groups they touch the 0/1 mask of the product is block-sparse and whole k-steps drop out:
 *   eap_so3_dense_point_keys   memb [b,p,16] -> keys int32 [b,p], bit g = the list of p names a row slot of 16 g .. 16 g + 15; the host sorts
 *                             a cloud's points by it and builds memb / pt / the masks in that order (the order is a column map for
 *                             eap_so3_dense_split_f32 and eap_so3_dense_untranspose_map*_f32)
 *   eap_so3_dense_steps_words / eap_so3_dense_steps   from the mask table of a direction: int32 [b][column blocks of 256][k-steps + 1] =
 *                             count (>= 1) and the ascending k-steps in which the block generates a weight that is not masked out
 *                             (skip = 0: every k-step; dir 1 with n_rows: of the cloud's prefix)
 *   eap_so3_dense_product_steps_f32   eap_so3_dense_product_f32 whose column blocks run their listed k-steps only (steps may be null; lists
 *                             longer than 512 k-steps are ignored: every k-step runs).  Skipped k-steps would have added exact zeros:
 *                             bit-equal to running them all in the same point order.  so3conv/functional.py:L1221-1261 */
int eap_so3_dense_supported(int p, int na, int ks, int rp, int o);
int eap_so3_dense_form(int form);
int eap_so3_dense_block_rows(int rows);      /* 0 (default): 256-row blocks where o % 256 == 0; 128: always 128-row blocks; -> old setting */
int eap_so3_dense_member(int b, int p, int n, int nn, int rp, int rows_ld, const int32_t *idx, const int32_t *rows,
                         const int32_t *n_rows, int32_t *slot_of, uint32_t *memb, int32_t *flags, eap_stream_t stream);
int64_t eap_so3_dense_mask_words(int b, int p, int ks, int rp, int dir);
int eap_so3_dense_masks(int b, int p, int ks, int rp, int dir, const uint32_t *memb, uint64_t *mask, eap_stream_t stream);
int eap_so3_dense_tables_f32(int b, int p, int n, int na, int ks, int rp, int rows_ld, float sigma, const float *q_xyz,
                             const float *s_xyz, const int32_t *rows, const float *rk, const float *row_rot, float *centre, float *pt,
                             float *kr, eap_stream_t stream);
int eap_so3_dense_split_f32(int b, int m, int l, int na, int seg, int64_t seg_pitch, int mapped, const int32_t *n_rows, const uint32_t *rowmax,
                            const int32_t *colmap, const float *src, float *scale, void *planes, eap_stream_t stream);
int eap_so3_dense_product_f32(int dir, int b, int o, int p, int na, int ks, int rp, int64_t ldz, float sigma, const int32_t *n_rows, const void *planes,
                              const float *scale,
                              const float *pt, const float *kr, const uint64_t *mask, float *out, eap_stream_t stream);
int eap_so3_dense_untranspose_f32(int b, int o, int p, int na, const float *yt, float *y, float *psum, float *psq, eap_stream_t stream);
int eap_so3_dense_untranspose_map_f32(int b, int o, int p, int na, int p_dst, const int32_t *map, const float *yt, float *y,
                                      eap_stream_t stream);
int eap_so3_dense_untranspose_map_stats_f32(int b, int o, int p, int na, int p_dst, const int32_t *map, const int32_t *pivot_pos, const float *yt,
                                            float *y, float *psum, float *psq, eap_stream_t stream);
/* conv + training-mode BatchNorm + leaky_relu as ONE node (round 6; `x = conv(x); feat = relu(norm(x.feats))`,
 * SPConvNets/utils/base_so3poseconv.py:L205-222, and its autograd): the forward's re-ordering pass applies the normalisation on the way out
 * (eap_so3_dense_untranspose_bnact_f32; the moments come from eap_bn_stats_f32 over Yt viewed as [b na, o, p]); the backward never writes
 * the gradient behind the BatchNorm: eap_bn_act_bwd_reduce_fromy_f32 (csrc/bn_act.hip) reduces sum(g), sum(g xhat) and leaves per-row bounds,
 * eap_so3_dense_split_bn_f32 forms gx = k1 g - k2 - k3 xhat while it splits it into the product's planes.  Both recover the
 * pre-activation from the layer's OUTPUT (leaky_relu with a positive slope is invertible): the conv output is not kept. */
int eap_so3_dense_split_bn_f32(int b, int m, int l, int l_src, int na, const uint32_t *rowbound, const int32_t *colmap, const float *grad,
                               const float *act, const float *coef, float slope, float *scale, void *planes, eap_stream_t stream);
int eap_so3_dense_untranspose_bnact_f32(int b, int o, int p, int na, int p_dst, const int32_t *map, const float *yt, const float *bn_scale,
                                        const float *bn_shift, float slope, float *y, eap_stream_t stream);
/* the forward's stored operand G[o,(k,r),a] = sum_c W[o,c,k] F[c,r,a] made and written as the product's planes by one kernel (round 6; before:
 * a GEMM that wrote G in fp32 and eap_so3_dense_split_f32 that read it back): W3 float [o ks][c], Ft float [b][na][rp][c] (referenced feature
 * rows, empty slots zero), bound float [b][na][o] >= max |G| of the output row, n_rows [b] (may be null) -> scale [b][na][o], planes.
 * so3conv/functional.py:L1221-1261 + so3conv/modules.py:L48-55 re-associated (csrc/so3_dense.hip header) */
int eap_so3_dense_gplanes_supported(int o, int c, int na, int ks, int rp);
int eap_so3_dense_gplanes_f32(int b, int o, int c, int na, int ks, int rp, const float *W3, const float *Ft, const int32_t *n_rows,
                              const float *bound, float *scale, void *planes, eap_stream_t stream);
int eap_so3_dense_point_keys(int b, int p, const uint32_t *memb, int32_t *keys, eap_stream_t stream);
int64_t eap_so3_dense_steps_words(int b, int p, int ks, int rp, int dir);
int eap_so3_dense_steps(int b, int p, int ks, int rp, int dir, int skip, const int32_t *n_rows, const uint64_t *mask, int32_t *steps,
                        eap_stream_t stream);
int eap_so3_dense_product_steps_f32(int dir, int b, int o, int p, int na, int ks, int rp, int64_t ldz, float sigma, const int32_t *n_rows,
                                    const void *planes, const float *scale, const float *pt, const float *kr, const uint64_t *mask,
                                    const int32_t *steps, float *out, eap_stream_t stream);

/* ---- SO(3) intra convolution -------------------------------------------------------------- */

/* so3_intra_group_fwd: intra_so3conv_grouping, so3conv/functional.py:L2553-2602.
 * feats [b,c,p,na], intra_idx int32 [na,t] -> out [b,c,t,p,na] = feats[b,c,p,intra_idx[a,t]]. */
int eap_so3_intra_group_fwd_f32(int b, int c, int p, int na, int t, const float *feats,
                                const int32_t *intra_idx, float *out, eap_stream_t stream);
/* transpose: gout [b,c,t,p,na] -> gfeats [b,c,p,na] (deterministic, inverse index). */
int eap_so3_intra_group_bwd_f32(int b, int c, int p, int na, int t, const float *gout,
                                const int32_t *intra_idx, float *gfeats, eap_stream_t stream);

/* ---- dense contraction (BasicSO3Conv.forward, so3conv/modules.py:L48-55) ------------------- */

/* Batched fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32), row-major:
 *   C_z[M,N] = op(A_z)[M,K] * op(B_z)[K,N]       z = 0..batch-1
 * transA = 0: A_z is [M,K] with leading dimension lda; 1: A_z is stored [K,M].  Same for B.
 * stride* are element offsets between consecutive batch items (0 = shared operand).
 * BasicSO3Conv:  C = y [b][O, P*A], A = W [O, C*K] (strideA 0), B = x [b][C*K, P*A]. */
int eap_gemm_f32(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                 int64_t strideA, const float *B, int64_t ldb, int64_t strideB, float *C,
                 int64_t ldc, int64_t strideC, int batch, eap_stream_t stream);

/* Split-K / batch-reduced GEMM for the weight gradient:
 *   C[M,N] = sum_z sum_k op(A_z)[M,k] * op(B_z)[k,N]
 * `workspace` holds batch*splits partial [M,N] slabs (eap_gemm_f32_reduce_workspace floats);
 * the reduction over slabs is a second deterministic kernel. */
int64_t eap_gemm_f32_reduce_workspace(int M, int N, int K, int batch);
int eap_gemm_f32_reduce(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                        int64_t strideA, const float *B, int64_t ldb, int64_t strideB, float *C,
                        int64_t ldc, int batch, float *workspace, eap_stream_t stream);

/* C_z[M,N] = A[M,K] * B_z[N,K]^T (the forward contraction of BasicSO3Conv.forward, vgtk/vgtk/so3conv/modules.py:L48-55,
 * with the grouped tensor kept transposed: both operands k-contiguous, A shared by the batch) with fp32 operands, fp32
 * accumulation and fp32-accurate products on the bf16 matrix cores: every operand value is split exactly into three bf16
 * values on the fly and each product taken as the six largest of the nine partial products (the dropped ones are below
 * the rounding of one fp32 product) -- csrc/gemm_bf16x3.hip.  Needs M >= 128, N >= 256, M and N multiples of 128, K % 16 == 0, lda / ldb /
 * strideB multiples of 4, 16-byte aligned bases: eap_gemm_bf16x3_f32_supported tells (1 / 0). */
int eap_gemm_bf16x3_f32_supported(int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb,
                                  int64_t strideB);
int eap_gemm_bf16x3_f32(int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t strideB,
                        float *C, int64_t ldc, int64_t strideC, int batch, eap_stream_t stream);
/* 1 (default): for K >= 1024 the operand shared by the batch (the weights) is split once per call into a stream-ordered
 * scratch buffer (hipMallocAsync on the launch stream, 8 bytes per element, released after the product) instead of once per
 * k-tile by every workgroup; 2: for every shape; 0: never (split inside the k-loop like the other operand).  Bit-identical
 * results.  Returns the previous setting; any other argument only queries. */
int eap_gemm_bf16x3_presplit(int on);
/* the same product with B_z row-major [K, N] ("NN": W [O, C] times x_z [C, P*A], the pointwise contraction behind every
 * 1 x 1 conv of the blocks and heads -- torch.matmul / nn.Conv2d(1) in SPConvNets/utils/base_so3conv.py) */
int eap_gemm_bf16x3_nn_f32_supported(int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb,
                                     int64_t strideB);
int eap_gemm_bf16x3_nn_f32(int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t strideB,
                           float *C, int64_t ldc, int64_t strideC, int batch, eap_stream_t stream);
/* The two products above with a per-row epilogue: C = leaky_relu(scale[row] * (A B) + shift[row], slope) (+ residual [batch][M][ldc],
 * may be null): an inference-mode nn.BatchNorm2d + activation (+ the separable block's skip sum,
 * SPConvNets/utils/base_so3poseconv.py:L214-221, L319-328) folded into the contraction.  transB = 1: B k-contiguous (as
 * eap_gemm_bf16x3_f32), 0: B row-major (as eap_gemm_bf16x3_nn_f32); same operand requirements. */
int eap_gemm_bf16x3_ep_f32(int transB, int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t strideB,
                           float *C, int64_t ldc, int64_t strideC, int batch, const float *scale, const float *shift, float slope,
                           const float *residual, int64_t strideRes, eap_stream_t stream);
/* C[M,N] = sum_z A_z[M,K] B_z[N,K]^T on the split kernel (the weight gradient of the pointwise contraction, dW = sum over the
 * clouds of dY_z x_z^T): every (item, k-slab) pair writes its partial into `workspace` (eap_gemm_bf16x3_reduce_workspace
 * floats, 16-byte aligned), a second kernel sums them in a fixed order. */
int eap_gemm_bf16x3_reduce_f32_supported(int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B,
                                         int64_t ldb, int64_t strideB, int64_t ldc);
int64_t eap_gemm_bf16x3_reduce_workspace(int M, int N, int K, int batch);
int eap_gemm_bf16x3_reduce_f32(int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B, int64_t ldb,
                               int64_t strideB, float *C, int64_t ldc, int batch, float *workspace, eap_stream_t stream);
/* eap_so3_intra_conv_f32 on the split kernel (nt = 12; o, p*na multiples of 128; c*nt a multiple of 16) */
int eap_so3_intra_conv_bf16x3_f32_supported(int b, int o, int c, int p, int na, int nt);
int eap_so3_intra_conv_bf16x3_f32(int b, int o, int c, int p, int na, int nt, const float *W, const float *feats,
                                  const int32_t *intra_idx, float *out, eap_stream_t stream);
/* The same contractions (torch.matmul in BasicSO3Conv.forward, vgtk/vgtk/so3conv/modules.py:L48-55; the 1 x 1 convs; the intra
 * conv, so3conv/functional.py:L2553-2602) with TWO fp16 planes per operand instead of three bf16 planes: three matrix
 * instructions per k-tile instead of six (round 4).  Every ROW of A and every COLUMN of B_z is first multiplied by a power of
 * two that puts its largest magnitude (or a bound on it) at 2^14..2^15 -- the scales come off again per output row and column
 * in the epilogue --, then split x = h + l, h = fp16(x), l = fp16(x - h), round to nearest: |x - h - l| <= 2^-23 |x| (rms
 * 2^-25 |x|) for elements down to 2^-17 of that magnitude, <= 2^-40 of it below; the products h h' + h l' + l h' are exact in
 * the fp32 accumulator, l l' (<= 2^-22 |x x'|) is dropped.  tests/test_gpu_split_planes.py bounds the error against fp64 by that
 * of the fp32-MFMA kernel on the same operands, per output element.  The magnitudes are device words holding the bit pattern
 * of a non-negative float (unsigned maxima: order-independent), so nothing waits for the host:
 *   eap_absmax_rows_f32        x [batch][rows][cols] (row pitch ld, item stride `stride`; cols / ld / stride multiples of 4, 16-byte
 *                              aligned) -> out [batch][rows]: largest magnitude of every row
 *   eap_absmax_colgroups_f32   the same tensor -> out [batch][cols / grp]: largest magnitude over the rows and over each group of grp
 *                              consecutive columns (grp a multiple of 4 dividing cols)
 *   eap_so3_grouped_bound_f32  a bound on the inter conv's grouped tensor per point, without a pass over it: X[c,k,p,a] is a sum over
 *                              the nn neighbours of a feature times a weight in [0, 1] (so3conv/functional.py:L1112-1261), so
 *                              |X[.,.,p,.]| <= sum_n point_max[idx[p,n]] with point_max [b][n_sup] = eap_absmax_colgroups_f32 of
 *                              feats [b][c][n_sup*na] in groups of na; idx int32 [b,p,nn] (entries >= n_sup: no neighbour) -> out [b][p]
 *   eap_gemm_f16x2_f32         C_z = A B_z, trans_b = 1: B_z [N,K] k-contiguous (operands as eap_gemm_bf16x3_f32_supported says), 0: B_z
 *                              [K,N] row-major (eap_gemm_bf16x3_nn_f32_supported); abs_a [M] = eap_absmax_rows_f32 of A; abs_b
 *                              [batch][N / grp_b]: float(word) * mult_b >= the largest magnitude in those grp_b columns of C's operand
 *                              B_z; scale / shift / slope / residual: the row epilogue of eap_gemm_bf16x3_ep_f32, or scale = shift = NULL
 *   eap_so3_intra_conv_f16x2_f32   eap_so3_intra_conv_bf16x3_f32 likewise: abs_w [o] of W's rows, abs_f [b][p] = per-point maxima of feats
 *                              (eap_absmax_colgroups_f32 in groups of na: a column gathers 12 anchors of its point) */
int eap_absmax_rows_f32(const float *x, int batch, int rows, int cols, int64_t ld, int64_t stride, int32_t *out_bits, eap_stream_t stream);
int eap_absmax_colgroups_f32(const float *x, int batch, int rows, int cols, int64_t ld, int64_t stride, int grp, int32_t *out_bits,
                             eap_stream_t stream);
int eap_so3_grouped_bound_f32(int b, int p, int nn, int n_sup, const int32_t *point_max_bits, const int32_t *idx, int32_t *out_bits,
                              eap_stream_t stream);
int eap_gemm_f16x2_f32(int trans_b, int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t strideB,
                       float *C, int64_t ldc, int64_t strideC, int batch, const int32_t *abs_a, const int32_t *abs_b, int grp_b, float mult_b,
                       const float *scale, const float *shift, float slope, const float *residual, int64_t strideRes,
                       eap_stream_t stream);
int eap_so3_intra_conv_f16x2_f32(int b, int o, int c, int p, int na, int nt, const float *W, const float *feats,
                                 const int32_t *intra_idx, float *out, const int32_t *abs_w, const int32_t *abs_f, eap_stream_t stream);


/* The same GEMMs with both operands fed by global -> LDS DMA through a three-stage ring (csrc/gemm_dma_f32.hip);
 * conventions of eap_gemm_f32 / eap_gemm_f32_reduce.  Needs K % 16 == 0, leading dimensions and batch strides
 * multiples of 4, 16-byte aligned bases, and a multiple of 4 rows for an operand whose rows are contiguous
 * (transA = 1: M % 4 == 0; transB = 0: N % 4 == 0); eap_gemm_dma_f32_supported tells (1 / 0). */
int eap_gemm_dma_f32_supported(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                               int64_t strideA, const float *B, int64_t ldb, int64_t strideB);
int eap_gemm_dma_f32(int transA, int transB, int M, int N, int K, const float *A, int64_t lda, int64_t strideA,
                     const float *B, int64_t ldb, int64_t strideB, float *C, int64_t ldc, int64_t strideC,
                     int batch, eap_stream_t stream);
int64_t eap_gemm_dma_f32_reduce_workspace(int M, int N, int K, int batch);
int eap_gemm_dma_f32_reduce(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                            int64_t strideA, const float *B, int64_t ldb, int64_t strideB, float *C, int64_t ldc,
                            int batch, float *workspace, eap_stream_t stream);
/* C[M,N] = sum_z A_z[M,K] B_z[N,K]^T for a SMALL output and a long contraction (M <= 64, N <= 32, K >= 4096; both operands
 * k-contiguous, 16-byte aligned, pitches and strides multiples of 4 floats): the first layer's weight gradient
 * dW[64, 24] = sum_b dY_b X_b^T (textbook backward of so3conv/modules.py:L48-55 at C = 1).  A streaming reduction -- 16-byte
 * loads straight into the MFMA operand registers, per-wave partial sums added in a fixed order (bit-reproducible); the tiled
 * kernels pad this product to 5 % useful work.  workspace: eap_gemm_skinny_reduce_workspace floats. */
int eap_gemm_skinny_reduce_f32_supported(int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B,
                                         int64_t ldb, int64_t strideB);
int64_t eap_gemm_skinny_reduce_workspace(int M, int N, int K, int batch);
int eap_gemm_skinny_reduce_f32(int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B, int64_t ldb,
                               int64_t strideB, float *C, int64_t ldc, int batch, float *workspace, eap_stream_t stream);

/* ---- blocked intermediate: the forward's grouped tensor X with coalesced row-end stores --------
 * X_blocked[b][p][a/4][c][k][4] holds the same numbers as X[b][c][k][p][a].  The grouping kernels
 * write 384 contiguous bytes per (channel, anchor quad) instead of 16-byte pieces 983 KB apart, and
 * the contraction GEMMs read it as a B operand "blocked by 4" along P*A.  Internal to the fused
 * conv (vgtk.so3conv.functional._InterConv); the reference-layout entries above stay as they are. */
int eap_so3_inter_group_fwd_can_block(int c, int n, int na, int ks, int has_mult, int has_flag);
int eap_so3_inter_group_fwd_xb_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma,
                                   const float *feats, const int32_t *idx, const float *gx,
                                   const float *rk, const uint8_t *mult, const int32_t *nonident,
                                   float *out, eap_stream_t stream);
/* same, output transposed: out[b][p*na + a][c*ks + k] (a plain row-major [P*A, C*K] matrix) */
int eap_so3_inter_group_fwd_t_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma,
                                  const float *feats, const int32_t *idx, const float *gx,
                                  const float *rk, const uint8_t *mult, const int32_t *nonident,
                                  float *out, eap_stream_t stream);
/* eap_gemm_f32 / eap_gemm_f32_reduce with B blocked by 4: element (row r of b_block_rows, position x)
 * at (x >> 2) * b_block_rows * 4 + r * 4 + (x & 3); b_block_rows = K (transB = 0) or N (transB = 1). */
int eap_gemm_f32_xb(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                    int64_t strideA, const float *B, int64_t b_block_rows, int64_t strideB, float *C,
                    int64_t ldc, int64_t strideC, int batch, eap_stream_t stream);
int eap_gemm_f32_reduce_xb(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                           int64_t strideA, const float *B, int64_t b_block_rows, int64_t strideB,
                           float *C, int64_t ldc, int batch, float *workspace, eap_stream_t stream);

/* Intra SO(3) conv, forward, as an implicit GEMM (no [b,c,t,p,na] gathered tensor):
 * out[b,o,p,a] = sum_{c,t} W[o, c*nt + t] * feats[b, c, p, intra_idx[a*nt + t]]
 * (so3conv/functional.py:L2553-2602 intra_so3conv_grouping + so3conv/modules.py:L48-55 BasicSO3Conv).
 * W [o, c*nt], feats [b,c,p,na], intra_idx int32 [na,nt] (na a multiple of 4, <= 64; nt <= 16), out [b,o,p,na]. */
int eap_so3_intra_conv_f32(int b, int o, int c, int p, int na, int nt, const float *W, const float *feats,
                           const int32_t *intra_idx, float *out, eap_stream_t stream);

/* ---- chamfer distance (extensions/chamfer_dist) ------------------------------------------- */

/* chamfer.forward: chamfer_cuda.cpp:L22-25, chamfer.cu:L15-171.
 * xyz1 [b,n,3], xyz2 [b,m,3] -> dist1 [b,n], dist2 [b,m], idx1 int32 [b,n], idx2 int32 [b,m]. */
int eap_chamfer_fwd_f32(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist1,
                        float *dist2, int32_t *idx1, int32_t *idx2, eap_stream_t stream);
/* chamfer.backward: chamfer_cuda.cpp:L27-34, chamfer.cu:L173-231.
 * -> gxyz1 [b,n,3], gxyz2 [b,m,3] (zero-initialised by the call). */
int eap_chamfer_bwd_f32(int b, int n, int m, const float *xyz1, const float *xyz2,
                        const int32_t *idx1, const int32_t *idx2, const float *g1, const float *g2,
                        float *gxyz1, float *gxyz2, eap_stream_t stream);

/* ---- block-layer epilogue: BatchNorm2d (batch statistics) + leaky_relu ------------------------- */
/* SPConvNets/utils/base_so3poseconv.py:L214-221 (`feat = self.norm(x.feats); feat = self.relu(feat)`),
 * SURVEY.md 8(f) row 1.  x, y, gy, gx: [b, c, n] (n = P*A, a multiple of 4), contiguous.
 * Reductions are returned as per-block partials [c][b][eap_bn_act_segments(n)] for the caller to
 * sum in a fixed order (deterministic). */
int eap_bn_act_segments(int64_t n);
/* partial sums of (x - pivot_c) and (x - pivot_c)^2, pivot_c = x[0, c, 0] */
int eap_bn_stats_f32(int b, int c, int64_t n, const float *x, float *psum, float *psq, eap_stream_t stream);
/* y = leaky_relu(x * scale[c] + shift[c], slope) */
int eap_bn_act_fwd_f32(int b, int c, int64_t n, float slope, const float *x, const float *scale,
                       const float *shift, float *y, eap_stream_t stream);
/* y = leaky_relu(x * scale[c] + shift[c], slope) + res: the separable block's skip-add
 * (SPConvNets/utils/base_so3poseconv.py:L319-328) in the same pass; res [b, c, n] */
int eap_bn_act_add_fwd_f32(int b, int c, int64_t n, float slope, const float *x, const float *scale,
                           const float *shift, const float *res, float *y, eap_stream_t stream);
/* partial sums of g and g * xhat,  g = gy * (x*scale+shift > 0 ? 1 : slope),  xhat = (x - mean) * invstd */
int eap_bn_act_bwd_reduce_f32(int b, int c, int64_t n, float slope, const float *gy, const float *x,
                              const float *scale, const float *shift, const float *mean,
                              const float *invstd, float *pg, float *pgx, eap_stream_t stream);
/* gx = scale[c] * g - k2[c] - xhat * k3[c] */
int eap_bn_act_bwd_apply_f32(int b, int c, int64_t n, float slope, const float *gy, const float *x,
                             const float *scale, const float *shift, const float *mean,
                             const float *invstd, const float *k2, const float *k3, float *gx,
                             eap_stream_t stream);
/* eap_bn_act_bwd_apply_f32 for rows of [points][na] that ALSO leaves max |gx| per (cloud, channel, anchor) in rowmax uint32 [b,c,na]
 * (float bit patterns): what eap_so3_dense_split_f32 takes instead of its own pass over the gradient. */
int eap_bn_act_bwd_apply_rowmax_f32(int b, int c, int64_t n, int na, float slope, const float *gy, const float *x,
                                    const float *scale, const float *shift, const float *mean, const float *invstd,
                                    const float *k2, const float *k3, float *gx, uint32_t *rowmax, eap_stream_t stream);
/* the reduction pass of the BatchNorm + leaky_relu backward from the layer's OUTPUT y' (see the conv + BatchNorm node above): partials of sum(g),
 * sum(g xhat) [c][b * eap_bn_act_fromy_blocks(n, na)] and the largest |g|, |xhat| per (cloud, channel, anchor) [b,c,na] (float bit patterns) */
int eap_bn_act_fromy_blocks(int64_t n, int na);
int eap_bn_act_bwd_reduce_fromy_f32(int b, int c, int64_t n, int na, float slope, const float *gy, const float *y, const float *beta,
                                    const float *inv_gamma, float *pg, float *pgx, uint32_t *gmax, uint32_t *xmax, eap_stream_t stream);

/* Per-cloud statistics over a point subset: the pose heads run their unary stacks once per cloud on the cloud's member
 * points (`for i_bz in range(bz): ...` SPConvNets/models/..pn_38_multi_stage.py:L706-830, the head's BatchNorm2d layers
 * SPConvNets/utils/base_so3conv.py: SO3OutBlockRTWithMaskSep), i.e. BatchNorm statistics per (cloud, channel) over the
 * members.  Batched form: a row is [points][na], mask [b, points] holds 0 / 1; scale, shift, mean, invstd, k2, k3 are
 * [b, c]; every point is normalised, the statistics and the backward's two correction terms see members only. */
/* partial sums [c][b][segments] of mask * (x - pivot_c) and mask * (x - pivot_c)^2, pivot_c = x[0, c, 0] */
int eap_bn_stats_masked_f32(int b, int c, int64_t n, int na, const float *x, const float *mask, float *psum, float *psq,
                            eap_stream_t stream);
/* y = leaky_relu(x * scale[b,c] + shift[b,c], slope) */
int eap_bn_act_cloud_fwd_f32(int b, int c, int64_t n, float slope, const float *x, const float *scale, const float *shift,
                             float *y, eap_stream_t stream);
/* partial sums of g and g * xhat over ALL points of the row, statistics per (cloud, channel) */
int eap_bn_act_cloud_bwd_reduce_f32(int b, int c, int64_t n, float slope, const float *gy, const float *x, const float *scale,
                                    const float *shift, const float *mean, const float *invstd, float *pg, float *pgx,
                                    eap_stream_t stream);
/* gx = scale[b,c] * g - mask * (k2[b,c] + xhat * k3[b,c]);  mask may be null (all ones) */
int eap_bn_act_cloud_bwd_apply_f32(int b, int c, int64_t n, int na, float slope, const float *gy, const float *x,
                                   const float *scale, const float *shift, const float *mean, const float *invstd,
                                   const float *k2, const float *k3, const float *mask, float *gx, eap_stream_t stream);

/* ---- pointwise contraction with 1-4 output channels (csrc/narrow_contract.hip) --------------------------------- */
/* y[b,o,n] = sum_c W[o,c] x[b,c,n], o <= 4, c <= 2048, n a multiple of 4: the last layer of the pose head's dense translation
 * branch (nn.Conv2d(c, 3 * num_heads, 1), SPConvNets/models/model_utils.py:L537-552) and the attention logit of
 * InvPPOutBlockOurs (nn.Conv2d(c, 1, 1), SPConvNets/utils/base_so3conv.py:L905-912) as streaming passes over x. */
int eap_narrow_contract_supported(int b, int o, int c, int64_t n);
int eap_narrow_contract_fwd_f32(int b, int o, int c, int64_t n, const float *W, const float *x, float *y, eap_stream_t stream);
/* dx[b,c,n] = sum_o W[o,c] g[b,o,n] */
int eap_narrow_contract_dx_f32(int b, int o, int c, int64_t n, const float *W, const float *g, float *dx, eap_stream_t stream);
/* dW[o,c] = sum_{b,n} g[b,o,n] x[b,c,n] as eap_narrow_contract_dw_slabs(n) * b partials [slab][b][o][c] for the caller to sum
 * in order (deterministic) */
int eap_narrow_contract_dw_slabs(int64_t n);
int eap_narrow_contract_dw_f32(int b, int o, int c, int64_t n, const float *g, const float *x, float *partial, eap_stream_t stream);

/* ---- heads on the backbone's [b,c,n,na] feature map ------------------------ */
/* Anchor attention pooling, InvPPOutBlockOurs.forward (SPConvNets/utils/base_so3conv.py:L905-912):
 * conf[b,n,a] = softmax_a(logits[b,n,a] * temperature), out[b,c,n] = sum_a x[b,c,n,a] conf[b,n,a].  na % 4 == 0, <= 64.
 * Backward: dx [b,c,n,na], dlogits [b,n,na] from g [b,c,n]. */
int eap_anchor_attn_pool_fwd_f32(int b, int c, int n, int na, float temperature, const float *x, const float *logits,
                                 float *out, float *conf, eap_stream_t stream);
int eap_anchor_attn_pool_bwd_f32(int b, int c, int n, int na, float temperature, const float *x, const float *logits,
                                 const float *g, float *dx, float *dlogits, eap_stream_t stream);
/* Masked point averages of every slot in one pass (SO3OutBlockRTWithMaskSep, SPConvNets/models/model_utils.py:L470-484,
 * L549-552): out[b,s,c,a] = inv_den[b,s] * sum_n mask[b,s,n] x[b,c,n,a];  ns <= 8.  Backward w.r.t. x. */
int eap_slot_masked_mean_fwd_f32(int b, int ns, int c, int n, int na, const float *x, const float *mask,
                                 const float *inv_den, float *out, eap_stream_t stream);
int eap_slot_masked_mean_bwd_f32(int b, int ns, int c, int n, int na, const float *g, const float *mask,
                                 const float *inv_den, float *dx, eap_stream_t stream);
/* The pose heads' 'max' pooling on a point subset, `(x * mask).max(2)` (SPConvNets/models/model_utils.py:L79-80, L470-484):
 * out[b,c,a] = max_p mask[b,p] * x[b,c,p,a], arg[b,c,a] = the lowest point index attaining it (int32); one pass over x.
 * Backward: dx[b,c,p,a] = (p == arg[b,c,a]) * mask[b,p] * g[b,c,a], written for every point. */
int eap_masked_max_fwd_f32(int b, int c, int n, int na, const float *x, const float *mask, float *out, int32_t *arg,
                           eap_stream_t stream);
int eap_masked_max_bwd_f32(int b, int c, int n, int na, const float *g, const int32_t *arg, const float *mask, float *dx,
                           eap_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EAP_HIP_H */
