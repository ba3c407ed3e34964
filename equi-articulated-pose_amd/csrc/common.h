// csrc/common.h -- shared helpers for the gfx950 kernels behind include/eap_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/eap_hip.h"

namespace eap {

void set_error(const char *msg);
// name (with template arguments, as rocprofv3 prints it) of the dominant kernel the calling thread launched last -- set by
// the launchers of the hot kernels, read by bench.py through eap_last_kernel() to attribute time per KERNEL, not per entry
void set_kernel(const char *name);

static inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        char buf[256];
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
        set_error(buf);
        return (int)e;
    }
    return 0;
}

static inline int bad_arg(const char *what) {
    set_error(what);
    return (int)hipErrorInvalidValue;
}

static inline int hip_fail(hipError_t e, const char *what) {
    if (e == hipSuccess) return 0;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    set_error(buf);
    return (int)e;
}

static inline hipStream_t S(eap_stream_t s) { return (hipStream_t)s; }

static inline unsigned cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace eap

#ifdef __HIPCC__
// Consecutive points must land on the SAME XCD: each (channel, k) output row of a point is only
// 4*na bytes, so neighbouring points share cache lines; with the default round-robin dispatch
// (block b -> XCD b % 8) they would sit half-written in eight different L2s and reach HBM as
// partial lines (measured: X written at ~1 TB/s).  Remap so that XCD x gets a contiguous range
// of points (bijective for any P).
__device__ __forceinline__ int xcd_point(int bx, int p) {
    const int q = p >> 3, r = p & 7, xcd = bx & 7, j = bx >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// csrc/so3_inter_lists.hip: the two-workgroups-per-CU grouping kernel (no anchor permutation)
namespace eap {
bool group_lists_supported(int na, int ks);
int group_lists_fwd(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                    const int32_t *idx, const float *gx, const float *rk, const int32_t *nonident, int blocked, float *out,
                    hipStream_t s);
int group_lists_inv(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap, float sigma, const float *gy,
                    const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p,
                    const float *ent_gx, const float *rk, float *z, hipStream_t s);
// csrc/so3_inter_lists2.hip: the same kernel with two channel tiles per wave sharing one weight evaluation (64-channel
// blocks); group_lists2_preferred: the channel count fills the wider blocks and the variant has not been switched off
// with eap_so3_group_lists_tiles(1)
bool group_lists2_preferred(int c, int na, int ks, int layout);
#ifdef EAP_EXPERIMENTS   // tools/experiments/kernels/so3_inter_lists3.hip (`make EXPERIMENTS=1`): the same on the bf16 matrix cores (3 x bf16 split operands)
bool group_lists3_preferred(int c, int na, int ks, int layout);
// tools/experiments/kernels/so3_inter_lists_h2.hip: the same on the fp16 matrix cores (two fp16 planes per operand), eap_so3_group_lists_tiles(4)
bool group_listsh_preferred(int c, int na, int ks, int layout);
int group_listsh_fwd(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                     const int32_t *idx, const float *gx, const float *rk, const int32_t *nonident, int layout, float *out,
                     hipStream_t s);
int group_listsh_inv(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap, float sigma, const float *gy,
                     const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p,
                     const float *ent_gx, const float *rk, float *z, hipStream_t s);
int group_lists3_fwd(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                     const int32_t *idx, const float *gx, const float *rk, const int32_t *nonident, int layout, float *out,
                     hipStream_t s);
int group_lists3_inv(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap, float sigma, const float *gy,
                     const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p,
                     const float *ent_gx, const float *rk, float *z, hipStream_t s);
#endif
int group_lists2_fwd(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                     const int32_t *idx, const float *gx, const float *rk, const int32_t *nonident, int layout, float *out,
                     hipStream_t s);
int group_lists2_inv(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap, float sigma, const float *gy,
                     const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p,
                     const float *ent_gx, const float *rk, float *z, hipStream_t s);
// csrc/zpconv_rows.hip: native inter zpconv forward near HBM speed (shared neighbour list per point)
bool inter_zpconv_rows_supported(int np, int nq, int na, int ks, int nn, int c);
// (only_flagged != nullptr: clouds whose flag is zero are left untouched)
int inter_zpconv_rows_fwd(int b, int np, int nq, int na, int ks, int nn, int c, const int32_t *idx, const float *w,
                          const float *feats, float *out, const int32_t *only_flagged, hipStream_t s);
// csrc/zpconv_mfma.hip: the same op on the matrix cores for clouds with one neighbour list per point (skip[b] == 0)
#ifdef EAP_EXPERIMENTS   // tools/experiments/kernels/zpconv_mfma2.hip
int zp_fwd_kernel();                                                                   // eap_inter_zpconv_fwd_kernel's setting
bool inter_zpconv_mfma2_supported(int np, int nq, int na, int ks, int nn, int c);     // csrc/zpconv_mfma2.hip

int inter_zpconv_mfma2_fwd(int b, int np, int nq, int na, int ks, int nn, int c, const int32_t *idx0, const float *w,
                           const float *feats, const int32_t *skip, float *out, hipStream_t s);
#endif
bool inter_zpconv_mfma_supported(int np, int nq, int na, int ks, int nn, int c);
int inter_zpconv_mfma_fwd(int b, int np, int nq, int na, int ks, int nn, int c, const int32_t *idx0, const float *w,
                          const float *feats, const int32_t *skip, float *out, hipStream_t s);
// index pattern check shared by the zpconv forward and backward (csrc/zpconv_mfma.hip), the flag-gated scatter backward
// (csrc/zpconv.hip)
// (idx0 == nullptr: only the comparison)
int zpconv_index_check(int b, int np, int per_point, int nn, const int32_t *idx, int32_t *idx0, float *eid, int32_t *flag, hipStream_t s);
// idx0[b,p,:] = the first (a,k) row of every point's 5-D index (1 MB per cloud): what the matrix kernels walk while the full
// comparison above still streams on the side stream
int zpconv_first_rows(int b, int np, int per_point, int nn, const int32_t *idx, int32_t *idx0, hipStream_t s);
// csrc/abi.hip: a side stream per device (fork: it waits for `s` so far; join: `s` waits for it)
int side_fork(hipStream_t s, hipStream_t *side);
int side_join(hipStream_t s);
// Joins on EVERY exit path after a fork: an early error return must not leave the side stream reading buffers the caller frees
// once the entry has failed.  (One side stream and one fork / join event pair per device, shared by every caller: entries that
// fork are meant to be called from ONE host thread per device -- the autograd thread of a process that owns the GPU, as the
// reference's extensions are; two host threads forking on the same device would share the events.)
struct SideJoin {
    hipStream_t s;
    bool armed = true;
    explicit SideJoin(hipStream_t stream) : s(stream) {}
    ~SideJoin() { if (armed) side_join(s); }
    int join() { armed = false; return side_join(s); }
};
int inter_zpconv_bwd_flagged(int b, int np, int nq, int na, int ks, int ann, int c, const int32_t *idx, const float *w,
                             const float *grad, float *gfeats, const int32_t *only_flagged, hipStream_t s);
// csrc/so3_inter_mfma.hip with the clouds already served by group_lists_fwd skipped
int group_fwd_perm_lists(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats, const int32_t *idx,
                         const float *gx, const float *rk, const uint8_t *mult, const int32_t *nonident, int blocked, float *out,
                         hipStream_t s);      // csrc/so3_inter_inv.hip; -1 = shape not taken
int group_fwd_mfma(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                   const int32_t *idx, const float *gx, const float *rk, const uint8_t *mult,
                   const int32_t *nonident, int skip_plain, int blocked, float *out, hipStream_t s);
}
#endif
