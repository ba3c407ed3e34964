// csrc/so3_intra.hip -- intra-SO(3) (group) convolution, grouping stage
// (vgtk/vgtk/so3conv/functional.py:L2553-2602):
//     out[b,c,t,p,a] = feats[b,c,p,intra_idx[a,t]]
// The reference does index_select + permute + contiguous (two materialised copies); here one
// kernel reads each 4*na-byte feature row once (lanes along the anchor dimension) and writes the
// `t` permuted copies as coalesced rows.  The backward is the deterministic transpose: every
// block first inverts the (tiny) index table in LDS, then each lane sums the rows that read it.
#include "common.h"

namespace {

constexpr int ROWS = 4;  // (c,p) rows per block = waves per block

__global__ __launch_bounds__(64 * ROWS) void so3_intra_group_fwd_kernel(
    long long rows, int c, int p, int na, int t, const float *__restrict__ feats,
    const int32_t *__restrict__ intra_idx, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * ROWS + wave;  // over (b,c,p)
    if (row >= rows || lane >= na) return;
    const long long bc = row / p;
    const int pi = (int)(row - bc * p);
    const float *f = feats + row * na;
    float *o = out + (bc * t * p + pi) * na + lane;
    for (int ti = 0; ti < t; ++ti) o[(long long)ti * p * na] = f[intra_idx[lane * t + ti]];
}

__global__ __launch_bounds__(64 * ROWS) void so3_intra_group_bwd_kernel(
    long long rows, int rows_per_block, int c, int p, int na, int t,
    const float *__restrict__ gout, const int32_t *__restrict__ intra_idx,
    float *__restrict__ gfeats) {
    extern __shared__ int32_t s_tab[];   // [na*t] index | [na+1] offsets | [na*t] CSR of (a*t+ti)
    int32_t *s_idx = s_tab, *s_off = s_tab + na * t, *s_csr = s_off + na + 1;
    const int total = na * t;
    for (int i = threadIdx.x; i < total; i += blockDim.x) s_idx[i] = intra_idx[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        int cnt = 0;
        if (lane < na)
            for (int e = 0; e < total; ++e) cnt += (s_idx[e] == lane);
        int incl = cnt;   // wave-wide inclusive scan
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane < na) {
            s_off[lane + 1] = incl;
            if (lane == 0) s_off[0] = 0;
            int w = incl - cnt;
            for (int e = 0; e < total; ++e)
                if (s_idx[e] == lane) s_csr[w++] = e;
        }
    }
    __syncthreads();
    if (lane >= na) return;
    const int beg = s_off[lane], end = s_off[lane + 1];
    const long long row0 = (long long)blockIdx.x * rows_per_block;
    for (int r = wave; r < rows_per_block; r += ROWS) {
        const long long row = row0 + r;
        if (row >= rows) break;
        const long long bc = row / p;
        const int pi = (int)(row - bc * p);
        const float *g = gout + (bc * t * p + pi) * na;
        float acc = 0.f;
        for (int j = beg; j < end; ++j) {
            const int e = s_csr[j];
            const int a = e / t, ti = e - a * t;
            acc += g[(long long)ti * p * na + a];
        }
        gfeats[row * na + lane] = acc;
    }
}

}  // namespace

extern "C" int eap_so3_intra_group_fwd_f32(int b, int c, int p, int na, int t, const float *feats,
                                           const int32_t *intra_idx, float *out, eap_stream_t stream) {
    const long long rows = (long long)b * c * p;
    if (rows <= 0 || t <= 0 || na <= 0) return 0;
    if (na > 64) return eap::bad_arg("so3_intra_group_fwd: at most 64 anchors");
    hipLaunchKernelGGL(so3_intra_group_fwd_kernel, dim3(eap::cdiv(rows, ROWS)), dim3(64 * ROWS), 0,
                       eap::S(stream), rows, c, p, na, t, feats, intra_idx, out);
    return eap::check_launch("so3_intra_group_fwd");
}

extern "C" int eap_so3_intra_group_bwd_f32(int b, int c, int p, int na, int t, const float *gout,
                                           const int32_t *intra_idx, float *gfeats, eap_stream_t stream) {
    const long long rows = (long long)b * c * p;
    if (rows <= 0 || na <= 0) return 0;
    if (na > 64) return eap::bad_arg("so3_intra_group_bwd: at most 64 anchors");
    if (t <= 0)
        return eap::hip_fail(hipMemsetAsync(gfeats, 0, sizeof(float) * rows * na, eap::S(stream)), "so3_intra_group_bwd memset");
    const int rows_per_block = 64;
    const size_t shmem = sizeof(int32_t) * ((size_t)2 * na * t + na + 1);
    hipLaunchKernelGGL(so3_intra_group_bwd_kernel, dim3(eap::cdiv(rows, rows_per_block)), dim3(64 * ROWS),
                       shmem, eap::S(stream), rows, rows_per_block, c, p, na, t, gout, intra_idx, gfeats);
    return eap::check_launch("so3_intra_group_bwd");
}
