// csrc/inv_lists.hip -- inverse neighbour lists on the device.
//
// The re-associated backward of the SO(3) inter convolution (csrc/so3_inter_inv.hip, so3_inter_lists.hip:
// autograd of vgtk/vgtk/so3conv/functional.py:L1221-1261) walks, for every REFERENCED support row q, the
// list of (query point p, slot n) pairs with idx[p,n] == q, in (p,n) order (a fixed order: the sums are
// bit-reproducible).  Round 1 built those lists with torch.sort + argsort + scatter_add + cumsum + gathers
// and one blocking host read per layer; here it is three small kernels and no host round trip:
//
//   eap_inv_lists_rows   counts[b,q] (LDS histogram per block, integer atomics), then per cloud a bitonic
//                        sort of (count, q) -- referenced rows first, longest list first, ties by index --
//                        and an exclusive scan:  rows / cnt / off [b,n], n_rows [b]
//   eap_inv_lists_fill   one workgroup per (cloud, referenced row): stream the cloud's idx (1 MB, L2
//                        resident), keep the matches in order (ballot + mbcnt ranks), write ent_p / ent_gx
//
// The fill reads R * P*NN*4 bytes per cloud from L2 for R referenced rows; the lists are only used when
// R <= P/4 (vgtk.so3conv.functional.INV_ROW_FRACTION), typically R = 100-300 of 4096.
#include "common.h"

namespace {

constexpr int RT = 1024;      // threads of the per-cloud sort block
constexpr int MAXN = 16384;   // support points per cloud the LDS sort holds (128 KB of 8-byte keys)

__global__ __launch_bounds__(256) void inv_count_kernel(int per_cloud, int n, const int32_t *__restrict__ idx,
                                                        int32_t *__restrict__ counts) {
    extern __shared__ int s_hist[];
    const int bi = blockIdx.y, t = threadIdx.x;
    for (int i = t; i < n; i += 256) s_hist[i] = 0;
    __syncthreads();
    const int chunk = (per_cloud + gridDim.x - 1) / gridDim.x;
    const int beg = blockIdx.x * chunk, end = min(per_cloud, beg + chunk);
    const int32_t *src = idx + (size_t)bi * per_cloud;
    for (int e = beg + t; e < end; e += 256) {
        const int q = src[e];
        if ((unsigned)q < (unsigned)n) atomicAdd(&s_hist[q], 1);
    }
    __syncthreads();
    for (int i = t; i < n; i += 256) {
        const int v = s_hist[i];
        if (v) atomicAdd(&counts[(size_t)bi * n + i], v);
    }
}

__global__ __launch_bounds__(RT) void inv_rows_kernel(int n, int n2, const int32_t *__restrict__ counts,
                                                      int32_t *__restrict__ rows, int32_t *__restrict__ off,
                                                      int32_t *__restrict__ cnt, int32_t *__restrict__ n_rows) {
    extern __shared__ unsigned long long s_key[];           // [n2]; then reused for the scan
    __shared__ int s_part[RT];
    const int bi = blockIdx.x, t = threadIdx.x;
    for (int i = t; i < n2; i += RT) {
        const unsigned c = i < n ? (unsigned)counts[(size_t)bi * n + i] : 0u;
        s_key[i] = c ? (((unsigned long long)c << 32) | (0xFFFFFFFFu - (unsigned)i)) : 0ull;
    }
    __syncthreads();
    // bitonic sort, descending
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < n2; i += RT) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long a = s_key[i], b = s_key[l];
                    const bool desc = (i & k) == 0;
                    if (desc ? a < b : a > b) { s_key[i] = b; s_key[l] = a; }
                }
            }
            __syncthreads();
        }
    }
    // exclusive scan of the counts in sorted order: thread t owns the consecutive slots [t*per, (t+1)*per)
    const int per = n2 / RT > 0 ? n2 / RT : 1;
    int local = 0, nz = 0;
    for (int u = 0; u < per; ++u) {
        const int i = t * per + u;
        if (i < n2) { const unsigned c = (unsigned)(s_key[i] >> 32); local += (int)c; nz += c != 0; }
    }
    s_part[t] = local;
    __syncthreads();
    for (int d = 1; d < RT; d <<= 1) {                      // Hillis-Steele over the 1024 partials
        const int v = t >= d ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - local;
    for (int u = 0; u < per; ++u) {
        const int i = t * per + u;
        if (i < n) {
            const unsigned long long key = s_key[i];
            const unsigned c = (unsigned)(key >> 32);
            rows[(size_t)bi * n + i] = c ? (int)(0xFFFFFFFFu - (unsigned)key) : -1;
            cnt[(size_t)bi * n + i] = (int)c;
            off[(size_t)bi * n + i] = run;
            run += (int)c;
        }
    }
    // number of referenced rows
    __syncthreads();
    s_part[t] = nz;
    __syncthreads();
    for (int d = RT / 2; d > 0; d >>= 1) {
        if (t < d) s_part[t] += s_part[t + d];
        __syncthreads();
    }
    if (t == 0) n_rows[bi] = s_part[0];
}

// one workgroup (4 waves) per (referenced row slot, cloud); wave w scans the w-th quarter of the cloud's
// entries twice: count, then place
__global__ __launch_bounds__(256) void inv_fill_kernel(int per_cloud, int nn, int n, int rld,
                                                       const int32_t *__restrict__ idx, const float4 *__restrict__ gx,
                                                       const int32_t *__restrict__ rows, const int32_t *__restrict__ off,
                                                       int32_t *__restrict__ ent_p, float4 *__restrict__ ent_gx) {
    __shared__ int s_wave[4];
    const int r = blockIdx.x, bi = blockIdx.y;
    const int q = rows[(size_t)bi * rld + r];
    if (q < 0) return;                                       // block-uniform
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int quads = per_cloud >> 2;                        // per_cloud is a multiple of 4 (launcher)
    const int qper = (quads + 3) / 4;
    const int qbeg = wave * qper, qend = min(quads, qbeg + qper);
    const int4 *src = reinterpret_cast<const int4 *>(idx + (size_t)bi * per_cloud);
    int mine = 0;
    for (int i = qbeg + lane; i < qend; i += 64) {
        const int4 v = src[i];
        mine += (v.x == q) + (v.y == q) + (v.z == q) + (v.w == q);
    }
    for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
    if (lane == 0) s_wave[wave] = mine;
    __syncthreads();
    int base = off[(size_t)bi * rld + r];
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    int32_t *dp = ent_p + (size_t)bi * per_cloud;
    float4 *dg = ent_gx + (size_t)bi * per_cloud;
    const float4 *sg = gx + (size_t)bi * per_cloud;
    for (int i0 = qbeg; i0 < qend; i0 += 64) {               // wave-uniform trip count
        const int i = i0 + lane;
        int4 v = make_int4(-1, -1, -1, -1);
        if (i < qend) v = src[i];
        const bool m0 = v.x == q, m1 = v.y == q, m2 = v.z == q, m3 = v.w == q;
        const unsigned long long b0 = __ballot(m0), b1 = __ballot(m1), b2 = __ballot(m2), b3 = __ballot(m3);
        if ((b0 | b1 | b2 | b3) == 0ull) continue;
        // matches of lower lanes come first (lane l holds entries 4l .. 4l+3), then this lane's own in order
        int pos = base;
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0));
        pos += __builtin_amdgcn_mbcnt_hi((unsigned)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b3, 0));
        const int e = 4 * i;
        if (m0) { dp[pos] = e / nn; dg[pos] = sg[e]; ++pos; }
        if (m1) { dp[pos] = (e + 1) / nn; dg[pos] = sg[e + 1]; ++pos; }
        if (m2) { dp[pos] = (e + 2) / nn; dg[pos] = sg[e + 2]; ++pos; }
        if (m3) { dp[pos] = (e + 3) / nn; dg[pos] = sg[e + 3]; ++pos; }
        base += __popcll(b0) + __popcll(b1) + __popcll(b2) + __popcll(b3);
    }
}

// scatter / gather of [b,c,rcap,na] row blocks <-> [b,c,n,na] (the referenced rows of a feature tensor)
__global__ __launch_bounds__(256) void rows_gather_kernel(int c, int n, int na4, int rcap, int rld,
                                                          const int32_t *__restrict__ rows, const float4 *__restrict__ src,
                                                          float4 *__restrict__ dst) {
    const int bi = blockIdx.z, ci = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= rcap * na4) return;
    const int r = e / na4, a = e - r * na4;
    const int q = rows[(size_t)bi * rld + r];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q >= 0) v = src[(((size_t)bi * c + ci) * n + q) * na4 + a];
    dst[(((size_t)bi * c + ci) * rcap + r) * na4 + a] = v;
}

__global__ __launch_bounds__(256) void rows_scatter_kernel(int c, int n, int na4, int rcap, int rld,
                                                           const int32_t *__restrict__ rows, const float4 *__restrict__ src,
                                                           float4 *__restrict__ dst) {
    const int bi = blockIdx.z, ci = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= rcap * na4) return;
    const int r = e / na4, a = e - r * na4;
    const int q = rows[(size_t)bi * rld + r];
    if (q >= 0) dst[(((size_t)bi * c + ci) * n + q) * na4 + a] = src[(((size_t)bi * c + ci) * rcap + r) * na4 + a];
}

}  // namespace

extern "C" int eap_inv_lists_rows(int b, int p, int n, int nn, const int32_t *idx, int32_t *counts, int32_t *rows,
                                  int32_t *off, int32_t *cnt, int32_t *n_rows, eap_stream_t stream) {
    if (b <= 0 || p <= 0 || nn <= 0 || n <= 0) return 0;
    if (n > MAXN) return eap::bad_arg("inv_lists_rows: more than 16384 support points per cloud");
    if ((long long)p * nn >= (1ll << 31)) return eap::bad_arg("inv_lists_rows: more than 2^31 entries per cloud");
    hipStream_t s = eap::S(stream);
    int e = eap::hip_fail(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)b * n, s), "inv_lists_rows memset");
    if (e) return e;
    const int per_cloud = p * nn;
    const int blocks = (int)eap::cdiv(per_cloud, 8192) > 0 ? (int)eap::cdiv(per_cloud, 8192) : 1;
    hipLaunchKernelGGL(inv_count_kernel, dim3(blocks, b), dim3(256), sizeof(int) * n, s, per_cloud, n, idx, counts);
    e = eap::check_launch("inv_lists_rows (count)");
    if (e) return e;
    int n2 = RT;
    while (n2 < n) n2 <<= 1;
    const size_t shmem = sizeof(unsigned long long) * n2;
    e = eap::hip_fail(hipFuncSetAttribute((const void *)inv_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                      "inv_lists_rows shared memory");
    if (e) return e;
    hipLaunchKernelGGL(inv_rows_kernel, dim3(b), dim3(RT), shmem, s, n, n2, counts, rows, off, cnt, n_rows);
    return eap::check_launch("inv_lists_rows (sort)");
}

extern "C" int eap_inv_lists_fill(int b, int p, int n, int nn, int rcap, const int32_t *idx, const float *gx,
                                  const int32_t *rows, const int32_t *off, int32_t *ent_p, float *ent_gx,
                                  eap_stream_t stream) {
    if (b <= 0 || p <= 0 || nn <= 0 || rcap <= 0) return 0;
    if (((long long)p * nn & 3) != 0) return eap::bad_arg("inv_lists_fill: p * nn must be a multiple of 4");
    if (rcap > n) return eap::bad_arg("inv_lists_fill: rcap exceeds the number of support points");
    hipLaunchKernelGGL(inv_fill_kernel, dim3(rcap, b), dim3(256), 0, eap::S(stream), p * nn, nn, n, n, idx,
                       reinterpret_cast<const float4 *>(gx), rows, off, ent_p, reinterpret_cast<float4 *>(ent_gx));
    return eap::check_launch("inv_lists_fill");
}

extern "C" int eap_rows_gather_f32(int b, int c, int n, int na, int rcap, int rows_ld, const int32_t *rows,
                                   const float *src, float *dst, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || rcap <= 0) return 0;
    if ((na & 3) != 0 || c > 65535 || b > 65535) return eap::bad_arg("rows_gather: na must be a multiple of 4; c, b <= 65535");
    hipLaunchKernelGGL(rows_gather_kernel, dim3(eap::cdiv((long long)rcap * (na / 4), 256), c, b), dim3(256), 0, eap::S(stream),
                       c, n, na / 4, rcap, rows_ld, rows, reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst));
    return eap::check_launch("rows_gather");
}

extern "C" int eap_rows_scatter_f32(int b, int c, int n, int na, int rcap, int rows_ld, const int32_t *rows,
                                    const float *src, float *dst, eap_stream_t stream) {
    if (b <= 0 || c <= 0) return 0;
    if ((na & 3) != 0 || c > 65535 || b > 65535) return eap::bad_arg("rows_scatter: na must be a multiple of 4; c, b <= 65535");
    hipStream_t s = eap::S(stream);
    int e = eap::hip_fail(hipMemsetAsync(dst, 0, sizeof(float) * (size_t)b * c * n * na, s), "rows_scatter memset");
    if (e || rcap <= 0) return e;
    hipLaunchKernelGGL(rows_scatter_kernel, dim3(eap::cdiv((long long)rcap * (na / 4), 256), c, b), dim3(256), 0, s,
                       c, n, na / 4, rcap, rows_ld, rows, reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst));
    return eap::check_launch("rows_scatter");
}
