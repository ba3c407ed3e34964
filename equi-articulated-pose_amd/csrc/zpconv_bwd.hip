// csrc/zpconv_bwd.hip -- the native inter "zpconv" backward (zpconv_cuda.cpp:L58-75, kernel
// zpconv_cuda_kernel.cu:L77-116) without atomics, for the index every reference caller builds (one neighbour
// list per point broadcast over (a,k): spconv/functional.py:L232-249).
//
//   gfeats[b,c,q,a] = sum over (p,n) with idx[b,p,.,.,n] == q of  sum_k w[b,p,a,k,n] * grad[b,c,k,p,a]
//
// The reference scatters with atomicAdd (one per (c,p,a,k,n) tuple).  Gathering instead -- per support point q,
// over the (p,n) pairs that reference it -- would re-read grad[:, :, p, :] (368 KB at C = 64) for each of the 64
// neighbours of p.  So the product is formed in FORWARD order, where everything of a point is read once, and the
// scatter is replaced by a sorted sum:
//   1. index check (csrc/zpconv_mfma.hip): one list per point? -> idx0 [b,p,nn], flag[b], entry ids;
//   2. zpconv_bwd_t_kernel: T[b,p,a,n,c] = sum_k grad[b,c,k,p,a] w[b,p,a,k,n] -- per (p,a) a [C x K] x [K x NN]
//      product on v_mfma_f32_32x32x2_f32 (M = channels, N = neighbours, K = kernel points: 24 = 12 k-steps, no
//      padding anywhere at C = 64, NN = 64).  Waves are independent (no barrier): a wave owns (p, anchor quad)
//      units, loads its A operand grad[c, k, p, 4 anchors] as 16-byte words straight into registers (one set
//      serves the unit's four anchors) and the weight rows w[p,a,k,:] as coalesced 128-byte rows; the 64 x 64
//      result of an anchor goes through a wave-private LDS tile so that T is written in 1 KB runs;
//   3. inverse neighbour lists on the device (csrc/inv_lists.hip: per support row q the entries (p,n) in a
//      fixed order -> deterministic sums);
//   4. zpconv_bwd_sum_kernel: gfeats[b,:,q,4 anchors] = sum of the T rows of q's entries, transposed through LDS to
//      16-byte stores.
// Clouds with any other index (flag set) go through the scatter kernel of csrc/zpconv.hip.
// T costs 4*P*NN*C*A bytes per cloud (4 GB at 4096 points, C = 64): the caller provides it (workspace) and may
// split the batch.
#include "common.h"
#include <type_traits>
#include <utility>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NWV = 8, TM = 64 * NWV;
constexpr int KS2 = 12;                    // MFMA k-steps: up to 24 kernel points
constexpr int TP = 68;                     // LDS tile pitch (floats): 16-byte aligned rows, 4 banks apart

// uniform 64-bit base + 32-bit byte offset per lane: hipcc then uses the SGPR-base addressing form and no 64-bit
// vector arithmetic (the first version spent 260 v_lshl_add_u64 per unit and, under the register pressure they
// caused, waited for every single load)
template <typename V>
__device__ __forceinline__ V ld_off(const float *ubase, unsigned voff) {
    return *reinterpret_cast<const V *>(reinterpret_cast<const char *>(ubase) + voff);
}
template <typename V>
__device__ __forceinline__ void st_off(float *ubase, unsigned voff, V v) {
    *reinterpret_cast<V *>(reinterpret_cast<char *>(ubase) + voff) = v;
}

// FULL: ks == 24, nn == 64, channel slice complete -- no clamps, no masks
template <bool FULL>
__global__ __launch_bounds__(TM, 2) void zpconv_bwd_t_kernel(int C, int na, int ks, int P, int nn, int c0, int upw, int nb,
                                                             const float *__restrict__ g, const float *__restrict__ w,
                                                             const int32_t *__restrict__ skip, float *__restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *tile = smem + (size_t)wave * 64 * TP;                   // wave-private [64 n][TP]
    const int naq = na >> 2;
    const long long per_cloud = (long long)P * naq, total = per_cloud * nb;
    // the 8 waves of a workgroup take 8 consecutive anchor quads (mostly of one point) at a time: their A loads touch
    // the same 128-byte lines of grad[:, :, p, :] within a few hundred cycles and share them in L1 (with a wave walking
    // 4 quads of a point on its own every quad re-fetched the point's 1536 lines from L2)

    const size_t g_ks = (size_t)P * na, g_cs = (size_t)ks * P * na;
    // per-lane byte offsets (the launcher bounds a 64-channel slice of grad and the rows of T below 4 GB)
    unsigned offA[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) offA[mt] = (unsigned)(((size_t)(min(c0 + 32 * mt + li, C - 1) - c0) * g_cs + (FULL ? (size_t)lh * g_ks : 0)) * 4);
    const unsigned offB = (unsigned)(((FULL ? lh * nn : 0) + li) * 4);
    const unsigned offT = (unsigned)(((lane >> 4) * C + (lane & 15) * 4) * 4);
    const int tw = li * TP + 4 * lh, trd = (lane >> 4) * TP + (lane & 15) * 4;      // tile write / read positions (floats)

    for (int ui = 0; ui < upw; ++ui) {
        const long long u = ((long long)blockIdx.x * upw + ui) * NWV + wave;
        if (u >= total) break;
        const int bi = (int)(u / per_cloud);
        if (__builtin_amdgcn_readfirstlane(skip[bi]) != 0) continue;
        const int rem = (int)(u - (long long)bi * per_cloud), p = rem / naq, aq = rem - p * naq;

        // A operand: grad[b, c, k, p, 4aq .. 4aq+3] for lane (c = 32 mt + li, k = 2 s + lh): one register set for the
        // unit's four anchors
        f32x4 A[2][KS2];
        {
            const float *gb = g + ((size_t)bi * C + c0) * g_cs + (size_t)p * na + 4 * aq;          // uniform
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const float *gs = FULL ? gb + (size_t)(2 * s) * g_ks : gb;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    A[mt][s] = ld_off<f32x4>(gs, FULL ? offA[mt] : offA[mt] + (unsigned)(min(2 * s + lh, ks - 1) * g_ks * 4));
            }
        }
        // eight steps (anchor j = st >> 1, neighbour half nt = st & 1).  ALL the weights of the unit are requested up front
        // (96 registers): loads and stores share one in-order counter on this part, so a wave that waits for a load
        // issued after a store waits for that store's round trip too -- with the weights of step st + 1 requested
        // during step st every step paid one (4.8 ms for 2 clouds)
        const float *wp = w + (((size_t)bi * P + p) * na + 4 * aq) * ks * nn;                       // uniform
        const size_t wa = (size_t)ks * nn;
        float B[8 * KS2];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const float *wr = wp + (size_t)(st >> 1) * wa;
            if (FULL) {
#pragma unroll
                for (int s = 0; s < KS2; ++s) B[st * KS2 + s] = ld_off<float>(wr + (size_t)(2 * s) * nn + 32 * (st & 1), offB);
            } else {
                const int n = 32 * (st & 1) + li;
#pragma unroll
                for (int s = 0; s < KS2; ++s) {
                    const float v = ld_off<float>(wr, (unsigned)((min(2 * s + lh, ks - 1) * nn + min(n, nn - 1)) * 4));
                    B[st * KS2 + s] = (2 * s + lh < ks && n < nn) ? v : 0.f;
                }
            }
        }
        float *tb = T + (((size_t)bi * P + p) * na + 4 * aq) * (size_t)nn * C + c0;                 // uniform
        // (a macro: clang's constant evaluator crashes on the equivalent pack-expanded lambda)
#define ZPB_STEP(ST)                                                                                                 \
        do {                                                                                                         \
            constexpr int J = (ST) >> 1, nt = (ST) & 1;                                                              \
            f32x16 acc[2];                                                                                           \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                         \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;                                     \
            _Pragma("unroll") for (int s = 0; s < KS2; ++s)                                                          \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                     \
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[mt][s][J], B[(ST) * KS2 + s], acc[mt], 0, 0, 0); \
            /* D[i = channel][j = neighbour]: lane column j = li, rows i = (r&3) + 8 (r>>2) + 4 lh.  Through the     \
               wave's LDS tile [n][c] and out as rows of T[b,p,a,n,:] */                                             \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                         \
                _Pragma("unroll") for (int i4 = 0; i4 < 4; ++i4)                                                     \
                    *reinterpret_cast<float4 *>(tile + 32 * nt * TP + tw + 32 * mt + 8 * i4) =                       \
                        make_float4(acc[mt][4 * i4], acc[mt][4 * i4 + 1], acc[mt][4 * i4 + 2], acc[mt][4 * i4 + 3]); \
            float *tr = tb + (size_t)J * nn * C + (size_t)(32 * nt) * C;                 /* uniform */               \
            _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                       \
                const float4 v = *reinterpret_cast<const float4 *>(tile + (32 * nt + 4 * it) * TP + trd);            \
                if (FULL || (32 * nt + 4 * it + (lane >> 4) < nn && c0 + (lane & 15) * 4 < C))                       \
                    st_off<float4>(tr + (size_t)(4 * it) * C, offT, v);                                              \
            }                                                                                                        \
        } while (0)
        ZPB_STEP(0); ZPB_STEP(1); ZPB_STEP(2); ZPB_STEP(3); ZPB_STEP(4); ZPB_STEP(5); ZPB_STEP(6); ZPB_STEP(7);
#undef ZPB_STEP
    }
}

// gfeats[b, c, q, 16 ag .. +15] = sum over q's entries e = (p, n) of T[b, p, a, n, c]: thread (anchor al = t >> 4,
// channel quad c4 = t & 15) reads 16 bytes per entry (a wave = 4 anchors x 256 contiguous bytes), four entries in flight,
// summed in entry order; the [16 anchors][64 channels] result goes through LDS to 16-byte stores along the anchors
__global__ __launch_bounds__(256) void zpconv_bwd_sum_kernel(int C, int na, int nq, int P, int nn, const int32_t *__restrict__ rows,
                                                             const int32_t *__restrict__ off, const int32_t *__restrict__ cnt,
                                                             const float4 *__restrict__ ent_e, const int32_t *__restrict__ skip,
                                                             const float *__restrict__ T, float *__restrict__ gf) {
    __shared__ float s_x[16][65];
    const int ag = blockIdx.x, r = blockIdx.y, bi = blockIdx.z, t = threadIdx.x;
    if (skip[bi] != 0) return;
    const int q = rows[(size_t)bi * nq + r];
    if (q < 0) return;
    const int n_ent = cnt[(size_t)bi * nq + r];
    const float4 *ent = ent_e + (size_t)bi * P * nn + off[(size_t)bi * nq + r];
    const int al = t >> 4, c4 = (t & 15) * 4, a = min(16 * ag + al, na - 1);
    const size_t a_off = (size_t)a * nn * C, p_stride = (size_t)na * nn * C;
    const float *Tb = T + (size_t)bi * P * p_stride;
    for (int cb = 0; cb < C; cb += 64) {
        const int c = min(cb + c4, C - 4);
        auto row = [&](int j) {
            const unsigned e = __float_as_uint(ent[j].x);
            return *reinterpret_cast<const float4 *>(Tb + (size_t)(e / nn) * p_stride + a_off + (size_t)(e % nn) * C + c);
        };
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        auto add = [](float4 &s, const float4 v) { s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; };
        int j = 0;
        for (; j + 4 <= n_ent; j += 4) {
            const float4 v0 = row(j), v1 = row(j + 1), v2 = row(j + 2), v3 = row(j + 3);
            add(s0, v0); add(s1, v1); add(s2, v2); add(s3, v3);
        }
        for (; j < n_ent; ++j) add(s0, row(j));
        s_x[al][c4] = (s0.x + s1.x) + (s2.x + s3.x);
        s_x[al][c4 + 1] = (s0.y + s1.y) + (s2.y + s3.y);
        s_x[al][c4 + 2] = (s0.z + s1.z) + (s2.z + s3.z);
        s_x[al][c4 + 3] = (s0.w + s1.w) + (s2.w + s3.w);
        __syncthreads();
        {
            const int cc = t >> 2, aq = t & 3;                       // 64 channels x 4 anchor quads
            if (cb + cc < C && 16 * ag + 4 * aq < na)
                *reinterpret_cast<float4 *>(gf + (((size_t)bi * C + cb + cc) * nq + q) * na + 16 * ag + 4 * aq) =
                    make_float4(s_x[4 * aq][cc], s_x[4 * aq + 1][cc], s_x[4 * aq + 2][cc], s_x[4 * aq + 3][cc]);
        }
        __syncthreads();
    }
}

}  // namespace

namespace eap {
bool inter_zpconv_bwd_matrix_supported(int np, int nq, int na, int ks, int nn, int c) {
    return na > 0 && na <= 64 && (na & 3) == 0 && ks > 0 && ks <= 2 * KS2 && nn > 0 && nn <= 64 && c >= 16 && (c & 3) == 0 && nq <= 16384 &&
           ((long long)np * nn & 3) == 0 && 64ll * ks * np * na * 4 < (1ll << 32) && 64ll * c * 4 < (1ll << 32);
}
}  // namespace eap

// Workspace layout, shared by the size query and the launcher: every chunk starts on a 256-byte boundary.
//   flags [b] | idx0 [b,np,ann] | entry ids float4 [b,np,ann] | counts, rows, off, cnt [b,nq] | n_rows [b] | ent_p [b,np*ann] |
//   ent_e float4 [b,np*ann] | T float [b,np,na,ann,c]
namespace {
struct BwdWorkspace {
    int64_t flag, idx0, eid, counts, rows, off, cnt, n_rows, ent_p, ent_e, T, total;
    BwdWorkspace(int b, int np, int nq, int na, int ann, int c) {
        const int64_t ent = (int64_t)b * np * ann, fl = 64 * (((int64_t)b + 63) / 64);
        int64_t at = 0;
        auto take = [&](int64_t bytes) { const int64_t r = at; at += (bytes + 255) / 256 * 256; return r; };
        flag = take(4 * fl);
        idx0 = take(4 * ent);
        eid = take(16 * ent);
        counts = take(4ll * b * nq);
        rows = take(4ll * b * nq);
        off = take(4ll * b * nq);
        cnt = take(4ll * b * nq);
        n_rows = take(4 * fl);
        ent_p = take(4 * ent);
        ent_e = take(16 * ent);
        T = take(4 * ent * na * c);
        total = at;
    }
};
}  // namespace

extern "C" int64_t eap_inter_zpconv_bwd_workspace(int b, int np, int nq, int na, int ann, int c) {
    if (b <= 0 || np <= 0 || nq < 0 || na <= 0 || ann <= 0 || c <= 0) return 256;
    return BwdWorkspace(b, np, nq, na, ann, c).total;
}

extern "C" int eap_inter_zpconv_bwd_ws_f32(int b, int np, int nq, int na, int ks, int ann, int c, const int32_t *idx,
                                           const float *w, const float *grad, float *gfeats, void *workspace,
                                           eap_stream_t stream) {
    if (b <= 0 || np <= 0 || c <= 0) return eap_inter_zpconv_bwd_f32(b, np, nq, na, ks, ann, c, idx, w, grad, gfeats, stream);
    hipStream_t s = eap::S(stream);
    const bool matrix = workspace != nullptr && eap::inter_zpconv_bwd_matrix_supported(np, nq, na, ks, ann, c) && (ann & 3) == 0 &&
                        (long long)na * ks * ann < (1ll << 31) &&
                        ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(gfeats) |
                          reinterpret_cast<uintptr_t>(workspace)) & 15) == 0;
    if (!matrix) return eap_inter_zpconv_bwd_f32(b, np, nq, na, ks, ann, c, idx, w, grad, gfeats, stream);
    const BwdWorkspace L(b, np, nq, na, ann, c);
    char *wsb = reinterpret_cast<char *>(workspace);
    int32_t *flag = reinterpret_cast<int32_t *>(wsb + L.flag);
    int32_t *idx0 = reinterpret_cast<int32_t *>(wsb + L.idx0);
    float *eid = reinterpret_cast<float *>(wsb + L.eid);
    int32_t *counts = reinterpret_cast<int32_t *>(wsb + L.counts);
    int32_t *rows = reinterpret_cast<int32_t *>(wsb + L.rows);
    int32_t *off = reinterpret_cast<int32_t *>(wsb + L.off);
    int32_t *cnt = reinterpret_cast<int32_t *>(wsb + L.cnt);
    int32_t *n_rows = reinterpret_cast<int32_t *>(wsb + L.n_rows);
    int32_t *ent_p = reinterpret_cast<int32_t *>(wsb + L.ent_p);
    float *ent_e = reinterpret_cast<float *>(wsb + L.ent_e);
    float *T = reinterpret_cast<float *>(wsb + L.T);

    int e = eap::hip_fail(hipMemsetAsync(flag, 0, sizeof(int32_t) * b, s), "inter_zpconv_backward flags");
    if (e) return e;
    e = eap::hip_fail(hipMemsetAsync(gfeats, 0, sizeof(float) * (size_t)b * c * nq * na, s), "inter_zpconv_backward memset");
    if (e) return e;
    e = eap::zpconv_index_check(b, np, na * ks * ann, ann, idx, idx0, eid, flag, s);
    if (e) return e;
    // T, 64 channels per pass
    {
        const size_t shmem = sizeof(float) * NWV * 64 * TP;
        e = eap::hip_fail(hipFuncSetAttribute((const void *)zpconv_bwd_t_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                          "inter_zpconv_backward shared memory");
        if (e) return e;
        e = eap::hip_fail(hipFuncSetAttribute((const void *)zpconv_bwd_t_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                          "inter_zpconv_backward shared memory");
        if (e) return e;
        const long long total = (long long)b * np * (na >> 2);
        const int upw = 4;
        const long long blocks = (total + (long long)NWV * upw - 1) / ((long long)NWV * upw);
        if (blocks >= (1ll << 31)) return eap::bad_arg("inter_zpconv_backward: too many workgroups");
        for (int c0 = 0; c0 < c; c0 += 64) {
            const bool full = ks == 2 * KS2 && ann == 64 && c0 + 64 <= c;
            hipLaunchKernelGGL(full ? zpconv_bwd_t_kernel<true> : zpconv_bwd_t_kernel<false>, dim3((unsigned)blocks), dim3(TM), shmem, s, c, na, ks,
                               np, ann, c0, upw, b, grad, w, flag, T);
            e = eap::check_launch("inter_zpconv_backward (products)");
            if (e) return e;
        }
    }
    e = eap_inv_lists_rows(b, np, nq, ann, idx0, counts, rows, off, cnt, n_rows, stream);
    if (e) return e;
    e = eap_inv_lists_fill(b, np, nq, ann, nq, idx0, eid, rows, off, ent_p, ent_e, stream);
    if (e) return e;
    hipLaunchKernelGGL(zpconv_bwd_sum_kernel, dim3((na + 15) / 16, nq, b), dim3(256), 0, s, c, na, nq, np, ann, rows, off, cnt,
                       reinterpret_cast<const float4 *>(ent_e), flag, T, gfeats);
    e = eap::check_launch("inter_zpconv_backward (sums)");
    if (e) return e;
    return eap::inter_zpconv_bwd_flagged(b, np, nq, na, ks, ann, c, idx, w, grad, gfeats, flag, s);
}
