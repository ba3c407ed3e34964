// csrc/so3_inter_bwd.hip -- backward of the fused SO(3) grouping w.r.t. the input features,
// atomics-free ("private slab" formulation).
//
//   gfeats[b,c,q,perm_n(a)] += sum_k w(p,a,k,n) * gout[b,c,k,p,a]      q = idx[b,p,n]
// (autograd of vgtk/vgtk/so3conv/functional.py:L1221-1261; the reference's own CUDA zpconv
//  backward scatters with atomicAdd, zpconv_cuda_kernel.cu:L77-116.)
//
// Measured on MI355X: fp32 row atomics (60 lanes) retire at ~4.5 G rows/s whatever the address
// pattern (tools/microbench/atomics.hip), i.e. 7.5 ms for the 33.6 M (point, neighbour, channel)
// rows of one 4096-point cloud at C=128 -- slower than the contraction GEMM of the same layer.
// Instead every block (one batch item x 32-channel chunk x contiguous range of points) owns a
// PRIVATE slab [q][32][na] in HBM/L2 and updates it with plain load-add-store:
//   * lanes = anchors (240-byte rows, coalesced), the 8 waves own 4 channels each, so a slab row
//     is only ever touched by one wave in program order -- no atomics, bit-reproducible;
//   * kernel weights are evaluated once per block and (point, anchor, k, n): each wave evaluates
//     one neighbour of an 8-neighbour chunk, parks the values in LDS ([n][anchor][k], k
//     contiguous, pitch 28 floats so the 6 ds_read_b128 per neighbour are conflict-free) and every
//     wave then reads all of them back for its own channels (register tile gout[4][24]);
//   * the slab rows of a whole chunk are requested before the weight phase, so the L2 round trip
//     hides behind it; chunks with a repeated row (repeat-padded lists) take an ordered slow path;
//   * slab loads bypass L1 (agent-scope relaxed load) so a wave always sees its own earlier
//     stores; slabs are summed over the point ranges by a second, deterministic kernel.
#include "common.h"

namespace {

constexpr int CC = 32;      // channels per block
constexpr int NW = 8;       // waves per block
constexpr int CW = CC / NW; // channels per wave (4): lane = (channel-in-wave, anchor quad)
constexpr int KS = 24;      // kernel points held in registers
constexpr int NB = 8;       // neighbours per weight chunk (= NW: one per wave in phase 1)
constexpr int RP = KS * 3 + 1;  // LDS pitch of one anchor's rotated kernel points (odd: conflict-free)
constexpr int T_ = 64 * NW;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld_l2(const float4 *p) {
    // one 16-byte non-temporal load: bypasses L1 (served by L2), so a wave always sees its own
    // earlier (write-through) stores to the slab; waits are left to the compiler
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

__global__ __launch_bounds__(T_) void so3_inter_group_bwd_slab_kernel(
    int c, int p, int n_sup, int nn, int na, int ks, float inv_sigma, int nch, int ps, int ppb,
    int identity_anchor, const float *__restrict__ gout, const int32_t *__restrict__ idx,
    const float4 *__restrict__ gx, const float *__restrict__ rk, const uint8_t *__restrict__ mult,
    float *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *s_w = reinterpret_cast<float *>(smem);                          // [2][NB][KS][64]
    float *s_t = s_w + 2 * NB * KS * 64;                                    // [NW][CW][64] permutation staging
    float *s_rk = s_t + NW * CW * 64;                                       // [64][RP]
    float4 *s_g = reinterpret_cast<float4 *>(s_rk + 64 * RP + (4 - (64 * RP) % 4) % 4);   // [nn]
    int32_t *s_q = reinterpret_cast<int32_t *>(s_g + nn);                   // [nn]
    int32_t *s_dup = s_q + nn;                                              // [nn/NB + 1]
    uint8_t *s_mult = reinterpret_cast<uint8_t *>(s_dup + nn / NB + 1);     // [na*na]

    const int split = blockIdx.x, chunk = blockIdx.y, bi = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq4 = na >> 2;                           // anchor quads per row (na % 4 == 0)
    const int cg = min(lane / nq4, CW - 1), aq = lane - (lane / nq4) * nq4;
    const bool act = lane < nq4 * CW;                  // 60 of 64 lanes at na = 60
    const int p_beg = split * ppb, p_end = min(p, p_beg + ppb);
    float *slab = ws + (((size_t)bi * nch + chunk) * ps + split) * ((size_t)n_sup * CC * na);

    for (int i = threadIdx.x; i < na * KS * 3; i += T_) {
        const int a = i / (KS * 3), r = i - a * KS * 3, k = r / 3;
        s_rk[a * RP + r] = k < ks ? rk[((size_t)a * ks + k) * 3 + (r - k * 3)] : 1e18f;   // k >= ks: weight 0
    }
    if (mult != nullptr)
        for (int i = threadIdx.x; i < na * na; i += T_) s_mult[i] = mult[i];

    const int ch = chunk * CC + wave * CW + cg;        // this lane's channel
    const int chl = wave * CW + cg;                    // ... within the slab row group
    const size_t o_ks = (size_t)p * na, o_cs = (size_t)ks * p * na;
    const int nchunks = (nn + NB - 1) / NB;

    for (int pi = p_beg; pi < p_end; ++pi) {
        __syncthreads();   // previous point's phase-2 reads of s_g/s_q/s_w are done
        const size_t pn = ((size_t)bi * p + pi) * nn;
        for (int i = threadIdx.x; i < nchunks; i += T_) s_dup[i] = 0;
        for (int i = threadIdx.x; i < nn; i += T_) {
            s_g[i] = gx[pn + i];
            const int q = idx[pn + i];
            s_q[i] = q < n_sup ? q : -1;
        }
        // register tile of output gradients: go[k] = gout[b, ch, k, pi, 4*aq .. 4*aq+3]
        float4 go[KS];
        {
            const float *src = gout + (size_t)bi * c * o_cs + (size_t)min(ch, c - 1) * o_cs + (size_t)pi * na + 4 * aq;
            const bool live = act && ch < c;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                go[k] = *reinterpret_cast<const float4 *>(src + (size_t)min(k, ks - 1) * o_ks);
                if (!live || k >= ks) go[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        __syncthreads();
        // a chunk whose neighbours are not pairwise distinct (repeat-padded lists of very sparse
        // balls) must not prefetch its slab rows: flag it
        for (int i = threadIdx.x; i < nn; i += T_) {
            const int q = s_q[i], c0 = (i / NB) * NB;
            bool dup = false;
            for (int j = c0; j < min(c0 + NB, nn); ++j) dup |= (j != i && s_q[j] == q && q >= 0);
            if (dup) s_dup[i / NB] = 1;
        }
        __syncthreads();

        for (int n0 = 0; n0 < nn; n0 += NB) {
            float *wbuf = s_w + ((n0 / NB) & 1) * (NB * KS * 64);
            const bool prefetch = s_dup[n0 / NB] == 0;
            // ---- slab rows of the whole chunk requested up front (hidden behind phase 1 + barrier)
            float4 old[NB];
            if (prefetch) {
#pragma unroll
                for (int nl = 0; nl < NB; ++nl) {
                    const int q = max(s_q[min(n0 + nl, nn - 1)], 0);   // clamped; discarded if invalid
                    old[nl] = ld_l2(reinterpret_cast<const float4 *>(slab + ((size_t)q * CC + chl) * na + 4 * aq));
                }
            }
            // ---- phase 1: wave `w` evaluates the weights of neighbour n0 + w, lane = anchor
            {
                const int nl = wave, n = n0 + nl;
                if (n < nn && lane < na) {
                    const float4 g = s_g[n];
                    const float *kr = s_rk + lane * RP;
#pragma unroll
                    for (int k = 0; k < KS; ++k) {
                        const float dx = g.x - kr[k * 3], dy = g.y - kr[k * 3 + 1], dz = g.z - kr[k * 3 + 2];
                        wbuf[(nl * KS + k) * 64 + lane] = fmaxf(1.0f - (dx * dx + dy * dy + dz * dz) * inv_sigma, 0.0f);
                    }
                }
            }
            __syncthreads();
            // ---- phase 2: every wave, its own CW channels (lane = channel x anchor quad)
#pragma unroll
            for (int nl = 0; nl < NB; ++nl) {
                const int n = n0 + nl;
                const int q = n < nn ? s_q[n] : -1;
                if (q < 0) continue;   // wave-uniform
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                const float *wsrc = wbuf + nl * KS * 64 + 4 * aq;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(wsrc + k * 64);
                    t.x = fmaf(w4.x, go[k].x, t.x);
                    t.y = fmaf(w4.y, go[k].y, t.y);
                    t.z = fmaf(w4.z, go[k].z, t.z);
                    t.w = fmaf(w4.w, go[k].w, t.w);
                }
                const int r = __float_as_int(s_g[n].w);
                if (mult != nullptr && r != identity_anchor) {
                    // anchor permutation: out anchor a contributes to input anchor mult[r][a];
                    // transpose through LDS so the slab access stays one float4 per lane
                    float *tb = s_t + (wave * CW + cg) * 64;
                    if (act) {
                        const uint8_t *m = s_mult + r * na + 4 * aq;
                        tb[m[0]] = t.x; tb[m[1]] = t.y; tb[m[2]] = t.z; tb[m[3]] = t.w;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    t = *reinterpret_cast<const float4 *>(tb + 4 * aq);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                if (act) {
                    float4 *row = reinterpret_cast<float4 *>(slab + ((size_t)q * CC + chl) * na + 4 * aq);
                    const float4 o = prefetch ? old[nl] : ld_l2(row);   // repeated rows: ordered slow path
                    *row = make_float4(o.x + t.x, o.y + t.y, o.z + t.z, o.w + t.w);
                }
            }
            // the other weight buffer is written next; the barrier after the next chunk's phase 1
            // orders those writes against this chunk's reads
        }
    }
}

// gfeats[b,c,q,a] = sum over point ranges of slab[b,chunk,split][q][c%32][a]
__global__ void so3_inter_group_bwd_reduce_kernel(long long total, int c, int n_sup, int na, int nch,
                                                  int ps, const float *__restrict__ ws,
                                                  float *__restrict__ gfeats) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over (b,c,q,a)
    if (e >= total) return;
    const int a = (int)(e % na);
    long long r = e / na;
    const int q = (int)(r % n_sup); r /= n_sup;
    const int ci = (int)(r % c);
    const int bi = (int)(r / c);
    const int chunk = ci / CC, cc = ci - chunk * CC;
    const size_t slab_sz = (size_t)n_sup * CC * na;
    const float *src = ws + ((size_t)bi * nch + chunk) * ps * slab_sz + ((size_t)q * CC + cc) * na + a;
    float s = 0.f;
    for (int z = 0; z < ps; ++z) s += src[(size_t)z * slab_sz];
    gfeats[e] = s;
}

void plan(int b, int c, int p, int *nch, int *ps, int *ppb) {
    *nch = (c + CC - 1) / CC;
    int want = (256 + b * *nch - 1) / (b * *nch);   // >= one block per CU
    int s = 1;
    while (s < want) s *= 2;
    const int max_s = (p + 15) / 16;                // at least 16 points per block
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    *ppb = (p + s - 1) / s;
    *ps = (p + *ppb - 1) / *ppb;
}

}  // namespace

extern "C" int64_t eap_so3_inter_group_bwd_workspace(int b, int c, int p, int n, int na) {
    int nch, ps, ppb;
    plan(b, c, p, &nch, &ps, &ppb);
    return (int64_t)b * nch * ps * n * CC * na;
}

extern "C" int eap_so3_inter_group_bwd_slab_f32(int b, int c, int p, int n, int nn, int na, int ks,
                                                float sigma, const float *gout, const int32_t *idx,
                                                const float *gx, const float *rk, const uint8_t *mult,
                                                int identity_anchor, float *gfeats, float *workspace,
                                                eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0 || na <= 0) return 0;
    if (na > 64) return eap::bad_arg("so3_inter_group_bwd_slab: at most 64 anchors");
    if (ks > KS) return eap::bad_arg("so3_inter_group_bwd_slab: at most 24 kernel points (use the atomic variant)");
    if (na % 4 != 0) return eap::bad_arg("so3_inter_group_bwd_slab: the anchor count must be a multiple of 4 (use the atomic variant)");
    hipStream_t s = eap::S(stream);
    if (p <= 0 || nn <= 0 || ks <= 0)
        return eap::hip_fail(hipMemsetAsync(gfeats, 0, sizeof(float) * (size_t)b * c * n * na, s), "so3_inter_group_bwd memset");
    int nch, ps, ppb;
    plan(b, c, p, &nch, &ps, &ppb);
    const size_t ws_floats = (size_t)b * nch * ps * n * CC * na;
    int e = eap::hip_fail(hipMemsetAsync(workspace, 0, sizeof(float) * ws_floats, s), "so3_inter_group_bwd_slab memset");
    if (e) return e;
    const size_t shmem = sizeof(float) * (2 * NB * KS * 64 + NW * CW * 64 + 64 * RP + 4) + 20 * (size_t)nn + 4 * ((size_t)nn / NB + 1) + (mult ? (size_t)na * na : 0);
    e = eap::hip_fail(hipFuncSetAttribute((const void *)so3_inter_group_bwd_slab_kernel,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                      "so3_inter_group_bwd_slab shared memory");
    if (e) return e;
    hipLaunchKernelGGL(so3_inter_group_bwd_slab_kernel, dim3(ps, nch, b), dim3(T_), shmem, s, c, p, n, nn, na, ks,
                       1.0f / sigma, nch, ps, ppb, identity_anchor, gout, idx, reinterpret_cast<const float4 *>(gx), rk, mult, workspace);
    e = eap::check_launch("so3_inter_group_bwd_slab");
    if (e) return e;
    const long long total = (long long)b * c * n * na;
    hipLaunchKernelGGL(so3_inter_group_bwd_reduce_kernel, dim3(eap::cdiv(total, 256)), dim3(256), 0, s, total, c, n,
                       na, nch, ps, workspace, gfeats);
    return eap::check_launch("so3_inter_group_bwd_reduce");
}
