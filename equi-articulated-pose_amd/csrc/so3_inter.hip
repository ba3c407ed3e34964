// csrc/so3_inter.hip -- SE(3)-equivariant inter point convolution, grouping stage, on gfx950.
//
// Reference path (vgtk/vgtk/so3conv/functional.py, stride-1 branch L1025-1261):
//   ball_idx, grouped_xyz = ball_query(xyz, xyz, r, nn)                         L1061
//   R_rel = R_p R_n^T ; g = R_rel (x_n - x_p)                                   L1065-1078
//   w[b,p,a,k,n] = relu(1 - |g - A_a kappa_k|^2 / sigma)                        L1112 (L2508-2549)
//   perm[b,p,n,a] = argmax_j tr(R_rel^T A_a A_j^T)                              L1199-1204
//   G[b,c,p,n,a] = feats[b,c,idx[b,p,n],perm[b,p,n,a]]                          L1221-1252
//   out[b,c,k,p,a] = sum_n G[b,c,p,n,a] w[b,p,a,k,n]                            L1261
// The reference materialises w ([B,P,60,24,64] = 12 GB at B=8,P=4096), a 60x60x3x3 trace
// tensor per (point, neighbour) and two gathered feature copies.  Here:
//   * so3_prep computes g and ONE nearest-anchor index r per (point, neighbour).  Because the
//     anchors form a group, argmax_j tr(R^T A_a A_j^T) = mult[r][a] with
//     r = argmax_g tr(R A_g): right-multiplying by A_a^T is an isometry of the Frobenius inner
//     product, so the 60x60 search collapses to one 60-way search plus a table lookup
//     (identical result except for exact fp ties between two anchors).
//   * so3_inter_group_fwd fuses weight evaluation, anchor permutation, neighbour gather and the
//     weighted sum; w never exists in memory.  Lanes run along the anchor dimension (the
//     contiguous dimension of feats [b,c,q,a] and of out [b,c,k,p,a]), the four waves of a block
//     split the kernel points, and each lane keeps a CC x KPW register tile of outputs.
//   * so3_inter_weights / so3_anchor_perm materialise w / perm only when a caller asks for them
//     (the module API returns inter_w; parity tests compare them with the oracle).
#include "common.h"

namespace {

constexpr int G_THREADS = 256;

// ---------------------------------------------------------------------------------------------
// prep: gx[b,p,n] = (g.x, g.y, g.z, r)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void so3_prep_kernel(
    int p, int n_sup, int nn, int na, const float *__restrict__ q_xyz,
    const float *__restrict__ s_xyz, const int32_t *__restrict__ idx,
    const float *__restrict__ q_pose, const float *__restrict__ s_pose,
    const float *__restrict__ anchors, int identity_anchor, float4 *__restrict__ gx,
    int32_t *__restrict__ nonident) {
    __shared__ float s_anchor[64 * 9];
    for (int i = threadIdx.x; i < na * 9; i += blockDim.x) s_anchor[i] = anchors[i];
    __syncthreads();
    const int bi = blockIdx.y;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)p * nn) return;
    const int pi = (int)(e / nn);
    const int q = idx[(size_t)bi * p * nn + e];
    const float *Q = q_xyz + (size_t)bi * 3 * p;
    const float *X = s_xyz + (size_t)bi * 3 * n_sup;
    // a neighbour index == n_sup addresses the reference's shadow point at 1e4 (spconv/functional.py:L83-87)
    const float sx = q < n_sup ? X[q] : 1e4f, sy = q < n_sup ? X[n_sup + q] : 1e4f,
                sz = q < n_sup ? X[2 * n_sup + q] : 1e4f;
    float dx = __fsub_rn(sx, Q[pi]), dy = __fsub_rn(sy, Q[p + pi]), dz = __fsub_rn(sz, Q[2 * p + pi]);
    int r = identity_anchor;
    if (q_pose != nullptr) {
        const float *Rp = q_pose + ((size_t)bi * p + pi) * 16;
        const float *Rn = s_pose + ((size_t)bi * n_sup + min(q, n_sup - 1)) * 16;
        float R[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)   // R_rel = R_p R_n^T
                R[i][j] = Rp[i * 4 + 0] * Rn[j * 4 + 0] + Rp[i * 4 + 1] * Rn[j * 4 + 1] + Rp[i * 4 + 2] * Rn[j * 4 + 2];
        const float gx_ = R[0][0] * dx + R[0][1] * dy + R[0][2] * dz;
        const float gy_ = R[1][0] * dx + R[1][1] * dy + R[1][2] * dz;
        const float gz_ = R[2][0] * dx + R[2][1] * dy + R[2][2] * dz;
        dx = gx_; dy = gy_; dz = gz_;
        float best = -1e30f;
        for (int g = 0; g < na; ++g) {
            const float *A = s_anchor + g * 9;   // tr(R A_g) = sum_ij R[i][j] A[j][i]
            const float t = R[0][0] * A[0] + R[0][1] * A[3] + R[0][2] * A[6] +
                            R[1][0] * A[1] + R[1][1] * A[4] + R[1][2] * A[7] +
                            R[2][0] * A[2] + R[2][1] * A[5] + R[2][2] * A[8];
            if (t > best) { best = t; r = g; }
        }
    }
    gx[(size_t)bi * p * nn + e] = make_float4(dx, dy, dz, __int_as_float(r));
    // per-cloud flag for the grouping kernels: with identity rotations everywhere the anchor
    // permutation is the identity and its table lookups are skipped
    if (nonident != nullptr && __any(r != identity_anchor) && (threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1)
        atomicOr(nonident + bi, 1);
}

// ---------------------------------------------------------------------------------------------
// materialised kernel weights  w[b,p,a,k,n]
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float kernel_weight_exact(float gx, float gy, float gz, float kx, float ky,
                                                     float kz, float sigma) {
#pragma clang fp contract(off)
    // same operation order as torch: sum((g - rk)^2) then 1 - d/sigma, one rounding each
    const float dx = __fsub_rn(gx, kx), dy = __fsub_rn(gy, ky), dz = __fsub_rn(gz, kz);
    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    return fmaxf(__fsub_rn(1.0f, __fdiv_rn(d2, sigma)), 0.0f);
}

__global__ __launch_bounds__(256) void so3_inter_weights_kernel(
    int p, int nn, int na, int ks, float sigma, const float4 *__restrict__ gx,
    const float *__restrict__ rk, float *__restrict__ w) {
    extern __shared__ float4 s_g[];
    const int pi = blockIdx.x, bi = blockIdx.y;
    const size_t pn = ((size_t)bi * p + pi) * nn;
    for (int i = threadIdx.x; i < nn; i += blockDim.x) s_g[i] = gx[pn + i];
    __syncthreads();
    float *wp = w + pn * na * ks;
    const int total = na * ks * nn;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int n = e % nn, ak = e / nn;
        const float4 g = s_g[n];
        const float *k = rk + (size_t)ak * 3;
        wp[e] = kernel_weight_exact(g.x, g.y, g.z, k[0], k[1], k[2], sigma);
    }
}

__global__ void so3_anchor_perm_kernel(long long total, int na, const float4 *__restrict__ gx,
                                       const uint8_t *__restrict__ mult, int64_t *__restrict__ perm) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over (b,p,n,a)
    if (e >= total) return;
    const int a = (int)(e % na);
    const int r = __float_as_int(gx[e / na].w);
    perm[e] = mult[r * na + a];
}

// ---------------------------------------------------------------------------------------------
// fused grouping, forward:  out[b,c,k,p,a] = sum_n feats[b,c,idx_n,perm_n(a)] * w(p,a,k,n)
//   block = one point, 4 waves = 4 groups of KPW kernel points, lane = anchor
// ---------------------------------------------------------------------------------------------
template <int KPW, int CC>
__global__ __launch_bounds__(G_THREADS) void so3_inter_group_fwd_kernel(
    int c, int p, int n_sup, int nn, int na, int ks, float inv_sigma,
    const float *__restrict__ feats, const int32_t *__restrict__ idx,
    const float4 *__restrict__ gx, const float *__restrict__ rk,
    const uint8_t *__restrict__ mult, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *s_g = reinterpret_cast<float4 *>(smem);                 // [nn]
    int32_t *s_q = reinterpret_cast<int32_t *>(smem + 16 * (size_t)nn);  // [nn]
    uint8_t *s_mult = smem + 20 * (size_t)nn;                      // [na*na] (only if mult)

    const int pi = xcd_point(blockIdx.x, p), bi = blockIdx.y;
    const size_t pn = ((size_t)bi * p + pi) * nn;
    for (int i = threadIdx.x; i < nn; i += G_THREADS) {
        s_g[i] = gx[pn + i];
        const int q = idx[pn + i];
        s_q[i] = q < n_sup ? q : -1;   // shadow row (all zeros in the reference) -> skipped
    }
    if (mult != nullptr)
        for (int i = threadIdx.x; i < na * na; i += G_THREADS) s_mult[i] = mult[i];
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int k0 = wave * KPW;
    if (lane >= na || k0 >= ks) return;

    float kx[KPW], ky[KPW], kz[KPW];
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
        const int k = min(k0 + kk, ks - 1);
        const float *r3 = rk + ((size_t)lane * ks + k) * 3;
        kx[kk] = r3[0]; ky[kk] = r3[1]; kz[kk] = r3[2];
    }
    const float *fb = feats + (size_t)bi * c * n_sup * na;
    const size_t f_cs = (size_t)n_sup * na;
    float *ob = out + (size_t)bi * c * ks * p * na + (size_t)pi * na + lane;
    const size_t o_ks = (size_t)p * na, o_cs = (size_t)ks * p * na;

    for (int c0 = 0; c0 < c; c0 += CC) {
        float acc[CC][KPW];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
#pragma unroll
            for (int kk = 0; kk < KPW; ++kk) acc[cc][kk] = 0.f;

        for (int n = 0; n < nn; ++n) {
            const int q = s_q[n];
            if (q < 0) continue;   // wave-uniform
            const float4 g = s_g[n];
            float wv[KPW];
#pragma unroll
            for (int kk = 0; kk < KPW; ++kk) {
                const float dx = g.x - kx[kk], dy = g.y - ky[kk], dz = g.z - kz[kk];
                wv[kk] = fmaxf(1.0f - (dx * dx + dy * dy + dz * dz) * inv_sigma, 0.0f);
            }
            const int a_src = mult != nullptr ? (int)s_mult[__float_as_int(g.w) * na + lane] : lane;
            const float *f = fb + (size_t)q * na + a_src;
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
                const float fv = (c0 + cc < c) ? f[(size_t)(c0 + cc) * f_cs] : 0.f;
#pragma unroll
                for (int kk = 0; kk < KPW; ++kk) acc[cc][kk] = fmaf(fv, wv[kk], acc[cc][kk]);
            }
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc)
            if (c0 + cc < c) {
#pragma unroll
                for (int kk = 0; kk < KPW; ++kk)
                    if (k0 + kk < ks) ob[(size_t)(c0 + cc) * o_cs + (size_t)(k0 + kk) * o_ks] = acc[cc][kk];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// fused grouping, backward w.r.t. feats (scatter; fp32 atomics on 4*na-byte row segments):
//   gfeats[b,c,idx_n,perm_n(a)] += sum_k w(p,a,k,n) * gout[b,c,k,p,a]
//   block = one point; lane = anchor; the 4 waves split the CHANNELS of a chunk so that the sum
//   over k is complete before the atomic is issued (one atomic per (c,n,a)).
// ---------------------------------------------------------------------------------------------
template <int KS_MAX, int CW>
__global__ __launch_bounds__(G_THREADS) void so3_inter_group_bwd_kernel(
    int c, int p, int n_sup, int nn, int na, int ks, float inv_sigma,
    const float *__restrict__ gout, const int32_t *__restrict__ idx,
    const float4 *__restrict__ gx, const float *__restrict__ rk,
    const uint8_t *__restrict__ mult, float *__restrict__ gfeats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *s_g = reinterpret_cast<float4 *>(smem);
    int32_t *s_q = reinterpret_cast<int32_t *>(smem + 16 * (size_t)nn);
    uint8_t *s_mult = smem + 20 * (size_t)nn;

    const int pi = blockIdx.x, bi = blockIdx.y;
    const size_t pn = ((size_t)bi * p + pi) * nn;
    for (int i = threadIdx.x; i < nn; i += G_THREADS) {
        s_g[i] = gx[pn + i];
        const int q = idx[pn + i];
        s_q[i] = q < n_sup ? q : -1;
    }
    if (mult != nullptr)
        for (int i = threadIdx.x; i < na * na; i += G_THREADS) s_mult[i] = mult[i];
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane >= na) return;
    float kx[KS_MAX], ky[KS_MAX], kz[KS_MAX];
#pragma unroll
    for (int k = 0; k < KS_MAX; ++k) {
        const float *r3 = rk + ((size_t)lane * ks + min(k, ks - 1)) * 3;
        kx[k] = r3[0]; ky[k] = r3[1]; kz[k] = r3[2];
    }
    float *fb = gfeats + (size_t)bi * c * n_sup * na;
    const size_t f_cs = (size_t)n_sup * na;
    const float *ob = gout + (size_t)bi * c * ks * p * na + (size_t)pi * na + lane;
    const size_t o_ks = (size_t)p * na, o_cs = (size_t)ks * p * na;

    for (int c0 = wave * CW; c0 < c; c0 += 4 * CW) {
        float go[CW][KS_MAX];
#pragma unroll
        for (int cc = 0; cc < CW; ++cc)
#pragma unroll
            for (int k = 0; k < KS_MAX; ++k)
                go[cc][k] = (c0 + cc < c && k < ks) ? ob[(size_t)(c0 + cc) * o_cs + (size_t)k * o_ks] : 0.f;
        for (int n = 0; n < nn; ++n) {
            const int q = s_q[n];
            if (q < 0) continue;
            const float4 g = s_g[n];
            float t[CW];
#pragma unroll
            for (int cc = 0; cc < CW; ++cc) t[cc] = 0.f;
#pragma unroll
            for (int k = 0; k < KS_MAX; ++k) {
                const float dx = g.x - kx[k], dy = g.y - ky[k], dz = g.z - kz[k];
                const float wv = fmaxf(1.0f - (dx * dx + dy * dy + dz * dz) * inv_sigma, 0.0f);
#pragma unroll
                for (int cc = 0; cc < CW; ++cc) t[cc] = fmaf(wv, go[cc][k], t[cc]);
            }
            const int a_src = mult != nullptr ? (int)s_mult[__float_as_int(g.w) * na + lane] : lane;
            float *f = fb + (size_t)q * na + a_src;
#pragma unroll
            for (int cc = 0; cc < CW; ++cc)
                if (c0 + cc < c) atomicAdd(f + (size_t)(c0 + cc) * f_cs, t[cc]);
        }
    }
}

}  // namespace

extern "C" int eap_so3_prep_f32(int b, int p, int n, int nn, int na, const float *q_xyz,
                                const float *s_xyz, const int32_t *idx, const float *q_pose,
                                const float *s_pose, const float *anchors, int identity_anchor,
                                float *gx, int32_t *nonident, eap_stream_t stream) {
    if (b <= 0 || p <= 0 || nn <= 0) return 0;
    if (na > 64 || na <= 0) return eap::bad_arg("so3_prep: 1..64 anchors supported");
    if ((q_pose == nullptr) != (s_pose == nullptr)) return eap::bad_arg("so3_prep: pass both poses or none");
    if (nonident != nullptr) {
        int e = eap::hip_fail(hipMemsetAsync(nonident, 0, sizeof(int32_t) * b, eap::S(stream)), "so3_prep memset");
        if (e) return e;
    }
    dim3 grid(eap::cdiv((long long)p * nn, 256), b);
    hipLaunchKernelGGL(so3_prep_kernel, grid, dim3(256), 0, eap::S(stream), p, n, nn, na, q_xyz, s_xyz,
                       idx, q_pose, s_pose, anchors, identity_anchor, reinterpret_cast<float4 *>(gx), nonident);
    return eap::check_launch("so3_prep");
}

extern "C" int eap_so3_inter_weights_f32(int b, int p, int nn, int na, int ks, float sigma,
                                         const float *gx, const float *rk, float *w,
                                         eap_stream_t stream) {
    if (b <= 0 || p <= 0 || nn <= 0 || na <= 0 || ks <= 0) return 0;
    hipLaunchKernelGGL(so3_inter_weights_kernel, dim3(p, b), dim3(256), 16 * (size_t)nn, eap::S(stream),
                       p, nn, na, ks, sigma, reinterpret_cast<const float4 *>(gx), rk, w);
    return eap::check_launch("so3_inter_weights");
}

extern "C" int eap_so3_anchor_perm(int b, int p, int nn, int na, const float *gx, const uint8_t *mult,
                                   int64_t *perm, eap_stream_t stream) {
    const long long total = (long long)b * p * nn * na;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(so3_anchor_perm_kernel, dim3(eap::cdiv(total, 256)), dim3(256), 0, eap::S(stream),
                       total, na, reinterpret_cast<const float4 *>(gx), mult, perm);
    return eap::check_launch("so3_anchor_perm");
}

namespace {
template <int KPW>
int launch_group_fwd(int b, int c, int p, int n, int nn, int na, int ks, float sigma,
                     const float *feats, const int32_t *idx, const float *gx, const float *rk,
                     const uint8_t *mult, float *out, hipStream_t s) {
    const size_t shmem = 20 * (size_t)nn + (mult ? (size_t)na * na : 0);
    dim3 grid(p, b), block(G_THREADS);
    const float inv_sigma = 1.0f / sigma;
    const float4 *g4 = reinterpret_cast<const float4 *>(gx);
    if (c == 1)
        hipLaunchKernelGGL((so3_inter_group_fwd_kernel<KPW, 1>), grid, block, shmem, s, c, p, n, nn, na, ks, inv_sigma, feats, idx, g4, rk, mult, out);
    else if (c <= 8)
        hipLaunchKernelGGL((so3_inter_group_fwd_kernel<KPW, 4>), grid, block, shmem, s, c, p, n, nn, na, ks, inv_sigma, feats, idx, g4, rk, mult, out);
    else
        hipLaunchKernelGGL((so3_inter_group_fwd_kernel<KPW, 16>), grid, block, shmem, s, c, p, n, nn, na, ks, inv_sigma, feats, idx, g4, rk, mult, out);
    return eap::check_launch("so3_inter_group_fwd");
}
}  // namespace

// can the forward write the blocked layout for these sizes?  (>= 16 channels, the entry-list kernel's
// anchor / kernel-point limits, a multiple of 4 anchors, and either no permutation table or the
// per-cloud flag array)
extern "C" int eap_so3_inter_group_fwd_can_block(int c, int n, int na, int ks, int has_mult, int has_flag) {
    return c >= 16 && (na & 3) == 0 && eap::group_lists_supported(na, ks) && (long long)c * n * na < (1ll << 31) &&
           (!has_mult || has_flag);
}

static int group_fwd_dispatch(int blocked, int b, int c, int p, int n, int nn, int na, int ks,
                              float sigma, const float *feats, const int32_t *idx,
                              const float *gx, const float *rk, const uint8_t *mult,
                              const int32_t *nonident, float *out, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || p <= 0 || na <= 0 || ks <= 0) return 0;
    if (na > 64) return eap::bad_arg("so3_inter_group_fwd: at most 64 anchors");
    if (ks > 32) return eap::bad_arg("so3_inter_group_fwd: at most 32 kernel points (use the zpconv op)");
    if (blocked && !eap_so3_inter_group_fwd_can_block(c, n, na, ks, mult != nullptr, nonident != nullptr))
        return eap::bad_arg("so3_inter_group_fwd_xb: blocked output not available for these sizes (ask eap_so3_inter_group_fwd_can_block)");
    if (nn <= 0)
        return eap::hip_fail(hipMemsetAsync(out, 0, sizeof(float) * (size_t)b * c * ks * p * na, eap::S(stream)), "so3_inter_group_fwd memset");
    // >= 16 channels: matrix-core formulation; fewer: the VALU kernel below (a 32-channel MFMA tile
    // would be mostly padding)
    if (c >= 16) {
        // no permutation (no pose, or a cloud whose relative rotations are all the identity -- the
        // flag eap_so3_prep_f32 leaves in nonident): the two-workgroups-per-CU kernel
        // (csrc/so3_inter_lists.hip); permuted clouds: csrc/so3_inter_mfma.hip.  With a flag array
        // both kernels are launched and each skips the other's clouds -- no host round trip.
        const bool lists = eap::group_lists_supported(na, ks) && (long long)c * n * na < (1ll << 31);
        if (lists && (!mult || nonident)) {
            int e =
#ifdef EAP_EXPERIMENTS
                    eap::group_listsh_preferred(c, na, ks, blocked)
                        ? eap::group_listsh_fwd(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult ? nonident : nullptr, blocked, out, eap::S(stream)) :
                    eap::group_lists3_preferred(c, na, ks, blocked)
                        ? eap::group_lists3_fwd(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult ? nonident : nullptr, blocked, out, eap::S(stream)) :
#endif
                    eap::group_lists2_preferred(c, na, ks, blocked)
                        ? eap::group_lists2_fwd(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult ? nonident : nullptr, blocked, out, eap::S(stream))
                        : eap::group_lists_fwd(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult ? nonident : nullptr, blocked, out, eap::S(stream));
            if (e || !mult) return e;
            // the permuted clouds: the entry-list kernel of csrc/so3_inter_inv.hip in its forward mode, else the register-staged one
            e = eap::group_fwd_perm_lists(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, nonident, blocked, out, eap::S(stream));
            if (e >= 0) return e;
            return eap::group_fwd_mfma(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, nonident, 1, blocked, out, eap::S(stream));
        }
        return eap::group_fwd_mfma(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, nonident, 0, 0, out, eap::S(stream));
    }
    if (ks <= 24) return launch_group_fwd<6>(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, out, eap::S(stream));
    return launch_group_fwd<8>(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, out, eap::S(stream));
}

extern "C" int eap_so3_inter_group_fwd_f32(int b, int c, int p, int n, int nn, int na, int ks,
                                           float sigma, const float *feats, const int32_t *idx,
                                           const float *gx, const float *rk, const uint8_t *mult,
                                           const int32_t *nonident, float *out, eap_stream_t stream) {
    return group_fwd_dispatch(0, b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, nonident, out, stream);
}

// same, output transposed: out[b][p*na + a][c*ks + k] -- the plain [P*A, C*K] matrix (B^T of the contraction)
extern "C" int eap_so3_inter_group_fwd_t_f32(int b, int c, int p, int n, int nn, int na, int ks,
                                             float sigma, const float *feats, const int32_t *idx,
                                             const float *gx, const float *rk, const uint8_t *mult,
                                             const int32_t *nonident, float *out, eap_stream_t stream) {
    return group_fwd_dispatch(2, b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, nonident, out, stream);
}

// same, output blocked by anchor quads: out[b][p][a/4][c][k][4] (read by eap_gemm_f32_xb)
extern "C" int eap_so3_inter_group_fwd_xb_f32(int b, int c, int p, int n, int nn, int na, int ks,
                                              float sigma, const float *feats, const int32_t *idx,
                                              const float *gx, const float *rk, const uint8_t *mult,
                                              const int32_t *nonident, float *out, eap_stream_t stream) {
    return group_fwd_dispatch(1, b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, nonident, out, stream);
}

extern "C" int eap_so3_inter_group_fwd_valu_f32(int b, int c, int p, int n, int nn, int na, int ks,
                                                float sigma, const float *feats, const int32_t *idx,
                                                const float *gx, const float *rk, const uint8_t *mult,
                                                float *out, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || p <= 0 || na <= 0 || ks <= 0) return 0;
    if (na > 64 || ks > 32) return eap::bad_arg("so3_inter_group_fwd_valu: at most 64 anchors / 32 kernel points");
    if (nn <= 0)
        return eap::hip_fail(hipMemsetAsync(out, 0, sizeof(float) * (size_t)b * c * ks * p * na, eap::S(stream)), "so3_inter_group_fwd memset");
    if (ks <= 24) return launch_group_fwd<6>(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, out, eap::S(stream));
    return launch_group_fwd<8>(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, out, eap::S(stream));
}

extern "C" int eap_so3_inter_group_bwd_f32(int b, int c, int p, int n, int nn, int na, int ks,
                                           float sigma, const float *gout, const int32_t *idx,
                                           const float *gx, const float *rk, const uint8_t *mult,
                                           float *gfeats, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0 || na <= 0) return 0;
    if (na > 64) return eap::bad_arg("so3_inter_group_bwd: at most 64 anchors");
    if (ks > 32) return eap::bad_arg("so3_inter_group_bwd: at most 32 kernel points");
    hipStream_t s = eap::S(stream);
    int e = eap::hip_fail(hipMemsetAsync(gfeats, 0, sizeof(float) * (size_t)b * c * n * na, s), "so3_inter_group_bwd memset");
    if (e || p <= 0 || nn <= 0 || ks <= 0) return e;
    const size_t shmem = 20 * (size_t)nn + (mult ? (size_t)na * na : 0);
    const float4 *g4 = reinterpret_cast<const float4 *>(gx);
    dim3 grid(p, b), block(G_THREADS);
    if (ks <= 24)
        hipLaunchKernelGGL((so3_inter_group_bwd_kernel<24, 2>), grid, block, shmem, s, c, p, n, nn, na, ks, 1.0f / sigma, gout, idx, g4, rk, mult, gfeats);
    else
        hipLaunchKernelGGL((so3_inter_group_bwd_kernel<32, 2>), grid, block, shmem, s, c, p, n, nn, na, ks, 1.0f / sigma, gout, idx, g4, rk, mult, gfeats);
    return eap::check_launch("so3_inter_group_bwd");
}
