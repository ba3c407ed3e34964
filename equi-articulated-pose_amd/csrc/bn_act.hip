// csrc/bn_act.hip -- block-layer epilogue: BatchNorm2d (training statistics) + leaky_relu in one
// pass over the [B,C,P,A] feature tensor (SPConvNets/utils/base_so3poseconv.py:L214-221:
// `feat = self.norm(x.feats); feat = self.relu(feat)`).
//
//   forward : y = leaky(x * scale[c] + shift[c]),  scale = gamma * invstd,  shift = beta - mean * scale
//   backward: g  = gy * (pre > 0 ? 1 : slope),     pre = x * scale + shift
//             gx = gamma * invstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  xhat = (x - mean) * invstd
//             dgamma = sum_c(g * xhat),  dbeta = sum_c(g)
//
// torch runs this as four kernels that each stream the 4 GB tensor (MIOpen BN forward, leaky_relu,
// leaky_relu backward, MIOpen BN backward: 15 ms per bench step); here the forward is a statistics
// pass + one apply pass, the backward one reduction pass + one apply pass.  All of it is
// HBM-bound streaming: 16-byte loads and stores, one (cloud, channel) row segment per block,
// per-block partial sums written out and reduced in a fixed order by the caller (deterministic).
//
// CLOUD variants (the eap_bn_act_cloud_* entries): the pose heads call their unary stacks once per cloud on a point subset
// (`for i_bz in range(bz)` of ...pn_38_multi_stage.py:L706-830), i.e. BatchNorm statistics per (cloud, channel) over the
// member points only.  Batched here: scale / shift / mean / invstd / k2 / k3 are indexed [cloud][channel], the statistics
// weigh every element by its point's 0/1 membership, every point is normalised, and the backward's two correction terms
// reach the member points only (d mean / dx_i = m_i / cnt, d var / dx_i = 2 m_i (x_i - mean) / cnt).
#include "common.h"

namespace {

constexpr int TB = 256;       // threads per block
constexpr int VPT = 8;        // float4 per thread
constexpr int SEG = TB * VPT * 4;   // floats per block (8192)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block -> (segment, channel, cloud); row = x + (b*C + c)*n
__device__ __forceinline__ void block_reduce2(float a, float b, float *pa, float *pb) {
    __shared__ float sa[TB / 64], sb[TB / 64];
    a = wave_sum(a); b = wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sb[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ta = 0.f, tb = 0.f;
#pragma unroll
        for (int i = 0; i < TB / 64; ++i) { ta += sa[i]; tb += sb[i]; }
        *pa = ta; *pb = tb;
    }
}

// partial[(c*B + b)*nseg + seg] = (sum, sumsq) of the segment, relative to a per-channel pivot
// (the first element of the channel's first row) so that E[x^2] - E[x]^2 does not cancel
// MASK: every element weighs mask[cloud][point] (0 / 1), point = element / na (na a multiple of 4: a float4 never
// straddles two points)
template <bool MASK>
__global__ __launch_bounds__(TB) void bn_stats_kernel(int c, long n, int nseg, const float *__restrict__ x,
                                                     float *__restrict__ psum, float *__restrict__ psq,
                                                     const float *__restrict__ mask, int na) {
    const int seg = blockIdx.x, ci = blockIdx.y, bi = blockIdx.z;
    const float pivot = x[(size_t)ci * n];
    const float *row = x + ((size_t)bi * c + ci) * n;
    const float *mrow = MASK ? mask + (size_t)bi * (unsigned)(n / na) : nullptr;
    float s = 0.f, q = 0.f;
    const long base = (long)seg * SEG;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const long i = base + ((long)v * TB + threadIdx.x) * 4;
        if (i + 3 < n) {
            const float4 a = *reinterpret_cast<const float4 *>(row + i);
            const float d0 = a.x - pivot, d1 = a.y - pivot, d2 = a.z - pivot, d3 = a.w - pivot;
            const float ps = (d0 + d1) + (d2 + d3), pq = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            if (MASK) {
                const float m = mrow[(unsigned)i / (unsigned)na];
                s = fmaf(m, ps, s); q = fmaf(m, pq, q);
            } else {
                s += ps; q += pq;
            }
        } else {
            for (long j = i; j < n; ++j) {
                const float d = row[j] - pivot, m = MASK ? mrow[(unsigned)j / (unsigned)na] : 1.f;
                s += m * d; q += m * d * d;
            }
        }
    }
    const size_t o = ((size_t)ci * gridDim.z + bi) * nseg + seg;
    block_reduce2(s, q, psum + o, psq + o);
}

// RES: y = leaky_relu(x * scale + shift) + res -- the separable block's skip connection
// (`x.feats + skip_feature`, SPConvNets/utils/base_so3poseconv.py:L319-328) folded into the epilogue pass
template <bool RES, bool CLOUD>
__global__ __launch_bounds__(TB) void bn_act_fwd_kernel(int c, long n, float slope, const float *__restrict__ x,
                                                       const float *__restrict__ scale, const float *__restrict__ shift,
                                                       const float *__restrict__ res, float *__restrict__ y) {
    const int seg = blockIdx.x, ci = blockIdx.y, bi = blockIdx.z;
    const int si = CLOUD ? bi * c + ci : ci;
    const float sc = scale[si], sh = shift[si];
    const size_t r0 = ((size_t)bi * c + ci) * n;
    const long base = (long)seg * SEG;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const long i = base + ((long)v * TB + threadIdx.x) * 4;
        if (i + 3 < n) {
            float4 a = *reinterpret_cast<const float4 *>(x + r0 + i);
            a.x = fmaf(a.x, sc, sh); a.y = fmaf(a.y, sc, sh); a.z = fmaf(a.z, sc, sh); a.w = fmaf(a.w, sc, sh);
            a.x = a.x > 0.f ? a.x : a.x * slope; a.y = a.y > 0.f ? a.y : a.y * slope;
            a.z = a.z > 0.f ? a.z : a.z * slope; a.w = a.w > 0.f ? a.w : a.w * slope;
            if (RES) {
                const float4 r = *reinterpret_cast<const float4 *>(res + r0 + i);
                a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
            }
            *reinterpret_cast<float4 *>(y + r0 + i) = a;
        } else {
            for (long j = i; j < n; ++j) {
                const float p = fmaf(x[r0 + j], sc, sh);
                y[r0 + j] = (p > 0.f ? p : p * slope) + (RES ? res[r0 + j] : 0.f);
            }
        }
    }
}

// partials of sum(g) and sum(g * xhat)
template <bool CLOUD>
__global__ __launch_bounds__(TB) void bn_act_bwd_reduce_kernel(int c, long n, int nseg, float slope,
                                                              const float *__restrict__ gy, const float *__restrict__ x,
                                                              const float *__restrict__ scale, const float *__restrict__ shift,
                                                              const float *__restrict__ mean, const float *__restrict__ invstd,
                                                              float *__restrict__ pg, float *__restrict__ pgx) {
    const int seg = blockIdx.x, ci = blockIdx.y, bi = blockIdx.z;
    const int si = CLOUD ? bi * c + ci : ci;
    const float sc = scale[si], sh = shift[si], mu = mean[si], is = invstd[si];
    const size_t r0 = ((size_t)bi * c + ci) * n;
    const long base = (long)seg * SEG;
    float s = 0.f, q = 0.f;
    auto one = [&](float g, float xv) {
        const float pre = fmaf(xv, sc, sh);
        const float gg = pre > 0.f ? g : g * slope;
        s += gg;
        q += gg * ((xv - mu) * is);
    };
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const long i = base + ((long)v * TB + threadIdx.x) * 4;
        if (i + 3 < n) {
            const float4 g = *reinterpret_cast<const float4 *>(gy + r0 + i);
            const float4 a = *reinterpret_cast<const float4 *>(x + r0 + i);
            one(g.x, a.x); one(g.y, a.y); one(g.z, a.z); one(g.w, a.w);
        } else {
            for (long j = i; j < n; ++j) one(gy[r0 + j], x[r0 + j]);
        }
    }
    const size_t o = ((size_t)ci * gridDim.z + bi) * nseg + seg;
    block_reduce2(s, q, pg + o, pgx + o);
}

// gx = k1[c] * g - k2[c] - xhat * k3[c]   with k1 = gamma*invstd, k2 = k1*mean(g), k3 = k1*mean(g*xhat)
// CLOUD: coefficients per (cloud, channel); with a mask the two correction terms reach member points only
template <bool CLOUD>
__global__ __launch_bounds__(TB) void bn_act_bwd_apply_kernel(int c, long n, float slope, const float *__restrict__ gy,
                                                             const float *__restrict__ x, const float *__restrict__ scale,
                                                             const float *__restrict__ shift, const float *__restrict__ mean,
                                                             const float *__restrict__ invstd, const float *__restrict__ k2,
                                                             const float *__restrict__ k3, float *__restrict__ gx,
                                                             const float *__restrict__ mask, int na) {
    const int seg = blockIdx.x, ci = blockIdx.y, bi = blockIdx.z;
    const int si = CLOUD ? bi * c + ci : ci;
    const float sc = scale[si], sh = shift[si], mu = mean[si], is = invstd[si], c2 = k2[si], c3 = k3[si];
    const size_t r0 = ((size_t)bi * c + ci) * n;
    const float *mrow = (CLOUD && mask) ? mask + (size_t)bi * (unsigned)(n / na) : nullptr;
    const long base = (long)seg * SEG;
    auto one = [&](float g, float xv, float m) {
        const float pre = fmaf(xv, sc, sh);
        const float gg = pre > 0.f ? g : g * slope;
        if (CLOUD) return fmaf(-m, fmaf((xv - mu) * is, c3, c2), gg * sc);
        return fmaf(gg, sc, -c2) - ((xv - mu) * is) * c3;      // scale = gamma * invstd = k1
    };
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
        const long i = base + ((long)v * TB + threadIdx.x) * 4;
        if (i + 3 < n) {
            const float4 g = *reinterpret_cast<const float4 *>(gy + r0 + i);
            const float4 a = *reinterpret_cast<const float4 *>(x + r0 + i);
            const float m = mrow ? mrow[(unsigned)i / (unsigned)na] : 1.f;
            *reinterpret_cast<float4 *>(gx + r0 + i) = make_float4(one(g.x, a.x, m), one(g.y, a.y, m), one(g.z, a.z, m), one(g.w, a.w, m));
        } else {
            for (long j = i; j < n; ++j) gx[r0 + j] = one(gy[r0 + j], x[r0 + j], mrow ? mrow[(unsigned)j / (unsigned)na] : 1.f);
        }
    }
}

// The same pass for a row that is [points][na] (na a multiple of 4), which ALSO leaves the largest magnitude of gx per (cloud,
// channel, anchor) in rowmax [b, c, na] (bit patterns of non-negative floats, zero-initialised by the caller; unsigned maximum =
// float maximum: order-independent, so the atomics make nothing run-dependent).  The consumer -- the stored-operand split of the
// dense backward product, csrc/so3_dense.hip -- otherwise reads the whole gradient once more just for these numbers.  Thread t of
// the nq * (256 / nq) active ones walks the float4 pieces t, t + T, t + 2T, ... of its segment: T is a multiple of nq = na / 4,
// so a thread meets ONE anchor quad and keeps its four maxima in registers.
constexpr int RM_PIECES = 16;           // pieces per thread and block
__global__ __launch_bounds__(256) void bn_act_bwd_apply_rowmax_kernel(int c, long n4, int nq, float slope, const float4 *__restrict__ gy,
                                                                      const float4 *__restrict__ x, const float *__restrict__ scale,
                                                                      const float *__restrict__ shift, const float *__restrict__ mean,
                                                                      const float *__restrict__ invstd, const float *__restrict__ k2,
                                                                      const float *__restrict__ k3, float4 *__restrict__ gx,
                                                                      unsigned *__restrict__ rowmax) {
    __shared__ unsigned s_max[256][4];
    const int ci = blockIdx.y, bi = blockIdx.z, t = threadIdx.x;
    const int T = nq * (256 / nq);
    const float sc = scale[ci], sh = shift[ci], mu = mean[ci], is = invstd[ci], c2 = k2[ci], c3 = k3[ci];
    const size_t r0 = ((size_t)bi * c + ci) * (size_t)n4;
    const long base = (long)blockIdx.x * RM_PIECES * T;
    auto one = [&](float g, float xv) {
        const float pre = fmaf(xv, sc, sh);
        const float gg = pre > 0.f ? g : g * slope;
        return fmaf(gg, sc, -c2) - ((xv - mu) * is) * c3;
    };
    unsigned m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    if (t < T) {
        float4 g[RM_PIECES], a[RM_PIECES];
#pragma unroll
        for (int v = 0; v < RM_PIECES; ++v) {
            const long i = base + (long)v * T + t;
            if (i < n4) { g[v] = gy[r0 + i]; a[v] = x[r0 + i]; }
        }
#pragma unroll
        for (int v = 0; v < RM_PIECES; ++v) {
            const long i = base + (long)v * T + t;
            if (i < n4) {
                const float4 o = make_float4(one(g[v].x, a[v].x), one(g[v].y, a[v].y), one(g[v].z, a[v].z), one(g[v].w, a[v].w));
                gx[r0 + i] = o;
                m0 = max(m0, __float_as_uint(o.x) & 0x7fffffffu); m1 = max(m1, __float_as_uint(o.y) & 0x7fffffffu);
                m2 = max(m2, __float_as_uint(o.z) & 0x7fffffffu); m3 = max(m3, __float_as_uint(o.w) & 0x7fffffffu);
            }
        }
    }
    s_max[t][0] = m0; s_max[t][1] = m1; s_max[t][2] = m2; s_max[t][3] = m3;
    __syncthreads();
    if (t < 4 * nq) {                                           // anchor t = 4 (t >> 2) + (t & 3): the threads of quad t >> 2 are t >> 2, + nq, ...
        unsigned v = 0;
        for (int u = t >> 2; u < T; u += nq) v = max(v, s_max[u][t & 3]);
        if (v) atomicMax(rowmax + ((size_t)bi * c + ci) * (4 * nq) + t, v);
    }
}

// The backward's reduction pass when the layer kept its OUTPUT y' = leaky(gamma xhat + beta) instead of its input (the conv + BatchNorm
// node of vgtk/so3conv/functional.py: the pre-activation is recovered from y', leaky_relu with a positive slope being invertible):
//     pre = y' > 0 ? y' : y' / slope,  xhat = (pre - beta) / gamma,  g = y' > 0 ? dy' : slope dy'
// -> partials of sum(g) and sum(g xhat) per block at [(c B + b) nblk + block] (reduced in a fixed order by the caller), and per (cloud,
// channel, anchor) the largest |g| and |xhat| (gmax, xmax [b, c, na]: float bit patterns, zero-initialised here; unsigned maximum = float
// maximum, order-independent) -- from them the caller bounds the rows of gx = k1 g - k2 - k3 xhat for the stored-operand split that forms gx
// on the way in (csrc/so3_dense.hip dense_split_kernel<true>).  Rows are [points][na], na a multiple of 4; thread t walks ONE anchor quad.
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_fromy_kernel(int c, long n4, int nq, int nblk, float slope, float inv_slope, const float4 *__restrict__ gy,
                                                                      const float4 *__restrict__ y, const float *__restrict__ beta,
                                                                      const float *__restrict__ inv_gamma, float *__restrict__ pg, float *__restrict__ pgx,
                                                                      unsigned *__restrict__ gmax, unsigned *__restrict__ xmax) {
    __shared__ unsigned s_max[256][8];
    const int ci = blockIdx.y, bi = blockIdx.z, t = threadIdx.x;
    const int T = nq * (256 / nq);
    const float be = beta[ci], ig = inv_gamma[ci];
    const size_t r0 = ((size_t)bi * c + ci) * (size_t)n4;
    const long base = (long)blockIdx.x * RM_PIECES * T;
    float s = 0.f, q = 0.f;
    unsigned mg[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0};
    if (t < T) {
        float4 g[RM_PIECES], a[RM_PIECES];
#pragma unroll
        for (int v = 0; v < RM_PIECES; ++v) {
            const long i = base + (long)v * T + t;
            if (i < n4) { g[v] = gy[r0 + i]; a[v] = y[r0 + i]; }
        }
#pragma unroll
        for (int v = 0; v < RM_PIECES; ++v) {
            const long i = base + (long)v * T + t;
            if (i < n4) {
                const float gv[4] = {g[v].x, g[v].y, g[v].z, g[v].w}, yv[4] = {a[v].x, a[v].y, a[v].z, a[v].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool pos = yv[e] > 0.f;
                    const float gg = pos ? gv[e] : gv[e] * slope, xh = ((pos ? yv[e] : yv[e] * inv_slope) - be) * ig;
                    s += gg;
                    q = fmaf(gg, xh, q);
                    mg[e] = max(mg[e], __float_as_uint(gg) & 0x7fffffffu);
                    mx[e] = max(mx[e], __float_as_uint(xh) & 0x7fffffffu);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s_max[t][e] = mg[e]; s_max[t][4 + e] = mx[e]; }
    const size_t o = ((size_t)ci * gridDim.z + bi) * nblk + blockIdx.x;
    block_reduce2(s, q, pg + o, pgx + o);                       // (its __syncthreads also publishes s_max)
    if (t < 4 * nq) {
        unsigned vg = 0, vx = 0;
        for (int u = t >> 2; u < T; u += nq) { vg = max(vg, s_max[u][t & 3]); vx = max(vx, s_max[u][4 + (t & 3)]); }
        const size_t at = ((size_t)bi * c + ci) * (4 * nq) + t;
        if (vg) atomicMax(gmax + at, vg);
        if (vx) atomicMax(xmax + at, vx);
    }
}

inline int nseg_of(long n) { return (int)((n + SEG - 1) / SEG); }
inline bool ok_dims(int b, int c, long n) { return b > 0 && c > 0 && n > 0 && c <= 65535 && b <= 65535; }

}  // namespace

extern "C" int eap_bn_act_segments(int64_t n) { return nseg_of((long)n); }

extern "C" int eap_bn_stats_f32(int b, int c, int64_t n, const float *x, float *psum, float *psq, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!ok_dims(b, c, n) || (n & 3) != 0) return eap::bad_arg("bn_stats: row length must be a multiple of 4; at most 65535 channels / clouds");
    const int nseg = nseg_of(n);
    hipLaunchKernelGGL(bn_stats_kernel<false>, dim3(nseg, c, b), dim3(TB), 0, eap::S(stream), c, (long)n, nseg, x, psum, psq, (const float *)nullptr, 1);
    return eap::check_launch("bn_stats");
}

extern "C" int eap_bn_act_fwd_f32(int b, int c, int64_t n, float slope, const float *x, const float *scale,
                                  const float *shift, float *y, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!ok_dims(b, c, n) || (n & 3) != 0) return eap::bad_arg("bn_act_fwd: row length must be a multiple of 4; at most 65535 channels / clouds");
    hipLaunchKernelGGL((bn_act_fwd_kernel<false, false>), dim3(nseg_of(n), c, b), dim3(TB), 0, eap::S(stream), c, (long)n, slope, x, scale, shift,
                       (const float *)nullptr, y);
    return eap::check_launch("bn_act_fwd");
}

extern "C" int eap_bn_act_add_fwd_f32(int b, int c, int64_t n, float slope, const float *x, const float *scale,
                                      const float *shift, const float *res, float *y, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!ok_dims(b, c, n) || (n & 3) != 0) return eap::bad_arg("bn_act_add_fwd: row length must be a multiple of 4; at most 65535 channels / clouds");
    hipLaunchKernelGGL((bn_act_fwd_kernel<true, false>), dim3(nseg_of(n), c, b), dim3(TB), 0, eap::S(stream), c, (long)n, slope, x, scale, shift, res, y);
    return eap::check_launch("bn_act_add_fwd");
}

extern "C" int eap_bn_act_bwd_reduce_f32(int b, int c, int64_t n, float slope, const float *gy, const float *x,
                                         const float *scale, const float *shift, const float *mean,
                                         const float *invstd, float *pg, float *pgx, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!ok_dims(b, c, n) || (n & 3) != 0) return eap::bad_arg("bn_act_bwd_reduce: row length must be a multiple of 4; at most 65535 channels / clouds");
    const int nseg = nseg_of(n);
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<false>, dim3(nseg, c, b), dim3(TB), 0, eap::S(stream), c, (long)n, nseg, slope, gy, x,
                       scale, shift, mean, invstd, pg, pgx);
    return eap::check_launch("bn_act_bwd_reduce");
}

extern "C" int eap_bn_act_bwd_apply_f32(int b, int c, int64_t n, float slope, const float *gy, const float *x,
                                        const float *scale, const float *shift, const float *mean,
                                        const float *invstd, const float *k2, const float *k3, float *gx,
                                        eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!ok_dims(b, c, n) || (n & 3) != 0) return eap::bad_arg("bn_act_bwd_apply: row length must be a multiple of 4; at most 65535 channels / clouds");
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<false>, dim3(nseg_of(n), c, b), dim3(TB), 0, eap::S(stream), c, (long)n, slope, gy, x,
                       scale, shift, mean, invstd, k2, k3, gx, (const float *)nullptr, 1);
    return eap::check_launch("bn_act_bwd_apply");
}

extern "C" int eap_bn_act_bwd_apply_rowmax_f32(int b, int c, int64_t n, int na, float slope, const float *gy, const float *x,
                                               const float *scale, const float *shift, const float *mean,
                                               const float *invstd, const float *k2, const float *k3, float *gx, uint32_t *rowmax,
                                               eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!ok_dims(b, c, n) || na <= 0 || (na & 3) != 0 || na > 256 || n % na != 0 ||
        ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gx)) & 15) != 0)
        return eap::bad_arg("bn_act_bwd_apply_rowmax: rows of [points][na], na a multiple of 4 up to 256, 16-byte aligned tensors");
    hipStream_t s = eap::S(stream);
    if (int e = eap::hip_fail(hipMemsetAsync(rowmax, 0, sizeof(uint32_t) * (size_t)b * c * na, s), "bn_act_bwd_apply_rowmax memset")) return e;
    const int nq = na / 4, T = nq * (256 / nq);
    const long n4 = (long)(n / 4);
    hipLaunchKernelGGL(bn_act_bwd_apply_rowmax_kernel, dim3((unsigned)((n4 + (long)RM_PIECES * T - 1) / ((long)RM_PIECES * T)), c, b), dim3(256), 0, s,
                       c, n4, nq, slope, reinterpret_cast<const float4 *>(gy), reinterpret_cast<const float4 *>(x), scale, shift, mean, invstd, k2, k3,
                       reinterpret_cast<float4 *>(gx), rowmax);
    return eap::check_launch("bn_act_bwd_apply_rowmax");
}

// blocks per (cloud, channel) row of eap_bn_act_bwd_reduce_fromy_f32: the partial arrays are [c][b * blocks]
extern "C" int eap_bn_act_fromy_blocks(int64_t n, int na) {
    if (na <= 0 || (na & 3) != 0 || na > 256) return 0;
    const int nq = na / 4, T = nq * (256 / nq);
    const long n4 = (long)(n / 4);
    return (int)((n4 + (long)RM_PIECES * T - 1) / ((long)RM_PIECES * T));
}

// see bn_act_bwd_reduce_fromy_kernel: gy, y [b,c,n] with rows [points][na]; beta, inv_gamma [c]; pg, pgx float [c][b * eap_bn_act_fromy_blocks(n, na)];
// gmax, xmax uint32 [b,c,na].  SPConvNets/utils/base_so3poseconv.py:L214-221 (autograd of norm + relu).
extern "C" int eap_bn_act_bwd_reduce_fromy_f32(int b, int c, int64_t n, int na, float slope, const float *gy, const float *y, const float *beta,
                                               const float *inv_gamma, float *pg, float *pgx, uint32_t *gmax, uint32_t *xmax, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!ok_dims(b, c, n) || na <= 0 || (na & 3) != 0 || na > 256 || n % na != 0 || !(slope > 0.f) ||
        ((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(y)) & 15) != 0)
        return eap::bad_arg("bn_act_bwd_reduce_fromy: rows of [points][na], na a multiple of 4 up to 256, 16-byte aligned tensors, a positive slope");
    hipStream_t s = eap::S(stream);
    if (int e = eap::hip_fail(hipMemsetAsync(gmax, 0, sizeof(uint32_t) * (size_t)b * c * na, s), "bn_act_bwd_reduce_fromy memset")) return e;
    if (int e = eap::hip_fail(hipMemsetAsync(xmax, 0, sizeof(uint32_t) * (size_t)b * c * na, s), "bn_act_bwd_reduce_fromy memset")) return e;
    const int nq = na / 4, nblk = eap_bn_act_fromy_blocks(n, na);
    hipLaunchKernelGGL(bn_act_bwd_reduce_fromy_kernel, dim3((unsigned)nblk, c, b), dim3(256), 0, s, c, (long)(n / 4), nq, nblk, slope, 1.0f / slope,
                       reinterpret_cast<const float4 *>(gy), reinterpret_cast<const float4 *>(y), beta, inv_gamma, pg, pgx, gmax, xmax);
    return eap::check_launch("bn_act_bwd_reduce_fromy");
}

// ---- per-cloud statistics over a point subset (the pose heads' batched per-cloud calls) ---------------------------
namespace {
inline int cloud_dims(const char *who, int b, int c, long n, int na, bool masked) {
    char buf[200];
    if (!ok_dims(b, c, n) || (n & 3) != 0 || n >= (1l << 31)) {
        snprintf(buf, sizeof(buf), "%s: row length must be a multiple of 4 below 2^31; at most 65535 channels / clouds", who);
        return eap::bad_arg(buf);
    }
    if (masked && (na <= 0 || (na & 3) != 0 || n % na != 0)) {
        snprintf(buf, sizeof(buf), "%s: with a mask a row is [points][na], na a multiple of 4", who);
        return eap::bad_arg(buf);
    }
    return 0;
}
}  // namespace

extern "C" int eap_bn_stats_masked_f32(int b, int c, int64_t n, int na, const float *x, const float *mask, float *psum, float *psq,
                                       eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (!mask) return eap::bad_arg("bn_stats_masked: mask is null");
    if (int e = cloud_dims("bn_stats_masked", b, c, (long)n, na, true)) return e;
    const int nseg = nseg_of(n);
    hipLaunchKernelGGL(bn_stats_kernel<true>, dim3(nseg, c, b), dim3(TB), 0, eap::S(stream), c, (long)n, nseg, x, psum, psq, mask, na);
    return eap::check_launch("bn_stats_masked");
}

extern "C" int eap_bn_act_cloud_fwd_f32(int b, int c, int64_t n, float slope, const float *x, const float *scale, const float *shift,
                                        float *y, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (int e = cloud_dims("bn_act_cloud_fwd", b, c, (long)n, 0, false)) return e;
    hipLaunchKernelGGL((bn_act_fwd_kernel<false, true>), dim3(nseg_of(n), c, b), dim3(TB), 0, eap::S(stream), c, (long)n, slope, x, scale, shift,
                       (const float *)nullptr, y);
    return eap::check_launch("bn_act_cloud_fwd");
}

extern "C" int eap_bn_act_cloud_bwd_reduce_f32(int b, int c, int64_t n, float slope, const float *gy, const float *x, const float *scale,
                                               const float *shift, const float *mean, const float *invstd, float *pg, float *pgx,
                                               eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (int e = cloud_dims("bn_act_cloud_bwd_reduce", b, c, (long)n, 0, false)) return e;
    const int nseg = nseg_of(n);
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<true>, dim3(nseg, c, b), dim3(TB), 0, eap::S(stream), c, (long)n, nseg, slope, gy, x,
                       scale, shift, mean, invstd, pg, pgx);
    return eap::check_launch("bn_act_cloud_bwd_reduce");
}

extern "C" int eap_bn_act_cloud_bwd_apply_f32(int b, int c, int64_t n, int na, float slope, const float *gy, const float *x,
                                              const float *scale, const float *shift, const float *mean, const float *invstd,
                                              const float *k2, const float *k3, const float *mask, float *gx, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (int e = cloud_dims("bn_act_cloud_bwd_apply", b, c, (long)n, na, mask != nullptr)) return e;
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<true>, dim3(nseg_of(n), c, b), dim3(TB), 0, eap::S(stream), c, (long)n, slope, gy, x,
                       scale, shift, mean, invstd, k2, k3, gx, mask, mask ? na : 1);
    return eap::check_launch("bn_act_cloud_bwd_apply");
}
