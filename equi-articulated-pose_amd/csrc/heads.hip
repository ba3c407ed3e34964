// csrc/heads.hip -- the anchor-axis reductions of the heads that read the backbone's [B,C,N,A] feature map
// (SURVEY.md section 8(f) rows 2 and 3): one pass over the map instead of the reference's 3-4 torch passes.
//
//   anchor attention pooling   InvPPOutBlockOurs.forward, SPConvNets/utils/base_so3conv.py:L905-912:
//        conf[b,n,a] = softmax_a(logit[b,n,a] * T);   out[b,c,n] = sum_a x[b,c,n,a] conf[b,n,a]
//   slot-masked point mean     the pose head's masked averages over the points of every slot at once
//        (SPConvNets/models/model_utils.py:L470-472 + L484 / L549-552, called once per slot by
//        ...pn_38_multi_stage.py:L695-1015):   out[b,s,c,a] = sum_n m[b,s,n] x[b,c,n,a] / max(sum_n m[b,s,n], 1e-8)
//
// Mapping: 16 lanes per point (15 anchor quads of a 60-anchor row + one idle lane), 4 consecutive points per
// wave: one 16-byte load per lane covers 960 contiguous bytes of x; reductions over the anchors are 4 shuffle
// steps inside the 16-lane group.  HBM-bound: x is read once (and dx written once in the backward).
#include "common.h"

namespace {

__device__ __forceinline__ float group16_sum(float v) {
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float group16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 8)); v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 1));
    return v;
}

// block = 4 waves = 16 consecutive points of one cloud; every wave walks all channels of its 4 points
template <bool BWD>
__global__ __launch_bounds__(256) void anchor_attn_pool_kernel(int c, int n, int na, float temperature,
                                                              const float *__restrict__ x, const float *__restrict__ logits,
                                                              const float *__restrict__ g, float *__restrict__ out,
                                                              float *__restrict__ conf_out, float *__restrict__ dx,
                                                              float *__restrict__ dlogits) {
    const int bi = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pt = blockIdx.x * 16 + wave * 4 + (lane >> 4), quad = lane & 15, nq = na >> 2;
    const bool on = pt < n && quad < nq;
    const size_t row = ((size_t)bi * n + min(pt, n - 1)) * na + 4 * min(quad, nq - 1);
    // softmax over the anchors of this point
    float4 l = *reinterpret_cast<const float4 *>(logits + row);
    l.x *= temperature; l.y *= temperature; l.z *= temperature; l.w *= temperature;
    const float neg = -3.0e38f;
    if (quad >= nq) l = make_float4(neg, neg, neg, neg);
    const float m = group16_max(fmaxf(fmaxf(l.x, l.y), fmaxf(l.z, l.w)));
    float4 e = make_float4(__expf(l.x - m), __expf(l.y - m), __expf(l.z - m), __expf(l.w - m));
    if (quad >= nq) e = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv = 1.0f / group16_sum(e.x + e.y + e.z + e.w);
    const float4 cf = make_float4(e.x * inv, e.y * inv, e.z * inv, e.w * inv);
    if (!BWD && on && conf_out) *reinterpret_cast<float4 *>(conf_out + row) = cf;
    const size_t cs = (size_t)n * na;                        // channel stride of x
    const float *xp = x + (size_t)bi * c * cs + ((size_t)min(pt, n - 1) * na + 4 * min(quad, nq - 1));
    if (!BWD) {
        float *op = out + (size_t)bi * c * n + min(pt, n - 1);
        for (int ci = 0; ci < c; ++ci) {
            const float4 v = *reinterpret_cast<const float4 *>(xp + (size_t)ci * cs);
            const float s = group16_sum(quad < nq ? v.x * cf.x + v.y * cf.y + v.z * cf.z + v.w * cf.w : 0.f);
            if (on && quad == 0) op[(size_t)ci * n] = s;
        }
    } else {
        const float *gp = g + (size_t)bi * c * n + min(pt, n - 1);
        float *dxp = dx + (size_t)bi * c * cs + ((size_t)min(pt, n - 1) * na + 4 * min(quad, nq - 1));
        float4 dc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ci = 0; ci < c; ++ci) {
            const float4 v = *reinterpret_cast<const float4 *>(xp + (size_t)ci * cs);
            const float gv = gp[(size_t)ci * n];
            dc.x = fmaf(gv, v.x, dc.x); dc.y = fmaf(gv, v.y, dc.y); dc.z = fmaf(gv, v.z, dc.z); dc.w = fmaf(gv, v.w, dc.w);
            if (on) *reinterpret_cast<float4 *>(dxp + (size_t)ci * cs) = make_float4(gv * cf.x, gv * cf.y, gv * cf.z, gv * cf.w);
        }
        // softmax backward: dlogit = T conf (dconf - sum_a conf dconf)
        const float s = group16_sum(quad < nq ? cf.x * dc.x + cf.y * dc.y + cf.z * dc.z + cf.w * dc.w : 0.f);
        if (on) *reinterpret_cast<float4 *>(dlogits + row) = make_float4(temperature * cf.x * (dc.x - s), temperature * cf.y * (dc.y - s),
                                                                         temperature * cf.z * (dc.z - s), temperature * cf.w * (dc.w - s));
    }
}

constexpr int MAXS = 8;     // slots per launch

// forward: block = (channel, cloud); 4 waves stride over the points, 4 points per wave-iteration; per slot one
// float4 accumulator per lane; the 4 point groups and 4 waves are folded through LDS at the end (fixed order)
__global__ __launch_bounds__(256) void slot_mean_fwd_kernel(int c, int n, int na, int ns, const float *__restrict__ x,
                                                            const float *__restrict__ m, const float *__restrict__ inv_den,
                                                            float *__restrict__ out) {
    __shared__ float4 s_part[MAXS][16][16];                   // [slot][wave * 4 + point group][quad]
    const int ci = blockIdx.x, bi = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, quad = lane & 15, nq = na >> 2;
    float4 acc[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) acc[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *xp = x + ((size_t)bi * c + ci) * n * na + 4 * min(quad, nq - 1);
    const float *mp = m + (size_t)bi * ns * n;
    for (int p0 = wave * 4 + grp; p0 < n; p0 += 16) {
        const float4 v = *reinterpret_cast<const float4 *>(xp + (size_t)p0 * na);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < ns) {
                const float w = mp[(size_t)s * n + p0];
                acc[s].x = fmaf(w, v.x, acc[s].x); acc[s].y = fmaf(w, v.y, acc[s].y);
                acc[s].z = fmaf(w, v.z, acc[s].z); acc[s].w = fmaf(w, v.w, acc[s].w);
            }
    }
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
        if (s < ns) s_part[s][wave * 4 + grp][quad] = acc[s];
    __syncthreads();
    for (int e = threadIdx.x; e < ns * nq; e += 256) {
        const int s = e / nq, q = e - s * nq;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < 16; ++j) { const float4 u = s_part[s][j][q]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        const float d = inv_den[(size_t)bi * ns + s];
        *reinterpret_cast<float4 *>(out + (((size_t)bi * ns + s) * c + ci) * na + 4 * q) = make_float4(t.x * d, t.y * d, t.z * d, t.w * d);
    }
}

// backward w.r.t. x: dx[b,c,n,a] = sum_s g[b,s,c,a] m[b,s,n] inv_den[b,s]
__global__ __launch_bounds__(256) void slot_mean_bwd_kernel(int c, int n, int na, int ns, const float *__restrict__ g,
                                                            const float *__restrict__ m, const float *__restrict__ inv_den,
                                                            float *__restrict__ dx) {
    const int ci = blockIdx.x, bi = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, quad = lane & 15, nq = na >> 2;
    float4 gv[MAXS];
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        gv[s] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < ns && quad < nq) {
            const float d = inv_den[(size_t)bi * ns + s];
            const float4 t = *reinterpret_cast<const float4 *>(g + (((size_t)bi * ns + s) * c + ci) * na + 4 * quad);
            gv[s] = make_float4(t.x * d, t.y * d, t.z * d, t.w * d);
        }
    }
    float *dp = dx + ((size_t)bi * c + ci) * n * na + 4 * min(quad, nq - 1);
    const float *mp = m + (size_t)bi * ns * n;
    for (int p0 = wave * 4 + grp; p0 < n; p0 += 16) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int s = 0; s < MAXS; ++s)
            if (s < ns) {
                const float w = mp[(size_t)s * n + p0];
                t.x = fmaf(w, gv[s].x, t.x); t.y = fmaf(w, gv[s].y, t.y); t.z = fmaf(w, gv[s].z, t.z); t.w = fmaf(w, gv[s].w, t.w);
            }
        if (quad < nq) *reinterpret_cast<float4 *>(dp + (size_t)p0 * na) = t;
    }
}

// masked max over the points, the pose head's 'max' pooling on a point subset (`(x * mask).max(2)`: SO3OutBlockRWithMask /
// SO3OutBlockRTWithMaskSep with pooling_method = 'max', SPConvNets/models/model_utils.py:L79-80, L470-484): out[b,c,a] = max_p
// mask[b,p] * x[b,c,p,a] and the point that attains it (the lowest index among equals: deterministic).  One pass over x.
// block = (channel, cloud); lane = (point group of 4, anchor quad); groups and waves are folded through LDS in a fixed order
__global__ __launch_bounds__(256) void masked_max_fwd_kernel(int c, int n, int na, const float *__restrict__ x, const float *__restrict__ m,
                                                             float *__restrict__ out, int32_t *__restrict__ arg) {
    __shared__ float4 s_val[16][16];
    __shared__ int4 s_idx[16][16];
    const int ci = blockIdx.x, bi = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, quad = lane & 15, nq = na >> 2;
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int4 at = make_int4(0, 0, 0, 0);
    const float *xp = x + ((size_t)bi * c + ci) * n * na + 4 * min(quad, nq - 1);
    const float *mp = m + (size_t)bi * n;
    for (int p0 = wave * 4 + grp; p0 < n; p0 += 16) {                      // ascending within a lane: ">" keeps the first of equals
        const float4 v = *reinterpret_cast<const float4 *>(xp + (size_t)p0 * na);
        const float w = mp[p0];
        const float a0 = w * v.x, a1 = w * v.y, a2 = w * v.z, a3 = w * v.w;
        if (a0 > best.x) { best.x = a0; at.x = p0; }
        if (a1 > best.y) { best.y = a1; at.y = p0; }
        if (a2 > best.z) { best.z = a2; at.z = p0; }
        if (a3 > best.w) { best.w = a3; at.w = p0; }
    }
    s_val[wave * 4 + grp][quad] = best;
    s_idx[wave * 4 + grp][quad] = at;
    __syncthreads();
    if (threadIdx.x < nq) {
        const int q = threadIdx.x;
        float4 bv = s_val[0][q];
        int4 bi4 = s_idx[0][q];
        for (int j = 1; j < 16; ++j) {
            const float4 u = s_val[j][q];
            const int4 k = s_idx[j][q];
            if (u.x > bv.x || (u.x == bv.x && k.x < bi4.x)) { bv.x = u.x; bi4.x = k.x; }
            if (u.y > bv.y || (u.y == bv.y && k.y < bi4.y)) { bv.y = u.y; bi4.y = k.y; }
            if (u.z > bv.z || (u.z == bv.z && k.z < bi4.z)) { bv.z = u.z; bi4.z = k.z; }
            if (u.w > bv.w || (u.w == bv.w && k.w < bi4.w)) { bv.w = u.w; bi4.w = k.w; }
        }
        *reinterpret_cast<float4 *>(out + ((size_t)bi * c + ci) * na + 4 * q) = bv;
        *reinterpret_cast<int4 *>(arg + ((size_t)bi * c + ci) * na + 4 * q) = bi4;
    }
}

// dx[b,c,p,a] = (p == arg[b,c,a]) * mask[b,p] * g[b,c,a], written for every point (no memset + scatter)
__global__ __launch_bounds__(256) void masked_max_bwd_kernel(int c, int n, int na, const float *__restrict__ g, const int32_t *__restrict__ arg,
                                                             const float *__restrict__ m, float *__restrict__ dx) {
    const int ci = blockIdx.x, bi = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 4, quad = lane & 15, nq = na >> 2;
    if (quad >= nq) return;
    const float4 gv = *reinterpret_cast<const float4 *>(g + ((size_t)bi * c + ci) * na + 4 * quad);
    const int4 at = *reinterpret_cast<const int4 *>(arg + ((size_t)bi * c + ci) * na + 4 * quad);
    float *dp = dx + ((size_t)bi * c + ci) * n * na + 4 * quad;
    const float *mp = m + (size_t)bi * n;
    for (int p0 = wave * 4 + grp; p0 < n; p0 += 16) {
        const float w = mp[p0];
        *reinterpret_cast<float4 *>(dp + (size_t)p0 * na) =
            make_float4(p0 == at.x ? w * gv.x : 0.f, p0 == at.y ? w * gv.y : 0.f, p0 == at.z ? w * gv.z : 0.f, p0 == at.w ? w * gv.w : 0.f);
    }
}

}  // namespace

extern "C" int eap_anchor_attn_pool_fwd_f32(int b, int c, int n, int na, float temperature, const float *x, const float *logits,
                                            float *out, float *conf, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (na <= 0 || na > 64 || (na & 3) != 0 || b > 65535) return eap::bad_arg("anchor_attn_pool: anchors must be a multiple of 4, at most 64; b <= 65535");
    hipLaunchKernelGGL(anchor_attn_pool_kernel<false>, dim3(eap::cdiv(n, 16), b), dim3(256), 0, eap::S(stream), c, n, na, temperature, x,
                       logits, nullptr, out, conf, nullptr, nullptr);
    return eap::check_launch("anchor_attn_pool_fwd");
}

extern "C" int eap_anchor_attn_pool_bwd_f32(int b, int c, int n, int na, float temperature, const float *x, const float *logits,
                                            const float *g, float *dx, float *dlogits, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (na <= 0 || na > 64 || (na & 3) != 0 || b > 65535) return eap::bad_arg("anchor_attn_pool: anchors must be a multiple of 4, at most 64; b <= 65535");
    hipLaunchKernelGGL(anchor_attn_pool_kernel<true>, dim3(eap::cdiv(n, 16), b), dim3(256), 0, eap::S(stream), c, n, na, temperature, x,
                       logits, g, nullptr, nullptr, dx, dlogits);
    return eap::check_launch("anchor_attn_pool_bwd");
}

extern "C" int eap_slot_masked_mean_fwd_f32(int b, int ns, int c, int n, int na, const float *x, const float *mask,
                                            const float *inv_den, float *out, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0 || ns <= 0) return 0;
    if (na <= 0 || na > 64 || (na & 3) != 0 || ns > MAXS || b > 65535)
        return eap::bad_arg("slot_masked_mean: anchors must be a multiple of 4, at most 64; at most 8 slots; b <= 65535");
    hipLaunchKernelGGL(slot_mean_fwd_kernel, dim3(c, b), dim3(256), 0, eap::S(stream), c, n, na, ns, x, mask, inv_den, out);
    return eap::check_launch("slot_masked_mean_fwd");
}

extern "C" int eap_slot_masked_mean_bwd_f32(int b, int ns, int c, int n, int na, const float *g, const float *mask,
                                            const float *inv_den, float *dx, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0 || ns <= 0) return 0;
    if (na <= 0 || na > 64 || (na & 3) != 0 || ns > MAXS || b > 65535)
        return eap::bad_arg("slot_masked_mean: anchors must be a multiple of 4, at most 64; at most 8 slots; b <= 65535");
    hipLaunchKernelGGL(slot_mean_bwd_kernel, dim3(c, b), dim3(256), 0, eap::S(stream), c, n, na, ns, g, mask, inv_den, dx);
    return eap::check_launch("slot_masked_mean_bwd");
}

extern "C" int eap_masked_max_fwd_f32(int b, int c, int n, int na, const float *x, const float *mask, float *out, int32_t *arg,
                                      eap_stream_t stream) {
    if (b <= 0 || c <= 0) return 0;
    if (n <= 0) return eap::bad_arg("masked_max: no points");
    if (na <= 0 || na > 64 || (na & 3) != 0 || b > 65535) return eap::bad_arg("masked_max: anchors must be a multiple of 4, at most 64; b <= 65535");
    hipLaunchKernelGGL(masked_max_fwd_kernel, dim3(c, b), dim3(256), 0, eap::S(stream), c, n, na, x, mask, out, arg);
    return eap::check_launch("masked_max_fwd");
}

extern "C" int eap_masked_max_bwd_f32(int b, int c, int n, int na, const float *g, const int32_t *arg, const float *mask, float *dx,
                                      eap_stream_t stream) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    if (na <= 0 || na > 64 || (na & 3) != 0 || b > 65535) return eap::bad_arg("masked_max: anchors must be a multiple of 4, at most 64; b <= 65535");
    hipLaunchKernelGGL(masked_max_bwd_kernel, dim3(c, b), dim3(256), 0, eap::S(stream), c, n, na, g, arg, mask, dx);
    return eap::check_launch("masked_max_bwd");
}
