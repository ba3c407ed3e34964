// csrc/zpconv.hip -- the reference's native "zpconv" operators on gfx950
// (vgtk/vgtk/cuda/zpconv_cuda.cpp L41-110, kernels zpconv_cuda_kernel.cu L33-195).
//
//   inter fwd : out[b,c,k,p,a]   = sum_n w[b,p,a,k,n] * feats[b,c,idx[b,p,a,k,n],a]
//   inter bwd : gfeats[b,c,q,a] += w * gout[b,c,k,p,a]          (q = idx[b,p,a,k,n])
//   intra fwd : out[b,c,k,p,a]   = sum_n w[a,k,n] * feats[b,c,p,idx[a,n]]
//   intra bwd : gfeats[b,c,p,a'] += w * gout[b,c,k,p,a]          (a' = idx[a,n])
//
// The reference gives every (p,a,k,n) tuple a thread that loops over channels with strided
// atomicAdds into the OUTPUT.  Here the forward kernels are atomics-free gathers:
//   * lanes run along the anchor dimension `a`, which is the contiguous dimension of both
//     feats [b,c,q,a] and out [b,c,k,p,a]: every feature read is one 4*na-byte row segment
//     when the neighbour index is shared across anchors (the only way the Python layer builds
//     it) and every output store is a coalesced row;
//   * idx / w are [b,p,a,k,n] -- contiguous along (k,n), strided along a -- so each block first
//     streams its (point, k-pair) slab of idx/w into LDS with fully coalesced loads and the
//     lanes then read their own anchor's row from LDS (row pitch odd => conflict-free).
// HBM traffic per launch is the algorithmic minimum: idx + w read once, out written once,
// feats served from L2/MALL -- but every (a,k,n) tuple chases its own feature row through L2
// (3-5 % of the HBM roofline).  The f32 forward with >= 8 channels therefore goes to
// csrc/zpconv_rows.hip (30-34 %); this file serves f64, the backward and tiny channel counts.  The backward kernels keep scatter semantics (fp atomics on the
// gradient of feats), lanes along `a` as well.
#include "common.h"

namespace {

constexpr int KC = 2;        // kernel points per block
constexpr int ZP_THREADS = 256;

template <typename T, int CC, bool BWD>
__global__ __launch_bounds__(ZP_THREADS) void inter_zpconv_kernel(
    int np, int nq, int na, int ks, int ann, int c, const int32_t *__restrict__ idx,
    const T *__restrict__ w, const T *__restrict__ src, T *__restrict__ dst, const int32_t *__restrict__ only_flagged) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (only_flagged != nullptr && only_flagged[blockIdx.z] == 0) return;     // served by csrc/zpconv_bwd.hip
    const int pitch = KC * ann + 1;
    T *s_w = reinterpret_cast<T *>(smem);
    int32_t *s_idx = reinterpret_cast<int32_t *>(smem + sizeof(T) * (size_t)na * pitch);

    // one block per point walks over all kernel-point pairs: the feature rows it gathers again
    // for the next pair are then served by this XCD's L2 instead of being re-fetched by a block
    // on another XCD
    const int p = xcd_point(blockIdx.x, np), bn = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kk = wave & (KC - 1), chalf = wave >> 1;
    const size_t row0 = ((size_t)bn * np + p) * na;
    // feats / gfeats: [b,c,nq,na]; out / gout: [b,c,ks,np,na]
    const size_t f_b = (size_t)bn * c * nq * na;
    const size_t f_cs = (size_t)nq * na, o_cs = (size_t)ks * np * na;

    for (int k0 = 0; k0 < ks; k0 += KC) {
        const int kcnt = min(KC, ks - k0);
        const int run = kcnt * ann;  // contiguous elements per anchor row in global memory
        __syncthreads();
        for (int i = threadIdx.x; i < na * run; i += ZP_THREADS) {
            const int a = i / run, e = i - a * run;
            const size_t g = ((row0 + a) * ks + k0) * ann + e;
            s_idx[a * pitch + e] = idx[g];
            s_w[a * pitch + e] = w[g];
        }
        __syncthreads();
        const int k = k0 + kk;
        if (kk >= kcnt || lane >= na) continue;
        const int32_t *my_idx = s_idx + lane * pitch + kk * ann;
        const T *my_w = s_w + lane * pitch + kk * ann;
        const size_t o_b = (size_t)bn * c * ks * np * na + ((size_t)k * np + p) * na + lane;

        for (int c0 = chalf * CC; c0 < c; c0 += 2 * CC) {
            if (!BWD) {
                T acc[CC];
#pragma unroll
                for (int cc = 0; cc < CC; ++cc) acc[cc] = 0;
                for (int n = 0; n < ann; ++n) {
                    const T wv = my_w[n];
                    const T *f = src + f_b + (size_t)my_idx[n] * na + lane;
#pragma unroll
                    for (int cc = 0; cc < CC; ++cc) acc[cc] += f[(size_t)min(c0 + cc, c - 1) * f_cs] * wv;
                }
#pragma unroll
                for (int cc = 0; cc < CC; ++cc)
                    if (c0 + cc < c) dst[o_b + (size_t)(c0 + cc) * o_cs] = acc[cc];
            } else {
                T g[CC];
#pragma unroll
                for (int cc = 0; cc < CC; ++cc)
                    g[cc] = (c0 + cc < c) ? src[o_b + (size_t)(c0 + cc) * o_cs] : (T)0;
                for (int n = 0; n < ann; ++n) {
                    const T wv = my_w[n];
                    T *f = dst + f_b + (size_t)my_idx[n] * na + lane;
#pragma unroll
                    for (int cc = 0; cc < CC; ++cc)
                        if (c0 + cc < c) atomicAdd(f + (size_t)(c0 + cc) * f_cs, g[cc] * wv);
                }
            }
        }
    }
}

template <typename T, bool BWD>
int launch_inter(int b, int np, int nq, int na, int ks, int ann, int c, const int32_t *idx,
                 const T *w, const T *src, T *dst, hipStream_t s, const int32_t *only_flagged = nullptr) {
    if (b <= 0 || c <= 0) return 0;
    if (na > 64) return eap::bad_arg("inter_zpconv: at most 64 anchors are supported");
    if (BWD && only_flagged == nullptr) {
        int e = eap::hip_fail(hipMemsetAsync(dst, 0, sizeof(T) * (size_t)b * c * nq * na, s),
                              "inter_zpconv_backward memset");
        if (e) return e;
    }
    if (np <= 0 || ks <= 0 || na <= 0) return 0;
    if (ann <= 0) {  // empty neighbourhoods: the output is all zeros
        if (!BWD) return eap::hip_fail(hipMemsetAsync(dst, 0, sizeof(T) * (size_t)b * c * ks * np * na, s),
                                       "inter_zpconv_forward memset");
        return 0;
    }
    const size_t shmem = (sizeof(T) + sizeof(int32_t)) * (size_t)na * (KC * ann + 1);
    if (shmem > 160 * 1024) return eap::bad_arg("inter_zpconv: neighbourhood too large for LDS staging");
    auto kern = inter_zpconv_kernel<T, 8, BWD>;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                          "inter_zpconv shared memory");
    if (e) return e;
    dim3 grid(np, 1, b);
    hipLaunchKernelGGL(kern, grid, dim3(ZP_THREADS), shmem, s, np, nq, na, ks, ann, c, idx, w, src, dst, only_flagged);
    return eap::check_launch(BWD ? "inter_zpconv_backward" : "inter_zpconv_forward");
}

// ---------------------------------------------------------------------------------------------
// intra: idx [na_out,ann], w [na_out,ks,ann] are tiny and shared by every point; one lane per
// output anchor, one block row per (b, c, p-tile).
// ---------------------------------------------------------------------------------------------
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void intra_zpconv_kernel(
    int np, int na_in, int na_out, int ks, int ann, int c, const int32_t *__restrict__ idx,
    const T *__restrict__ w, const T *__restrict__ src, T *__restrict__ dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = blockIdx.x * 4 + wave, ci = blockIdx.y, bn = blockIdx.z;
    if (p >= np || lane >= na_out) return;
    const size_t f_row = (((size_t)bn * c + ci) * np + p) * na_in;        // feats / gfeats row
    const size_t o_row = (((size_t)bn * c + ci) * ks * np + p) * na_out;  // + k*np*na_out
    for (int k = 0; k < ks; ++k) {
        const T *wk = w + ((size_t)lane * ks + k) * ann;
        if (!BWD) {
            T acc = 0;
            for (int n = 0; n < ann; ++n) acc += src[f_row + idx[lane * ann + n]] * wk[n];
            dst[o_row + (size_t)k * np * na_out + lane] = acc;
        } else {
            const T g = src[o_row + (size_t)k * np * na_out + lane];
            for (int n = 0; n < ann; ++n) atomicAdd(dst + f_row + idx[lane * ann + n], g * wk[n]);
        }
    }
}

template <typename T, bool BWD>
int launch_intra(int b, int np, int na_in, int na_out, int ks, int ann, int c, const int32_t *idx,
                 const T *w, const T *src, T *dst, hipStream_t s) {
    if (b <= 0 || c <= 0 || np <= 0) return 0;
    if (na_out > 64) return eap::bad_arg("intra_zpconv: at most 64 output anchors are supported");
    if (BWD) {
        int e = eap::hip_fail(hipMemsetAsync(dst, 0, sizeof(T) * (size_t)b * c * np * na_in, s),
                              "intra_zpconv_backward memset");
        if (e) return e;
    }
    if (ks <= 0 || na_out <= 0) return 0;
    dim3 grid(eap::cdiv(np, 4), c, b);
    hipLaunchKernelGGL((intra_zpconv_kernel<T, BWD>), grid, dim3(256), 0, s, np, na_in, na_out, ks,
                       ann, c, idx, w, src, dst);
    return eap::check_launch(BWD ? "intra_zpconv_backward" : "intra_zpconv_forward");
}

}  // namespace

namespace eap {
// the scatter kernel for the clouds whose flag is non-zero only, accumulating into an output the caller has zeroed
int inter_zpconv_bwd_flagged(int b, int np, int nq, int na, int ks, int ann, int c, const int32_t *idx, const float *w,
                             const float *grad, float *gfeats, const int32_t *only_flagged, hipStream_t s) {
    return launch_inter<float, true>(b, np, nq, na, ks, ann, c, idx, w, grad, gfeats, s, only_flagged);
}
}  // namespace eap

#define EAP_INTER(NAME, T, BWD)                                                                  \
    extern "C" int NAME(int b, int np, int nq, int na, int ks, int ann, int c, const int32_t *idx, \
                        const T *w, const T *src, T *dst, eap_stream_t stream) {                 \
        return launch_inter<T, BWD>(b, np, nq, na, ks, ann, c, idx, w, src, dst, eap::S(stream)); \
    }
extern "C" int eap_inter_zpconv_fwd_f32(int b, int np, int nq, int na, int ks, int ann, int c, const int32_t *idx,
                                        const float *w, const float *src, float *dst, eap_stream_t stream) {
    // HBM-speed path when the sizes fit (csrc/zpconv_rows.hip; it checks the index pattern itself)
    if (b > 0 && np > 0 && eap::inter_zpconv_rows_supported(np, nq, na, ks, ann, c))
        return eap::inter_zpconv_rows_fwd(b, np, nq, na, ks, ann, c, idx, w, src, dst, nullptr, eap::S(stream));
    return launch_inter<float, false>(b, np, nq, na, ks, ann, c, idx, w, src, dst, eap::S(stream));
}
EAP_INTER(eap_inter_zpconv_fwd_f64, double, false)
EAP_INTER(eap_inter_zpconv_bwd_f32, float, true)
EAP_INTER(eap_inter_zpconv_bwd_f64, double, true)

#define EAP_INTRA(NAME, T, BWD)                                                                  \
    extern "C" int NAME(int b, int np, int na_in, int na_out, int ks, int ann, int c,            \
                        const int32_t *idx, const T *w, const T *src, T *dst, eap_stream_t stream) { \
        return launch_intra<T, BWD>(b, np, na_in, na_out, ks, ann, c, idx, w, src, dst, eap::S(stream)); \
    }
EAP_INTRA(eap_intra_zpconv_fwd_f32, float, false)
EAP_INTRA(eap_intra_zpconv_fwd_f64, double, false)
EAP_INTRA(eap_intra_zpconv_bwd_f32, float, true)
EAP_INTRA(eap_intra_zpconv_bwd_f64, double, true)
