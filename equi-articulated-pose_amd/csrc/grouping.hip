// csrc/grouping.hip -- neighbour search for the SO(3) point convolution on gfx950.
//
// Replaces the reference's grouping extension (vgtk/vgtk/cuda/grouping_cuda.cpp L71-86,
// kernels grouping_cuda_kernel.cu L68-113).  The reference runs ONE block per cloud with a
// thread striding over the queries; here every query point gets its own lane, the support
// cloud streams through LDS in tiles (each support point is a wave-wide broadcast read) and a
// block retires as soon as all of its 64 queries have found `nsample` neighbours -- with the
// large radii of the deeper layers (first-nsample-in-index-order semantics) that is after the
// first tile.
//
// Bit-exactness: d2 is evaluated as ((dx*dx + dy*dy) + dz*dz) with one rounding per operation
// (__fmul_rn/__fadd_rn never contract to FMA), the same order as grouping_cuda_kernel.cu:L93-94
// and as the CPU oracle; the neighbour lists are therefore identical integer-for-integer.
#include "common.h"

namespace {

constexpr int BQ_THREADS = 64;
constexpr int BQ_TILE = 1024;

__device__ __forceinline__ float d2_exact(float ax, float ay, float az, float bx, float by, float bz) {
    const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}
__device__ __forceinline__ double d2_exact(double ax, double ay, double az, double bx, double by, double bz) {
    const double dx = __dsub_rn(ax, bx), dy = __dsub_rn(ay, by), dz = __dsub_rn(az, bz);
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

template <typename T>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(
    int n, int m, T radius2, int nsample, const T *__restrict__ new_xyz,
    const T *__restrict__ xyz, int32_t *__restrict__ idx) {
    __shared__ T tile[3][BQ_TILE];
    const int bi = blockIdx.y;
    xyz += (size_t)bi * 3 * n;
    new_xyz += (size_t)bi * 3 * m;
    idx += (size_t)bi * m * nsample;

    const int j = blockIdx.x * BQ_THREADS + threadIdx.x;
    const bool live = j < m;
    T qx = 0, qy = 0, qz = 0;
    if (live) { qx = new_xyz[j]; qy = new_xyz[m + j]; qz = new_xyz[2 * m + j]; }
    int32_t *out = idx + (size_t)(live ? j : 0) * nsample;

    int cnt = live ? 0 : nsample;
    for (int k0 = 0; k0 < n; k0 += BQ_TILE) {
        const int len = min(BQ_TILE, n - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < len; t += BQ_THREADS) {
            tile[0][t] = xyz[k0 + t];
            tile[1][t] = xyz[n + k0 + t];
            tile[2][t] = xyz[2 * n + k0 + t];
        }
        __syncthreads();
        for (int t = 0; t < len && cnt < nsample; ++t) {
            const T d2 = d2_exact(qx, qy, qz, tile[0][t], tile[1][t], tile[2][t]);
            if (d2 < radius2) { out[cnt] = k0 + t; ++cnt; }
        }
        if (__syncthreads_and(cnt >= nsample)) break;
    }
    if (!live) return;
    // grouping_cuda_kernel.cu:L100-105: cyclic repeat-padding only when cnt < nsample-1; with
    // exactly nsample-1 hits the last slot keeps the host wrapper's zero initialisation.
    if (cnt < nsample - 1) {
        if (cnt == 0) {
            for (int t = 0; t < nsample; ++t) out[t] = 0;
        } else {
            for (int t = cnt; t < nsample; ++t) out[t] = out[t - cnt];
        }
    } else if (cnt == nsample - 1) {
        out[nsample - 1] = 0;
    }
}

template <typename T>
int launch_ball_query(int b, int n, int m, float radius, int nsample, const T *new_xyz,
                      const T *xyz, int32_t *idx, hipStream_t s) {
    if (b <= 0 || m <= 0 || nsample <= 0) return 0;
    if (n <= 0) return eap::bad_arg("ball_query: empty support cloud");
    dim3 grid(eap::cdiv(m, BQ_THREADS), b);
    const T r2 = (T)(radius * radius);  // float product first, grouping_cuda_kernel.cu:L82
    hipLaunchKernelGGL(ball_query_kernel<T>, grid, dim3(BQ_THREADS), 0, s, n, m, r2, nsample,
                       new_xyz, xyz, idx);
    return eap::check_launch("ball_query");
}


// ---------------------------------------------------------------------------------------------
// furthest point sampling (grouping_cuda.cpp:L160-174, kernel grouping_cuda_kernel.cu:L352-466).
// Inherently sequential in m; one block per cloud.  Off the shipped models' path (stride is 1),
// kept for API completeness.  The winner of each round must match the reference exactly, ties
// included: per-thread first strict maximum over k = tid, tid+T, ..., then a halving tree that
// keeps the lower slot on ties -- the same reduction shape, with T = the reference's block size
// (largest power of two <= n, capped at 1024).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fps_kernel(int n, int m, const float *__restrict__ xyz,
                                                  float *__restrict__ temp, int32_t *__restrict__ idx) {
    __shared__ float s_d[1024];
    __shared__ int s_i[1024];
    const int bi = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
    xyz += (size_t)bi * 3 * n;
    temp += (size_t)bi * n;
    idx += (size_t)bi * m;
    for (int k = tid; k < n; k += T) temp[k] = 1e10f;
    int old = 0;
    if (tid == 0) idx[0] = 0;
    __syncthreads();
    for (int j = 1; j < m; ++j) {
        const float x1 = xyz[old], y1 = xyz[n + old], z1 = xyz[2 * n + old];
        float best = -1.f;
        int besti = 0;
        for (int k = tid; k < n; k += T) {
            const float x2 = xyz[k], y2 = xyz[n + k], z2 = xyz[2 * n + k];
            const float mag = __fadd_rn(__fadd_rn(__fmul_rn(x2, x2), __fmul_rn(y2, y2)), __fmul_rn(z2, z2));
            if (mag <= 1e-3f) continue;
            const float d = d2_exact(x2, y2, z2, x1, y1, z1);
            const float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            if (d2 > best) { best = d2; besti = k; }
        }
        s_d[tid] = best;
        s_i[tid] = besti;
        __syncthreads();
        for (int s = T >> 1; s >= 1; s >>= 1) {
            if (tid < s && s_d[tid + s] > s_d[tid]) { s_d[tid] = s_d[tid + s]; s_i[tid] = s_i[tid + s]; }
            __syncthreads();
        }
        old = s_i[0];
        if (tid == 0) idx[j] = old;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// anchor_query, S^2 variant (grouping_cuda.cpp:L88-108, kernel .cu:L181-247):
//   w[b,p,a,k,n] = (kw - |x|)^2 + ((kh - theta) |x|)^2,  theta = acos(x . anchor_a / |x|)
// ---------------------------------------------------------------------------------------------
__global__ void anchor_query_kernel(int np, int nn, int na, int ks, const float *__restrict__ gxyz,
                                    const float *__restrict__ anchors, const float *__restrict__ kpts,
                                    float *__restrict__ w) {
    const int bi = blockIdx.y;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long long)np * nn) return;
    const int pi = (int)(e / nn), ni = (int)(e % nn);
    const float *g = gxyz + (size_t)bi * 3 * np * nn;
    const float x = g[e], y = g[(size_t)np * nn + e], z = g[(size_t)2 * np * nn + e];
    const float norm = __fadd_rn(sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z))), 1e-6f);
    float *wp = w + (((size_t)bi * np + pi) * na) * ks * nn + ni;
    for (int ai = 0; ai < na; ++ai) {
        const float dot = __fadd_rn(__fadd_rn(__fmul_rn(x, anchors[ai * 3]), __fmul_rn(y, anchors[ai * 3 + 1])),
                                    __fmul_rn(z, anchors[ai * 3 + 2]));
        const float theta = acosf(__fdiv_rn(dot, norm));
        for (int ki = 0; ki < ks; ++ki) {
            const float a = __fsub_rn(kpts[ki * 2], norm);
            const float c = __fmul_rn(__fsub_rn(kpts[ki * 2 + 1], theta), norm);
            wp[((size_t)ai * ks + ki) * nn] = __fadd_rn(__fmul_rn(a, a), __fmul_rn(c, c));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// initial_anchor_query (grouping_cuda.cpp:L138-158, kernel .cu:L117-167): kernel-point occupancy
// of a fragment around each centre.  The reference scatters with two atomicAdds per hit; here
// every output element (kernel point, centre, anchor) owns a lane that sums over the fragment
// points in index order -- deterministic, no atomics.
// ---------------------------------------------------------------------------------------------
__global__ void initial_anchor_query_kernel(int nc, int m, int na, int ks, float radius, float sigma,
                                            const float *__restrict__ centers, const float *__restrict__ xyz,
                                            const float *__restrict__ kpts, float *__restrict__ w,
                                            float *__restrict__ cnt) {
    const int pn = blockIdx.x, bi = blockIdx.y;
    const float *C = centers + (size_t)bi * 3 * nc;
    const float cx = C[pn], cy = C[nc + pn], cz = C[2 * nc + pn];
    for (int e = threadIdx.x; e < ks * na; e += blockDim.x) {
        const int kn = e / na, an = e - kn * na;
        const float kx = __fadd_rn(kpts[e * 3], cx), ky = __fadd_rn(kpts[e * 3 + 1], cy), kz = __fadd_rn(kpts[e * 3 + 2], cz);
        float sw = 0.f, sc = 0.f;
        for (int pm = 0; pm < m; ++pm) {
            const float x = xyz[3 * pm], y = xyz[3 * pm + 1], z = xyz[3 * pm + 2];
            const float dc = sqrtf(d2_exact(cx, cy, cz, x, y, z));
            if (dc <= radius) {
                const float dk = sqrtf(d2_exact(kx, ky, kz, x, y, z));
                const float wt = __fsub_rn(1.f, __fdiv_rn(__fmul_rn(dk, dk), sigma));
                if (wt > 0.f) sw = __fadd_rn(sw, wt);
                sc = __fadd_rn(sc, 1.f);
            }
        }
        const size_t o = (((size_t)bi * ks + kn) * nc + pn) * na + an;
        w[o] = sw;
        cnt[o] = sc;
    }
}

}  // namespace

extern "C" int eap_ball_query_f32(int b, int n, int m, float radius, int nsample,
                                  const float *new_xyz, const float *xyz, int32_t *idx,
                                  eap_stream_t stream) {
    return launch_ball_query<float>(b, n, m, radius, nsample, new_xyz, xyz, idx, eap::S(stream));
}
extern "C" int eap_ball_query_f64(int b, int n, int m, float radius, int nsample,
                                  const double *new_xyz, const double *xyz, int32_t *idx,
                                  eap_stream_t stream) {
    return launch_ball_query<double>(b, n, m, radius, nsample, new_xyz, xyz, idx, eap::S(stream));
}

extern "C" int eap_furthest_point_sampling_f32(int b, int n, int m, const float *xyz, float *temp,
                                               int32_t *idx, eap_stream_t stream) {
    if (b <= 0 || m <= 0) return 0;
    if (n <= 0) return eap::bad_arg("furthest_point_sampling: empty cloud");
    int threads = 1;
    while (threads * 2 <= n && threads < 1024) threads *= 2;   // opt_n_threads, grouping_cuda_kernel.cu:L29-33
    hipLaunchKernelGGL(fps_kernel, dim3(b), dim3(threads), 0, eap::S(stream), n, m, xyz, temp, idx);
    return eap::check_launch("furthest_point_sampling");
}

extern "C" int eap_anchor_query_f32(int b, int np, int nn, int na, int ks, const float *grouped_xyz,
                                    const float *anchors, const float *kernel_pts, float *w,
                                    eap_stream_t stream) {
    if (b <= 0 || np <= 0 || nn <= 0) return 0;
    hipLaunchKernelGGL(anchor_query_kernel, dim3(eap::cdiv((long long)np * nn, 256), b), dim3(256), 0,
                       eap::S(stream), np, nn, na, ks, grouped_xyz, anchors, kernel_pts, w);
    return eap::check_launch("anchor_query");
}

extern "C" int eap_initial_anchor_query_f32(int b, int nc, int m, int na, int ks, float radius, float sigma,
                                            const float *centers, const float *xyz, const float *kernel_pts,
                                            float *w, float *cnt, eap_stream_t stream) {
    if (b <= 0 || nc <= 0 || na <= 0 || ks <= 0) return 0;
    hipLaunchKernelGGL(initial_anchor_query_kernel, dim3(nc, b), dim3(256), 0, eap::S(stream), nc, m, na, ks,
                       radius, sigma, centers, xyz, kernel_pts, w, cnt);
    return eap::check_launch("initial_anchor_query");
}
