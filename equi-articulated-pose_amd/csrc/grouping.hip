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
