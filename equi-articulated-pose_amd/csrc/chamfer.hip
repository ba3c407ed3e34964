// csrc/chamfer.hip -- chamfer distance forward / backward on gfx950
// (reference: extensions/chamfer_dist/chamfer.cu L15-145 forward, L173-231 backward;
//  bound as chamfer.forward / chamfer.backward in chamfer_cuda.cpp L22-39).
//
//   forward : dist1[b,i] = min_j |xyz1[b,i] - xyz2[b,j]|^2, idx1 = first arg-min; same with the
//             clouds swapped for dist2 / idx2.
//   backward: gxyz1[b,i] += 2 g1 (x1 - x2[idx1]);  gxyz2[b,idx1] -= the same; and symmetrically.
//
// One lane per query point, the other cloud streams through LDS in 1024-point tiles (every
// candidate is a wave-wide broadcast read).  Squared distances are evaluated with one rounding
// per operation in the reference's source order, so distances and arg-min indices are bit-exact
// against the CPU oracle (first minimum wins, like the reference's strict `<`).
#include "common.h"

namespace {

constexpr int CH_THREADS = 256;
constexpr int CH_TILE = 1024;

__global__ __launch_bounds__(CH_THREADS) void chamfer_nn_kernel(
    int n, int m, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    float *__restrict__ dist, int32_t *__restrict__ index) {
    __shared__ float tile[CH_TILE * 3];
    const int bi = blockIdx.y;
    const int i = blockIdx.x * CH_THREADS + threadIdx.x;
    const float *p1 = xyz1 + ((size_t)bi * n + min(i, n - 1)) * 3;
    const float x1 = p1[0], y1 = p1[1], z1 = p1[2];
    const float *c2 = xyz2 + (size_t)bi * m * 3;
    float best = 0.f;
    int besti = 0;
    for (int k0 = 0; k0 < m; k0 += CH_TILE) {
        const int len = min(CH_TILE, m - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < len * 3; t += CH_THREADS) tile[t] = c2[(size_t)k0 * 3 + t];
        __syncthreads();
        for (int k = 0; k < len; ++k) {
            const float dx = __fsub_rn(tile[k * 3 + 0], x1), dy = __fsub_rn(tile[k * 3 + 1], y1),
                        dz = __fsub_rn(tile[k * 3 + 2], z1);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            if ((k0 + k) == 0 || d < best) { best = d; besti = k0 + k; }
        }
    }
    if (i < n) {
        dist[(size_t)bi * n + i] = best;
        index[(size_t)bi * n + i] = besti;
    }
}

__global__ void chamfer_grad_kernel(int n, int m, const float *__restrict__ xyz1,
                                    const float *__restrict__ xyz2, const float *__restrict__ grad_dist1,
                                    const int32_t *__restrict__ idx1, float *__restrict__ gxyz1,
                                    float *__restrict__ gxyz2) {
    const int bi = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t o1 = ((size_t)bi * n + i) * 3;
    const int j2 = idx1[(size_t)bi * n + i];
    const size_t o2 = ((size_t)bi * m + j2) * 3;
    const float g = grad_dist1[(size_t)bi * n + i] * 2;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float v = g * (xyz1[o1 + d] - xyz2[o2 + d]);
        atomicAdd(gxyz1 + o1 + d, v);
        atomicAdd(gxyz2 + o2 + d, -v);
    }
}

}  // namespace

extern "C" int eap_chamfer_fwd_f32(int b, int n, int m, const float *xyz1, const float *xyz2,
                                   float *dist1, float *dist2, int32_t *idx1, int32_t *idx2,
                                   eap_stream_t stream) {
    if (b <= 0) return 0;
    if (n <= 0 || m <= 0) return eap::bad_arg("chamfer_forward: empty cloud");
    hipStream_t s = eap::S(stream);
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3(eap::cdiv(n, CH_THREADS), b), dim3(CH_THREADS), 0, s, n, m, xyz1, xyz2, dist1, idx1);
    hipLaunchKernelGGL(chamfer_nn_kernel, dim3(eap::cdiv(m, CH_THREADS), b), dim3(CH_THREADS), 0, s, m, n, xyz2, xyz1, dist2, idx2);
    return eap::check_launch("chamfer_forward");
}

extern "C" int eap_chamfer_bwd_f32(int b, int n, int m, const float *xyz1, const float *xyz2,
                                   const int32_t *idx1, const int32_t *idx2, const float *g1,
                                   const float *g2, float *gxyz1, float *gxyz2, eap_stream_t stream) {
    if (b <= 0) return 0;
    hipStream_t s = eap::S(stream);
    int e = eap::hip_fail(hipMemsetAsync(gxyz1, 0, sizeof(float) * (size_t)b * n * 3, s), "chamfer_backward memset");
    if (!e) e = eap::hip_fail(hipMemsetAsync(gxyz2, 0, sizeof(float) * (size_t)b * m * 3, s), "chamfer_backward memset");
    if (e || n <= 0 || m <= 0) return e;
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3(eap::cdiv(n, 256), b), dim3(256), 0, s, n, m, xyz1, xyz2, g1, idx1, gxyz1, gxyz2);
    hipLaunchKernelGGL(chamfer_grad_kernel, dim3(eap::cdiv(m, 256), b), dim3(256), 0, s, m, n, xyz2, xyz1, g2, idx2, gxyz2, gxyz1);
    return eap::check_launch("chamfer_backward");
}
