// csrc/gathering.hip -- point gather / scatter-add (vgtk/vgtk/cuda/gathering_cuda.cpp L29-60,
// kernels gathering_cuda_kernel.cu L43-98).
//
// Layout: pts [b,c,n], idx [b,m], out [b,c,m].  Lanes run along m (the contiguous output
// dimension) so stores are coalesced; the reference maps lanes to channels (stride m).
#include "common.h"

namespace {

template <typename T>
__global__ void gather_fwd_kernel(int c, int n, int m, const T *__restrict__ pts,
                                  const int32_t *__restrict__ idx, float *__restrict__ out) {
    const int bi = blockIdx.z, ci = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int a = idx[(size_t)bi * m + j];
    out[((size_t)bi * c + ci) * m + j] = (float)pts[((size_t)bi * c + ci) * n + a];
}

template <typename T>
__global__ void gather_bwd_kernel(int c, int n, int m, const T *__restrict__ grad_out,
                                  const int32_t *__restrict__ idx, T *__restrict__ grad_pts) {
    const int bi = blockIdx.z, ci = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const int a = idx[(size_t)bi * m + j];
    atomicAdd(grad_pts + ((size_t)bi * c + ci) * n + a, grad_out[((size_t)bi * c + ci) * m + j]);
}

template <typename T>
int launch_bwd(int b, int c, int n, int m, const T *g, const int32_t *idx, T *gp, hipStream_t s) {
    if (b <= 0 || c <= 0 || n <= 0) return 0;
    int e = eap::hip_fail(hipMemsetAsync(gp, 0, sizeof(T) * (size_t)b * c * n, s), "gather_bwd memset");
    if (e || m <= 0) return e;
    hipLaunchKernelGGL(gather_bwd_kernel<T>, dim3(eap::cdiv(m, 256), c, b), dim3(256), 0, s, c, n, m, g, idx, gp);
    return eap::check_launch("gather_points_backward");
}

}  // namespace

extern "C" int eap_gather_points_fwd_f32(int b, int c, int n, int m, const float *pts,
                                         const int32_t *idx, float *out, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || m <= 0) return 0;
    hipLaunchKernelGGL(gather_fwd_kernel<float>, dim3(eap::cdiv(m, 256), c, b), dim3(256), 0,
                       eap::S(stream), c, n, m, pts, idx, out);
    return eap::check_launch("gather_points_forward");
}
extern "C" int eap_gather_points_bwd_f32(int b, int c, int n, int m, const float *grad_out,
                                         const int32_t *idx, float *grad_pts, eap_stream_t stream) {
    return launch_bwd<float>(b, c, n, m, grad_out, idx, grad_pts, eap::S(stream));
}
extern "C" int eap_gather_points_bwd_f64(int b, int c, int n, int m, const double *grad_out,
                                         const int32_t *idx, double *grad_pts, eap_stream_t stream) {
    return launch_bwd<double>(b, c, n, m, grad_out, idx, grad_pts, eap::S(stream));
}
