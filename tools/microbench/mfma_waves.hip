// tools/microbench/mfma_waves.hip -- fp32 matrix-pipe utilisation of the grouping kernels' inner loop WITHOUT memory:
// every MFMA's B operand is a kernel weight evaluated on the VALU (5 ops), A operands sit in registers.  Swept: waves
// per SIMD (1, 2, 4), accumulators per wave, and how the VALU work is clustered (CL MFMAs, then the CL weights of the
// next group).  hipcc --offload-arch=gfx950 -O3 -o mfma_waves mfma_waves.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WPS, int NACC, int CL, int NV>
__global__ __launch_bounds__(256 * WPS, 1) void k(float *out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[NACC], w0[NACC], w1[NACC], kx[NACC], ky[NACC], kz[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] = seed + lane + i; w0[i] = seed * i; w1[i] = seed - i; kx[i] = seed + i; ky[i] = seed * 2 + i; kz[i] = seed * 3 - i; }
    float gx = seed, gy = seed * 2, gz = seed * 3, base = seed * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float *wc = half ? w1 : w0, *wn = half ? w0 : w1;
#pragma unroll
            for (int g = 0; g < NACC / CL; ++g) {
#pragma unroll
                for (int j = 0; j < CL; ++j) {
                    const int i = g * CL + j;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], wc[i], acc[i], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < CL; ++j) {
                    const int i = g * CL + j;
                    float x = fmaf(gx, kx[i], base);
                    if (NV >= 2) x = fmaf(gy, ky[i], x);
                    if (NV >= 3) x = fmaf(gz, kz[i], x);
                    if (NV >= 4) x = x + a[i];
                    if (NV >= 5) x = fmaxf(x, 0.f);
                    wn[i] = x;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            gx += 1e-9f; gy -= 1e-9f;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s + w0[1] + w1[2];
}

template <int WPS, int NACC, int CL, int NV>
void run(float *out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WPS, NACC, CL, NV>), dim3(256), dim3(256 * WPS), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<WPS, NACC, CL, NV>), dim3(256), dim3(256 * WPS), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 4 * WPS * iters * 2.0 * NACC * (2.0 * 32 * 32 * 2);
    printf("waves/SIMD %d  acc/wave %2d  cluster %d  VALU/MFMA %d: %8.3f ms  %6.1f TFLOP/s  (%.1f %% of 157.3)\n", WPS, NACC, CL, NV, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    const int it = 20000;
    run<1, 8, 1, 5>(out, it); run<1, 8, 2, 5>(out, it); run<1, 8, 4, 5>(out, it); run<1, 8, 8, 5>(out, it);
    run<1, 8, 4, 2>(out, it); run<1, 8, 8, 2>(out, it); run<1, 8, 1, 1>(out, it); run<1, 8, 1, 2>(out, it);
    run<2, 8, 1, 5>(out, it / 2); run<2, 8, 2, 5>(out, it / 2); run<2, 8, 4, 5>(out, it / 2); run<2, 8, 8, 5>(out, it / 2);
    run<2, 4, 1, 5>(out, it); run<2, 4, 4, 5>(out, it);
    run<4, 4, 1, 5>(out, it / 2); run<4, 4, 2, 5>(out, it / 2); run<4, 4, 4, 5>(out, it / 2);
    run<4, 4, 1, 2>(out, it / 2); run<4, 4, 4, 2>(out, it / 2);
    return 0;
}
