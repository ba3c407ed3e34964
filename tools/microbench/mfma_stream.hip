// tools/microbench/mfma_stream.hip -- what a LONE wave per SIMD can push through the fp32 matrix pipe, by where its
// accumulators live and what rides between the MFMAs.  hipcc --offload-arch=gfx950 -O3 -o mfma_stream mfma_stream.hip
//   variant 0: 8 accumulators in architectural VGPRs (compiler's choice under a 256-register budget), bare MFMAs
//   variant 1: the same, accumulators pinned to AGPRs ("+a")
//   variant 2: VGPR accumulators, 5 dependent VALU ops per MFMA whose result OVERWRITES the B operand just used (WAR)
//   variant 3: VGPR accumulators, 5 VALU ops per MFMA writing a DIFFERENT register (double-buffered operands)
//   variant 4: AGPR accumulators, 5 VALU ops per MFMA, double-buffered operands
//   variant 5: variant 4 + one ds_read_b128 per 4 MFMAs
//   variant 6: AGPR accumulators, 5 INDEPENDENT VALU ops per MFMA (five chains advanced one op each)
//   variant 7: variant 6 with 10 independent VALU ops per MFMA
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, float seed) {
    __shared__ float4 lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[8], b0[8], b1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = seed + lane + i; b0[i] = seed * i; b1[i] = seed - i; }
    float gx = seed, gy = seed * 2, gz = seed * 3, base = seed * 0.5f;
    float4 ld = make_float4(0.f, 0.f, 0.f, 0.f);
    float ch[10];
#pragma unroll
    for (int c = 0; c < 10; ++c) ch[c] = seed * c;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float *bc = half ? b1 : b0, *bn = half ? b0 : b1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (V == 1 || V >= 4) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i]), "v"(bc[i]));
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bc[i], acc[i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (V == 6 || V == 7) {
#pragma unroll
                    for (int c = 0; c < (V == 7 ? 10 : 5); ++c) ch[c] = fmaf(gx, ch[c], a[(i + c) & 7]);
                } else if (V >= 2) {
                    float x = fmaf(gx, a[i], base);
                    x = fmaf(gy, bn[(i + 1) & 7], x);
                    x = fmaf(gz, x, gx);
                    x = x + base;
                    x = fmaxf(x, 0.f);
                    if (V == 2) bc[i] = x; else bn[i] = x;
                }
                if (V == 5 && (i & 3) == 1) { ld = lds[(lane + i * 64 + it) & 1023]; }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (V == 5) { gx += ld.x; }
        }
    }
    if (V == 1 || V >= 4) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int c = 0; c < 10; ++c) s += ch[c];
    out[blockIdx.x * 256 + threadIdx.x] = s + b0[3] + b1[5];
}

template <int V>
void run(float *out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 4 * iters * 16.0 * (2.0 * 32 * 32 * 2);
    printf("variant %d: %.3f ms  %.1f TFLOP/s  (%.1f %% of 157.3)\n", V, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *out; hipMalloc(&out, 256 * 256 * 4);
    const int iters = 20000;
    run<0>(out, iters); run<1>(out, iters); run<2>(out, iters); run<3>(out, iters); run<4>(out, iters); run<5>(out, iters); run<6>(out, iters); run<7>(out, iters);
    return 0;
}
