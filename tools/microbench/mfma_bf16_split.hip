// tools/microbench/mfma_bf16_split.hip -- feasibility of a 3 x bf16 split contraction: rate of v_mfma_f32_32x32x16_bf16 from
// one wave per SIMD with 16 accumulators (the 128 x 128 wave tile of csrc/gemm_dma_f32.hip), bare and with NV vector
// instructions per MFMA riding along (the on-the-fly fp32 -> 3 x bf16 operand split: v_cvt_pk_bf16_f32, shifts, subtracts),
// and the accuracy of the 6-product split against fp64 on random data.  hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_split mfma_bf16_split.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int NV>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = (u32x4){0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u + i, 0x3f803f80u}; b[i] = a[i] + 1u; }
    float x0 = seed + lane, x1 = seed * 2.f, y = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 6; ++rep) {          // six products of the split
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[4 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[4 * i + j], 0, 0, 0);
                    // riders: the split of two fp32 values costs cvt_pk + 2 unpack + 2 sub per plane
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        if (v % 5 == 0) { const unsigned p = pk_bf16(x0, x1); y = __uint_as_float(p << 16); }
                        else if (v % 5 == 1) x0 = x0 - y;
                        else if (v % 5 == 2) { y = __uint_as_float(__float_as_uint(y) & 0xffff0000u); }
                        else if (v % 5 == 3) x1 = x1 - y;
                        else x0 = x0 + 1e-9f;
                    }
                }
        }
    }
    float s = x0 + x1 + y;
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV>), dim3(256), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = 256.0 * 4 * iters * 96;                 // wave-level MFMA instructions
    const double flops = mfma * 2.0 * 32 * 32 * 16;
    printf("%d vector riders per MFMA: %.3f ms, %.0f TFLOP/s bf16 = %.0f TFLOP/s of fp32-equivalent products (6 MFMAs each), %.1f cycles per MFMA at 2.4 GHz\n",
           NV, ms, flops / ms / 1e9, flops / 6 / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 96.0));
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 256 * 4);
    run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<6>(out);
    return 0;
}
