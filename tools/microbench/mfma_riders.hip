// tools/microbench/mfma_riders.hip -- what each kind of instruction costs a stream of fp32 MFMAs (v_mfma_f32_32x32x2_f32,
// 64 cycles each) when it rides between them: plain VALU (v_fma_f32), packed VALU (v_pk_fma_f32), LDS reads
// (ds_read_b128), LDS writes (ds_write_b128) and scalar ALU -- N riders per MFMA, 1 / 2 / 4 waves per SIMD.
// Operands in registers, no global memory.  hipcc --offload-arch=gfx950 -O3 -o mfma_riders mfma_riders.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND: 0 none, 1 v_fma_f32, 2 v_pk_fma_f32, 3 ds_read_b128, 4 ds_write_b128, 5 s_add_u32 (scalar)
template <int WPS, int KIND, int N>
__global__ __launch_bounds__(256 * WPS, 1) void k(float *out, int iters, float seed) {
    __shared__ float4 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = make_float4(seed, seed, seed, seed);
    __syncthreads();
    constexpr int NACC = 4;
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { a[i] = seed + lane + i; b[i] = seed * i + 1.f; }
    float ch[8]; f32x2 pk[8]; float4 ld[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) { ch[c] = seed * c; pk[c] = (f32x2){seed * c, seed + c}; }
#pragma unroll
    for (int c = 0; c < 4; ++c) ld[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned sacc = 0;
    const float4 *lp = lds + ((threadIdx.x * 7) & 1023);
    float4 *wp = lds + 1024 + (threadIdx.x & 1023);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < N; ++c) {
                if (KIND == 1) ch[c] = fmaf(seed, ch[c], a[i]);
                if (KIND == 2) pk[c] = __builtin_elementwise_fma((f32x2){seed, seed}, pk[c], (f32x2){a[i], a[i]});
                if (KIND == 3) ld[c & 3] = lp[(c + i * 8) & 63];
                if (KIND == 4) wp[0] = make_float4(ch[c], ch[c], a[i], b[i]);
                if (KIND == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 3) { a[0] += ld[0].x * 1e-30f + ld[1].y * 1e-30f + ld[2].z * 1e-30f + ld[3].w * 1e-30f; }
    }
    float s = (float)sacc;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int c = 0; c < 8; ++c) s += ch[c] + pk[c].x + pk[c].y;
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s + lds[lane].x;
}

template <int WPS, int KIND, int N>
void run(float *out, int iters, const char *name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<WPS, KIND, N>), dim3(256), dim3(256 * WPS), 0, 0, out, 10, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<WPS, KIND, N>), dim3(256), dim3(256 * WPS), 0, 0, out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * 4 * WPS * iters * 4.0 * (2.0 * 32 * 32 * 2);
    const double frac = flops / ms / 1e9 / 157.3;
    printf("waves/SIMD %d  %-14s x %d per MFMA: %6.1f TFLOP/s  %5.1f %% of peak  -> %5.1f cycles per MFMA slot (64 = free)\n", WPS, name, N,
           flops / ms / 1e9, frac * 100, 64.0 / frac);
}

template <int WPS>
void sweep(float *out, int it) {
    run<WPS, 0, 0>(out, it, "bare");
    run<WPS, 1, 1>(out, it, "v_fma_f32"); run<WPS, 1, 2>(out, it, "v_fma_f32"); run<WPS, 1, 4>(out, it, "v_fma_f32"); run<WPS, 1, 8>(out, it, "v_fma_f32");
    run<WPS, 2, 1>(out, it, "v_pk_fma_f32"); run<WPS, 2, 2>(out, it, "v_pk_fma_f32"); run<WPS, 2, 4>(out, it, "v_pk_fma_f32"); run<WPS, 2, 8>(out, it, "v_pk_fma_f32");
    run<WPS, 3, 1>(out, it, "ds_read_b128"); run<WPS, 3, 2>(out, it, "ds_read_b128"); run<WPS, 3, 4>(out, it, "ds_read_b128");
    run<WPS, 4, 1>(out, it, "ds_write_b128"); run<WPS, 4, 2>(out, it, "ds_write_b128");
    run<WPS, 5, 4>(out, it, "s_add_u32"); run<WPS, 5, 8>(out, it, "s_add_u32");
}

int main() {
    float *out; (void)hipMalloc(&out, 256 * 1024 * 4);
    sweep<1>(out, 20000); sweep<2>(out, 10000); sweep<4>(out, 5000);
    return 0;
}
