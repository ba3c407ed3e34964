// tools/microbench/glds_offset.hip -- where does the immediate `offset:` of global_load_lds_dwordx4 go: into the global
// address only, or into the LDS destination as well?  One wave moves 1 KB with offset:1024 from a buffer holding its own
// byte index / 4, M0 = 2048; the LDS is then dumped.  Answer (MI355X): printed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k(const unsigned *src, unsigned *out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xffffffffu;
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)lds;
    unsigned keep;
    const unsigned voff = threadIdx.x * 16u;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(lds0 + 2048u) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 64) out[i] = lds[i];
}

int main() {
    unsigned *src, *out, h[4096];
    hipMalloc(&src, 65536); hipMalloc(&out, 16384);
    unsigned init[16384];
    for (int i = 0; i < 16384; ++i) init[i] = i;
    hipMemcpy(src, init, 65536, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out);
    hipMemcpy(h, out, 16384, hipMemcpyDeviceToHost);
    int first = -1, last = -1;
    for (int i = 0; i < 4096; ++i) if (h[i] != 0xffffffffu) { if (first < 0) first = i; last = i; }
    printf("LDS words written: [%d, %d] (bytes %d..%d), first value %u (global byte %u)\n", first, last, first * 4, last * 4 + 3, h[first], h[first] * 4);
    printf("M0 = 2048, offset:1024 -> LDS destination %s the immediate; global source %s it\n", first * 4 == 3072 ? "INCLUDES" : (first * 4 == 2048 ? "does NOT include" : "??"),
           h[first] * 4 == 1024 ? "includes" : "does not include");
    return 0;
}
