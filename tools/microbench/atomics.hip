// Micro-benchmark: cost of 240-byte row atomics (fp32 atomicAdd, lanes = 60 anchors) with hot /
// uniform row distributions.  Dev utility -- informs the backward scatter design.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void row_atomics(int P, int NN, int C, int Q, const int *__restrict__ idx, float *g, int mode) {
    // block = point, 4 waves split channels; lane = anchor
    const int p = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane >= 60) return;
    for (int c = wave; c < C; c += 4)
        for (int n = 0; n < NN; ++n) {
            const int q = idx[p * NN + n];
            float *dst = g + ((size_t)c * Q + q) * 60 + lane;
            if (mode == 0) atomicAdd(dst, 1.0f);
            else if (mode == 1) __hip_atomic_fetch_add(dst, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else unsafeAtomicAdd(dst, 1.0f);
        }
}

int main() {
    const int P = 4096, NN = 64, C = 128, Q = 4096;
    std::vector<int> hot(P * NN), uni(P * NN);
    srand(1);
    for (int i = 0; i < P * NN; ++i) { hot[i] = rand() % 114; uni[i] = rand() % Q; }
    int *d_idx; float *g;
    hipMalloc(&d_idx, sizeof(int) * P * NN);
    hipMalloc(&g, sizeof(float) * (size_t)C * Q * 60);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 2; ++pat) {
        hipMemcpy(d_idx, pat == 0 ? hot.data() : uni.data(), sizeof(int) * P * NN, hipMemcpyHostToDevice);
        for (int mode = 0; mode < 3; ++mode) {
            hipMemset(g, 0, sizeof(float) * (size_t)C * Q * 60);
            row_atomics<<<P, 256>>>(P, NN, C, Q, d_idx, g, mode);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            row_atomics<<<P, 256>>>(P, NN, C, Q, d_idx, g, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s rows, mode %d: %.3f ms for %.1f M row-atomics (%.2f G rows/s)\n", pat == 0 ? "hot(114)" : "uniform", mode, ms,
                   (double)P * NN * C / 1e6, (double)P * NN * C / ms / 1e6);
        }
    }
    return 0;
}
