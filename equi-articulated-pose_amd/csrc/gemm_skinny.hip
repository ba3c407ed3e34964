// csrc/gemm_skinny.hip -- C[M,N] = sum_z A_z[M,K] B_z[N,K]^T for a SMALL output (M <= 64, N <= 32) and a long contraction:
// the first layer's weight gradient dW[64, 24] = sum over clouds of dY[64, P*A] X[24, P*A]^T (the textbook backward of
// vgtk/vgtk/so3conv/modules.py:L48-55 at C = 1; K = 245 760 per cloud).  The tiled kernels pad such a product to their
// 256 x 128 tile -- 5 % of the matrix work useful, 1.1 ms for 0.7 GB of operands; this is a streaming reduction instead:
// every wave walks its own slice of K with 16-byte loads straight into the MFMA operand registers (no LDS staging: each
// operand element is used once), v_mfma_f32_32x32x2_f32 with rows = M (one or two tiles), columns = N, and the per-wave,
// per-workgroup partial results are summed in a fixed order (bit-reproducible).
//
// Operand registers: lane (i = lane & 31, h = lane >> 5) of the A operand holds row i, contraction index h.  A lane loads
// float4 A[i][k + 4 h .. + 3]; component c of the two halves is the pair (k + c, k + 4 + c) -- some pair of contraction
// indices, the same one on the B side, which is all a sum needs.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NT = 256;          // 4 waves
constexpr int KSTEP = 32;        // contraction indices per wave and iteration (4 float4 per lane and tile)

template <int MT>
__global__ __launch_bounds__(NT) void gemm_skinny_kernel(int M, int N, int K, int kchunk, int splits, const float *__restrict__ A,
                                                         long long lda, long long sA, const float *__restrict__ B, long long ldb,
                                                         long long sB, float *__restrict__ ws) {
    __shared__ float s_part[3][MT][16][64];                          // partial tiles of waves 1..3
    const int sp = blockIdx.x, z = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int kbeg0 = sp * kchunk, kend0 = min(K, kbeg0 + kchunk);
    // the workgroup's range in four wave slices, multiples of KSTEP (the last wave takes the remainder)
    const int per = ((kend0 - kbeg0 + 4 * KSTEP - 1) / (4 * KSTEP)) * KSTEP;
    const int kbeg = min(kend0, kbeg0 + wave * per), kend = min(kend0, kbeg + per);
    const float *Ab = A + (long long)z * sA, *Bb = B + (long long)z * sB;
    const float *arow[MT];
    bool aval[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        aval[mt] = 32 * mt + li < M;
        arow[mt] = Ab + (long long)min(32 * mt + li, M - 1) * lda;
    }
    const bool bval = li < N;
    const float *brow = Bb + (long long)min(li, N - 1) * ldb;
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = kbeg;
    float4 a[2][MT][4], b[2][4];                                     // two register stages: the next step's loads fly during this step's MFMAs
    auto load = [&](int st, int kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[st][mt][j] = aval[mt] ? *reinterpret_cast<const float4 *>(arow[mt] + kk + 8 * j + 4 * h) : zero4;
            b[st][j] = bval ? *reinterpret_cast<const float4 *>(brow + kk + 8 * j + 4 * h) : zero4;
        }
    };
    auto mac = [&](int st) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][mt][j].x, b[st][j].x, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][mt][j].y, b[st][j].y, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][mt][j].z, b[st][j].z, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][mt][j].w, b[st][j].w, acc[mt], 0, 0, 0);
            }
    };
    if (k + KSTEP <= kend) load(0, k);
    for (; k + 2 * KSTEP <= kend; k += 2 * KSTEP) {
        load(1, k + KSTEP);
        mac(0);
        if (k + 3 * KSTEP <= kend) load(0, k + 2 * KSTEP);
        mac(1);
    }
    if (k + KSTEP <= kend) {                                         // an odd step left: stage 0 holds it
        mac(0);
        k += KSTEP;
    }
    for (; k < kend; k += 2) {                                       // the slice's tail, two indices at a time
        const bool in = k + h < kend;
        const float bv = (bval && in) ? brow[k + h] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float av = (aval[mt] && in) ? arow[mt][k + h] : 0.f;
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mt], 0, 0, 0);
        }
    }
    // waves 1..3 -> LDS, wave 0 adds them in order and writes the workgroup's slab
    if (wave > 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) s_part[wave - 1][mt][r][lane] = acc[mt][r];
    }
    __syncthreads();
    if (wave == 0) {
        float *slab = ws + ((long long)z * splits + sp) * M * N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][r];
                v += s_part[0][mt][r][lane];
                v += s_part[1][mt][r][lane];
                v += s_part[2][mt][r][lane];
                const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * h;      // D: column = lane & 31, row = (r&3) + 8 (r>>2) + 4 h
                if (m < M && li < N) slab[(long long)m * N + li] = v;
            }
    }
}

// one wave per output element: lane l adds slabs l, l + 64, ... in order, then a fixed butterfly over the lanes
__global__ __launch_bounds__(256) void skinny_sum_kernel(int mn, int N, int slabs, const float *__restrict__ ws, float *__restrict__ C, long long ldc) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= mn) return;
    float s = 0.f;
    for (int z = lane; z < slabs; z += 64) s += ws[(long long)z * mn + e];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) C[(long long)(e / N) * ldc + (e % N)] = s;
}

int pick_splits(int K, int batch) {
    // about four workgroups per CU, at least 4 * KSTEP * 4 contraction indices each
    int splits = (1024 + batch - 1) / batch;
    const int most = K / (16 * KSTEP) > 0 ? K / (16 * KSTEP) : 1;
    if (splits > most) splits = most;
    return splits < 1 ? 1 : splits;
}

}  // namespace

// A_z [M,K] and B_z [N,K], both k-contiguous (transA = 0, transB = 1 of eap_gemm_f32_reduce): M <= 64, N <= 32, lda / ldb /
// strides multiples of 4 floats, 16-byte aligned bases
extern "C" int eap_gemm_skinny_reduce_f32_supported(int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B,
                                                    int64_t ldb, int64_t strideB) {
    return M > 0 && N > 0 && M <= 64 && N <= 32 && K >= 4096 && (lda & 3) == 0 && (ldb & 3) == 0 && (strideA & 3) == 0 && (strideB & 3) == 0 &&
           (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
}

extern "C" int64_t eap_gemm_skinny_reduce_workspace(int M, int N, int K, int batch) {
    return (int64_t)M * N * batch * pick_splits(K, batch);
}

// C[M,N] (row pitch ldc) = sum_z A_z B_z^T; workspace: eap_gemm_skinny_reduce_workspace floats.  Bit-reproducible.
extern "C" int eap_gemm_skinny_reduce_f32(int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B, int64_t ldb,
                                          int64_t strideB, float *C, int64_t ldc, int batch, float *workspace, eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (!eap_gemm_skinny_reduce_f32_supported(M, N, K, A, lda, strideA, B, ldb, strideB))
        return eap::bad_arg("gemm_skinny_reduce: M <= 64, N <= 32, K >= 4096, 16-byte aligned k-contiguous operands");
    hipStream_t s = eap::S(stream);
    int splits = pick_splits(K, batch);
    int kchunk = (K + splits - 1) / splits;
    kchunk = ((kchunk + 4 * KSTEP - 1) / (4 * KSTEP)) * (4 * KSTEP);
    splits = (K + kchunk - 1) / kchunk;                       // (never more than pick_splits: the workspace covers it)
    if (M > 32)
        hipLaunchKernelGGL(gemm_skinny_kernel<2>, dim3(splits, batch), dim3(NT), 0, s, M, N, K, kchunk, splits, A, (long long)lda, (long long)strideA, B,
                           (long long)ldb, (long long)strideB, workspace);
    else
        hipLaunchKernelGGL(gemm_skinny_kernel<1>, dim3(splits, batch), dim3(NT), 0, s, M, N, K, kchunk, splits, A, (long long)lda, (long long)strideA, B,
                           (long long)ldb, (long long)strideB, workspace);
    int e = eap::check_launch("gemm_skinny_reduce");
    if (e) return e;
    const int mn = M * N;
    hipLaunchKernelGGL(skinny_sum_kernel, dim3((mn + 3) / 4), dim3(256), 0, s, mn, N, batch * splits, workspace, C, (long long)ldc);
    eap::set_kernel(M > 32 ? "gemm_skinny_kernel<2>" : "gemm_skinny_kernel<1>");
    return eap::check_launch("gemm_skinny_reduce (sum)");
}
