// csrc/gemm_dma_f32.hip -- the dense contractions of the SO(3) convolution on the fp32 matrix cores, operands fed
// by global -> LDS DMA through a three-stage ring.
//
//     C_z[M,N] = op(A_z)[M,K] * op(B_z)[K,N]            row-major, z = batch item (optionally x k-split)
//
//   BasicSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:L48-55):  y[b][O, P*A] = W[O, C*K] * X^T[b][P*A, C*K]^T, the
//   grouped tensor X kept TRANSPOSED by the grouping kernel (csrc/so3_inter_lists.hip, layout 2): both operands
//   k-contiguous;  the re-associated backward's  dF = W2[C, O*K] * Z[O*K, R*A]  (B row-contiguous) and
//   dW = sum_b Z_b[O*K, R*A] * Fc_b[C, R*A]^T  (both k-contiguous, k-split + fixed-order slab reduction);  the
//   textbook backward's dX = W^T dY (A stored [K,M], B row-contiguous).
// Exact fp32: v_mfma_f32_32x32x2_f32 is an fmaf chain, no reduced-precision path.
//
// Why a second GEMM kernel: csrc/gemm_f32.hip stages operands through registers (global -> VGPR -> ds_write, 12
// scalar LDS stores per k-tile for k-contiguous operands) and drains everything at one __syncthreads per 16-deep
// k-tile; rocprofv3 counters (profiles/r02_a_*): matrix pipe busy 83 %, waves parked 14 % of their cycles.  Here
//   * operands go global -> LDS by DMA (global_load_lds_dwordx4): no staging registers, no LDS store pass;
//   * three LDS stages, the DMA runs two k-tiles ahead, waits are COUNTED (vmcnt(N), never 0 inside the loop) and
//     the one barrier per k-tile is a raw s_barrier, so loads stay in flight across it;
//   * k-contiguous operand: LDS image [row][4 k-blocks of 16 bytes] with the k-block slot XOR-ed by (row >> 2) & 3
//     -- applied on the per-lane SOURCE address of the DMA (its destination is lane-linear) and on the
//     ds_read_b128 address -- so the fragment reads (32 rows x 16 bytes per half-wave) are bank-conflict free; one
//     ds_read_b128 per row feeds TWO MFMA k-steps (elements {h, 2 + h} for the half-wave h);
//   * row-contiguous operand ([K][rows] in memory): LDS image [k][rows], fragment = 32 consecutive floats per
//     half-wave (conflict-free ds_read_b32).  Both images use the same k <-> (step, half-wave) assignment.
// Block tile (64 NWM) x 128 x 16, NWM x 2 waves, wave tile 64 x 64 (2 x 2 MFMA tiles); 72 KB of LDS at NWM = 4 and
// <= 128 VGPRs: two workgroups per CU.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16, BN = 128, NWN = 2, STAGES = 3;

__device__ inline unsigned lds_addr(const void *ptr) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)ptr;
}
// wave-wide 16-byte-per-lane global -> LDS DMA (lane l's bytes land at lds_dst + 16 l); inline asm so that
// hipcc's waitcnt bookkeeping does not drain it at the next LDS read -- waits are placed by hand below
__device__ inline void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct DmaArgs {
    int M, N, K;
    const float *A; long long lda, sA;
    const float *B; long long ldb, sB;
    float *C; long long ldc, sC;
    int tiles_m, tiles_n;
    int splits, kchunk;          // k-splits per batch item (1 = plain GEMM), K elements per split (multiple of BK)
};

struct Frag { float4 q[2]; float s[2][2]; };

// AKM: A stored [K, M] (m contiguous) instead of [M, K];  BKN: B stored [K, N] (n contiguous) instead of [N, K]
template <int NWM, bool AKM, bool BKN>
__global__ __launch_bounds__(64 * NWM * NWN, 2) void gemm_dma_f32_kernel(DmaArgs g) {
    constexpr int BM = 64 * NWM, NT = 64 * NWM * NWN;
    constexpr int A_PIECES = BM * 4 / NT, B_PIECES = BN * 4 / NT;      // 16-byte DMA pieces per thread and stage
    constexpr unsigned STAGE_BYTES = (BM + BN) * BK * 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // XCD-aware tile map: the tiles_m row tiles of one column panel run back to back on one XCD (block b lands
    // on XCD b % 8), so the streamed B panel is fetched from HBM once
    int id = blockIdx.x, tm, tn;
    {
        const int groups = g.tiles_n / 8 * 8;
        const int xcd = id & 7, slot = id >> 3;
        const int panel = (slot / g.tiles_m) * 8 + xcd;
        if (panel < groups && id < groups * g.tiles_m) { tn = panel; tm = slot % g.tiles_m; }
        else { const int r = id - groups * g.tiles_m; tn = groups + r / g.tiles_m; tm = r % g.tiles_m; }
    }
    const int z = blockIdx.y;
    const int bz = z / g.splits, sp = z - bz * g.splits;
    const int kbeg = sp * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int m0 = tm * BM, n0 = tn * BN;
    const float *A = g.A + (long long)bz * g.sA;
    const float *B = g.B + (long long)bz * g.sB;
    float *C = g.C + (long long)z * g.sC;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int li = lane & 31, lh = lane >> 5;

    // ---- DMA sources ----------------------------------------------------------------------------------------
    // k-contiguous tile: piece q = (row q >> 2, LDS slot q & 3) holds k-block (q & 3) ^ ((row >> 2) & 3)
    // row-contiguous tile: piece q = (k q / (R/4), 4 rows starting at 4 (q % (R/4)))
    const float *srcA[A_PIECES], *srcB[B_PIECES];
#pragma unroll
    for (int u = 0; u < A_PIECES; ++u) {
        const int q = u * NT + t;
        if (!AKM) {
            const int row = q >> 2, j = (q & 3) ^ ((row >> 2) & 3);
            srcA[u] = A + (long long)min(m0 + row, g.M - 1) * g.lda + kbeg + 4 * j;
        } else {
            const int k = q / (BM / 4), c4 = (q % (BM / 4)) * 4;
            srcA[u] = A + (long long)(kbeg + k) * g.lda + min(m0 + c4, g.M - 4);
        }
    }
#pragma unroll
    for (int u = 0; u < B_PIECES; ++u) {
        const int q = u * NT + t;
        if (!BKN) {
            const int row = q >> 2, j = (q & 3) ^ ((row >> 2) & 3);
            srcB[u] = B + (long long)min(n0 + row, g.N - 1) * g.ldb + kbeg + 4 * j;
        } else {
            const int k = q / (BN / 4), c4 = (q % (BN / 4)) * 4;
            srcB[u] = B + (long long)(kbeg + k) * g.ldb + min(n0 + c4, g.N - 4);
        }
    }
    const long long stepA = AKM ? (long long)BK * g.lda : BK;
    const long long stepB = BKN ? (long long)BK * g.ldb : BK;
    const unsigned lds0 = lds_addr(smem);
    const unsigned dstA = lds0 + (unsigned)wave * 1024u, dstB = lds0 + (unsigned)BM * 64u + (unsigned)wave * 1024u;
    // one DMA piece (u < A_PIECES: of the A tile, else of the B tile) of the k-tile that goes to `stage`
    auto issue_piece = [&](int u, int stage) {
        const unsigned sb = (unsigned)stage * STAGE_BYTES;
        if (u < A_PIECES) {
            glds16(srcA[u], __builtin_amdgcn_readfirstlane(dstA + sb + (unsigned)u * (NT * 16u)));
            srcA[u] += stepA;
        } else {
            const int v = u - A_PIECES;
            glds16(srcB[v], __builtin_amdgcn_readfirstlane(dstB + sb + (unsigned)v * (NT * 16u)));
            srcB[v] += stepB;
        }
    };
    auto issue = [&](int stage) {
#pragma unroll
        for (int u = 0; u < A_PIECES + B_PIECES; ++u) issue_piece(u, stage);
    };

    // ---- fragment read offsets (floats).  k-contiguous image: row (li), slot j ^ ((li >> 2) & 3), the second MFMA
    // tile of the operand 32 rows = 512 floats further.  row-contiguous image [k][R]: element (k, row) ----
    int offA[4], offB[4];
    {
        const int c = (li >> 2) & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            offA[j] = AKM ? 0 : (wm * 64 + li) * 16 + ((j ^ c) << 2);
            offB[j] = BKN ? 0 : BM * 16 + (wn * 64 + li) * 16 + ((j ^ c) << 2);
        }
    }
    const int rowA = lh * BM + wm * 64 + li;                    // [k][BM] image: + k0 * BM, k0 = 4j (+2)
    const int rowB = BM * 16 + lh * BN + wn * 64 + li;          // [k][BN] image

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // operands of the two MFMA k-steps of k-block j: step 0 takes k = 4j + h, step 1 k = 4j + 2 + h (h = half-wave)
    auto load_a = [&](const float *sf, int j, Frag &f) {
        if (!AKM) {
            f.q[0] = *reinterpret_cast<const float4 *>(sf + offA[j]);
            f.q[1] = *reinterpret_cast<const float4 *>(sf + offA[j] + 512);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) f.s[s][i] = sf[rowA + (4 * j + 2 * s) * BM + i * 32];
        }
    };
    auto load_b = [&](const float *sf, int j, Frag &f) {
        if (!BKN) {
            f.q[0] = *reinterpret_cast<const float4 *>(sf + offB[j]);
            f.q[1] = *reinterpret_cast<const float4 *>(sf + offB[j] + 512);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) f.s[s][i] = sf[rowB + (4 * j + 2 * s) * BN + i * 32];
        }
    };
    auto pick = [&](const Frag &f, bool rowmajor, int s, int i) -> float {
        if (rowmajor) return f.s[s][i];
        const float4 v = f.q[i];
        return s == 0 ? (lh ? v.y : v.x) : (lh ? v.w : v.z);
    };

    const int nt = (kend - kbeg) / BK;
    if (nt > 0) issue(0);
    if (nt > 1) issue(1);
    int stage = 0;
    for (int it = 0; it < nt; ++it) {
        // my pieces of k-tile `it` have landed (those of it + 1 may still be in flight) ...
        if (it + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_PIECES + B_PIECES) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and so have everyone's; everyone has finished reading k-tile it - 1, whose stage is refilled next
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                                   // no LDS read may move above the barrier
        // k-tile it + 2 goes to the stage everyone has just left; its pieces are requested one per k-block, from
        // the MIDDLE of the block's MFMAs: eight waves bursting 24 DMA instructions right behind the barrier
        // queue on the CU's one address path and nobody reaches an MFMA until it drains (13 % of the kernel in
        // the first version of this loop)
        const bool more = it + 2 < nt;
        const int nstage = stage >= 1 ? stage - 1 : STAGES - 1;          // (it + 2) % 3 == (stage + 2) % 3
        const float *sf = reinterpret_cast<const float *>(smem) + (size_t)stage * (STAGE_BYTES / 4);
        // fragments of k-block j + 1 are requested before the MFMAs of k-block j are issued
        Frag fa[2], fb[2];
        load_a(sf, 0, fa[0]);
        load_b(sf, 0, fb[0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cur = j & 1, nxt = cur ^ 1;
            if (j < 3) {
                load_a(sf, j + 1, fa[nxt]);
                load_b(sf, j + 1, fb[nxt]);
            }
            float a[2][2], b[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i) { a[s][i] = pick(fa[cur], AKM, s, i); b[s][i] = pick(fb[cur], BKN, s, i); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][0], b[s][0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][0], b[s][1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][1], b[s][0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][1], b[s][1], acc[1][1], 0, 0, 0);
                if (s == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (more && j < A_PIECES + B_PIECES) issue_piece(j, nstage);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stage = stage + 1 == STAGES ? 0 : stage + 1;
    }

    // ---- epilogue: D[i][j] of a 32x32 tile sits at col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5):
    // per register the two half-waves write one 128-byte row segment each
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M && col < g.N) C[(long long)row * g.ldc + col] = acc[i][j][r];
            }
        }
}

// sum `slabs` partial [M,N] slabs (contiguous, pitch M*N) into C (leading dimension ldc), in slab order
__global__ void dma_reduce_slabs_kernel(long long mn, int N, int slabs, const float *__restrict__ ws,
                                        float *__restrict__ C, long long ldc) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= mn) return;
    float s = 0.f;
    for (int z = 0; z < slabs; ++z) s += ws[(long long)z * mn + e];
    C[(e / N) * ldc + (e % N)] = s;
}

template <int NWM, bool AKM, bool BKN>
int launch_one(DmaArgs g, int zcount, hipStream_t s) {
    constexpr int BM = 64 * NWM;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const size_t shmem = (size_t)STAGES * (BM + BN) * BK * 4;
    auto kern = gemm_dma_f32_kernel<NWM, AKM, BKN>;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                          "gemm_dma_f32 shared memory");
    if (e) return e;
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, zcount), dim3(64 * NWM * NWN), shmem, s, g);
    return eap::check_launch("gemm_dma_f32");
}

int launch(bool akm, bool bkn, const DmaArgs &g, int zcount, hipStream_t s) {
    if (zcount > 65535) return eap::bad_arg("gemm_dma_f32: batch * splits exceeds 65535");
    if (g.M > 128) {
        if (akm) return bkn ? launch_one<4, true, true>(g, zcount, s) : launch_one<4, true, false>(g, zcount, s);
        return bkn ? launch_one<4, false, true>(g, zcount, s) : launch_one<4, false, false>(g, zcount, s);
    }
    if (akm) return bkn ? launch_one<2, true, true>(g, zcount, s) : launch_one<2, true, false>(g, zcount, s);
    return bkn ? launch_one<2, false, true>(g, zcount, s) : launch_one<2, false, false>(g, zcount, s);
}

bool supported(int transA, int transB, int M, int N, int K, const float *A, int64_t lda, int64_t sA, const float *B,
               int64_t ldb, int64_t sB) {
    if (M <= 0 || N <= 0 || K < BK || (K % BK) != 0) return false;
    if ((lda & 3) || (ldb & 3) || (sA & 3) || (sB & 3)) return false;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return false;
    if (transA && ((M & 3) || M < 4)) return false;           // row-contiguous operands move in 4-row pieces
    if (!transB && ((N & 3) || N < 4)) return false;
    return true;
}

int pick_splits(int M, int N, int K, int batch) {
    // enough blocks to fill 256 CUs about three times over, at least 8 k-tiles per split
    const int tiles = ((M + 255) / 256) * ((N + BN - 1) / BN) * batch;
    int splits = (1536 + tiles - 1) / tiles;
    const int max_splits = K / (8 * BK) > 0 ? K / (8 * BK) : 1;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    return splits;
}

}  // namespace

// transA = 0: A is [M,K] (k contiguous); 1: A stored [K,M].  transB = 0: B is [K,N] (n contiguous); 1: B stored [N,K].
// (the conventions of eap_gemm_f32)
extern "C" int eap_gemm_dma_f32_supported(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                                          int64_t strideA, const float *B, int64_t ldb, int64_t strideB) {
    return supported(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB) ? 1 : 0;
}

extern "C" int eap_gemm_dma_f32(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                                int64_t strideA, const float *B, int64_t ldb, int64_t strideB, float *C, int64_t ldc,
                                int64_t strideC, int batch, eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (!supported(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB))
        return eap::bad_arg("gemm_dma_f32: K must be a multiple of 16, leading dimensions / strides multiples of 4, bases 16-byte "
                            "aligned, row-contiguous operands a multiple of 4 rows (use eap_gemm_f32 otherwise)");
    DmaArgs g{M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, 0, 0, 1, K};
    return launch(transA != 0, transB == 0, g, batch, eap::S(stream));
}

extern "C" int64_t eap_gemm_dma_f32_reduce_workspace(int M, int N, int K, int batch) {
    return (int64_t)M * N * batch * pick_splits(M, N, K, batch);
}

// C[M,N] = sum_z sum_k op(A_z)[M,k] op(B_z)[k,N]: batch * splits partial slabs in `workspace`, reduced in slab order
extern "C" int eap_gemm_dma_f32_reduce(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                                       int64_t strideA, const float *B, int64_t ldb, int64_t strideB, float *C,
                                       int64_t ldc, int batch, float *workspace, eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (!supported(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB))
        return eap::bad_arg("gemm_dma_f32_reduce: unsupported operand shape / alignment (use eap_gemm_f32_reduce)");
    hipStream_t s = eap::S(stream);
    const int splits = pick_splits(M, N, K, batch);
    const int kchunk = ((K / BK + splits - 1) / splits) * BK;
    DmaArgs g{M, N, K, A, lda, strideA, B, ldb, strideB, workspace, N, (long long)M * N, 0, 0, splits, kchunk};
    int e = launch(transA != 0, transB == 0, g, batch * splits, s);
    if (e) return e;
    const long long mn = (long long)M * N;
    hipLaunchKernelGGL(dma_reduce_slabs_kernel, dim3(eap::cdiv(mn, 256)), dim3(256), 0, s, mn, N, batch * splits, workspace, C,
                       (long long)ldc);
    return eap::check_launch("gemm_dma_f32_reduce");
}
