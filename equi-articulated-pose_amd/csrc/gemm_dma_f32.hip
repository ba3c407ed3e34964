// csrc/gemm_dma_f32.hip -- the dense contractions of the SO(3) convolution on the fp32 matrix cores: ONE wave per
// SIMD, operands fed by global -> LDS DMA through a four-stage ring.
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
// Design, from the counters (profiles/r02_gemm_pmc.json):
//   csrc/gemm_f32.hip (register staging, 2 x 8 waves per CU, one __syncthreads per k-tile): matrix pipe busy 81 %,
//   waves parked 21 % of their cycles.  A first DMA version with the same 2 x 8-wave geometry: 86 %.  The vendor
//   library's kernel on the same product: 96 % -- with ONE wave per SIMD that owns the matrix pipe.  Sixteen waves in
//   two barrier groups keep stalling each other: whenever a group waits for its slowest wave, the gaps of the other
//   group on that SIMD go unfilled.  So here:
//   * a workgroup is 4 waves, one per SIMD, each with a (32 MI) x (32 NI) tile of accumulators (up to 128 x 128 =
//     256 VGPRs; the file is 512 per lane with one wave per SIMD); block tile 256 x 256 (or 128 x 256 / 256 x 128);
//   * operands go global -> LDS by DMA (global_load_lds_dwordx4): no staging registers, no LDS store pass; the DMA
//     instructions are issued one per MFMA k-step from inside the MFMA stream, never in a burst;
//   * four LDS stages of one 16-deep k-tile each; the DMA runs three tiles ahead, waits are COUNTED (vmcnt(N)) and
//     the one barrier per k-tile is a raw s_barrier, so loads stay in flight across it.  The barrier of tile t
//     certifies tile t + 1, so the fragments of the next tile's first k-block are read BEFORE the current tile's
//     last MFMAs are issued: the matrix pipe never waits for LDS behind a barrier;
//   * k-contiguous operand: LDS image [row][4 k-blocks of 16 bytes] with the k-block slot XOR-ed by (row >> 2) & 3
//     -- applied on the per-lane SOURCE address of the DMA (its destination is lane-linear) and on the
//     ds_read_b128 address -- so the fragment reads (32 rows x 16 bytes per half-wave) are bank-conflict free
//     (SQ_LDS_BANK_CONFLICT = 0); one ds_read_b128 per row feeds TWO MFMA k-steps (elements {h, 2 + h} for the
//     half-wave h);
//   * row-contiguous operand ([K][rows] in memory): LDS image [k][rows], fragment = 32 consecutive floats per
//     half-wave (conflict-free ds_read_b32).  Both images use the same k <-> (step, half-wave) assignment.
#include "common.h"

#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16, STAGES = 4, NT = 256;

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

// fragments of one k-block of an operand: k-contiguous image -> NI_ x 16 bytes (q), row-contiguous -> 2 x NI_ floats (s)
template <int NI_>
struct Frag { float4 q[NI_]; float s[2][NI_]; };

// WM x WN = 4 waves, wave tile (32 MI) x (32 NI).  AKM: A stored [K, M] (m contiguous) instead of [M, K];
// BKN: B stored [K, N] (n contiguous) instead of [N, K]
template <int WM, int WN, int MI, int NI, bool AKM, bool BKN>
__global__ __launch_bounds__(NT, 1) void gemm_dma_f32_kernel(DmaArgs g) {
    static_assert(WM * WN == 4, "four waves: one per SIMD");
    constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
    constexpr int A_PIECES = BM * 4 / NT, B_PIECES = BN * 4 / NT, NP = A_PIECES + B_PIECES;   // 16-byte DMA pieces per thread and stage
    static_assert(NP <= 8, "one DMA piece per MFMA k-step");
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
    const int wm = wave / WN, wn = wave % WN;
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
    auto issue_tile = [&](int stage) {
#pragma unroll
        for (int u = 0; u < NP; ++u) issue_piece(u, stage);
    };

    // ---- fragment read offsets (floats).  k-contiguous image: row (li), slot j ^ ((li >> 2) & 3), the i-th MFMA
    // tile of the operand 32 rows = 512 floats further.  row-contiguous image [k][R]: element (k, row) ----
    int offA[4], offB[4];
    {
        const int c = (li >> 2) & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            offA[j] = AKM ? 0 : (wm * 32 * MI + li) * 16 + ((j ^ c) << 2);
            offB[j] = BKN ? 0 : BM * 16 + (wn * 32 * NI + li) * 16 + ((j ^ c) << 2);
        }
    }
    const int rowA = lh * BM + wm * 32 * MI + li;               // [k][BM] image: + k0 * BM, k0 = 4j (+2)
    const int rowB = BM * 16 + lh * BN + wn * 32 * NI + li;     // [k][BN] image

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // operands of the two MFMA k-steps of k-block j: step 0 takes k = 4j + h, step 1 k = 4j + 2 + h (h = half-wave)
    auto load_a = [&](const float *sf, int j, Frag<MI> &f) {
        if (!AKM) {
#pragma unroll
            for (int i = 0; i < MI; ++i) f.q[i] = *reinterpret_cast<const float4 *>(sf + offA[j] + i * 512);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < MI; ++i) f.s[s][i] = sf[rowA + (4 * j + 2 * s) * BM + i * 32];
        }
    };
    auto load_b = [&](const float *sf, int j, Frag<NI> &f) {
        if (!BKN) {
#pragma unroll
            for (int i = 0; i < NI; ++i) f.q[i] = *reinterpret_cast<const float4 *>(sf + offB[j] + i * 512);
        } else {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < NI; ++i) f.s[s][i] = sf[rowB + (4 * j + 2 * s) * BN + i * 32];
        }
    };
    auto pick = [&](const float4 &v, int s) -> float { return s == 0 ? (lh ? v.y : v.x) : (lh ? v.w : v.z); };

    const int nt = (kend - kbeg) / BK;
    const float *lds_f = reinterpret_cast<const float *>(smem);
    // two fragment sets, named (never indexed at run time: they must stay in registers)
    Frag<MI> fa0, fa1;
    Frag<NI> fb0, fb1;
    if (nt > 0) {
        issue_tile(0);
        if (nt > 1) issue_tile(1);
        if (nt > 2) issue_tile(2);
        if (nt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
        else if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_a(lds_f, 0, fa0);
        load_b(lds_f, 0, fb0);
    }
    int stage = 0;
    for (int it = 0; it < nt; ++it) {
        // my pieces of k-tile it + 1 have landed (those of it + 2 may still be in flight) ...
        if (it + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and so have everyone's; everyone has finished reading k-tile it - 1, whose stage takes k-tile it + 3
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                                   // no LDS read may move above the barrier
        const bool more = it + 3 < nt;
        const bool next_tile = it + 1 < nt;
        const int nstage = stage == 0 ? STAGES - 1 : stage - 1;          // (it + 3) % 4
        const int fstage = stage + 1 == STAGES ? 0 : stage + 1;          // (it + 1) % 4
        const float *sf = lds_f + (size_t)stage * (STAGE_BYTES / 4);
        const float *sn = lds_f + (size_t)fstage * (STAGE_BYTES / 4);
        // one k-block: request the NEXT block's fragments (of the next k-tile after the last block), then the
        // 2 x MI x NI MFMAs of this one with one DMA piece of k-tile it + 3 per k-step inside the MFMA stream
        auto kblock = [&](auto jc, const Frag<MI> &ca, const Frag<NI> &cb, Frag<MI> &na_, Frag<NI> &nb_) {
            constexpr int j = decltype(jc)::value;
            if (j < 3) {
                load_a(sf, j + 1, na_);
                load_b(sf, j + 1, nb_);
            } else if (next_tile) {
                load_a(sn, 0, na_);
                load_b(sn, 0, nb_);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float a[MI], b[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = AKM ? ca.s[s][i] : pick(ca.q[i], s);
#pragma unroll
                for (int i = 0; i < NI; ++i) b[i] = BKN ? cb.s[s][i] : pick(cb.q[i], s);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int jn = 0; jn < NI; ++jn) {
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[jn], acc[i][jn], 0, 0, 0);
                        if (i == 0 && jn == NI - 1 && 2 * j + s < NP) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) issue_piece(2 * j + s, nstage);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        kblock(std::integral_constant<int, 0>{}, fa0, fb0, fa1, fb1);
        kblock(std::integral_constant<int, 1>{}, fa1, fb1, fa0, fb0);
        kblock(std::integral_constant<int, 2>{}, fa0, fb0, fa1, fb1);
        kblock(std::integral_constant<int, 3>{}, fa1, fb1, fa0, fb0);
        stage = fstage;
    }

    // ---- epilogue: D[i][j] of a 32x32 tile sits at col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5):
    // per register the two half-waves write one 128-byte row segment each
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + wn * 32 * NI + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 32 * MI + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
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

template <int WM, int WN, int MI, int NI, bool AKM, bool BKN>
int launch_one(DmaArgs g, int zcount, hipStream_t s) {
    constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const size_t shmem = (size_t)STAGES * (BM + BN) * BK * 4;
    auto kern = gemm_dma_f32_kernel<WM, WN, MI, NI, AKM, BKN>;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                          "gemm_dma_f32 shared memory");
    if (e) return e;
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, zcount), dim3(NT), shmem, s, g);
    return eap::check_launch("gemm_dma_f32");
}

template <bool AKM, bool BKN>
int launch_shape(const DmaArgs &g, int zcount, hipStream_t s) {
    // block tile 256 x 256 (wave tile 128 x 128); 128 x 256 for M <= 128; 256 x 128 for N <= 128
    if (g.M <= 128) return launch_one<1, 4, 4, 2, AKM, BKN>(g, zcount, s);
    if (g.N <= 128) return launch_one<4, 1, 2, 4, AKM, BKN>(g, zcount, s);
    return launch_one<2, 2, 4, 4, AKM, BKN>(g, zcount, s);
}

int launch(bool akm, bool bkn, const DmaArgs &g, int zcount, hipStream_t s) {
    if (zcount > 65535) return eap::bad_arg("gemm_dma_f32: batch * splits exceeds 65535");
    if (akm) return bkn ? launch_shape<true, true>(g, zcount, s) : launch_shape<true, false>(g, zcount, s);
    return bkn ? launch_shape<false, true>(g, zcount, s) : launch_shape<false, false>(g, zcount, s);
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
    // one workgroup per CU: enough blocks to fill 256 CUs about four times over, at least 8 k-tiles per split
    const int bm = M <= 128 ? 128 : 256, bn = N <= 128 ? 128 : 256;
    const int tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch;
    int splits = (1024 + tiles - 1) / tiles;
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
