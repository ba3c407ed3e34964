// csrc/gemm_dma_f32.hip -- the dense contractions of the SO(3) convolution on the fp32 matrix cores: ONE wave per
// SIMD, 256 accumulator registers per wave, operands streamed through a three-stage LDS ring.
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
// Design, step by step from the counters (profiles/r02_gemm_pmc.json; SQ_VALU_MFMA_BUSY_CYCLES / cycles / SIMDs):
//   1. csrc/gemm_f32.hip -- register staging, 2 x 8 waves per CU, one __syncthreads per k-tile: matrix pipe busy
//      81 %, waves parked (SQ_WAIT_ANY) 21 % of their cycles.  The vendor library's kernel on the same product:
//      96 %, with ONE wave per SIMD that owns the matrix pipe and nobody to wait for.
//   2. global -> LDS DMA (global_load_lds_dwordx4) instead of register staging, same 2 x 8-wave geometry: 86 %.
//   3. one wave per SIMD, 128 x 128 accumulator tile per wave, DMA-fed 4-stage ring: parked time 21 % -> 5 %, but
//      still 86 % busy: with a single wave per SIMD nothing covers the issue cost of a DMA instruction (60-180
//      cycles each against the 64-cycle shadow of an MFMA; eight per k-tile).
//   4. this file: one wave per SIMD, and the operands travel global -> VGPR -> LDS.  A plain global_load_dwordx4
//      issues in a few cycles, its data is not needed until the NEXT k-tile (eight staging quads = 32 of the 512
//      VGPRs a lone wave owns), and the ds_write_b128 that parks it in LDS is one more filler in an MFMA shadow:
//      per MFMA k-step exactly one load and one store ride along.
// Pipeline (all waits by the compiler, no inline asm): tile t is multiplied out of stage t % 3 while tile t + 2 is
// being written into stage (t + 2) % 3 from registers loaded during tile t - 1, and tile t + 3 is being loaded.  One
// __syncthreads per k-tile; the barrier of tile t certifies stage (t + 1) % 3, so the fragments of the next tile's
// first k-block are read BEFORE the current tile's last MFMAs are issued: the pipe never waits for LDS behind a
// barrier.
//   * k-contiguous operand: LDS image [row][4 k-blocks of 16 bytes], the k-block slot XOR-ed by (row >> 2) & 3 on
//     the ds_write and the ds_read_b128 side: fragment reads (32 rows x 16 bytes per half-wave) are conflict-free
//     (SQ_LDS_BANK_CONFLICT = 0); one ds_read_b128 per row feeds TWO MFMA k-steps (elements {h, 2 + h} for the
//     half-wave h);
//   * row-contiguous operand ([K][rows] in memory): LDS image [k][rows], fragment = 32 consecutive floats per
//     half-wave (conflict-free ds_read_b32).  Both images use the same k <-> (step, half-wave) assignment.
// Block tile 256 x 256 (wave tile 128 x 128), or 128 x 256 / 256 x 128 for small M / N; 96 KB of LDS.
#include "common.h"

#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16, STAGES = 3, NT = 256;

struct DmaArgs {
    int M, N, K;
    const float *A; long long lda, sA;
    const float *B; long long ldb, sB;
    float *C; long long ldc, sC;
    int tiles_m, tiles_n;
    int splits, kchunk;          // k-splits per batch item (1 = plain GEMM), K elements per split (multiple of BK)
};

// the k-tile in flight: eight named 16-byte quads (an indexed array of vectors is not promoted to registers)
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Staging {
    f32x4 v0, v1, v2, v3, v4, v5, v6, v7, v8, v9;
    template <int U>
    __device__ __forceinline__ f32x4 &at() {
        if constexpr (U == 0) return v0; else if constexpr (U == 1) return v1; else if constexpr (U == 2) return v2;
        else if constexpr (U == 3) return v3; else if constexpr (U == 4) return v4; else if constexpr (U == 5) return v5;
        else if constexpr (U == 6) return v6; else if constexpr (U == 7) return v7; else if constexpr (U == 8) return v8;
        else return v9;
    }
};

template <typename F, int... U>
__device__ __forceinline__ void for_each_piece(F &&f, std::integer_sequence<int, U...>) {
    (f(std::integral_constant<int, U>{}), ...);
}

// fragments of one k-PAIR (8 k's, four MFMA k-steps) of an operand: k-contiguous image -> NI_ x 16 bytes (q: this
// half-wave's own k-block of the pair, all four elements used), row-contiguous -> 4 x NI_ floats (s)
template <int NI_>
struct Frag { float4 q[NI_]; float s[4][NI_]; };

// WM x WN = 4 waves, wave tile (32 MI) x (32 NI).  AKM: A stored [K, M] (m contiguous) instead of [M, K];
// BKN: B stored [K, N] (n contiguous) instead of [N, K]
// DBG (timing experiments only, results are wrong for DBG > 0; profiles/r02_gemm_ablation.txt): 1 = no staging
// traffic inside the k-loop, 2 = also no fragment reads (the bare MFMA stream: 152.6 TFLOP/s)
template <int WM, int WN, int MI, int NI, bool AKM, bool BKN, int DBG = 0>
__global__ __launch_bounds__(NT, 1) void gemm_dma_f32_kernel(DmaArgs g) {
    static_assert(WM * WN == 4, "four waves: one per SIMD");
    constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
    constexpr int A_PIECES = BM * 4 / NT, B_PIECES = BN * 4 / NT, NP = A_PIECES + B_PIECES;   // 16-byte DMA pieces per thread and stage
    static_assert(NP >= 4 && NP <= 10, "one staged piece per MFMA k-step (a second one in the first two for the 128 x 512 tile)");
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

    // ---- staging: thread t moves NP 16-byte pieces per k-tile, global -> VGPR -> LDS ------------------------------
    // k-contiguous tile: piece q = (row q >> 2, k-block q & 3), parked at LDS slot (q & 3) ^ ((row >> 2) & 3) of its row
    // row-contiguous tile: piece q = (k q / (R/4), 4 rows starting at 4 (q % (R/4))), LDS image [k][R]
    const float *src[NP];
    int ldsoff[NP];                                              // floats, within a stage
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const bool isA = u < A_PIECES;
        const int q = (isA ? u : u - A_PIECES) * NT + t;
        if (isA) {
            if (!AKM) {
                const int row = q >> 2, j = q & 3;
                src[u] = A + (long long)min(m0 + row, g.M - 1) * g.lda + kbeg + 4 * j;
                ldsoff[u] = row * 16 + ((j ^ ((row >> 2) & 3)) << 2);
            } else {
                const int k = q / (BM / 4), c4 = (q % (BM / 4)) * 4;
                src[u] = A + (long long)(kbeg + k) * g.lda + min(m0 + c4, g.M - 4);
                ldsoff[u] = k * BM + c4;
            }
        } else {
            if (!BKN) {
                const int row = q >> 2, j = q & 3;
                src[u] = B + (long long)min(n0 + row, g.N - 1) * g.ldb + kbeg + 4 * j;
                ldsoff[u] = BM * 16 + row * 16 + ((j ^ ((row >> 2) & 3)) << 2);
            } else {
                const int k = q / (BN / 4), c4 = (q % (BN / 4)) * 4;
                src[u] = B + (long long)(kbeg + k) * g.ldb + min(n0 + c4, g.N - 4);
                ldsoff[u] = BM * 16 + k * BN + c4;
            }
        }
    }
    const long long stepA = AKM ? (long long)BK * g.lda : BK;
    const long long stepB = BKN ? (long long)BK * g.ldb : BK;
    float *lds_w = reinterpret_cast<float *>(smem);
    Staging stg;                                                 // the k-tile in flight
    // (pieces are always named by compile-time constants: the arrays must stay in registers)
    // `advance` = 0 past the last k-tile: the piece is simply re-read (never used) so that the loop body has no
    // branch around a load and the compiler can count its waits (vmcnt(N), not vmcnt(0))
    auto load_piece = [&](auto uc, long long advance = 1) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        stg.template at<u>() = *reinterpret_cast<const f32x4 *>(src[u]);
        src[u] += (u < A_PIECES ? stepA : stepB) * advance;
    };
    auto park_piece = [&](auto uc, int stage) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        *reinterpret_cast<f32x4 *>(lds_w + (size_t)stage * (STAGE_BYTES / 4) + ldsoff[u]) = stg.template at<u>();
    };
#define EAP_EACH_PIECE(STMT)                                                                                     \
    do {                                                                                                         \
        { constexpr std::integral_constant<int, 0> uc{}; STMT; }                                                 \
        { constexpr std::integral_constant<int, 1> uc{}; STMT; }                                                 \
        { constexpr std::integral_constant<int, 2> uc{}; STMT; }                                                 \
        { constexpr std::integral_constant<int, 3> uc{}; STMT; }                                                 \
        if constexpr (NP > 4) { constexpr std::integral_constant<int, 4> uc{}; STMT; }                           \
        if constexpr (NP > 5) { constexpr std::integral_constant<int, 5> uc{}; STMT; }                           \
        if constexpr (NP > 6) { constexpr std::integral_constant<int, 6> uc{}; STMT; }                           \
        if constexpr (NP > 7) { constexpr std::integral_constant<int, 7> uc{}; STMT; }                           \
        if constexpr (NP > 8) { constexpr std::integral_constant<int, 8> uc{}; STMT; }                           \
        if constexpr (NP > 9) { constexpr std::integral_constant<int, 9> uc{}; STMT; }                           \
    } while (0)

    // ---- fragment reads.  A k-tile is two k-PAIRS of 8 k's.  In MFMA k-step s (0..3) of pair p the half-wave h
    // supplies k = 8p + 4h + s: for a k-contiguous image that is element s of the 16-byte k-block 2p + h of the
    // lane's row -- ONE ds_read_b128 per row and pair, every byte of it used, no selects (a first version read the
    // same k-block in both half-waves and used half of it: twice the LDS reads, and every ds_read_b128 of a lone
    // wave costs the matrix pipe ~25 cycles).  Slot of k-block j in row r: j ^ ((r >> 2) & 3); MFMA tile i of the
    // operand is 32 rows = 512 floats further.  Row-contiguous image [k][R]: element (k, row). ----
    int offA[2], offB[2];
    {
        const int c = (li >> 2) & 3;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            offA[p] = AKM ? 0 : (wm * 32 * MI + li) * 16 + (((2 * p + lh) ^ c) << 2);
            offB[p] = BKN ? 0 : BM * 16 + (wn * 32 * NI + li) * 16 + (((2 * p + lh) ^ c) << 2);
        }
    }
    const int rowA = 4 * lh * BM + wm * 32 * MI + li;           // [k][BM] image: + (8p + s) * BM
    const int rowB = BM * 16 + 4 * lh * BN + wn * 32 * NI + li; // [k][BN] image

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the r-th fragment read of k-pair p of the stage at `base` (A tiles first, then B tiles; a row-contiguous
    // operand has four reads per tile, one per k-step)
    constexpr int RA = AKM ? 4 * MI : MI, RB = BKN ? 4 * NI : NI, NREADS = RA + RB;
    auto frag_read = [&](const float *base, int p, int r, Frag<MI> &na_, Frag<NI> &nb_) __attribute__((always_inline)) {
        if (r < RA) {
            if (!AKM) na_.q[r] = *reinterpret_cast<const float4 *>(base + offA[p] + r * 512);
            else na_.s[r / MI][r % MI] = base[rowA + (8 * p + r / MI) * BM + (r % MI) * 32];
        } else {
            const int q = r - RA;
            if (!BKN) nb_.q[q] = *reinterpret_cast<const float4 *>(base + offB[p] + q * 512);
            else nb_.s[q / NI][q % NI] = base[rowB + (8 * p + q / NI) * BN + (q % NI) * 32];
        }
    };
    auto elem = [](const float4 &v, int s) __attribute__((always_inline)) -> float { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; };

    const int nt = (kend - kbeg) / BK;
    const float *lds_f = reinterpret_cast<const float *>(smem);
    // two fragment sets, named (never indexed at run time: they must stay in registers)
    Frag<MI> fa0, fa1;
    Frag<NI> fb0, fb1;
    if (nt > 0) {
        // tiles 0 and 1 into their stages, tile 2 into the staging registers
        EAP_EACH_PIECE(load_piece(uc, nt > 1 ? 1 : 0));
        EAP_EACH_PIECE(park_piece(uc, 0));
        EAP_EACH_PIECE(load_piece(uc, nt > 2 ? 1 : 0));
        EAP_EACH_PIECE(park_piece(uc, 1));
        EAP_EACH_PIECE(load_piece(uc, nt > 3 ? 1 : 0));
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NREADS; ++r) frag_read(lds_f, 0, r, fa0, fb0);
    }
    int stage = 0;
    for (int it = 0; it < nt; ++it) {
        // everyone has finished reading k-tile it - 1 (its stage takes k-tile it + 2 now) and has parked its pieces
        // of k-tile it + 1
        if (it > 0) __syncthreads();
        const bool next_tile = it + 1 < nt;
        const long long adv = it + 4 < nt ? 1 : 0;      // after loading k-tile it + 3 the pointers move on only if k-tile it + 4 exists
        const int pstage = stage == 0 ? STAGES - 1 : stage - 1;          // (it + 2) % 3
        const int fstage = stage + 1 == STAGES ? 0 : stage + 1;          // (it + 1) % 3
        const float *sf = lds_f + (size_t)stage * (STAGE_BYTES / 4);
        const float *sn = lds_f + (size_t)fstage * (STAGE_BYTES / 4);
        // one k-block: request the NEXT block's fragments (of the next k-tile after the last block), then the
        // 2 x MI x NI MFMAs of this one; per k-step one staged piece of k-tile it + 2 is parked in LDS and its
        // register reloaded with the same piece of k-tile it + 3, from inside the MFMA stream
        // one MFMA k-step (s = 0..3) of k-pair p.  Between the MFMAs ride, evenly spaced: the fragment reads of the
        // NEXT pair (of the next k-tile's first pair after the second), one parked staging piece and its reload
        auto kstep = [&](auto pc, auto sc, const Frag<MI> &ca, const Frag<NI> &cb, Frag<MI> &na_, Frag<NI> &nb_) __attribute__((always_inline)) {
            constexpr int p = decltype(pc)::value, s = decltype(sc)::value, u = 4 * p + s;
            const float *nbase = p == 0 ? sf : sn;
            constexpr int np = p == 0 ? 1 : 0;
            const bool do_reads = p == 0 || next_tile;
            float a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = AKM ? ca.s[s][i] : elem(ca.q[i], s);
#pragma unroll
            for (int i = 0; i < NI; ++i) b[i] = BKN ? cb.s[s][i] : elem(cb.q[i], s);
            __builtin_amdgcn_sched_barrier(0);                 // then the MFMA stream in this order
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int jn = 0; jn < NI; ++jn) {
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[jn], acc[i][jn], 0, 0, 0);
                    // Everything that is not an MFMA rides in ONE cluster per k-step, right here: a lone wave pays
                    // ~26 cycles of matrix-pipe time for every PLACE where other instructions sit between two MFMAs,
                    // almost regardless of how many there are (tools/microbench/mfma_stream.hip: 5 or 10 VALU ops per
                    // gap both cost 64 -> 90 cycles per MFMA).  Spread over the stream, the 24 riders of a k-tile cost
                    // 7.6 %; clustered per k-step (a quarter of the next pair's fragment reads, one parked staging
                    // piece, its reload) they cost 8 gaps.
                    if (i == 0 && jn == NI - 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (DBG < 2) {
#pragma unroll
                            for (int r = 0; r < NREADS; ++r)
                                if (r * 4 / NREADS == s && do_reads) frag_read(nbase, np, r, na_, nb_);
                        }
                        if constexpr (u < NP && DBG == 0) {
                            park_piece(std::integral_constant<int, u>{}, pstage);     // (a stage nobody reads once it + 2 >= nt)
                            load_piece(std::integral_constant<int, u>{}, adv);
                        }
                        if constexpr (u + 8 < NP && DBG == 0) {
                            park_piece(std::integral_constant<int, u + 8>{}, pstage);
                            load_piece(std::integral_constant<int, u + 8>{}, adv);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto kpair = [&](auto pc, const Frag<MI> &ca, const Frag<NI> &cb, Frag<MI> &na_, Frag<NI> &nb_) __attribute__((always_inline)) {
            kstep(pc, std::integral_constant<int, 0>{}, ca, cb, na_, nb_);
            kstep(pc, std::integral_constant<int, 1>{}, ca, cb, na_, nb_);
            kstep(pc, std::integral_constant<int, 2>{}, ca, cb, na_, nb_);
            kstep(pc, std::integral_constant<int, 3>{}, ca, cb, na_, nb_);
        };
        kpair(std::integral_constant<int, 0>{}, fa0, fb0, fa1, fb1);
        kpair(std::integral_constant<int, 1>{}, fa1, fb1, fa0, fb0);
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
    {
        char nm[96];
        snprintf(nm, sizeof(nm), "gemm_dma_f32_kernel<%d, %d, %d, %d, %s, %s>", WM, WN, MI, NI, AKM ? "true" : "false", BKN ? "true" : "false");
        eap::set_kernel(nm);
    }
    return eap::check_launch("gemm_dma_f32");
}

template <int DBG>
int launch_debug(DmaArgs g, int zcount, hipStream_t s) {
    g.tiles_m = (g.M + 255) / 256;
    g.tiles_n = (g.N + 255) / 256;
    const size_t shmem = (size_t)STAGES * 512 * BK * 4;
    auto kern = gemm_dma_f32_kernel<2, 2, 4, 4, false, false, DBG>;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem), "gemm_dma_f32 debug");
    if (e) return e;
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, zcount), dim3(NT), shmem, s, g);
    return eap::check_launch("gemm_dma_f32 (debug variant)");
}

template <bool AKM, bool BKN>
int launch_shape(const DmaArgs &g, int zcount, hipStream_t s) {
    // block tile 256 x 256 (wave tile 128 x 128); for M <= 128: 128 x 512 (the same wave tile) when that still leaves
    // four workgroups per CU, else 128 x 256; 256 x 128 for N <= 128; 64 x 512 / 512 x 64 for a side of at most 64
    // a 64-wide side: 64 x 512 / 512 x 64 tiles (half of a 128-row tile would be padding)
    if (g.M <= 64 && g.N > 64) return launch_one<1, 4, 2, 4, AKM, BKN>(g, zcount, s);
    if (g.N <= 64 && g.M > 64) return launch_one<4, 1, 4, 2, AKM, BKN>(g, zcount, s);
    if (g.M <= 128 && (long long)((g.N + 511) / 512) * zcount >= 1024) return launch_one<1, 4, 4, 4, AKM, BKN>(g, zcount, s);
    if (g.M <= 128) return launch_one<1, 4, 4, 2, AKM, BKN>(g, zcount, s);
    if (g.N <= 128) return launch_one<4, 1, 2, 4, AKM, BKN>(g, zcount, s);
    return launch_one<2, 2, 4, 4, AKM, BKN>(g, zcount, s);
}

int launch(bool akm, bool bkn, const DmaArgs &g, int zcount, hipStream_t s) {
    if (zcount > 65535) return eap::bad_arg("gemm_dma_f32: batch * splits exceeds 65535");
#ifdef EAP_ABLATION
    // Timing ablations (WRONG RESULTS), only in a library built with `make ABLATION=1`: EAP_GEMM_DEBUG=1,2
    // (tools/gemm_only.py, profiles/r02_gemm_ablation.txt).  A production build never reads the variable.
    if (!akm && !bkn && g.M > 128 && g.N > 128) {
        static const int dbg = getenv("EAP_GEMM_DEBUG") ? atoi(getenv("EAP_GEMM_DEBUG")) : 0;
        if (dbg == 1) return launch_debug<1>(g, zcount, s);
        if (dbg == 2) return launch_debug<2>(g, zcount, s);
    }
#endif
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
    const int bm = (M <= 64 && N > 64) ? 64 : (N <= 64 && M > 64) ? 512 : M <= 128 ? 128 : 256;
    const int bn = (N <= 64 && M > 64) ? 64 : (M <= 64 && N > 64) ? 512 : N <= 128 ? 128 : 256;
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
