// csrc/gemm_f32.hip -- batched fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// This is the dense anchor x channel contraction of the SO(3) convolution,
//   y[b,o,(p,a)] = sum_{(c,k)} W[o,(c,k)] * x[b,(c,k),(p,a)]     BasicSO3Conv.forward,
//                                                                vgtk/vgtk/so3conv/modules.py:L48-55
// (a torch.matmul in the reference) plus its two gradients (dX = W^T dY, dW = sum_b dY X^T).
// fp32-input MFMA is bit-for-bit an fmaf chain (no reduced-precision path), so results carry
// plain fp32 rounding -- required by the 1e-4 pose tolerance after three layers.
//
// Structure: NWM x NWN waves per block (default 4x2 = 512 threads, block tile 256 x 128 x 16), wave
// tile 64 x 64 made of 32x32 MFMA tiles, LDS double-buffered with register prefetch of the next k-tile (one
// barrier per k-tile).  LDS tiles are stored k-major ([BK][BM+4] / [BK][BN+4]) so a wave's
// fragment read (lane -> row/col l&31, k = l>>5) is 32 consecutive floats per half-wave:
// conflict-free ds_read_b32.  The block -> tile map is XCD-aware: blocks that share a column
// panel of the streamed operand are adjacent in dispatch order on the SAME XCD (blockIdx % 8),
// so the panel is fetched from HBM once and re-read from that XCD's L2.
#include "common.h"

#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16, PAD = 4;

template <int BM, int BN>
struct Smem {
    float a[2][BK][BM + PAD];
    float b[2][BK][BN + PAD];
};

struct GemmArgs {
    int M, N, K;
    const float *A; long long lda, sA;
    const float *B; long long ldb, sB;
    float *C; long long ldc, sC;
    int splits;      // k-splits per batch item (1 for the plain GEMM)
    int kchunk;      // K elements per split (multiple of BK)
    int tiles_m, tiles_n;
    // implicit intra conv (GATHER kernels): B row k = (channel c = k / gnt, tap t = k % gnt), column
    // n = (point p = n / gna, anchor a = n % gna)  ->  B[c*ldb + p*gna + gidx[a*gnt + t]]
    const int32_t *gidx; int gna, gnt;
    // B operand stored blocked by 4 along its contiguous logical dimension: element (row r of the `bblk`
    // (channel, kernel point) rows, position x along P*A) at  (x >> 2) * bblk * 4 + r * 4 + (x & 3)
    // -- the layout the grouping kernels can write with coalesced stores (csrc/so3_inter_lists.hip)
    long long bblk;
};

// Load 4 consecutive elements along the contiguous dimension `x` of a row-major [rows][cols]
// view; zero outside [0,rows) x [0,cols).
template <bool VEC>
__device__ __forceinline__ float4 load4(const float *__restrict__ base, long long ld, int r, int x,
                                        int rows, int cols) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) {
        const float *ptr = base + (long long)r * ld + x;
        if (VEC && x + 3 < cols) {
            v = *reinterpret_cast<const float4 *>(ptr);
        } else {
            if (x < cols) v.x = ptr[0];
            if (x + 1 < cols) v.y = ptr[1];
            if (x + 2 < cols) v.z = ptr[2];
            if (x + 3 < cols) v.w = ptr[3];
        }
    }
    return v;
}

// NWM x NWN waves per block, every wave owns a WTM x 64 tile (WTM = 64, or 32 for the small-M variant)
template <int NWM, int NWN, int WTM, bool TA, bool TB, bool VEC, bool GATHER = false>
__global__ __launch_bounds__(64 * NWM * NWN, (NWM * NWN >= 16 ? 4 : 2)) void gemm_f32_kernel(GemmArgs g) {
    constexpr int BM = WTM * NWM, BN = 64 * NWN, NT = 64 * NWM * NWN;
    __shared__ Smem<BM, BN> sm;
    __shared__ int s_gi[GATHER ? 64 * 16 : 1];              // the [anchor][tap] index table of the implicit intra conv
    constexpr int MT = WTM / 32;          // 32x32 MFMA tiles per wave along M
    constexpr int A_V4 = BM * BK / 4 / NT;  // float4 loads per thread for the A tile
    constexpr int B_V4 = BN * BK / 4 / NT;
    static_assert(A_V4 >= 1 && B_V4 >= 1, "tile too small for the thread count");

    // ---- XCD-aware tile mapping -------------------------------------------------------------
    // linear id -> (xcd = id % 8, slot = id / 8); the tiles_m row tiles of one column panel sit
    // in consecutive slots of one XCD.
    const int tiles = g.tiles_m * g.tiles_n;
    int id = blockIdx.x, tm, tn;
    {
        const int groups = g.tiles_n / 8 * 8;            // column panels handled in the XCD scheme
        const int xcd = id & 7, slot = id >> 3;
        const int panel = (slot / g.tiles_m) * 8 + xcd;
        if (panel < groups && id < groups * g.tiles_m) { tn = panel; tm = slot % g.tiles_m; }
        else { const int r = id - groups * g.tiles_m; tn = groups + r / g.tiles_m; tm = r % g.tiles_m; }
        (void)tiles;
    }
    const int z = blockIdx.y;
    const int bz = z / g.splits, sp = z - bz * g.splits;
    const int kbeg = sp * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int m0 = tm * BM, n0 = tn * BN;

    const float *A = g.A + (long long)bz * g.sA;
    const float *B = g.B + (long long)bz * g.sB;
    float *C = g.C + (long long)z * g.sC;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / NWN, wn = wave % NWN;

    float4 ra[A_V4], rb[B_V4];

    // implicit intra conv: this thread's 4 B columns are 4 consecutive anchors of one point
    int g_pcol = 0, g_a0 = 0;
    if (GATHER) {
        for (int i = t; i < g.gna * g.gnt; i += NT) s_gi[i] = g.gidx[i];
        const int n = n0 + (t % (BN / 4)) * 4;
        g_pcol = n / g.gna * g.gna;
        g_a0 = n - g_pcol;
        __syncthreads();
    }

    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < A_V4; ++u) {
            if (!TA) {   // A [M,K]: k contiguous.  thread -> (row, 4 k's)
                const int r = (t >> 2) + u * (NT / 4), kq = (t & 3) * 4;
                ra[u] = load4<VEC>(A, g.lda, m0 + r, k0 + kq, g.M, kend);
            } else {     // A stored [K,M]: m contiguous.  thread -> (k, 4 rows)
                const int k = (t / (BM / 4)) + u * (NT / (BM / 4)), iq = (t % (BM / 4)) * 4;
                ra[u] = load4<VEC>(A, g.lda, k0 + k, m0 + iq, kend, g.M);
            }
        }
#pragma unroll
        for (int u = 0; u < B_V4; ++u) {
            if (GATHER) {   // rows (channel, tap) of the never-materialised [C*T, P*A] operand
                const int k = k0 + (t / (BN / 4)) + u * (NT / (BN / 4)), n = g_pcol + g_a0;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < kend && n < g.N) {                  // N = P*A is a multiple of 4: all four columns valid
                    const int ci = k / g.gnt, tt = k - ci * g.gnt;
                    const float *row = B + (long long)ci * g.ldb + g_pcol;
                    v.x = row[s_gi[(g_a0 + 0) * g.gnt + tt]]; v.y = row[s_gi[(g_a0 + 1) * g.gnt + tt]];
                    v.z = row[s_gi[(g_a0 + 2) * g.gnt + tt]]; v.w = row[s_gi[(g_a0 + 3) * g.gnt + tt]];
                }
                rb[u] = v;
            } else if (!TB) {   // B [K,N]: n contiguous
                const int k = (t / (BN / 4)) + u * (NT / (BN / 4)), jq = (t % (BN / 4)) * 4;
                if (g.bblk) {   // rows = K (channel, kernel point), blocked along N
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k0 + k < kend && n0 + jq + 3 < g.N)
                        v = *reinterpret_cast<const float4 *>(B + (long long)((n0 + jq) >> 2) * g.bblk * 4 + (long long)(k0 + k) * 4);
                    rb[u] = v;
                } else
                    rb[u] = load4<VEC>(B, g.ldb, k0 + k, n0 + jq, kend, g.N);
            } else {     // B stored [N,K]: k contiguous
                const int r = (t >> 2) + u * (NT / 4), kq = (t & 3) * 4;
                if (g.bblk) {   // rows = N (channel, kernel point), blocked along K
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (n0 + r < g.N && k0 + kq + 3 < kend)
                        v = *reinterpret_cast<const float4 *>(B + (long long)((k0 + kq) >> 2) * g.bblk * 4 + (long long)(n0 + r) * 4);
                    rb[u] = v;
                } else
                    rb[u] = load4<VEC>(B, g.ldb, n0 + r, k0 + kq, g.N, kend);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < A_V4; ++u) {
            if (!TA) {
                const int r = (t >> 2) + u * (NT / 4), kq = (t & 3) * 4;
                sm.a[buf][kq + 0][r] = ra[u].x; sm.a[buf][kq + 1][r] = ra[u].y;
                sm.a[buf][kq + 2][r] = ra[u].z; sm.a[buf][kq + 3][r] = ra[u].w;
            } else {
                const int k = (t / (BM / 4)) + u * (NT / (BM / 4)), iq = (t % (BM / 4)) * 4;
                *reinterpret_cast<float4 *>(&sm.a[buf][k][iq]) = ra[u];
            }
        }
#pragma unroll
        for (int u = 0; u < B_V4; ++u) {
            if (!TB) {
                const int k = (t / (BN / 4)) + u * (NT / (BN / 4)), jq = (t % (BN / 4)) * 4;
                *reinterpret_cast<float4 *>(&sm.b[buf][k][jq]) = rb[u];
            } else {
                const int r = (t >> 2) + u * (NT / 4), kq = (t & 3) * 4;
                sm.b[buf][kq + 0][r] = rb[u].x; sm.b[buf][kq + 1][r] = rb[u].y;
                sm.b[buf][kq + 2][r] = rb[u].z; sm.b[buf][kq + 3][r] = rb[u].w;
            }
        }
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ntile = (kend - kbeg + BK - 1) / BK;
    if (ntile > 0) {
        fetch(kbeg);
        stash(0);
    }
    __syncthreads();
    const int li = lane & 31, lk = lane >> 5;
    for (int it = 0; it < ntile; ++it) {
        const int buf = it & 1;
        if (it + 1 < ntile) fetch(kbeg + (it + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            // the MT row tiles (2 column tiles) of a wave are interleaved: tile i owns rows MT*li + i,
            // so all operands of a k-step are one aligned 8-byte (or 4-byte) LDS read per side
            float af[MT], bf[2];
            if (MT == 2) {
                const float2 a2 = *reinterpret_cast<const float2 *>(&sm.a[buf][kk + lk][wm * WTM + 2 * li]);
                af[0] = a2.x; af[MT - 1] = a2.y;
            } else {
                af[0] = sm.a[buf][kk + lk][wm * WTM + li];
            }
            const float2 b2 = *reinterpret_cast<const float2 *>(&sm.b[buf][kk + lk][wn * 64 + 2 * li]);
            bf[0] = b2.x; bf[1] = b2.y;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < ntile) stash(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5);
    // with the interleaving above, tile (i, j) element (row, col) is C[MT*row + i][2*col + j]: the two
    // column tiles of a lane are adjacent -> one 8-byte store
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int col = n0 + wn * 64 + 2 * li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * WTM + MT * ((r & 3) + 8 * (r >> 2) + 4 * lk) + i;
            if (row < g.M) {
                float *dst = C + (long long)row * g.ldc + col;
                if (col + 1 < g.N && (g.ldc & 1) == 0 && ((reinterpret_cast<uintptr_t>(C) & 7) == 0))
                    *reinterpret_cast<float2 *>(dst) = make_float2(acc[i][0][r], acc[i][1][r]);
                else {
                    if (col < g.N) dst[0] = acc[i][0][r];
                    if (col + 1 < g.N) dst[1] = acc[i][1][r];
                }
            }
        }
    }
}

// sum `slabs` partial [M,N] slabs (contiguous, pitch M*N) into C (leading dimension ldc)
__global__ void reduce_slabs_kernel(long long mn, int N, int slabs, const float *__restrict__ ws,
                                    float *__restrict__ C, long long ldc) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= mn) return;
    float s = 0.f;
    for (int z = 0; z < slabs; ++z) s += ws[(long long)z * mn + e];
    C[(e / N) * ldc + (e % N)] = s;
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int NWM, int NWN, int WTM>
int launch(bool ta, bool tb, const GemmArgs &g, int zcount, hipStream_t s) {
    const bool vec = aligned16(g.A) && aligned16(g.B) && g.lda % 4 == 0 && g.ldb % 4 == 0 &&
                     g.sA % 4 == 0 && g.sB % 4 == 0;
    dim3 grid(g.tiles_m * g.tiles_n, zcount), block(64 * NWM * NWN);
#define EAP_GEMM_LAUNCH(TA, TB)                                                                              \
    do {                                                                                                     \
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<NWM, NWN, WTM, TA, TB, true>), grid, block, 0, s, g);   \
        else hipLaunchKernelGGL((gemm_f32_kernel<NWM, NWN, WTM, TA, TB, false>), grid, block, 0, s, g);      \
    } while (0)
    if (g.gidx) {   // implicit intra conv: A = W [M,K] row-major, B gathered
        if (vec) hipLaunchKernelGGL((gemm_f32_kernel<NWM, NWN, WTM, false, false, true, true>), grid, block, 0, s, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<NWM, NWN, WTM, false, false, false, true>), grid, block, 0, s, g);
    } else if (!ta && !tb) EAP_GEMM_LAUNCH(false, false);
    else if (ta && !tb) EAP_GEMM_LAUNCH(true, false);
    else if (!ta && tb) EAP_GEMM_LAUNCH(false, true);
    else EAP_GEMM_LAUNCH(true, true);
#undef EAP_GEMM_LAUNCH
    return eap::check_launch("gemm_f32");
}

int tile_config() {   // 0 (128x128) | 1 (256x128, default: +2% on the L2-layer shape) | 2 (128x256) | 3 (256x256)
#ifdef EAP_ABLATION      // only in a library built with `make ABLATION=1`; a production build never reads the variable
    static int cfg = -1;
    if (cfg < 0) { const char *e = getenv("EAP_GEMM_TILE"); cfg = e ? atoi(e) : 1; }
    return cfg;
#else
    return 1;
#endif
}

int run(bool ta, bool tb, GemmArgs g, int zcount, hipStream_t s) {
    if (zcount > 65535) return eap::bad_arg("gemm_f32: batch*splits exceeds 65535");
    int bm, bn;
    const int cfg = tile_config();
    if (g.M <= 64) { bm = 64; bn = 128; }
    else if (cfg == 1 && g.M >= 256) { bm = 256; bn = 128; }
    else if (cfg == 2 && g.N >= 256) { bm = 128; bn = 256; }
    else if (cfg == 3 && g.M >= 256 && g.N >= 256) { bm = 256; bn = 256; }
    else { bm = 128; bn = 128; }
    g.tiles_m = (g.M + bm - 1) / bm;
    g.tiles_n = (g.N + bn - 1) / bn;
    if (bm == 64) return launch<2, 2, 32>(ta, tb, g, zcount, s);
    if (bm == 256 && bn == 256) return launch<4, 4, 64>(ta, tb, g, zcount, s);
    if (bm == 256) return launch<4, 2, 64>(ta, tb, g, zcount, s);
    if (bn == 256) return launch<2, 4, 64>(ta, tb, g, zcount, s);
    return launch<2, 2, 64>(ta, tb, g, zcount, s);
}

int pick_splits(int M, int N, int K, int batch) {
    // enough blocks to fill 256 CUs twice over, at least 4 k-tiles per split
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128) * batch;
    int splits = (2048 + tiles - 1) / tiles;
    const int max_splits = (K + 4 * BK - 1) / (4 * BK);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    if (splits > 64) splits = 64;
    return splits;
}

}  // namespace

extern "C" int eap_gemm_f32(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                            int64_t strideA, const float *B, int64_t ldb, int64_t strideB, float *C,
                            int64_t ldc, int64_t strideC, int batch, eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    GemmArgs g{M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, 1, (K + BK - 1) / BK * BK, 0, 0, nullptr, 0, 0, 0};
    if (g.kchunk == 0) g.kchunk = BK;
    return run(transA != 0, transB != 0, g, batch, eap::S(stream));
}

// Intra SO(3) conv as an implicit GEMM: out[b,o,p,a] = sum_{c,t} W[o, c*T + t] * feats[b, c, p, intra_idx[a,t]]
// (so3conv/functional.py:L2553-2602 + modules.py:L48-55) without the [B,C,T,P,A] gathered tensor.
extern "C" int eap_so3_intra_conv_f32(int b, int o, int c, int p, int na, int nt, const float *W, const float *feats,
                                      const int32_t *intra_idx, float *out, eap_stream_t stream) {
    if (b <= 0 || o <= 0 || p <= 0 || na <= 0) return 0;
    if (na > 64 || nt > 16 || (na & 3) != 0) return eap::bad_arg("so3_intra_conv: at most 64 anchors (a multiple of 4) and 16 taps");
    const int K = c * nt, N = p * na;
    GemmArgs g{o, N, K, W, K, 0, feats, (long long)N, (long long)c * N, out, (long long)N, (long long)o * N, 1,
               (K + BK - 1) / BK * BK, 0, 0, intra_idx, na, nt, 0};
    if (g.kchunk == 0) g.kchunk = BK;
    return run(false, false, g, b, eap::S(stream));
}

// eap_gemm_f32 with B stored blocked by 4 (see GemmArgs::bblk); b_block_rows = number of B's (channel,
// kernel point) rows = K without transB, N with it; the blocked dimension must be a multiple of 4
extern "C" int eap_gemm_f32_xb(int transA, int transB, int M, int N, int K, const float *A, int64_t lda,
                               int64_t strideA, const float *B, int64_t b_block_rows, int64_t strideB, float *C,
                               int64_t ldc, int64_t strideC, int batch, eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (((transB ? K : N) & 3) != 0 || b_block_rows != (transB ? N : K))
        return eap::bad_arg("gemm_f32_xb: blocked dimension must be a multiple of 4 and b_block_rows the other one");
    GemmArgs g{M, N, K, A, lda, strideA, B, 4, strideB, C, ldc, strideC, 1, (K + BK - 1) / BK * BK, 0, 0, nullptr, 0, 0, b_block_rows};
    if (g.kchunk == 0) g.kchunk = BK;
    return run(transA != 0, transB != 0, g, batch, eap::S(stream));
}

extern "C" int eap_gemm_f32_reduce_xb(int transA, int transB, int M, int N, int K, const float *A,
                                      int64_t lda, int64_t strideA, const float *B, int64_t b_block_rows,
                                      int64_t strideB, float *C, int64_t ldc, int batch, float *workspace,
                                      eap_stream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    if (batch <= 0 || K <= 0) return eap_gemm_f32_reduce(transA, transB, M, N, K, A, lda, strideA, B, 4, strideB, C, ldc, batch, workspace, stream);
    if (((transB ? K : N) & 3) != 0 || b_block_rows != (transB ? N : K))
        return eap::bad_arg("gemm_f32_reduce_xb: blocked dimension must be a multiple of 4 and b_block_rows the other one");
    hipStream_t s = eap::S(stream);
    const int splits = pick_splits(M, N, K, batch);
    int kchunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    GemmArgs g{M, N, K, A, lda, strideA, B, 4, strideB, workspace, N, (long long)M * N, splits, kchunk, 0, 0, nullptr, 0, 0, b_block_rows};
    int e = run(transA != 0, transB != 0, g, batch * splits, s);
    if (e) return e;
    const long long mn = (long long)M * N;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(eap::cdiv(mn, 256)), dim3(256), 0, s, mn, N,
                       batch * splits, workspace, C, (long long)ldc);
    return eap::check_launch("gemm_f32_reduce");
}

extern "C" int64_t eap_gemm_f32_reduce_workspace(int M, int N, int K, int batch) {
    return (int64_t)M * N * batch * pick_splits(M, N, K, batch);
}

extern "C" int eap_gemm_f32_reduce(int transA, int transB, int M, int N, int K, const float *A,
                                   int64_t lda, int64_t strideA, const float *B, int64_t ldb,
                                   int64_t strideB, float *C, int64_t ldc, int batch, float *workspace,
                                   eap_stream_t stream) {
    if (M <= 0 || N <= 0) return 0;
    hipStream_t s = eap::S(stream);
    if (batch <= 0 || K <= 0) {
        for (int r = 0; r < M; ++r) {
            int e = eap::hip_fail(hipMemsetAsync(C + (long long)r * ldc, 0, sizeof(float) * N, s), "gemm reduce memset");
            if (e) return e;
        }
        return 0;
    }
    const int splits = pick_splits(M, N, K, batch);
    int kchunk = ((K + splits - 1) / splits + BK - 1) / BK * BK;
    GemmArgs g{M, N, K, A, lda, strideA, B, ldb, strideB, workspace, N, (long long)M * N, splits, kchunk, 0, 0, nullptr, 0, 0, 0};
    int e = run(transA != 0, transB != 0, g, batch * splits, s);
    if (e) return e;
    const long long mn = (long long)M * N;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(eap::cdiv(mn, 256)), dim3(256), 0, s, mn, N,
                       batch * splits, workspace, C, (long long)ldc);
    return eap::check_launch("gemm_f32_reduce");
}
