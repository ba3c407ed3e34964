// csrc/gemm_bf16x3.hip -- the forward contraction of the SO(3) convolution with fp32 operands, fp32 accumulation and
// fp32-accurate products on the bf16 matrix cores ("3 x bf16 split").
//
//     C_z[M,N] = A[M,K] * B_z[N,K]^T          both operands k-contiguous (row-major), z = batch item
//
//   BasicSO3Conv.forward (vgtk/vgtk/so3conv/modules.py:L48-55):  y[b][O, P*A] = W[O, C*K] * X^T[b][P*A, C*K]^T with the
//   grouped tensor X kept transposed by the grouping kernel (csrc/so3_inter_lists2.hip, layout 2).
//
// Why: v_mfma_f32_32x32x2_f32 peaks at 157 TFLOP/s and csrc/gemm_dma_f32.hip is at 0.89 of it; the bf16 matrix pipe is
// 16 x faster.  Every fp32 operand value x is split EXACTLY into three bf16 values, x = h + m + l (h = bf16(x),
// m = bf16(x - h), l = bf16(x - h - m): 3 x 8 significand bits), and a product a * b is taken as
//     ah bh + (ah bm + am bh) + (ah bl + al bh + am bm)
// -- six v_mfma_f32_32x32x16_bf16 per tile and k-block, each product of two bf16 values exact in the fp32 accumulator.
// The dropped terms (am bl + al bm + al bl) are <= 2^-23 |a b|: the same order as the rounding of ONE fp32 product
// (2^-24), i.e. the result is as accurate as the fmaf chain of the fp32 MFMA (tests: the same 1e-5 bars, and a
// comparison against fp64 in tests/test_gpu_lists_and_modules.py).  Measured ceiling of the instruction stream from one
// wave per SIMD with 16 accumulators: 1.9 PFLOP/s bf16 = 315 TFLOP/s of fp32-equivalent products, with up to three
// vector instructions per MFMA riding along for free (tools/microbench/mfma_bf16_split.hip) -- the split is done ON THE
// FLY on the way global -> VGPR -> LDS, so neither operand is ever stored in split form.
//
// Geometry: 8 waves (two per SIMD: the second wave issues matrix instructions while the first waits for its fragments or
// its staged loads), wave tile 128 x 64 (8 accumulator tiles = 128 registers), block tile 256 x 256 (128 x 256 for row
// counts that are not a multiple of 256), BK = 16.  Per k-tile every thread stages up to four 16-byte pieces (4 k's of one
// operand row) global -> VGPR four k-tiles ahead, splits them BETWEEN the tile's products and parks the three planes in the
// LDS stage of the tile after next ([plane][row][16 k] bf16, 32 bytes per row; the row's two 16-byte k-halves swapped by
// ((row >> 2) ^ (row >> 3)) & 1: conflict-free for the fragment ds_read_b128).  One __syncthreads per k-tile, three LDS
// stages (144 KB): the barrier that closes tile t certifies the stage of tile t + 2, so the first fragments of tile t + 1
// are read before tile t's last product.
//
// B operand layouts (template BMODE; A is always k-contiguous and shared by the batch):
//   0  B_z[N, K] k-contiguous rows: one 16-byte load per piece (the transposed intermediate of the inter conv)
//   1  B_z[K, N] row-major ("NN"): the pointwise contractions `so3_contract` (every 1 x 1 conv of the blocks and heads over
//      [b, C, P*A]); a piece = four dword loads from four k-rows, lanes along n (256 contiguous bytes per request)
//   2  implicit intra-SO(3) gather: B[(c, t), (p, a)] = F_z[c, p, idx[a, t]] (so3conv/functional.py:L2553-2602), the 60 x 12
//      table in LDS; a piece = the four taps t0..t0+3 of one channel, four dword loads inside the point's 240-byte row
#include "common.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 16, BN = 256;
constexpr unsigned PLANE_BYTES = 256 * BK * 2;                 // one plane of one operand tile: 8 KB
// PL = planes per operand: 3 (x = h + m + l in bf16, six products) or 2 (x s = h + l in fp16, three products; below)
constexpr unsigned oper_bytes(int pl) { return (unsigned)pl * PLANE_BYTES; }      // 24 KB / 16 KB
constexpr unsigned stage_bytes(int pl) { return 2 * oper_bytes(pl); }             // 48 KB / 32 KB
constexpr size_t shmem_bytes(int pl) { return 3 * (size_t)stage_bytes(pl); }      // 144 KB / 96 KB (+ TBL_BYTES for the gather table)
constexpr int WAVES_N = 4;                                     // 8 waves per workgroup (see the kernel's template comment)

struct Args {
    int M, N, K;
    const float *A; long long lda;
    const float *B; long long ldb, sB;
    float *C; long long ldc, sC;
    int tiles_m, tiles_n;
    const int *tbl; int na;             // BMODE 2: gather table [na][TAPS], anchors per point
    const unsigned *Apre;               // APRE: A split once by presplit_kernel, [M][K/4] pieces of 32 bytes (h0 h1 m0 m1 | l0 l1 - -)
    long long sA; int slabs, kslab;     // batch-reduce form (BMODE 0): z = item * slabs + slab; both operands start at k = slab * kslab
    // row epilogue (the block layer's inference-mode BatchNorm + leaky_relu (+ skip sum) folded into the contraction,
    // SPConvNets/utils/base_so3poseconv.py:L214-221, L319-328): C = lrelu(scale[row] * acc + shift[row]) (+ res); scale == nullptr: none
    const float *ep_scale, *ep_shift, *ep_res; float ep_slope; long long sRes;
    // PL = 2: the magnitudes the power-of-two operand scales are derived from, as bit patterns of non-negative floats (device
    // words): absA[row of A]; absB[z][column of C / grpB] x multB >= the largest magnitude in that column of B_z (one word
    // serves grpB consecutive columns: the 60 anchors of a point)
    const unsigned *absA, *absB; float multB; int grpB; long long sAbsB;
};
constexpr int TAPS = 12;                // intra_idx is [60, 12] (vgtk/so3conv/functional.py get_intra_idx)
constexpr unsigned TBL_BYTES = 64 * TAPS * 4;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // [15:0] = bf16(a), [31:16] = bf16(b), round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// the eight 16-byte pieces a thread stages per k-tile: piece u = 4 consecutive k's of operand row 64 (u & 3) + (t >> 2)
// (u < 4: A tile, u >= 4: B tile), k-chunk t & 3 -- four lanes cover the 64 contiguous bytes a row contributes to a k-tile
struct Row16 { f32x4 q0, q1; u32x2 p0, p1; };      // p0, p1: the l words of a pre-split A piece whose h, m words sit in q0 / q1 (APRE)

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    h = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);       // exact
    m = pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);       // exact
    l = pk_bf16(s0, s1);
}

// PL = 2: x (already multiplied by its tensor's power-of-two scale) = h + l + e with h = fp16(x), l = fp16(x - h), both
// rounded to nearest even: |x - h| <= 2^-11 |x| is exact in fp32, so |e| <= 2^-12 |x - h| <= 2^-23 |x| (rms 2^-25 |x|:
// the remainder is usually well below its bound) as long as l stays a normal fp16 number, i.e. |x| >= 2^-2; below that
// l is a subnormal and |e| <= 2^-25 absolute.  The scales put each tensor's largest magnitude in [2^14, 2^15) (or a
// bound on it at 2^15), so elements down to 2^-17 of the largest keep the relative bound and nothing overflows.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair_h(float x0, float x1, unsigned &h, unsigned &l) {
    const f16x2 hh = __builtin_convertvector((f32x2){x0, x1}, f16x2);               // v_cvt_pk_f16_f32
    const f16x2 ll = __builtin_convertvector((f32x2){x0 - (float)hh.x, x1 - (float)hh.y}, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}
// the scale 2^(14 - e) of a tensor whose largest magnitude (or a bound on it) is v in [2^e, 2^(e+1)); 1 for 0, inf, nan
__device__ __forceinline__ float pow2_scale(float v) {
    const unsigned b = __float_as_uint(v) & 0x7fffffffu;
    const int e = (int)(b >> 23) - 127;
    if (b == 0u || e == 128) return 1.0f;
    const int se = max(-120, min(120, 14 - max(e, -126)));
    return __uint_as_float((unsigned)(se + 127) << 23);
}

__device__ __forceinline__ void split_quad(const f32x4 &a, const f32x4 &b, u32x4 &h, u32x4 &m, u32x4 &l) {   // 8 k's
    unsigned h0, h1, h2, h3, m0, m1, m2, m3, l0, l1, l2, l3;
    split_pair(a.x, a.y, h0, m0, l0);
    split_pair(a.z, a.w, h1, m1, l1);
    split_pair(b.x, b.y, h2, m2, l2);
    split_pair(b.z, b.w, h3, m3, l3);
    h = (u32x4){h0, h1, h2, h3};
    m = (u32x4){m0, m1, m2, m3};
    l = (u32x4){l0, l1, l2, l3};
}

// DBG (timing ablations, `make ABLATION=1` + EAP_GEMM_SPLIT_DEBUG, WRONG results): 1 = no global loads inside the k-loop,
// 2 = no split / park, 4 = no fragment reads, 8 = no barrier
// WN = waves along N: 2 -> 4 waves (one per SIMD), wave tile 128 x 128, 256 accumulators; 4 -> 8 waves (two per SIMD),
// wave tile 128 x 64, 128 accumulators -- the second wave of a SIMD issues matrix instructions while the first waits for
// its LDS fragments or its staged loads
// MI = row tiles per wave: 4 -> block tile 256 x 256; 2 -> 128 x 256 (wave tile 64 x 64) for row counts that would leave
// half of a 256-row tile empty (the second layer's contraction: 128 output channels)
// PL = planes per operand (3: bf16, 2: fp16 after the tensor scales -- see split_pair_h)
template <int MI, int WN, int DBG, int BMODE, bool APRE, int PL>
__device__ __forceinline__ void split_gemm_body(const Args &g) {
    constexpr unsigned OPER_BYTES = oper_bytes(PL), STAGE_BYTES = stage_bytes(PL);
    constexpr int NT = 128 * WN, NI = 8 / WN, WM = 8 / WN, BM = 32 * MI * WM, RG = NT / 4;   // threads, column tiles per wave, waves along M, rows per block, rows per staging group
    constexpr int NPA = BM / RG, NPO = 256 / RG;                                              // staged pieces per thread: A tile, B tile
    static_assert(WN == 4 && NPO == 2 && (NPA == 1 || NPA == 2), "8 waves, 128-row staging groups");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // XCD-aware tile map (as csrc/gemm_dma_f32.hip): the row tiles of one column panel run back to back on one XCD
    int id = blockIdx.x, tm, tn;
    {
        const int groups = g.tiles_n / 8 * 8;
        const int xcd = id & 7, slot = id >> 3;
        const int panel = (slot / g.tiles_m) * 8 + xcd;
        if (panel < groups && id < groups * g.tiles_m) { tn = panel; tm = slot % g.tiles_m; }
        else { const int r = id - groups * g.tiles_m; tn = groups + r / g.tiles_m; tm = r % g.tiles_m; }
    }
    const int z = blockIdx.y;
    const int m0 = tm * BM, n0 = tn * BN;
    const int item = g.slabs > 1 ? z / g.slabs : z;
    const long long item_of_z = item;
    const long long kbeg = g.slabs > 1 ? (long long)(z - item * g.slabs) * g.kslab : 0;
    const float *A = g.A + item * g.sA + kbeg;
    const float *B = g.B + item * g.sB + (BMODE == 0 ? kbeg : 0);
    float *C = g.C + (long long)z * g.sC;

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    // ---- staging: per k-tile thread t moves eight 16-byte pieces global -> VGPR -> (split) -> LDS: pieces 0..3 of the A
    //      tile (rows 64 u + (t >> 2), u = 0..3), 4..7 of the B tile, k-chunk t & 3.  Request address = wave-uniform base of
    //      the row group (RG = 64 or 128 rows; clamped to the last group that exists: M, N multiples of RG) + one 32-bit offset per lane.
    const int c4 = t & 3, rq = t >> 2;
    // BMODE 1, 2: lanes along the columns (a request = 64 consecutive columns of one k-row), the k-chunk wave-uniform
    const int c4b = BMODE ? __builtin_amdgcn_readfirstlane(t >> 7) : c4, rqb = BMODE ? (t & 127) : rq;
    const float *baseA[NPA], *baseB[NPO];
#pragma unroll
    for (int u = 0; u < NPA; ++u) baseA[u] = A + (long long)min(m0 + RG * u, g.M - RG) * g.lda;
#pragma unroll
    for (int u = 0; u < NPO; ++u) baseB[u] = B + (long long)min(n0 + RG * u, g.N - RG) * g.ldb;
    // APRE: the weights were split once (presplit_kernel): piece (row, k / 4) = 32 bytes (h0 h1 m0 m1 | l0 l1 - -)
    const unsigned char *preA[NPA];
    constexpr int PRE_PIECE = PL == 3 ? 32 : 16;     // PL = 2: piece (row, k / 4) = 16 bytes (h0 h1 l0 l1)
    const long long pre_pitch = (long long)(g.K / 4) * PRE_PIECE;
#pragma unroll
    for (int u = 0; u < NPA; ++u) preA[u] = reinterpret_cast<const unsigned char *>(g.Apre) + (long long)min(m0 + RG * u, g.M - RG) * pre_pitch;
    const unsigned offP = (unsigned)((long long)rq * pre_pitch + PRE_PIECE * c4);
    // PL = 2: the operand scales (powers of two: every product below is exact) and what undoes them in the epilogue
    // PL = 2: the scales of the operand rows this thread stages (powers of two: every product below is exact); a workgroup's
    // rows are the same for the whole k-loop, so this is two or four words per thread and kernel
    float sclA[2] = {1.0f, 1.0f}, sclB[2] = {1.0f, 1.0f};
    const unsigned *absBz = g.absB + item_of_z * g.sAbsB;
    if constexpr (PL == 2) {
#pragma unroll
        for (int u = 0; u < NPA; ++u) sclA[u] = pow2_scale(__uint_as_float(g.absA[min(m0 + RG * u, g.M - RG) + rq]));
#pragma unroll
        for (int u = 0; u < NPO; ++u) sclB[u] = pow2_scale(__uint_as_float(absBz[(min(n0 + RG * u, g.N - RG) + rqb) / g.grpB]) * g.multB);
    }
    const unsigned offA = (unsigned)((long long)rq * g.lda + 4 * c4) * 4u, offB = (unsigned)((long long)rq * g.ldb + 4 * c4) * 4u;
    unsigned colB[NPO], tapB[NPO];      // BMODE 1: byte offset of the piece's column; BMODE 2: of its point's row, and of its anchor's table row
#pragma unroll
    for (int u = 0; u < NPO; ++u) {
        const int n = min(n0 + RG * u, g.N - RG) + rqb;
        if constexpr (BMODE == 2) {
            const int pt = n / g.na;
            colB[u] = (unsigned)(pt * g.na) * 4u;
            tapB[u] = (unsigned)(n - pt * g.na) * (TAPS * 4u);
        } else {
            colB[u] = (unsigned)n * 4u; tapB[u] = 0;
        }
    }
    // LDS: row r of a plane is 32 bytes (16 bf16); its two 16-byte k-halves are swapped by b(r) = ((r >> 2) ^ (r >> 3)) & 1:
    // conflict-free for the fragment ds_read_b128 (four 16-lane groups, 256-byte bank window); the 8-byte parking writes of
    // 16 contiguous lanes cover 4 whole rows = 128 contiguous bytes either way (BMODE 0; lanes along the columns write 8 of
    // every 32 bytes: two-way conflicts on a sixth of the LDS traffic).  rows RG u + rq: bits 2, 3 of the row = bits of rq
    auto wr_of = [](int r, int c) { return (unsigned)r * 32u + ((16u * (unsigned)(c >> 1)) ^ ((((unsigned)r >> 2) ^ ((unsigned)r >> 3)) & 1u) * 16u) + 8u * (unsigned)(c & 1); };
    const unsigned wr_off = wr_of(rq, c4), wr_offB = wr_of(rqb, c4b);
    const unsigned char *tbl = smem + 3 * STAGE_BYTES;
    if constexpr (BMODE == 2) {
        for (int i = t; i < g.na * TAPS; i += NT) reinterpret_cast<int *>(smem + 3 * STAGE_BYTES)[i] = g.tbl[i] * 4;   // byte offsets
        __syncthreads();
    }
    const int nk = (g.slabs > 1 ? g.kslab : g.K) / BK;

    Row16 ra0, rb0, ra1, rb1;                                           // (A pieces 0..3, B pieces 0..3) of two tiles in flight
    auto ld = [&](const float *ubase, unsigned voff) __attribute__((always_inline)) {
        return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(ubase) + voff);
    };
    auto ldw = [&](const float *ubase, unsigned voff) __attribute__((always_inline)) {
        return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(ubase) + voff);
    };
    auto load_tile = [&](int kt, Row16 &a, Row16 &b) __attribute__((always_inline)) {
        const int ko = kt * BK;
        if constexpr (APRE) {
            const unsigned char *p0 = preA[0] + kt * (4 * PRE_PIECE) + offP;
            a.q0 = *reinterpret_cast<const f32x4 *>(p0);
            if constexpr (PL == 3) a.p0 = *reinterpret_cast<const u32x2 *>(p0 + 16);
            if constexpr (NPA == 2) {
                const unsigned char *p1 = preA[1] + kt * (4 * PRE_PIECE) + offP;
                a.q1 = *reinterpret_cast<const f32x4 *>(p1);
                if constexpr (PL == 3) a.p1 = *reinterpret_cast<const u32x2 *>(p1 + 16);
            }
        } else {
            a.q0 = ld(baseA[0] + ko, offA);
            if constexpr (NPA == 2) a.q1 = ld(baseA[1] + ko, offA);
        }
        if constexpr (BMODE == 0) {
            b.q0 = ld(baseB[0] + ko, offB); b.q1 = ld(baseB[1] + ko, offB);
        } else if constexpr (BMODE == 1) {
            const float *r0 = B + (long long)(ko + 4 * c4b) * g.ldb, *r1 = r0 + g.ldb, *r2 = r1 + g.ldb, *r3 = r2 + g.ldb;   // wave-uniform
            b.q0 = (f32x4){ldw(r0, colB[0]), ldw(r1, colB[0]), ldw(r2, colB[0]), ldw(r3, colB[0])};
            b.q1 = (f32x4){ldw(r0, colB[1]), ldw(r1, colB[1]), ldw(r2, colB[1]), ldw(r3, colB[1])};
        } else {
            const int k0 = ko + 4 * c4b, ch = k0 / TAPS, t0 = k0 - ch * TAPS;                  // wave-uniform: channel, first tap
            const float *r = B + (long long)ch * g.ldb;
            const u32x4 o0 = *reinterpret_cast<const u32x4 *>(tbl + tapB[0] + 4 * t0);
            const u32x4 o1 = *reinterpret_cast<const u32x4 *>(tbl + tapB[1] + 4 * t0);
            b.q0 = (f32x4){ldw(r, colB[0] + o0.x), ldw(r, colB[0] + o0.y), ldw(r, colB[0] + o0.z), ldw(r, colB[0] + o0.w)};
            b.q1 = (f32x4){ldw(r, colB[1] + o1.x), ldw(r, colB[1] + o1.y), ldw(r, colB[1] + o1.z), ldw(r, colB[1] + o1.w)};
        }
    };
    // one piece (4 k's) -> three 8-byte words, parked at its place in row 64 u + rq of the operand tile
    auto park_piece = [&](unsigned char *oper, int u, const f32x4 &q, unsigned wr, float scl) __attribute__((always_inline)) {
        unsigned h0, m0_, l0, h1, m1, l1;
        if constexpr (PL == 2) {
            split_pair_h(q.x * scl, q.y * scl, h0, l0);
            split_pair_h(q.z * scl, q.w * scl, h1, l1);
            unsigned char *row = oper + (unsigned)u * ((unsigned)RG * 32u) + wr;
            *reinterpret_cast<u32x2 *>(row) = (u32x2){h0, h1};
            *reinterpret_cast<u32x2 *>(row + PLANE_BYTES) = (u32x2){l0, l1};
            return;
        }
        split_pair(q.x, q.y, h0, m0_, l0);
        split_pair(q.z, q.w, h1, m1, l1);
        unsigned char *row = oper + (unsigned)u * ((unsigned)RG * 32u) + wr;
        *reinterpret_cast<u32x2 *>(row) = (u32x2){h0, h1};
        *reinterpret_cast<u32x2 *>(row + PLANE_BYTES) = (u32x2){m0_, m1};
        *reinterpret_cast<u32x2 *>(row + 2 * PLANE_BYTES) = (u32x2){l0, l1};
    };
    // half of an operand's pieces of a tile (the split rides between the products in four portions)
    auto park_half = [&](unsigned char *oper, const Row16 &r, int half, int pieces, unsigned wr, float scl) __attribute__((always_inline)) {
        if (half < pieces) park_piece(oper, half, half ? r.q1 : r.q0, wr, scl);
    };
    // an A piece that arrived split: three 8-byte words straight to the planes
    auto park_half_a = [&](unsigned char *oper, const Row16 &r, int half, unsigned wr) __attribute__((always_inline)) {
        if constexpr (APRE) {
            if (half < NPA) {
                const f32x4 &q = half ? r.q1 : r.q0;
                const u32x2 &l = half ? r.p1 : r.p0;       // (PL = 3 only)
                unsigned char *row = oper + (unsigned)half * ((unsigned)RG * 32u) + wr;
                *reinterpret_cast<u32x2 *>(row) = (u32x2){__float_as_uint(q.x), __float_as_uint(q.y)};
                *reinterpret_cast<u32x2 *>(row + PLANE_BYTES) = (u32x2){__float_as_uint(q.z), __float_as_uint(q.w)};
                if constexpr (PL == 3) *reinterpret_cast<u32x2 *>(row + 2 * PLANE_BYTES) = l;
            }
        } else {
            park_half(oper, r, half, NPA, wr, sclA[half]);
        }
    };
    auto park = [&](unsigned char *oper, const Row16 &r, int pieces, unsigned wr) __attribute__((always_inline)) {      // (B tiles)
        park_half(oper, r, 0, pieces, wr, sclB[0]);
        park_half(oper, r, 1, pieces, wr, sclB[1]);
    };
    auto park_a = [&](unsigned char *oper, const Row16 &r, unsigned wr) __attribute__((always_inline)) {
        park_half_a(oper, r, 0, wr);
        park_half_a(oper, r, 1, wr);
    };

    // ---- fragments: lane (row li of a 32-row tile, k-half lh) reads 16 bytes = 8 bf16 ------------------------------
    // row r of the wave's A rows = 128 wm + 32 i + li: bits 2 and 3 of r are those of li
    const unsigned rd_half = 16u * ((unsigned)lh ^ ((((unsigned)li >> 2) ^ ((unsigned)li >> 3)) & 1u));
    const unsigned rdA = (unsigned)(32 * MI * wm + li) * 32u + rd_half;
    const unsigned rdB = OPER_BYTES + (unsigned)(32 * NI * wn + li) * 32u + rd_half;
    auto frag = [&](const unsigned char *st, unsigned base, int p, auto &f) __attribute__((always_inline)) {
        constexpr int NF = sizeof(f) / sizeof(u32x4);
        if constexpr (DBG & 4) {
#pragma unroll
            for (int i = 0; i < NF; ++i) f[i] = (u32x4){0x3f803f80u + (unsigned)p, 0x3f803f80u, 0x3f803f80u + base, 0x3f803f80u};
            return;
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) f[i] = *reinterpret_cast<const u32x4 *>(st + base + p * PLANE_BYTES + i * 32 * 32);
    };

    f32x16 acc[MI][NI];       // never zeroed: the first product of the first k-tile takes the constant 0 as its C operand
    u32x4 ah[MI], bl[NI];     // the first two plane fragments of the NEXT tile are read before the tile's closing barrier

    // One k-tile.  Stage `st` holds tile kt (certified by the barrier that closed tile kt - 1, like stage `s1` of tile
    // kt + 1); (sa, sb) hold tile kt + 2 in registers since tile kt - 1: they are split and parked in stage `s2` in four
    // pieces BETWEEN the products (vector instructions ride in the shadow of the matrix pipe,
    // tools/microbench/mfma_bf16_split.hip), then take tile kt + 4.  Plane fragments are read one product ahead of
    // their first use and dropped after their last.
    auto tile = [&](auto first, int kt, const unsigned char *st, const unsigned char *s1, unsigned char *s2, Row16 &sa, Row16 &sb)
                    __attribute__((always_inline)) {
        auto mm = [&](const u32x4 &fa, const u32x4 &fb, const f32x16 &c) __attribute__((always_inline)) {
            if constexpr (PL == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), c, 0, 0, 0);
            else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fb), c, 0, 0, 0);
        };
        auto product = [&](const u32x4 (&fa)[MI], const u32x4 (&fb)[NI]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mm(fa[i], fb[j], acc[i][j]);
        };
        auto product0 = [&](const u32x4 (&fa)[MI], const u32x4 (&fb)[NI]) __attribute__((always_inline)) {      // the first of the k-loop
            const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = mm(fa[i], fb[j], zc);
        };
#define SB() __builtin_amdgcn_sched_barrier(0)
        if constexpr (PL == 2) {
            // three products per tile: ah bl, ah bh, al bh (ah, bl arrive from the previous tile; the last product needs
            // neither, so the next tile's pair is read straight into them).  The split of tile kt + 2 rides in three portions.
            u32x4 bh[NI], al[MI];
            frag(st, rdB, 0, bh);
            SB();
            if constexpr (decltype(first)::value) product0(ah, bl); else product(ah, bl);
            if constexpr (!(DBG & 2)) park_a(s2, sa, wr_off);
            SB();
            frag(st, rdA, 1, al);
            SB();
            product(ah, bh);
            if constexpr (!(DBG & 2)) park_half(s2 + OPER_BYTES, sb, 0, NPO, wr_offB, sclB[0]);
            SB();
            frag(s1, rdA, 0, ah);                // the next tile's pair (its stage was certified a tile ago)
            frag(s1, rdB, 1, bl);
            SB();
            product(al, bh);
            if constexpr (!(DBG & 2)) park_half(s2 + OPER_BYTES, sb, 1, NPO, wr_offB, sclB[1]);
            if constexpr (!(DBG & 1)) load_tile(min(kt + 4, nk - 1), sa, sb);
            SB();
            if constexpr (!(DBG & 8)) __syncthreads();
            return;
        }
        u32x4 bm[NI], am[MI], bh[NI], al[MI];
        // issue order pinned per product: the LDS reads of the NEXT product's new plane first, then this product's MFMAs
        // with a quarter of the split riding between them (the compiler otherwise sinks the reads next to their use and
        // every product starts with an LDS round trip)
        frag(st, rdB, 1, bm);
        SB();
        if constexpr (decltype(first)::value) product0(ah, bl); else product(ah, bl);
        if constexpr (!(DBG & 2)) park_half_a(s2, sa, 0, wr_off);
        SB();
        frag(st, rdA, 1, am);
        SB();
        product(ah, bm);
        if constexpr (!(DBG & 2)) park_half_a(s2, sa, 1, wr_off);
        SB();
        frag(st, rdB, 0, bh);
        SB();
        product(am, bm);
        if constexpr (!(DBG & 2)) park_half(s2 + OPER_BYTES, sb, 0, NPO, wr_offB, sclB[0]);
        SB();
        frag(st, rdA, 2, al);
        SB();
        product(am, bh);
        if constexpr (!(DBG & 2)) park_half(s2 + OPER_BYTES, sb, 1, NPO, wr_offB, sclB[1]);
        SB();
        // branch-free on purpose (accumulators that cross a control-flow join get copied): past the last tiles the
        // staged registers are re-split into a stage nobody reads, the loads repeat the last tile, the fragment reads hit
        // a stale stage
        u32x4 ah2[MI];
        frag(s1, rdA, 0, ah2);               // the next tile's first two planes (its stage was certified a tile ago)
        SB();
        product(al, bh);
        if constexpr (!(DBG & 1)) load_tile(min(kt + 4, nk - 1), sa, sb);
        SB();
        frag(s1, rdB, 2, bl);
        SB();
        product(ah, bh);
        SB();
#undef SB
#pragma unroll
        for (int i = 0; i < MI; ++i) ah[i] = ah2[i];
        if constexpr (!(DBG & 8)) __syncthreads();
    };

    // ---- prologue --------------------------------------------------------------------------------------------------
    load_tile(0, ra0, rb0);
    load_tile(min(1, nk - 1), ra1, rb1);
    park_a(smem, ra0, wr_off);
    park(smem + OPER_BYTES, rb0, NPO, wr_offB);
    load_tile(min(2, nk - 1), ra0, rb0);
    park_a(smem + STAGE_BYTES, ra1, wr_off);
    park(smem + STAGE_BYTES + OPER_BYTES, rb1, NPO, wr_offB);
    load_tile(min(3, nk - 1), ra1, rb1);
    __syncthreads();
    frag(smem, rdA, 0, ah);
    frag(smem, rdB, PL - 1, bl);

    // stage of tile kt = kt % 3; register set of tile kt + 2 = kt % 2
    unsigned s0 = 0, s1 = STAGE_BYTES, s2 = 2 * STAGE_BYTES;
    tile(std::true_type{}, 0, smem + s0, smem + s1, smem + s2, ra0, rb0);
    int kt = 1;
    for (; kt + 1 < nk; kt += 2) {
        { const unsigned r = s0; s0 = s1; s1 = s2; s2 = r; }
        tile(std::false_type{}, kt, smem + s0, smem + s1, smem + s2, ra1, rb1);
        { const unsigned r = s0; s0 = s1; s1 = s2; s2 = r; }
        tile(std::false_type{}, kt + 1, smem + s0, smem + s1, smem + s2, ra0, rb0);
    }
    if (kt < nk) {
        { const unsigned r = s0; s0 = s1; s1 = s2; s2 = r; }
        tile(std::false_type{}, kt, smem + s0, smem + s1, smem + s2, ra1, rb1);
    }

    // ---- epilogue: D[i][j] of a 32 x 32 tile sits at col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) --------
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int col = n0 + 32 * NI * wn + 32 * j + li;
            float unsB = 1.0f;         // PL = 2: the scales come off again, in two factors (their product can leave the normal range)
            if constexpr (PL == 2) unsB = 1.0f / pow2_scale(__uint_as_float(absBz[min(col, g.N - 1) / g.grpB]) * g.multB);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 32 * MI * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < g.M && col < g.N) {
                    float v = acc[i][j][r];
                    if constexpr (PL == 2) v = (v * (1.0f / pow2_scale(__uint_as_float(g.absA[row])))) * unsB;
                    if (g.ep_scale != nullptr) {                      // kernel-uniform
                        v = fmaf(v, g.ep_scale[row], g.ep_shift[row]);
                        v = v >= 0.f ? v : v * g.ep_slope;
                        if (g.ep_res != nullptr) v += g.ep_res[(long long)z * g.sRes + (long long)row * g.ldc + col];
                    }
                    C[(long long)row * g.ldc + col] = v;
                }
            }
        }
}

template <int MI, int WN, int DBG, int BMODE, bool APRE>
__global__ __launch_bounds__(128 * WN, WN / 2) void gemm_bf16x3_kernel(Args g) { split_gemm_body<MI, WN, DBG, BMODE, APRE, 3>(g); }

// two fp16 planes per operand, three products per k-tile (see split_pair_h for what that costs in accuracy)
template <int MI, int WN, int DBG, int BMODE, bool APRE>
__global__ __launch_bounds__(128 * WN, WN / 2) void gemm_f16x2_kernel(Args g) { split_gemm_body<MI, WN, DBG, BMODE, APRE, 2>(g); }

}  // namespace

namespace {

// the shared operand (the weights) split once per call instead of once per k-tile by every workgroup: piece (row, k / 4) ->
// 32 bytes (h0 h1 m0 m1 | l0 l1 - -), the same roundings as the in-kernel split
__global__ __launch_bounds__(256) void presplit_kernel(int M, int K4, const float *__restrict__ A, long long lda, u32x4 *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * K4) return;
    const int row = (int)(i / K4), c = (int)(i - (long long)row * K4);
    const f32x4 q = *reinterpret_cast<const f32x4 *>(A + row * lda + 4 * c);
    unsigned h0, m0, l0, h1, m1, l1;
    split_pair(q.x, q.y, h0, m0, l0);
    split_pair(q.z, q.w, h1, m1, l1);
    out[2 * i] = (u32x4){h0, h1, m0, m1};
    out[2 * i + 1] = (u32x4){l0, l1, 0u, 0u};
}

// PL = 2: piece (row, k / 4) -> 16 bytes (h0 h1 l0 l1) of the SCALED weights, the same roundings as the in-kernel split
__global__ __launch_bounds__(256) void presplit_h_kernel(int M, int K4, const float *__restrict__ A, long long lda, const unsigned *__restrict__ absA,
                                                         u32x4 *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)M * K4) return;
    const int row = (int)(i / K4), c = (int)(i - (long long)row * K4);
    const float scl = pow2_scale(__uint_as_float(absA[row]));
    const f32x4 q = *reinterpret_cast<const f32x4 *>(A + row * lda + 4 * c);
    unsigned h0, l0, h1, l1;
    split_pair_h(q.x * scl, q.y * scl, h0, l0);
    split_pair_h(q.z * scl, q.w * scl, h1, l1);
    out[i] = (u32x4){h0, h1, l0, l1};
}

// ---- operand magnitudes for the two-plane kernel, as bit patterns of non-negative floats (unsigned maximum = float maximum:
// order-independent, so the atomics below make nothing run-dependent; a NaN compares above everything and turns the scale
// into 1, pow2_scale) --------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned absbits4(const f32x4 &q) {
    return max(max(__float_as_uint(q.x) & 0x7fffffffu, __float_as_uint(q.y) & 0x7fffffffu), max(__float_as_uint(q.z) & 0x7fffffffu, __float_as_uint(q.w) & 0x7fffffffu));
}
// one wave per row of a [batch][rows][cols] tensor: out[batch * rows]
__global__ __launch_bounds__(256) void absmax_rows_kernel(long long rows_total, int rows, int cols4, const float *__restrict__ x, long long ld,
                                                          long long stride, unsigned *__restrict__ out) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows_total) return;
    const long long item = r / rows;
    const f32x4 *row = reinterpret_cast<const f32x4 *>(x + item * stride + (r - item * rows) * ld);
    unsigned m = 0u;
    for (int c = threadIdx.x & 63; c < cols4; c += 64) m = max(m, absbits4(row[c]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) out[r] = m;
}
// largest magnitude over the rows and over groups of grp consecutive columns: out[batch][cols / grp] (zeroed by the caller).
// A thread owns one 16-byte column piece and walks a slice of the rows; lanes run along the columns.
__global__ __launch_bounds__(256) void absmax_colgroups_kernel(int rows, int cols4, int grp4, int row_slices, const float *__restrict__ x, long long ld,
                                                               long long stride, unsigned *__restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols4) return;
    const int item = blockIdx.z, sl = blockIdx.y;
    const int r0 = (int)((long long)rows * sl / row_slices), r1 = (int)((long long)rows * (sl + 1) / row_slices);
    const float *p = x + item * stride + 4 * (long long)c;
    unsigned m = 0u;
    for (int r = r0; r < r1; ++r) m = max(m, absbits4(*reinterpret_cast<const f32x4 *>(p + r * ld)));
    if (m != 0u) atomicMax(out + (long long)item * (cols4 / grp4) + c / grp4, m);
}
// the inter conv's grouped tensor X[c,k,p,a] = sum over the nn neighbours of feats[c, idx[p,n], a'] times a weight in [0, 1]
// (so3conv/functional.py:L1112-1261): |X[., ., p, .]| <= sum_n max_{c,a} |feats[c, idx[p,n], a]| -- a bound per point from the
// per-point maxima of the features, without a pass over X.  Summed in neighbour order (deterministic).
__global__ __launch_bounds__(256) void grouped_bound_kernel(long long total, int p, int nn, int n_sup, const unsigned *__restrict__ point_max,
                                                            const int32_t *__restrict__ idx, unsigned *__restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const unsigned *pm = point_max + (i / p) * n_sup;
    const int32_t *id = idx + i * nn;
    float s = 0.f;
    for (int n = 0; n < nn; ++n) {
        const int q = id[n];
        if ((unsigned)q < (unsigned)n_sup) s += __uint_as_float(pm[q]);
    }
    out[i] = __float_as_uint(s);
}

int g_presplit = 1;      // eap_gemm_bf16x3_presplit(0): split the weights in the k-loop like the other operand (A/B runs, tests)

// the stream-ordered pool keeps what it has been given: with the default release threshold (0) the scratch of every call goes
// back to the driver at the next synchronisation point and is mapped again by the next call
void keep_pool_memory() {
    static bool done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || done[dev]) return;
    done[dev] = true;
    hipMemPool_t pool;
    if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess) {
        uint64_t keep = ~0ull;
        (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    }
    (void)hipGetLastError();
}

template <int BMODE, int PL = 3>
int launch_split(Args &g, int batch, hipStream_t stream, const char *who) {
    // pays from a few thousand k-tiles per workgroup column upwards (+2.7 % on the deepest layer's contraction; on the 1-2 ms
    // pointwise contractions the extra launch and the allocation cost more than the saved vector work:
    // tools/split_modes_timing.py); the test switch value 2 forces it for every shape
    bool pre = g.sA == 0 && g.slabs <= 1 && (g_presplit == 2 || (g_presplit == 1 && g.K >= 1024));
    void *scratch = nullptr;
    if (pre) {
        const long long pieces = (long long)g.M * (g.K / 4);
        keep_pool_memory();
        if (hipMallocAsync(&scratch, (size_t)pieces * (PL == 3 ? 32 : 16), stream) != hipSuccess) {      // no stream-ordered pool on this device / out of
            (void)hipGetLastError();                                                      // memory: split in the k-loop instead
            scratch = nullptr;
            pre = false;
        }
    }
    if (pre) {
        const long long pieces = (long long)g.M * (g.K / 4);
        if constexpr (PL == 3)
            hipLaunchKernelGGL(presplit_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, g.M, g.K / 4, g.A, g.lda,
                               reinterpret_cast<u32x4 *>(scratch));
        else
            hipLaunchKernelGGL(presplit_h_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, g.M, g.K / 4, g.A, g.lda, g.absA,
                               reinterpret_cast<u32x4 *>(scratch));
        g.Apre = reinterpret_cast<const unsigned *>(scratch);
    }
    struct Release {           // stream-ordered: the buffer is released after the product that reads it
        void *p; hipStream_t s;
        ~Release() { if (p) (void)hipFreeAsync(p, s); }
    } release{scratch, stream};
    const bool tall = (g.M % 256) == 0;            // 256-row tiles; otherwise 128-row tiles (M is a multiple of 128)
    g.tiles_m = tall ? g.M / 256 : g.M / 128;
    g.tiles_n = (g.N + BN - 1) / BN;
    const size_t shmem = shmem_bytes(PL) + (BMODE == 2 ? TBL_BYTES : 0);
    auto launch = [&](auto kern) {
        int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem), who);
        if (e) return e;
        hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n, batch), dim3(128 * WAVES_N), shmem, stream, g);
        return 0;
    };
    auto run = [&](auto dbg_c) {
        constexpr int D = decltype(dbg_c)::value;
        if constexpr (PL == 2) {
            if (pre) return tall ? launch(gemm_f16x2_kernel<4, WAVES_N, D, BMODE, true>) : launch(gemm_f16x2_kernel<2, WAVES_N, D, BMODE, true>);
            return tall ? launch(gemm_f16x2_kernel<4, WAVES_N, D, BMODE, false>) : launch(gemm_f16x2_kernel<2, WAVES_N, D, BMODE, false>);
        } else {
            if (pre) return tall ? launch(gemm_bf16x3_kernel<4, WAVES_N, D, BMODE, true>) : launch(gemm_bf16x3_kernel<2, WAVES_N, D, BMODE, true>);
            return tall ? launch(gemm_bf16x3_kernel<4, WAVES_N, D, BMODE, false>) : launch(gemm_bf16x3_kernel<2, WAVES_N, D, BMODE, false>);
        }
    };
    int e;
#ifdef EAP_ABLATION
    const int dbg = getenv("EAP_GEMM_SPLIT_DEBUG") ? atoi(getenv("EAP_GEMM_SPLIT_DEBUG")) : 0;
    if constexpr (BMODE == 0) {
        switch (dbg) {
            case 1: e = run(std::integral_constant<int, 1>{}); break;
            case 2: e = run(std::integral_constant<int, 2>{}); break;
            case 3: e = run(std::integral_constant<int, 3>{}); break;
            case 4: e = run(std::integral_constant<int, 4>{}); break;
            case 7: e = run(std::integral_constant<int, 7>{}); break;
            case 15: e = run(std::integral_constant<int, 15>{}); break;
            default: e = run(std::integral_constant<int, 0>{});
        }
    } else {
        e = run(std::integral_constant<int, 0>{});
    }
#else
    e = run(std::integral_constant<int, 0>{});
#endif
    if (e) return e;
    static const char *names[3][2] = {{"gemm_bf16x3_kernel<2, 4>", "gemm_bf16x3_kernel<4, 4>"},
                                      {"gemm_bf16x3_kernel<2, 4, nn>", "gemm_bf16x3_kernel<4, 4, nn>"},
                                      {"gemm_bf16x3_kernel<2, 4, gather>", "gemm_bf16x3_kernel<4, 4, gather>"}};
    static const char *names2[3][2] = {{"gemm_f16x2_kernel<2, 4>", "gemm_f16x2_kernel<4, 4>"},
                                       {"gemm_f16x2_kernel<2, 4, nn>", "gemm_f16x2_kernel<4, 4, nn>"},
                                       {"gemm_f16x2_kernel<2, 4, gather>", "gemm_f16x2_kernel<4, 4, gather>"}};
    eap::set_kernel((PL == 2 ? names2 : names)[BMODE][tall ? 1 : 0]);
    return eap::check_launch(who);
}

inline bool tile_dims_ok(int M, int N, int K) {
    return M >= 128 && N >= 256 && (M % (32 * WAVES_N)) == 0 && (N % (32 * WAVES_N)) == 0 && K >= BK && (K % BK) == 0;
}

}  // namespace

// 1 (default): for K >= 1024 the shared operand is split once per call into a stream-ordered scratch buffer (hipMallocAsync,
// M * K * 8 bytes, released after the product); 2: for every shape; 0: both operands are split inside the k-loop.  Same values.
extern "C" int eap_gemm_bf16x3_presplit(int on) {
    const int was = g_presplit;
    if (on >= 0 && on <= 2) g_presplit = on;
    return was;
}

// can the split kernel take this product?  (both operands k-contiguous, K a multiple of 16, 16-byte aligned rows; it pays
// from a 256 x 256 tile per CU upwards)
extern "C" int eap_gemm_bf16x3_f32_supported(int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb,
                                             int64_t strideB) {
    if (!tile_dims_ok(M, N, K)) return 0;
    if ((long long)32 * WAVES_N * lda * 4 >= (1ll << 32) || (long long)32 * WAVES_N * ldb * 4 >= (1ll << 32)) return 0;
    if ((lda & 3) || (ldb & 3) || (strideB & 3)) return 0;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return 0;
    return 1;
}

extern "C" int eap_gemm_bf16x3_f32(int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t strideB,
                                   float *C, int64_t ldc, int64_t strideC, int batch, eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (!eap_gemm_bf16x3_f32_supported(M, N, K, A, lda, B, ldb, strideB)) return eap::bad_arg("gemm_bf16x3_f32: unsupported operands (ask eap_gemm_bf16x3_f32_supported)");
    if (batch > 65535) return eap::bad_arg("gemm_bf16x3_f32: batch exceeds 65535");
    Args g{};
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda;
    g.B = B; g.ldb = ldb; g.sB = strideB;
    g.C = C; g.ldc = ldc; g.sC = strideC;
    return launch_split<0>(g, batch, eap::S(stream), "gemm_bf16x3_f32");
}

// eap_gemm_bf16x3_f32 / _nn_f32 with a per-row epilogue: C = leaky_relu(scale[row] * (A B) + shift[row], slope) (+ residual, laid
// out like C): an inference-mode BatchNorm2d + activation (+ the separable block's skip sum) without a pass of their own.
// transB = 1: B k-contiguous [N,K] (the inter conv's transposed intermediate); 0: B row-major [K,N] (pointwise contraction).
extern "C" int eap_gemm_bf16x3_ep_f32(int transB, int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t strideB,
                                      float *C, int64_t ldc, int64_t strideC, int batch, const float *scale, const float *shift, float slope,
                                      const float *residual, int64_t strideRes, eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (!scale || !shift) return eap::bad_arg("gemm_bf16x3_ep_f32: scale and shift are required");
    const bool ok = transB ? eap_gemm_bf16x3_f32_supported(M, N, K, A, lda, B, ldb, strideB) : eap_gemm_bf16x3_nn_f32_supported(M, N, K, A, lda, B, ldb, strideB);
    if (!ok) return eap::bad_arg("gemm_bf16x3_ep_f32: unsupported operands (ask eap_gemm_bf16x3_f32_supported / _nn_f32_supported)");
    if (batch > 65535) return eap::bad_arg("gemm_bf16x3_ep_f32: batch exceeds 65535");
    Args g{};
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda;
    g.B = B; g.ldb = ldb; g.sB = strideB;
    g.C = C; g.ldc = ldc; g.sC = strideC;
    g.ep_scale = scale; g.ep_shift = shift; g.ep_slope = slope; g.ep_res = residual; g.sRes = strideRes;
    return transB ? launch_split<0>(g, batch, eap::S(stream), "gemm_bf16x3_ep_f32") : launch_split<1>(g, batch, eap::S(stream), "gemm_bf16x3_ep_f32");
}

// C_z[M,N] = A[M,K] * B_z[K,N]: A k-contiguous and shared by the batch, B row-major (the pointwise contraction
// so3_contract: W [O, C] times x_z [C, P*A])
extern "C" int eap_gemm_bf16x3_nn_f32_supported(int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb,
                                                int64_t strideB) {
    if (!tile_dims_ok(M, N, K)) return 0;
    if ((long long)32 * WAVES_N * lda * 4 >= (1ll << 32) || (long long)ldb * 4 >= (1ll << 32) || ldb < N) return 0;
    if ((lda & 3) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 3)) return 0;
    return 1;
}

extern "C" int eap_gemm_bf16x3_nn_f32(int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t strideB,
                                      float *C, int64_t ldc, int64_t strideC, int batch, eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (!eap_gemm_bf16x3_nn_f32_supported(M, N, K, A, lda, B, ldb, strideB)) return eap::bad_arg("gemm_bf16x3_nn_f32: unsupported operands (ask eap_gemm_bf16x3_nn_f32_supported)");
    if (batch > 65535) return eap::bad_arg("gemm_bf16x3_nn_f32: batch exceeds 65535");
    Args g{};
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda;
    g.B = B; g.ldb = ldb; g.sB = strideB;
    g.C = C; g.ldc = ldc; g.sC = strideC;
    return launch_split<1>(g, batch, eap::S(stream), "gemm_bf16x3_nn_f32");
}

// ---- two fp16 planes per operand (PL = 2): three matrix instructions per k-tile instead of six -------------------------
// x s = h + l in fp16 after a power-of-two scale s per ROW of A and per COLUMN of B (per output row / column: the scales
// come off again in the epilogue) that puts the row's largest magnitude -- or a bound on it -- at 2^14..2^15:
// representation error <= 2^-23 |x| (rms 2^-25) for elements down to 2^-17 of that magnitude, 2^-40 of it below; products
// h h' + h l' + l h' exact in the fp32 accumulator, l l' <= 2^-22 |x x'| dropped.  The error against fp64 is bounded by the
// tests at that of the fp32-MFMA kernel (fp32 operands, fp32 fmaf chain) on the same operands, per element of the output.
// The caller supplies the magnitudes as device words (bit patterns of non-negative floats), so nothing waits for the host:
//   abs_a [M]                     eap_absmax_rows_f32 of A
//   abs_b [batch][N / grp_b]      float(word) * mult_b >= the largest magnitude in these grp_b columns of B_z:
//                                 eap_absmax_rows_f32 of a k-contiguous B (grp_b = 1), eap_absmax_colgroups_f32 of a row-major B or
//                                 of the intra conv's features (grp_b = anchors per point), eap_so3_grouped_bound_f32 for the inter
//                                 conv's transposed intermediate (a bound per point from the features' per-point maxima, without
//                                 a pass over its 24 GB)
extern "C" int eap_absmax_rows_f32(const float *x, int batch, int rows, int cols, int64_t ld, int64_t stride, int32_t *out_bits, eap_stream_t stream) {
    if (batch <= 0 || rows <= 0) return 0;
    if (cols <= 0 || (cols & 3) || (ld & 3) || (stride & 3) || ld < cols || (reinterpret_cast<uintptr_t>(x) & 15))
        return eap::bad_arg("absmax_rows_f32: rows of whole 16-byte pieces, 16-byte aligned");
    const long long total = (long long)batch * rows;
    hipLaunchKernelGGL(absmax_rows_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, eap::S(stream), total, rows, cols / 4, x, (long long)ld,
                       (long long)stride, reinterpret_cast<unsigned *>(out_bits));
    return eap::check_launch("absmax_rows_f32");
}

extern "C" int eap_absmax_colgroups_f32(const float *x, int batch, int rows, int cols, int64_t ld, int64_t stride, int grp, int32_t *out_bits,
                                        eap_stream_t stream) {
    if (batch <= 0 || cols <= 0) return 0;
    if (grp <= 0 || (grp & 3) || (cols % grp) || (ld & 3) || (stride & 3) || ld < cols || (reinterpret_cast<uintptr_t>(x) & 15) || batch > 65535)
        return eap::bad_arg("absmax_colgroups_f32: groups of whole 16-byte pieces dividing the row, 16-byte aligned rows, batch <= 65535");
    if (int e = eap::hip_fail(hipMemsetAsync(out_bits, 0, (size_t)batch * (cols / grp) * 4, eap::S(stream)), "absmax_colgroups_f32")) return e;
    if (rows <= 0) return 0;
    const int cols4 = cols / 4, blocks = (cols4 + 255) / 256;
    // enough (column block, row slice, item) triples to fill the chip a few times over
    int slices = 1;
    while (slices < 64 && slices * 2 <= rows && (long long)blocks * batch * slices < 2048) slices *= 2;
    hipLaunchKernelGGL(absmax_colgroups_kernel, dim3(blocks, slices, batch), dim3(256), 0, eap::S(stream), rows, cols4, grp / 4, slices, x, (long long)ld,
                       (long long)stride, reinterpret_cast<unsigned *>(out_bits));
    return eap::check_launch("absmax_colgroups_f32");
}

extern "C" int eap_so3_grouped_bound_f32(int b, int p, int nn, int n_sup, const int32_t *point_max_bits, const int32_t *idx, int32_t *out_bits,
                                         eap_stream_t stream) {
    if (b <= 0 || p <= 0) return 0;
    if (nn <= 0 || n_sup <= 0) return eap::bad_arg("so3_grouped_bound_f32: empty neighbour lists");
    const long long total = (long long)b * p;
    hipLaunchKernelGGL(grouped_bound_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, eap::S(stream), total, p, nn, n_sup,
                       reinterpret_cast<const unsigned *>(point_max_bits), idx, reinterpret_cast<unsigned *>(out_bits));
    return eap::check_launch("so3_grouped_bound_f32");
}

// C_z = A B_z (+ the row epilogue of eap_gemm_bf16x3_ep_f32 when scale != NULL); trans_b = 1: B_z [N,K] k-contiguous, 0: B_z [K,N]
// row-major.  Shapes as eap_gemm_bf16x3_f32_supported / _nn_f32_supported say.
extern "C" int eap_gemm_f16x2_f32(int trans_b, int M, int N, int K, const float *A, int64_t lda, const float *B, int64_t ldb, int64_t strideB,
                                  float *C, int64_t ldc, int64_t strideC, int batch, const int32_t *abs_a, const int32_t *abs_b, int grp_b, float mult_b,
                                  const float *scale, const float *shift, float slope, const float *residual, int64_t strideRes,
                                  eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (!abs_a || !abs_b || !(mult_b > 0.f) || grp_b <= 0 || (N % grp_b) != 0)
        return eap::bad_arg("gemm_f16x2_f32: the operand magnitudes are required (eap_absmax_rows_f32 / _colgroups_f32), column groups dividing N");
    if ((scale == nullptr) != (shift == nullptr)) return eap::bad_arg("gemm_f16x2_f32: scale and shift come together");
    const bool ok = trans_b ? eap_gemm_bf16x3_f32_supported(M, N, K, A, lda, B, ldb, strideB) : eap_gemm_bf16x3_nn_f32_supported(M, N, K, A, lda, B, ldb, strideB);
    if (!ok) return eap::bad_arg("gemm_f16x2_f32: unsupported operands (ask eap_gemm_bf16x3_f32_supported / _nn_f32_supported)");
    if (batch > 65535) return eap::bad_arg("gemm_f16x2_f32: batch exceeds 65535");
    Args g{};
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda;
    g.B = B; g.ldb = ldb; g.sB = strideB;
    g.C = C; g.ldc = ldc; g.sC = strideC;
    g.ep_scale = scale; g.ep_shift = shift; g.ep_slope = slope; g.ep_res = scale ? residual : nullptr; g.sRes = strideRes;
    g.absA = reinterpret_cast<const unsigned *>(abs_a); g.absB = reinterpret_cast<const unsigned *>(abs_b); g.multB = mult_b;
    g.grpB = grp_b; g.sAbsB = N / grp_b;
    return trans_b ? launch_split<0, 2>(g, batch, eap::S(stream), "gemm_f16x2_f32") : launch_split<1, 2>(g, batch, eap::S(stream), "gemm_f16x2_f32");
}

// eap_so3_intra_conv_bf16x3_f32 with two fp16 planes: abs_w [o] = eap_absmax_rows_f32 of W, abs_f [b][p] = the features' per-point
// maxima (eap_absmax_colgroups_f32 over [b][c][p*na] with groups of na: a column gathers 12 anchors of its point)
extern "C" int eap_so3_intra_conv_f16x2_f32(int b, int o, int c, int p, int na, int nt, const float *W, const float *feats,
                                            const int32_t *intra_idx, float *out, const int32_t *abs_w, const int32_t *abs_f, eap_stream_t stream) {
    if (b <= 0 || o <= 0 || p <= 0) return 0;
    if (!abs_w || !abs_f) return eap::bad_arg("so3_intra_conv_f16x2_f32: the operand magnitudes are required (eap_absmax_rows_f32 / _colgroups_f32)");
    if (!eap_so3_intra_conv_bf16x3_f32_supported(b, o, c, p, na, nt) || (reinterpret_cast<uintptr_t>(W) & 15))
        return eap::bad_arg("so3_intra_conv_f16x2_f32: unsupported shape (ask eap_so3_intra_conv_bf16x3_f32_supported)");
    if (b > 65535) return eap::bad_arg("so3_intra_conv_f16x2_f32: batch exceeds 65535");
    Args g{};
    const long long pa = (long long)p * na;
    g.M = o; g.N = (int)pa; g.K = c * nt;
    g.A = W; g.lda = (long long)c * nt;
    g.B = feats; g.ldb = pa; g.sB = (long long)c * pa;
    g.C = out; g.ldc = pa; g.sC = (long long)o * pa;
    g.tbl = intra_idx; g.na = na;
    g.absA = reinterpret_cast<const unsigned *>(abs_w); g.absB = reinterpret_cast<const unsigned *>(abs_f); g.multB = 1.0f;
    g.grpB = na; g.sAbsB = p;
    return launch_split<2, 2>(g, b, eap::S(stream), "so3_intra_conv_f16x2_f32");
}

// the intra-SO(3) conv as an implicit GEMM on the split kernel: out[b,o,p,a] = sum_{c,t} W[o, c*12 + t] feats[b,c,p,idx[a,t]]
extern "C" int eap_so3_intra_conv_bf16x3_f32_supported(int b, int o, int c, int p, int na, int nt) {
    if (nt != TAPS || na <= 0 || na > 64 || (long long)p * na >= (1ll << 29)) return 0;
    return tile_dims_ok(o, p * na, c * nt) ? 1 : 0;
}

extern "C" int eap_so3_intra_conv_bf16x3_f32(int b, int o, int c, int p, int na, int nt, const float *W, const float *feats,
                                             const int32_t *intra_idx, float *out, eap_stream_t stream) {
    if (b <= 0 || o <= 0 || p <= 0) return 0;
    if (!eap_so3_intra_conv_bf16x3_f32_supported(b, o, c, p, na, nt) || (reinterpret_cast<uintptr_t>(W) & 15))
        return eap::bad_arg("so3_intra_conv_bf16x3_f32: unsupported shape (ask eap_so3_intra_conv_bf16x3_f32_supported)");
    if (b > 65535) return eap::bad_arg("so3_intra_conv_bf16x3_f32: batch exceeds 65535");
    Args g{};
    const long long pa = (long long)p * na;
    g.M = o; g.N = (int)pa; g.K = c * nt;
    g.A = W; g.lda = (long long)c * nt;
    g.B = feats; g.ldb = pa; g.sB = (long long)c * pa;
    g.C = out; g.ldc = pa; g.sC = (long long)o * pa;
    g.tbl = intra_idx; g.na = na;
    return launch_split<2>(g, b, eap::S(stream), "so3_intra_conv_bf16x3_f32");
}

// ---- batch-reduce form: C[M,N] = sum_z A_z[M,K] B_z[N,K]^T (the weight gradient of the pointwise contraction: dW = sum over
// clouds of dY_z x_z^T, both operands k-contiguous, K = P*A long).  Every (cloud, k-slab) pair is one z of the kernel above and
// writes its own [M,N] partial; a second kernel sums the partials in a fixed order (deterministic, no atomics).
namespace {

__global__ __launch_bounds__(256) void split_reduce_kernel(long long mn4, int parts, const f32x4 *__restrict__ ws, float *__restrict__ C, int N, long long ldc) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mn4) return;
    f32x4 acc = ws[i];
    for (int z = 1; z < parts; ++z) acc += ws[(long long)z * mn4 + i];
    const long long e = i * 4, row = e / N, col = e - row * N;
    *reinterpret_cast<f32x4 *>(C + row * ldc + col) = acc;
}

inline int reduce_slabs(int M, int N, int K, int batch) {
    // enough (item, slab) pairs to fill the chip twice over; slabs of whole k-tiles, at least 64 k-tiles each
    const int tiles = ((M % 256) == 0 ? M / 256 : M / 128) * ((N + BN - 1) / BN);
    int s = 1;
    while (s < 64 && (long long)tiles * batch * s < 512 && (K / BK) % (2 * s) == 0 && K / (2 * s) >= 64 * BK) s *= 2;
    return s;
}

}  // namespace

extern "C" int eap_gemm_bf16x3_reduce_f32_supported(int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B,
                                                    int64_t ldb, int64_t strideB, int64_t ldc) {
    if (!tile_dims_ok(M, N, K)) return 0;
    if ((long long)32 * WAVES_N * lda * 4 >= (1ll << 32) || (long long)32 * WAVES_N * ldb * 4 >= (1ll << 32)) return 0;
    if ((lda & 3) || (ldb & 3) || (strideA & 3) || (strideB & 3) || (ldc & 3)) return 0;
    if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return 0;
    return 1;
}

// floats of scratch eap_gemm_bf16x3_reduce_f32 needs
extern "C" int64_t eap_gemm_bf16x3_reduce_workspace(int M, int N, int K, int batch) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
    return (int64_t)M * N * batch * reduce_slabs(M, N, K, batch);
}

extern "C" int eap_gemm_bf16x3_reduce_f32(int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B, int64_t ldb,
                                          int64_t strideB, float *C, int64_t ldc, int batch, float *workspace, eap_stream_t stream) {
    if (M <= 0 || N <= 0 || batch <= 0) return 0;
    if (!eap_gemm_bf16x3_reduce_f32_supported(M, N, K, A, lda, strideA, B, ldb, strideB, ldc) || (reinterpret_cast<uintptr_t>(C) & 15) ||
        (reinterpret_cast<uintptr_t>(workspace) & 15))
        return eap::bad_arg("gemm_bf16x3_reduce_f32: unsupported operands (ask eap_gemm_bf16x3_reduce_f32_supported)");
    const int slabs = reduce_slabs(M, N, K, batch);
    if ((long long)batch * slabs > 65535) return eap::bad_arg("gemm_bf16x3_reduce_f32: batch x slabs exceeds 65535");
    Args g{};
    g.M = M; g.N = N; g.K = K;
    g.A = A; g.lda = lda; g.sA = strideA;
    g.B = B; g.ldb = ldb; g.sB = strideB;
    g.C = workspace; g.ldc = N; g.sC = (long long)M * N;
    g.slabs = slabs; g.kslab = K / slabs;          // one slab: z is the item and the kernel's k-loop covers K
    if (int e = launch_split<0>(g, batch * slabs, eap::S(stream), "gemm_bf16x3_reduce_f32")) return e;
    const long long mn4 = (long long)M * N / 4;
    hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)((mn4 + 255) / 256)), dim3(256), 0, eap::S(stream), mn4, batch * slabs,
                       reinterpret_cast<const f32x4 *>(workspace), C, N, (long long)ldc);
    return eap::check_launch("gemm_bf16x3_reduce_f32 (sum of the partials)");
}
