// csrc/zpconv_mfma.hip -- the native inter "zpconv" forward (zpconv_cuda.cpp:L41-56, kernel
// zpconv_cuda_kernel.cu:L33-73) on the matrix cores.
//
//   out[b,c,k,p,a] = sum_n w[b,p,a,k,n] * feats[b,c,idx[b,p,a,k,n],a]
//
// Every caller in the reference builds the 5-D index by broadcasting ONE neighbour list per point over (a,k)
// (spconv/functional.py:L232-249, so3conv/functional.py:L2508-2549).  Then, per anchor, the op is the
// [C x NN] x [NN x K] product the SO(3) grouping kernel runs (csrc/so3_inter_lists.hip) -- with the weights
// streamed from memory instead of evaluated.  Two launches:
//   1. zpconv_index_check_kernel: streams the whole index once at HBM speed (that IS the index read the op is
//      charged for), compares every (a,k) row of a point with its first row, writes the point's list idx0[b,p,:]
//      and raises flag[b] on the first difference;
//   2. zpconv_mfma_kernel: per (run of 8 points, 64 channels, anchor group of <= 32), v_mfma_f32_32x32x2_f32 with
//      M = channels (two tiles per wave: the streamed weights are read ONCE for 64 channels), N = kernel points,
//      K = neighbours; a wave owns 4 anchors.  Feature rows come by global -> LDS DMA in 128-byte pieces
//      (image [8 entries][64 channel rows][8 slots], slot = (piece + row) mod 8).  Weights: lane (k, h) owns
//      neighbours 8h .. 8h+7 of a block of 16 -- 32 contiguous bytes of row w[b,p,a,k,:] per anchor, so every
//      128-byte line of w is visited twice, not eight times (a first version on the 8-neighbour chunks of the
//      grouping kernel with 16-byte words spent its time refilling L1: 4.4 ms for 2 clouds against 3.6 for the
//      VALU kernel of csrc/zpconv_rows.hip).  One workgroup per CU (128 KB of LDS, 256 registers per lane).
//      Skips flagged clouds.
// Flagged clouds (an arbitrary 5-D index) are served by csrc/zpconv_rows.hip, which skips the others: no host
// round trip decides anything.
#include "common.h"
#include <type_traits>
#include <utility>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CT = 256;

// (eid != nullptr: also entry ids for the backward's inverse lists -- float4 whose x holds the bits of p * nn + n)
__global__ __launch_bounds__(CT) void zpconv_index_check_kernel(int np, int per_point, int nn, const int32_t *__restrict__ idx,
                                                                int32_t *__restrict__ idx0, float4 *__restrict__ eid,
                                                                int32_t *__restrict__ flag) {
    extern __shared__ int4 s_ref[];                                // the point's first row
    const int p = blockIdx.x, bi = blockIdx.y, t = threadIdx.x;
    const size_t pb = (size_t)bi * np + p;
    const int4 *src = reinterpret_cast<const int4 *>(idx + pb * per_point);
    const int qpr = nn >> 2, total = per_point >> 2;
    for (int q = t; q < qpr; q += CT) {
        const int4 v = src[q];
        s_ref[q] = v;
        if (idx0 != nullptr) reinterpret_cast<int4 *>(idx0 + pb * nn)[q] = v;
    }
    if (eid != nullptr)
        for (int n = t; n < nn; n += CT) eid[pb * nn + n] = make_float4(__uint_as_float((unsigned)(p * nn + n)), 0.f, 0.f, 0.f);
    __syncthreads();
    int mismatch = 0;
    // four independent 16-byte loads per thread in flight
    for (int f0 = t; f0 < total; f0 += 4 * CT) {
        int4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = src[min(f0 + j * CT, total - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int4 r = s_ref[min(f0 + j * CT, total - 1) % qpr];
            mismatch |= (v[j].x ^ r.x) | (v[j].y ^ r.y) | (v[j].z ^ r.z) | (v[j].w ^ r.w);
        }
    }
    if (__syncthreads_or(mismatch != 0) && t == 0) atomicOr(flag + bi, 1);
}

__global__ __launch_bounds__(256) void zpconv_first_rows_kernel(long long rows4, int qpr, long long per_point4, const int4 *__restrict__ idx,
                                                                int4 *__restrict__ idx0) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // (point, 16-byte piece of its first row)
    if (i >= rows4) return;
    const long long pt = i / qpr;
    idx0[i] = idx[pt * per_point4 + (i - pt * qpr)];
}

// ---------------------------------------------------------------------------------------------------------------
constexpr int CB = 64;        // channels per workgroup: two MFMA M tiles sharing one stream of weights
constexpr int NBK = 8;        // entries per LDS stage (4 MFMA k-steps)
constexpr int SBK = 16;       // entries per weight block (two stages)
constexpr int APW = 4;        // anchors per wave
constexpr int NWV = 8;
constexpr int TM = 64 * NWV;
constexpr int NSTD = 8;       // DMA instructions per thread and stage: NBK * CB * 8 pieces / TM
constexpr int PITCH = 32;     // floats per LDS row (8 pieces)
constexpr int RPB = 8;        // consecutive points per workgroup
constexpr unsigned BUF_BYTES = NBK * CB * PITCH * 4;      // 64 KB

__device__ inline unsigned lds_addr(const void *ptr) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)ptr;
}
// wave-wide 16-byte-per-lane global -> LDS DMA (wave-uniform 64-bit base in SGPRs + a 32-bit byte offset per
// lane), invisible to hipcc's waitcnt bookkeeping on purpose: the stage's vmcnt(0) before its barrier covers it
__device__ inline void glds16s(const void *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ inline void glds4(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ inline void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct WSet { f32x4 a0l, a0h, a1l, a1h, a2l, a2h, a3l, a3h; };    // [anchor][half of the lane's 8 neighbours]
template <int AI, int H>
__device__ __forceinline__ f32x4 &wref(WSet &w) {
    if constexpr (AI == 0) { if constexpr (H == 0) return w.a0l; else return w.a0h; }
    else if constexpr (AI == 1) { if constexpr (H == 0) return w.a1l; else return w.a1h; }
    else if constexpr (AI == 2) { if constexpr (H == 0) return w.a2l; else return w.a2h; }
    else { if constexpr (H == 0) return w.a3l; else return w.a3h; }
}

__global__ __launch_bounds__(TM, 2) void zpconv_mfma_kernel(
    int C, int PF, int na, int ks, int P, int nn, int AG, int gsz, int ny, int nb,
    const float *__restrict__ F, const int32_t *__restrict__ idx0, const float *__restrict__ w,
    const int32_t *__restrict__ skip, float *__restrict__ out, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- block -> (run of points, anchor group, cloud, channel slice).  The weights do not depend on the channel: the
    // slices of one unit run back to back on ONE XCD (block b lands on XCD b % 8) and share the weight lines in its
    // L2; XCDs own contiguous ranges of units (neighbouring points' output lines meet in one L2).
    const int nrun = (P + RPB - 1) / RPB, per_cloud = nrun * AG;
    const unsigned units = (unsigned)per_cloud * (unsigned)nb, upx = (units + 7u) >> 3;
    const unsigned xcd = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const unsigned unit = xcd * upx + j / (unsigned)ny;
    if (j / (unsigned)ny >= upx || unit >= units) return;
    const int bi = (int)(unit / (unsigned)per_cloud), qd = (int)(unit % (unsigned)per_cloud), cy = (int)(j % (unsigned)ny);
    if (__builtin_amdgcn_readfirstlane(skip[bi]) != 0) return;     // irregular index: csrc/zpconv_rows.hip serves this cloud
    const int run = qd / AG, ag = qd - run * AG;
    const int r_begin = run * RPB, rows_blk = min(RPB, P - r_begin), c0 = cy * CB;

    const int t = threadIdx.x, lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lk = lane & 31, lh = lane >> 5;
    const int a0 = ag * gsz, gcount = min(gsz, na - a0);           // anchors [a0, a0 + gcount) of this block
    const int npg = gcount >> 2;                                   // 16-byte pieces per feature row that exist
    const int al_beg = wave_u * APW;                               // this wave's anchors = piece wave_u
    const bool active = al_beg < gcount;                           // wave-uniform

    float *s_f = reinterpret_cast<float *>(smem);                  // [2][NBK][CB][PITCH]
    int *s_p = reinterpret_cast<int *>(s_f + 2 * NBK * CB * PITCH);   // [3][SBK] ring of neighbour rows

    const int spr = nn / SBK;                                      // weight blocks per point
    const int nblk = rows_blk * spr, nstage = 2 * nblk;
    const size_t e0 = ((size_t)bi * P + r_begin) * nn;             // the run's neighbours are contiguous in idx0

    // ---- DMA: thread t, instruction u -> flat piece u*512 + t of the stage image [8 entries][64 rows][8 slots]:
    // entry u, channel row (t>>3)&63, slot t&7 -> piece by the row's rotation (absent pieces masked).  The entry's row
    // is wave-uniform: it goes into the scalar base, the per-lane offset is one constant.
    const float *fb = F + ((size_t)bi * C + c0) * PF * na;
    const unsigned lds_f = lds_addr(s_f), lds_p = lds_addr(s_p);
    const int d_cl = (t >> 3) & 63, d_piece = ((t & 7) - d_cl) & 7;
    const bool d_valid = d_piece < npg;
    const unsigned dma_off = ((unsigned)(min(c0 + d_cl, C - 1) - c0) * (unsigned)PF * (unsigned)na + (unsigned)(a0 + 4 * min(d_piece, npg - 1))) * 4u;
    auto issue_idx = [&](int blk, int slot) {
        if (wave_u == 0 && lane < SBK) {
            const size_t e = e0 + (size_t)min(blk, max(nblk - 1, 0)) * SBK + lane;
            glds4(idx0 + e, __builtin_amdgcn_readfirstlane(lds_p + (unsigned)slot * SBK * 4u));
        }
    };
    // stage `par` of a block holds its entries {0..3, 8..11} (par 0) or {4..7, 12..15} (par 1): LDS entry e = 4h + s is
    // neighbour 8h + 4 par + s, the one lane half h contracts in MFMA k-step s of that stage
    unsigned src_row[NSTD];                                        // wave-uniform: the entries' feature rows
    auto prep_rows = [&](int slot, int par) {
#pragma unroll
        for (int u = 0; u < NSTD; ++u)
            src_row[u] = min((unsigned)__builtin_amdgcn_readfirstlane(s_p[slot * SBK + 8 * (u >> 2) + 4 * par + (u & 3)]), (unsigned)PF - 1u);
    };
    auto issue = [&](int u, int buf) {
        if (d_valid && !(dbg & 4))
            glds16s(fb + (size_t)src_row[u] * na, dma_off,
                    __builtin_amdgcn_readfirstlane(lds_f + (unsigned)buf * BUF_BYTES + (unsigned)(u * TM + wave_u * 64) * 16u));
    };

    if (nstage > 0) {
        issue_idx(0, 0);
        issue_idx(1, 1);
        dma_wait();
        __syncthreads();
        prep_rows(0, 0);
#pragma unroll
        for (int u = 0; u < NSTD; ++u) issue(u, 0);
    }

    // Row end.  The MFMA result of one register r is, per lane, 16 bytes (the wave's 4 anchors) of output row
    // (channel(r, lh), k = lk): stored directly, a wave-instruction would write 64 different cache lines of 16 bytes
    // each (measured: 1.3 of the kernel's 3.0 ms for 2 clouds).  Instead the 8 waves exchange through LDS -- the
    // feature stage just consumed is free until the next stage requests into it -- 8 registers per round as tiles
    // [2 channels x 32 kernel points][8 anchor pieces], and thread t writes piece t&7 of row t>>3: 8 consecutive
    // lanes = 128 contiguous bytes of out[b, c, k, p, a0 ..].  Four rounds, two barriers each.
    const size_t o_ks = (size_t)P * na, o_cs = (size_t)ks * P * na;
    float *ob = out + ((size_t)bi * C + c0) * o_cs + a0;
    const int x_row = t >> 3, x_piece = t & 7;
    const int x_rd = x_row * 32 + 4 * ((x_piece + x_row) & 7), x_wr = (lh * 32 + lk) * 32 + 4 * ((wave_u + lk) & 7);   // floats within a tile
    constexpr int XT = 64 * 32;                                    // floats per tile
    // (32-bit byte offset per lane; the launcher bounds one 64-channel slice of the output below 4 GB)
    const unsigned x_off = (unsigned)(((size_t)(4 * (x_row >> 5)) * o_cs + (size_t)min(x_row & 31, ks - 1) * o_ks + 4 * min(x_piece, npg - 1)) * 4);
    const bool x_on = (x_row & 31) < ks && x_piece < npg && !(dbg & 1);
    const int x_cmax = C - c0 - 4 * (x_row >> 5);                  // channels ch < x_cmax exist for this thread
    // tile j of the round -> register I = 8 g + j of the accumulators = channel tile I >> 4, register I & 15
    auto flush_store = [&](const float *tile0, int j, int row, int I) __attribute__((always_inline)) {
        const float4 v = *reinterpret_cast<const float4 *>(tile0 + j * XT + x_rd);
        const int ch = 32 * (I >> 4) + (I & 3) + 8 * ((I & 15) >> 2);
        char *rowp = reinterpret_cast<char *>(ob + (size_t)row * na + (size_t)ch * o_cs);       // uniform
        // (sc1 write-through stores, which drop the written lines from the XCD's L2, were measured here: 10.95 vs 10.90 ms,
        // profiles/r03_store_flavour_experiment.txt -- the output stream is not what evicts the weight lines)
        if (x_on && ch < x_cmax) *reinterpret_cast<float4 *>(rowp + x_off) = v;
    };

    int g0 = 0, g1 = 1, g2 = 2;                                    // ring slots of blocks sb, sb+1, sb+2
    if (!active) {
        // a wave without anchors (the last one of a 28-anchor group) feeds the DMA and takes its share of the row-end stores
        if (nstage > 0) dma_wait();
        __syncthreads();
        int blk_row = 0, row = r_begin;
        for (int st = 0; st < nstage; ++st) {
            const int par = st & 1;
            prep_rows(par ? g1 : g0, par ^ 1);
#pragma unroll
            for (int u = 0; u < NSTD; ++u) issue(u, (st & 1) ^ 1);
            dma_wait();
            if (par && ++blk_row == nn / SBK) {
                const float *tile0 = s_f + (st & 1) * (NBK * CB * PITCH);
                __syncthreads();                                   // everyone has read the stage's operands
                for (int g = 0; g < 4; ++g) {
                    __syncthreads();
                    for (int jj = 0; jj < 8; ++jj) flush_store(tile0, jj, row, 8 * g + jj);
                    if (g < 3) __syncthreads();
                }
                blk_row = 0;
                ++row;
            }
            __syncthreads();
            if (par) { const int gt = g0; g0 = g1; g1 = g2; g2 = gt; }
        }
        return;
    }

    // operand read: the wave's four anchors are ONE 16-byte piece (piece wave_u) of a channel row, at slot
    // (piece + row) mod 8: the 32 channel lanes of a read spread over the banks
    const float4 *fa_lane = reinterpret_cast<const float4 *>(s_f + (size_t)(4 * lh * CB + lk) * PITCH + 4 * ((wave_u + lk) & 7));

    // never zeroed: the first MFMA k-step of every point takes the constant 0 as its C operand (128 v_mov per point and
    // wave would be matrix time)
    f32x16 acc[2][APW];

    // streamed weights: lane (k, h) reads w[b, p, a, k, 16 blk + 8 h .. + 7] as two 16-byte words per anchor, by inline
    // asm like the DMA (hipcc cannot see the DMA requests and would otherwise wait for everything in flight)
    const float *wq = w + (((size_t)bi * P + r_begin) * na + a0 + al_beg) * ks * nn;      // uniform
    const unsigned wlane_b = (unsigned)(min(lk, ks - 1) * nn + 8 * lh) * 4u, astride = (unsigned)(ks * nn);
    const size_t row_jump = (size_t)na * ks * nn - (size_t)(spr - 1) * SBK;
    int wcc = 0;
    auto wload = [&](const float *wp, WSet &ws) __attribute__((always_inline)) {
        if (dbg & 2) wp = w;
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ws.a0l) : "v"(wlane_b), "s"(wp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ws.a0h) : "v"(wlane_b), "s"(wp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ws.a1l) : "v"(wlane_b), "s"(wp + astride) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ws.a1h) : "v"(wlane_b), "s"(wp + astride) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ws.a2l) : "v"(wlane_b), "s"(wp + 2 * (size_t)astride) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ws.a2h) : "v"(wlane_b), "s"(wp + 2 * (size_t)astride) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ws.a3l) : "v"(wlane_b), "s"(wp + 3 * (size_t)astride) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ws.a3h) : "v"(wlane_b), "s"(wp + 3 * (size_t)astride) : "memory");
    };

    auto flush_round = [&]<int G>(std::integral_constant<int, G>, float *tile0, int row) __attribute__((always_inline)) {
        [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
            ((*reinterpret_cast<float4 *>(tile0 + J * XT + x_wr) =
                  make_float4(acc[(8 * G + J) >> 4][0][(8 * G + J) & 15], acc[(8 * G + J) >> 4][1][(8 * G + J) & 15],
                              acc[(8 * G + J) >> 4][2][(8 * G + J) & 15], acc[(8 * G + J) >> 4][3][(8 * G + J) & 15])), ...);
        }(std::make_integer_sequence<int, 8>{});
        __syncthreads();
        [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
            (flush_store(tile0, J, row, 8 * G + J), ...);
        }(std::make_integer_sequence<int, 8>{});
        if (G < 3) __syncthreads();
    };
    auto store_row = [&](int row, int buf) __attribute__((always_inline)) {
        float *tile0 = s_f + buf * (NBK * CB * PITCH);
        __syncthreads();                                           // everyone has read the stage's operands
        flush_round(std::integral_constant<int, 0>{}, tile0, row);
        flush_round(std::integral_constant<int, 1>{}, tile0, row);
        flush_round(std::integral_constant<int, 2>{}, tile0, row);
        flush_round(std::integral_constant<int, 3>{}, tile0, row);
    };

    auto nothing = [] {};
    // MFMA k-step S of a stage with weight half H: 8 MFMAs (2 channel tiles x 4 anchors)
#define ZP_MFMA(TI, AI, FA, COMP, H, S, WS, C) acc[TI][AI] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA.COMP, wref<AI, H>(WS)[S], C, 0, 0, 0)
#define ZP_STEP(S, H, FA0, FA1, WS, MID, END, FIRST)                                                                 \
    do {                                                                                                             \
        __builtin_amdgcn_s_setprio(3);                                                                               \
        if (FIRST) {              /* block-uniform: the first k-step of a point starts from C = 0 */                  \
            const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      \
            ZP_MFMA(0, 0, FA0, x, H, S, WS, zc); ZP_MFMA(1, 0, FA1, x, H, S, WS, zc);                                \
            ZP_MFMA(0, 1, FA0, y, H, S, WS, zc); ZP_MFMA(1, 1, FA1, y, H, S, WS, zc);                                \
            MID();                                                                                                   \
            ZP_MFMA(0, 2, FA0, z, H, S, WS, zc); ZP_MFMA(1, 2, FA1, z, H, S, WS, zc);                                \
            ZP_MFMA(0, 3, FA0, w, H, S, WS, zc); ZP_MFMA(1, 3, FA1, w, H, S, WS, zc);                                \
        } else {                                                                                                     \
            ZP_MFMA(0, 0, FA0, x, H, S, WS, acc[0][0]); ZP_MFMA(1, 0, FA1, x, H, S, WS, acc[1][0]);                  \
            ZP_MFMA(0, 1, FA0, y, H, S, WS, acc[0][1]); ZP_MFMA(1, 1, FA1, y, H, S, WS, acc[1][1]);                  \
            MID();                                                                                                   \
            ZP_MFMA(0, 2, FA0, z, H, S, WS, acc[0][2]); ZP_MFMA(1, 2, FA1, z, H, S, WS, acc[1][2]);                  \
            ZP_MFMA(0, 3, FA0, w, H, S, WS, acc[0][3]); ZP_MFMA(1, 3, FA1, w, H, S, WS, acc[1][3]);                  \
        }                                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        END();                                                                                                       \
    } while (0)
    // one stage (8 entries): operands of a k-step are read two steps ahead; the next stage's 8 pieces per thread are
    // requested from inside the first two k-steps.  (A macro, not a lambda: instantiated four times.)
#define ZP_STAGE(ST, PAR, WS, WNEXT)                                                                                 \
    do {                                                                                                             \
        const int buf = (ST) & 1, nbuf = buf ^ 1;                                                                    \
        const float4 *fbuf = fa_lane + buf * (NBK * CB * PITCH / 4);                                                 \
        float4 fa00 = fbuf[0], fa01 = fbuf[32 * PITCH / 4];                                                          \
        float4 fa10 = fbuf[CB * PITCH / 4], fa11 = fbuf[CB * PITCH / 4 + 32 * PITCH / 4];                            \
        if ((PAR) == 0) {                                                                                            \
            /* weights of the next block (of the last one again past the end: no branch around the loads) */         \
            const float *wnext = wq;                                                                                 \
            if ((ST) + 2 < nstage) wnext = wq + (wcc + 1 == spr ? row_jump : (size_t)SBK);                            \
            wload(wnext, WNEXT);                                                                                     \
            if (++wcc == spr) wcc = 0;                                                                               \
            wq = wnext;                                                                                              \
            issue_idx(((ST) >> 1) + 2, g2);                                                                          \
        }                                                                                                            \
        prep_rows((PAR) ? g1 : g0, (PAR) ^ 1);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ZP_STEP(0, PAR, fa00, fa01, WS, [&] { issue(0, nbuf); issue(1, nbuf); }, [&] { issue(2, nbuf); issue(3, nbuf); }, ((PAR) == 0 && blk_row == 0)); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        fa00 = fbuf[2 * (CB * PITCH / 4)]; fa01 = fbuf[2 * (CB * PITCH / 4) + 32 * PITCH / 4];                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ZP_STEP(1, PAR, fa10, fa11, WS, [&] { issue(4, nbuf); issue(5, nbuf); }, [&] { issue(6, nbuf); issue(7, nbuf); }, false); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        fa10 = fbuf[3 * (CB * PITCH / 4)]; fa11 = fbuf[3 * (CB * PITCH / 4) + 32 * PITCH / 4];                       \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ZP_STEP(2, PAR, fa00, fa01, WS, nothing, nothing, false);                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ZP_STEP(3, PAR, fa10, fa11, WS, nothing, nothing, false);                                                    \
        dma_wait();                                                                                                  \
        if ((PAR) == 1) {                                                                                            \
            if (++blk_row == spr) {                       /* block-uniform */                                        \
                store_row(row, buf);                                                                                 \
                blk_row = 0;                                                                                         \
                ++row;                                                                                               \
            }                                                                                                        \
        }                                                                                                            \
        __syncthreads();                                                                                             \
        if ((PAR) == 1) { const int gt = g0; g0 = g1; g1 = g2; g2 = gt; }                                            \
    } while (0)

    WSet wA, wB;
    int blk_row = 0, row = r_begin;
    if (nstage > 0) {
        wload(wq, wA);
        dma_wait();                                                // first stage image, first weights
    }
    __syncthreads();
    for (int st = 0; st < nstage; st += 4) {
        ZP_STAGE(st, 0, wA, wB);
        ZP_STAGE(st + 1, 1, wA, wB);
        if (st + 2 < nstage) {
            ZP_STAGE(st + 2, 0, wB, wA);
            ZP_STAGE(st + 3, 1, wB, wA);
        }
    }
#undef ZP_STAGE
#undef ZP_STEP
#undef ZP_MFMA
}

}  // namespace

namespace eap {

int zpconv_index_check(int b, int np, int per_point, int nn, const int32_t *idx, int32_t *idx0, float *eid, int32_t *flag, hipStream_t s) {
    hipLaunchKernelGGL(zpconv_index_check_kernel, dim3(np, b), dim3(CT), (size_t)nn * 4, s, np, per_point, nn, idx, idx0,
                       reinterpret_cast<float4 *>(eid), flag);
    return eap::check_launch("inter_zpconv (index check)");
}

int zpconv_first_rows(int b, int np, int per_point, int nn, const int32_t *idx, int32_t *idx0, hipStream_t s) {
    const long long rows4 = (long long)b * np * (nn >> 2);
    hipLaunchKernelGGL(zpconv_first_rows_kernel, dim3(eap::cdiv(rows4, 256)), dim3(256), 0, s, rows4, nn >> 2, (long long)(per_point >> 2),
                       reinterpret_cast<const int4 *>(idx), reinterpret_cast<int4 *>(idx0));
    return eap::check_launch("inter_zpconv (first rows)");
}

bool inter_zpconv_mfma_supported(int np, int nq, int na, int ks, int nn, int c) {
    if (na <= 0 || na > 64 || (na & 3) != 0 || ks <= 0 || ks > 32 || nn <= 0 || (nn % SBK) != 0 || c < 16) return false;
    if ((long long)CB * nq * na * 4 >= (1ll << 32) || (long long)ks * nn * 4 >= (1ll << 24)) return false;
    if ((long long)CB * ks * np * na * 4 >= (1ll << 32)) return false;          // 32-bit byte offsets within a 64-channel slice of the output
    return true;
}

int inter_zpconv_mfma_fwd(int b, int np, int nq, int na, int ks, int nn, int c, const int32_t *idx0, const float *w,
                          const float *feats, const int32_t *skip, float *out, hipStream_t s) {
    const int AG = na > 32 ? 2 : 1, gsz = AG == 1 ? na : ((na / 2 + 3) & ~3);
    const size_t shmem = 2 * (size_t)BUF_BYTES + 4 * 3 * SBK;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)zpconv_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                          "inter_zpconv_forward (matrix path) shared memory");
    if (e) return e;
    const int ny = (c + CB - 1) / CB;
    const long long units = (long long)((np + RPB - 1) / RPB) * AG * b, blocks = 8 * ((units + 7) / 8) * ny;
    if (blocks >= (1ll << 31)) return eap::bad_arg("inter_zpconv_forward (matrix path): too many workgroups");
    // EAP_ZP_DEBUG (timing ablations only: wrong results): 1 = no output stores, 2 = every weight load hits the same
    // lines, 4 = no feature DMA
#ifdef EAP_ABLATION      // only in a library built with `make ABLATION=1`; a production build never reads the variable
    const int dbg = getenv("EAP_ZP_DEBUG") ? atoi(getenv("EAP_ZP_DEBUG")) : 0;
#else
    const int dbg = 0;
#endif
    hipLaunchKernelGGL(zpconv_mfma_kernel, dim3((unsigned)blocks), dim3(TM), shmem, s, c, nq, na, ks, np, nn, AG, gsz, ny, b, feats, idx0, w,
                       skip, out, dbg);
    return eap::check_launch("inter_zpconv_forward (matrix path)");
}

}  // namespace eap

extern "C" int64_t eap_inter_zpconv_fwd_workspace(int b, int np, int ann) {
    return 256 + 4ll * ((int64_t)b * np * ann) + 2 * 4ll * 64 * ((b + 63) / 64);        // flags | zeros | idx0
}

extern "C" int eap_inter_zpconv_fwd_ws_f32(int b, int np, int nq, int na, int ks, int ann, int c, const int32_t *idx,
                                           const float *w, const float *src, float *dst, void *workspace,
                                           eap_stream_t stream) {
    if (b <= 0 || np <= 0) return 0;
    hipStream_t s = eap::S(stream);
    const bool matrix = workspace != nullptr && eap::inter_zpconv_mfma_supported(np, nq, na, ks, ann, c) &&
                        eap::inter_zpconv_rows_supported(np, nq, na, ks, ann, c) && (long long)na * ks * ann < (1ll << 31) &&
                        ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0;
    if (!matrix) return eap_inter_zpconv_fwd_f32(b, np, nq, na, ks, ann, c, idx, w, src, dst, stream);
    // The matrix kernel walks one neighbour list per point: idx0 = the first (a,k) row of every point (a 1 MB gather).  Whether
    // every other row equals it is checked by streaming the whole index once -- on the side stream, BESIDE the matrix kernel
    // (an HBM stream next to a kernel bound by its barriers); a cloud that fails the check is recomputed afterwards by the
    // arbitrary-index kernel, which overwrites its output.
    const int fl = 64 * ((b + 63) / 64);
    int32_t *flag = reinterpret_cast<int32_t *>(workspace);
    int32_t *nobody = flag + fl;                                         // all zero: the matrix kernel skips no cloud
    int32_t *idx0 = flag + 2 * fl;
    int e = eap::hip_fail(hipMemsetAsync(flag, 0, sizeof(int32_t) * 2 * fl, s), "inter_zpconv_forward flags");
    if (e) return e;
    e = eap::zpconv_first_rows(b, np, na * ks * ann, ann, idx, idx0, s);
    if (e) return e;
    hipStream_t side;
    e = eap::side_fork(s, &side);
    if (e) return e;
    eap::SideJoin joiner(s);              // (also on the error returns below)
    e = eap::zpconv_index_check(b, np, na * ks * ann, ann, idx, nullptr, nullptr, flag, side);
    if (e) return e;
#ifdef EAP_EXPERIMENTS   // `make EXPERIMENTS=1`: the 32-neighbour re-cut of tools/experiments/kernels/zpconv_mfma2.hip behind eap_inter_zpconv_fwd_kernel(2)
    if (eap::zp_fwd_kernel() == 2 && eap::inter_zpconv_mfma2_supported(np, nq, na, ks, ann, c))
        e = eap::inter_zpconv_mfma2_fwd(b, np, nq, na, ks, ann, c, idx0, w, src, nobody, dst, s);
    else
#endif
    e = eap::inter_zpconv_mfma_fwd(b, np, nq, na, ks, ann, c, idx0, w, src, nobody, dst, s);
    if (e) return e;
    e = joiner.join();
    if (e) return e;
    return eap::inter_zpconv_rows_fwd(b, np, nq, na, ks, ann, c, idx, w, src, dst, flag, s);
}
