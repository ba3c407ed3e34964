// csrc/zpconv_mfma2.hip -- the matrix path of the native inter "zpconv" forward (csrc/zpconv_mfma.hip; reference
// zpconv_cuda.cpp:L41-56, zpconv_cuda_kernel.cu:L33-73) re-cut so that EVERY 128-BYTE LINE OF THE STREAMED WEIGHTS IS
// CONSUMED IN ONE VISIT (round 3).
//
//   out[b,c,k,p,a] = sum_n w[b,p,a,k,n] * feats[b,c,idx0[b,p,n],a]
//
// The first kernel gives lane (k, h) the neighbours 8h .. 8h+7 of a block of 16 for each of the wave's 4 anchors: 32
// bytes per (anchor, k) row and block, so a 128-byte line of w (32 neighbours) is requested by two consecutive blocks --
// and by the counters the second request misses the L2 (the XCD streams ~4 MB between them): 34 GB fetched per launch
// where the weights are 12.6 GB (profiles/r03_pmc_traffic.json).  Here a weight block is 32 neighbours x 2 anchors:
// lane (k, h) reads 64 contiguous bytes (neighbours 16h .. 16h+15) per anchor, the two lane halves together one whole
// line, with the same 32 registers per block and the same double buffer.  A point is then two passes over its
// neighbours, one per anchor pair of the wave; the feature rows are gathered (global -> LDS DMA, L2 hits) once per pass.
// Everything else is the first kernel's: 8 waves x 4 anchors x 2 channel tiles of accumulators, stages of 8 entries,
// operands read two k-steps ahead, row-end exchange through LDS into 128-byte output pieces.
//
// STATUS (measured on MI355X, 8 x 4096 points, C = 64; profiles/r03_ae_zpconv_fwd_recut.txt): the traffic is fixed --
// 33.8 -> 21.2 GB fetched per launch, L2 hit 0.55 -> 0.76 -- and the kernel is NOT faster: 10.3 ms against 9.7 under the
// profiler, 12.0-12.3 against 11.5 for the whole op.  With a third of its bytes gone the kernel turns out not to be
// bandwidth-bound: the matrix pipe is busy a third of the time in both (0.33 / 0.35), the rest is the per-stage
// barrier + DMA round trip (twice as many stages here, 16 MFMAs each instead of 32) and the row-end exchange, during
// which nothing else runs on a CU that holds ONE workgroup.  Correct and tested (tests/test_gpu_parity.py), kept behind
// eap_inter_zpconv_fwd_kernel(2); the first kernel stays the default.  A third cut -- the first kernel's scheme in 4-wave
// workgroups of 16 anchors, TWO per CU so that one's row end overlaps the other's matrix work -- was bit-identical and
// slower still (11.1 ms, 36.4 GB fetched + 15.7 GB written in 64-byte pieces, matrix pipe 0.32): not kept.
#include "common.h"
#include <type_traits>
#include <utility>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CB = 64;        // channels per workgroup: two MFMA M tiles sharing one stream of weights
constexpr int NBK = 8;        // entries per LDS stage (4 MFMA k-steps)
constexpr int WBK = 32;       // neighbours per weight block (four stages): one 128-byte line per (anchor, k) row
constexpr int NNMAX = 128;    // neighbours per point the index ring holds
constexpr int APW = 4;        // anchors per wave (two passes of two)
constexpr int NWV = 8;
constexpr int TM = 64 * NWV;
constexpr int NSTD = 8;       // DMA instructions per thread and stage: NBK * CB * 8 pieces / TM
constexpr int PITCH = 32;     // floats per LDS row (8 pieces)
constexpr int RPB = 8;        // consecutive points per workgroup
constexpr unsigned BUF_BYTES = NBK * CB * PITCH * 4;      // 64 KB

__device__ inline unsigned lds_addr(const void *ptr) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)ptr;
}
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

// one weight block of a wave: [anchor of the pair][quarter of the lane's 16 neighbours]
struct WSet { f32x4 a0q0, a0q1, a0q2, a0q3, a1q0, a1q1, a1q2, a1q3; };
template <int A>
__device__ __forceinline__ float pick(const float4 &v) {
    if constexpr (A == 0) return v.x; else if constexpr (A == 1) return v.y; else if constexpr (A == 2) return v.z; else return v.w;
}
template <int AJ, int Q>
__device__ __forceinline__ f32x4 &wref(WSet &w) {
    if constexpr (AJ == 0) {
        if constexpr (Q == 0) return w.a0q0; else if constexpr (Q == 1) return w.a0q1; else if constexpr (Q == 2) return w.a0q2; else return w.a0q3;
    } else {
        if constexpr (Q == 0) return w.a1q0; else if constexpr (Q == 1) return w.a1q1; else if constexpr (Q == 2) return w.a1q2; else return w.a1q3;
    }
}

// SPR = weight blocks per pass = neighbours / 32 (2 or 4), a template parameter: the stage sequence of a point is fully
// unrolled, so "first k-step of a pass" and "last stage of the point" are compile-time facts and no branch surrounds an MFMA
template <int SPR>
__global__ __launch_bounds__(TM, 2) void zpconv_mfma2_kernel(
    int C, int PF, int na, int ks, int P, int nn, int AG, int gsz, int ny, int nb,
    const float *__restrict__ F, const int32_t *__restrict__ idx0, const float *__restrict__ w,
    const int32_t *__restrict__ skip, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- block -> (run of points, anchor group, cloud, channel slice): as csrc/zpconv_mfma.hip ----
    const int nrun = (P + RPB - 1) / RPB, per_cloud = nrun * AG;
    const unsigned units = (unsigned)per_cloud * (unsigned)nb, upx = (units + 7u) >> 3;
    const unsigned xcd = blockIdx.x & 7u, jb = blockIdx.x >> 3;
    const unsigned unit = xcd * upx + jb / (unsigned)ny;
    if (jb / (unsigned)ny >= upx || unit >= units) return;
    const int bi = (int)(unit / (unsigned)per_cloud), qd = (int)(unit % (unsigned)per_cloud), cy = (int)(jb % (unsigned)ny);
    if (__builtin_amdgcn_readfirstlane(skip[bi]) != 0) return;     // irregular index: csrc/zpconv_rows.hip serves this cloud
    const int run = qd / AG, ag = qd - run * AG;
    const int r_begin = run * RPB, rows_blk = min(RPB, P - r_begin), c0 = cy * CB;

    const int t = threadIdx.x, lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lk = lane & 31, lh = lane >> 5;
    const int a0 = ag * gsz, gcount = min(gsz, na - a0);
    const int npg = gcount >> 2;
    const int al_beg = wave_u * APW;
    const bool active = al_beg < gcount;                           // wave-uniform

    float *s_f = reinterpret_cast<float *>(smem);                  // [2][NBK][CB][PITCH]
    int *s_p = reinterpret_cast<int *>(s_f + 2 * NBK * CB * PITCH);   // [2][NNMAX]: neighbour rows of this point and the next

    constexpr int spr = SPR;
    const int nstage = rows_blk * 2 * spr * 4;
    const size_t e0 = ((size_t)bi * P + r_begin) * nn;

    // ---- DMA (as the first kernel): thread t, instruction u -> entry u, channel row (t>>3)&63, slot t&7 ----
    const float *fb = F + ((size_t)bi * C + c0) * PF * na;
    const unsigned lds_f = lds_addr(s_f), lds_p = lds_addr(s_p);
    const int d_cl = (t >> 3) & 63, d_piece = ((t & 7) - d_cl) & 7;
    const bool d_valid = d_piece < npg;
    const unsigned dma_off = ((unsigned)(min(c0 + d_cl, C - 1) - c0) * (unsigned)PF * (unsigned)na + (unsigned)(a0 + 4 * min(d_piece, npg - 1))) * 4u;
    auto issue_idx = [&](int rowi) {                               // the neighbour list of point r_begin + rowi into ring slot rowi & 1
        if (rowi < rows_blk && wave_u * 64 < nn && wave_u * 64 + lane < nn)
            glds4(idx0 + e0 + (size_t)rowi * nn + wave_u * 64 + lane,
                  __builtin_amdgcn_readfirstlane(lds_p + (unsigned)((rowi & 1) * NNMAX + wave_u * 64) * 4u));
    };
    // stage `par` (0..3) of weight block `blk`: LDS entry e = 4h + s is neighbour 32 blk + 16 h + 4 par + s, the one lane
    // half h contracts in MFMA k-step s of that stage
    unsigned src_row[NSTD];
    auto prep_rows = [&](int rowi, int blk, int par) {
#pragma unroll
        for (int u = 0; u < NSTD; ++u)
            src_row[u] = min((unsigned)__builtin_amdgcn_readfirstlane(s_p[(rowi & 1) * NNMAX + WBK * blk + 16 * (u >> 2) + 4 * par + (u & 3)]), (unsigned)PF - 1u);
    };
    auto issue = [&](int u, int buf) {
        if (d_valid)
            glds16s(fb + (size_t)src_row[u] * na, dma_off,
                    __builtin_amdgcn_readfirstlane(lds_f + (unsigned)buf * BUF_BYTES + (unsigned)(u * TM + wave_u * 64) * 16u));
    };
    // the stage after (rowi, ap, blk, par); past the last one: the same stage again (a harmless reload into the idle buffer)
    auto next_stage = [&](int &rowi, int &ap, int &blk, int &par) {
        if (par < 3) { ++par; return; }
        if (blk + 1 < spr) { par = 0; ++blk; return; }
        if (ap == 0) { par = 0; blk = 0; ap = 1; return; }
        if (rowi + 1 < rows_blk) { par = 0; blk = 0; ap = 0; ++rowi; }
    };

    if (nstage > 0) {
        issue_idx(0);
        dma_wait();
        __syncthreads();
        prep_rows(0, 0, 0);
#pragma unroll
        for (int u = 0; u < NSTD; ++u) issue(u, 0);
    }

    // ---- row end: the 8 waves exchange the accumulators through LDS into 128-byte output pieces (first kernel) ----
    const size_t o_ks = (size_t)P * na, o_cs = (size_t)ks * P * na;
    float *ob = out + ((size_t)bi * C + c0) * o_cs + a0;
    const int x_row = t >> 3, x_piece = t & 7;
    const int x_rd = x_row * 32 + 4 * ((x_piece + x_row) & 7), x_wr = (lh * 32 + lk) * 32 + 4 * ((wave_u + lk) & 7);
    constexpr int XT = 64 * 32;
    const unsigned x_off = (unsigned)(((size_t)(4 * (x_row >> 5)) * o_cs + (size_t)min(x_row & 31, ks - 1) * o_ks + 4 * min(x_piece, npg - 1)) * 4);
    const bool x_on = (x_row & 31) < ks && x_piece < npg;
    const int x_cmax = C - c0 - 4 * (x_row >> 5);
    auto flush_store = [&](const float *tile0, int j, int row, int I) __attribute__((always_inline)) {
        const float4 v = *reinterpret_cast<const float4 *>(tile0 + j * XT + x_rd);
        const int ch = 32 * (I >> 4) + (I & 3) + 8 * ((I & 15) >> 2);
        char *rowp = reinterpret_cast<char *>(ob + (size_t)row * na + (size_t)ch * o_cs);       // uniform
        if (x_on && ch < x_cmax) *reinterpret_cast<float4 *>(rowp + x_off) = v;
    };

    if (!active) {
        // a wave without anchors (the last one of a 28-anchor group) feeds the DMA and takes its share of the row-end stores
        if (nstage > 0) dma_wait();
        __syncthreads();
        int rowi = 0, ap = 0, blk = 0, par = 0;
        for (int st = 0; st < nstage; ++st) {
            const bool first_of_point = ap == 0 && blk == 0 && par == 0, last_of_point = ap == 1 && blk == spr - 1 && par == 3;
            const int row = r_begin + rowi;
            if (first_of_point) issue_idx(rowi + 1);
            next_stage(rowi, ap, blk, par);
            prep_rows(rowi, blk, par);
#pragma unroll
            for (int u = 0; u < NSTD; ++u) issue(u, (st & 1) ^ 1);
            dma_wait();
            if (last_of_point) {
                const float *tile0 = s_f + (st & 1) * (NBK * CB * PITCH);
                __syncthreads();                                   // everyone has read the stage's operands
                for (int g = 0; g < 4; ++g) {
                    __syncthreads();
                    for (int jj = 0; jj < 8; ++jj) flush_store(tile0, jj, row, 8 * g + jj);
                    if (g < 3) __syncthreads();
                }
            }
            __syncthreads();
        }
        return;
    }

    // operand read: the wave's four anchors are ONE 16-byte piece (piece wave_u, slot (piece + row) mod 8: the conflict-free
    // ds_read_b128 pattern of the first kernel); a pass uses two of its four components
    const float4 *fa_lane = reinterpret_cast<const float4 *>(s_f + (size_t)(4 * lh * CB + lk) * PITCH + 4 * ((wave_u + lk) & 7));
    constexpr int ENT_F2 = CB * PITCH / 4, TILE_F2 = 32 * PITCH / 4, BUF_F2 = NBK * CB * PITCH / 4;

    f32x16 acc[2][APW];        // never zeroed: the first k-step of a pass starts from the constant 0

    // streamed weights: lane (k, h) reads w[b, p, a, k, 32 blk + 16 h .. + 15] -- 64 contiguous bytes, the two halves one line
    const float *wbase = w + (((size_t)bi * P + r_begin) * na + a0 + al_beg) * ks * nn;      // uniform
    const unsigned wlane_b = (unsigned)(min(lk, ks - 1) * nn + 16 * lh) * 4u;
    const size_t astride = (size_t)ks * nn, pstride = (size_t)na * ks * nn;
    auto wptr = [&](int rowi, int ap, int blk) { return wbase + (size_t)rowi * pstride + (size_t)(2 * ap) * astride + (size_t)(WBK * blk); };
    auto wload = [&](const float *wp, WSet &ws) __attribute__((always_inline)) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ws.a0q0) : "v"(wlane_b), "s"(wp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ws.a0q1) : "v"(wlane_b), "s"(wp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(ws.a0q2) : "v"(wlane_b), "s"(wp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(ws.a0q3) : "v"(wlane_b), "s"(wp) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ws.a1q0) : "v"(wlane_b), "s"(wp + astride) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(ws.a1q1) : "v"(wlane_b), "s"(wp + astride) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(ws.a1q2) : "v"(wlane_b), "s"(wp + astride) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(ws.a1q3) : "v"(wlane_b), "s"(wp + astride) : "memory");
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
    // MFMA k-step S of stage quarter Q: 4 MFMAs (2 channel tiles x the pass's 2 anchors)
#define ZP2_MFMA(TI, AP, AJ, FA, COMP, Q, S, WS, C) acc[TI][2 * (AP) + (AJ)] = __builtin_amdgcn_mfma_f32_32x32x2f32(pick<2 * (AP) + (AJ)>(FA), wref<AJ, Q>(WS)[S], C, 0, 0, 0)
#define ZP2_STEP(S, Q, AP, FA0, FA1, WS, MID, END, FIRST)                                                            \
    do {                                                                                                             \
        __builtin_amdgcn_s_setprio(3);                                                                               \
        if (FIRST) {              /* block-uniform: the first k-step of a pass starts from C = 0 */                   \
            const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      \
            ZP2_MFMA(0, AP, 0, FA0, x, Q, S, WS, zc); ZP2_MFMA(1, AP, 0, FA1, x, Q, S, WS, zc);                      \
            MID();                                                                                                   \
            ZP2_MFMA(0, AP, 1, FA0, y, Q, S, WS, zc); ZP2_MFMA(1, AP, 1, FA1, y, Q, S, WS, zc);                      \
        } else {                                                                                                     \
            ZP2_MFMA(0, AP, 0, FA0, x, Q, S, WS, acc[0][2 * (AP)]); ZP2_MFMA(1, AP, 0, FA1, x, Q, S, WS, acc[1][2 * (AP)]);          \
            MID();                                                                                                   \
            ZP2_MFMA(0, AP, 1, FA0, y, Q, S, WS, acc[0][2 * (AP) + 1]); ZP2_MFMA(1, AP, 1, FA1, y, Q, S, WS, acc[1][2 * (AP) + 1]);  \
        }                                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        END();                                                                                                       \
    } while (0)
    // one stage (8 entries) of block BLK of pass AP: operands of a k-step are read two steps ahead; the next stage's 8 pieces
    // per thread are requested from inside the first two k-steps.  st parity = PAR parity (four stages per block).
#define ZP2_STAGE(PAR, AP, BLK, WS, WNEXT)                                                                           \
    do {                                                                                                             \
        constexpr int buf = (PAR) & 1, nbuf = buf ^ 1;                                                               \
        const float4 *fbuf = fa_lane + buf * BUF_F2;                                                                 \
        float4 fa00 = fbuf[0], fa01 = fbuf[TILE_F2];                                                                 \
        float4 fa10 = fbuf[ENT_F2], fa11 = fbuf[ENT_F2 + TILE_F2];                                                   \
        int n_row = rowi, n_ap = (AP), n_blk = (BLK), n_par = (PAR);                                                 \
        next_stage(n_row, n_ap, n_blk, n_par);                                                                       \
        if ((PAR) == 0) {                                                                                            \
            /* weights of the next block (of this one again past the end: no branch around the loads) */             \
            int w_row = rowi, w_ap = (AP), w_blk = (BLK), w_par = 3;                                                 \
            next_stage(w_row, w_ap, w_blk, w_par);                                                                   \
            wload(wptr(w_row, w_ap, w_blk), WNEXT);                                                                  \
            if ((AP) == 0 && (BLK) == 0) issue_idx(rowi + 1);                                                        \
        }                                                                                                            \
        prep_rows(n_row, n_blk, n_par);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ZP2_STEP(0, PAR, AP, fa00, fa01, WS, [&] { issue(0, nbuf); issue(1, nbuf); }, [&] { issue(2, nbuf); issue(3, nbuf); }, ((PAR) == 0 && (BLK) == 0)); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        fa00 = fbuf[2 * ENT_F2]; fa01 = fbuf[2 * ENT_F2 + TILE_F2];                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ZP2_STEP(1, PAR, AP, fa10, fa11, WS, [&] { issue(4, nbuf); issue(5, nbuf); }, [&] { issue(6, nbuf); issue(7, nbuf); }, false); \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        fa10 = fbuf[3 * ENT_F2]; fa11 = fbuf[3 * ENT_F2 + TILE_F2];                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ZP2_STEP(2, PAR, AP, fa00, fa01, WS, nothing, nothing, false);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
        ZP2_STEP(3, PAR, AP, fa10, fa11, WS, nothing, nothing, false);                                               \
        dma_wait();                                                                                                  \
        if ((PAR) == 3 && (AP) == 1 && (BLK) == spr - 1) store_row(r_begin + rowi, buf);      /* block-uniform */    \
        __syncthreads();                                                                                             \
    } while (0)
#define ZP2_BLOCK(AP, BLK, WS, WNEXT)                                                                                \
    do {                                                                                                             \
        ZP2_STAGE(0, AP, BLK, WS, WNEXT); ZP2_STAGE(1, AP, BLK, WS, WNEXT);                                          \
        ZP2_STAGE(2, AP, BLK, WS, WNEXT); ZP2_STAGE(3, AP, BLK, WS, WNEXT);                                          \
    } while (0)

    WSet wA, wB;
    if (nstage > 0) {
        wload(wptr(0, 0, 0), wA);
        dma_wait();                                                // first stage image, first weights
    }
    __syncthreads();
    for (int rowi = 0; rowi < rows_blk; ++rowi) {                  // SPR is even: every pass starts on wA
        ZP2_BLOCK(0, 0, wA, wB);
        ZP2_BLOCK(0, 1, wB, wA);
        if constexpr (SPR == 4) {
            ZP2_BLOCK(0, 2, wA, wB);
            ZP2_BLOCK(0, 3, wB, wA);
        }
        ZP2_BLOCK(1, 0, wA, wB);
        ZP2_BLOCK(1, 1, wB, wA);
        if constexpr (SPR == 4) {
            ZP2_BLOCK(1, 2, wA, wB);
            ZP2_BLOCK(1, 3, wB, wA);
        }
    }
#undef ZP2_BLOCK
#undef ZP2_STAGE
#undef ZP2_STEP
#undef ZP2_MFMA
}

int g_zp_fwd_kernel = 1;      // eap_inter_zpconv_fwd_kernel: 1 = csrc/zpconv_mfma.hip (default, see STATUS above), 2 = this file where it applies

}  // namespace

namespace eap {

int zp_fwd_kernel() { return g_zp_fwd_kernel; }

bool inter_zpconv_mfma2_supported(int np, int nq, int na, int ks, int nn, int c) {
    return inter_zpconv_mfma_supported(np, nq, na, ks, nn, c) && (nn == 64 || nn == 128) &&
           (long long)na * ks * nn * 4 * RPB < (1ll << 40);
}

int inter_zpconv_mfma2_fwd(int b, int np, int nq, int na, int ks, int nn, int c, const int32_t *idx0, const float *w,
                           const float *feats, const int32_t *skip, float *out, hipStream_t s) {
    const int AG = na > 32 ? 2 : 1, gsz = AG == 1 ? na : ((na / 2 + 3) & ~3);
    const size_t shmem = 2 * (size_t)BUF_BYTES + 4 * 2 * NNMAX;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)zpconv_mfma2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                          "inter_zpconv_forward (matrix path 2) shared memory");
    if (e) return e;
    e = eap::hip_fail(hipFuncSetAttribute((const void *)zpconv_mfma2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                      "inter_zpconv_forward (matrix path 2) shared memory");
    if (e) return e;
    const int ny = (c + CB - 1) / CB;
    const long long units = (long long)((np + RPB - 1) / RPB) * AG * b, blocks = 8 * ((units + 7) / 8) * ny;
    if (blocks >= (1ll << 31)) return eap::bad_arg("inter_zpconv_forward (matrix path 2): too many workgroups");
    if (nn == 64)
        hipLaunchKernelGGL(zpconv_mfma2_kernel<2>, dim3((unsigned)blocks), dim3(TM), shmem, s, c, nq, na, ks, np, nn, AG, gsz, ny, b, feats, idx0, w, skip, out);
    else
        hipLaunchKernelGGL(zpconv_mfma2_kernel<4>, dim3((unsigned)blocks), dim3(TM), shmem, s, c, nq, na, ks, np, nn, AG, gsz, ny, b, feats, idx0, w, skip, out);
    return eap::check_launch("inter_zpconv_forward (matrix path 2)");
}

}  // namespace eap

// 1 (default): the first matrix kernel (16-neighbour weight blocks, every weight line visited by two blocks); 2: the re-cut
// with 32-neighbour blocks where the neighbour count is 64 or 128 (less traffic, not faster: see the file's STATUS).  Returns the previous setting; other values only query.
extern "C" int eap_inter_zpconv_fwd_kernel(int which) {
    const int was = g_zp_fwd_kernel;
    if (which == 1 || which == 2) g_zp_fwd_kernel = which;
    return was;
}
