// so3_inter_lists_h2.hip -- the entry-list grouping kernel (csrc/so3_inter_lists2.hip) with its products on the fp16 matrix
// cores: fp32 features, fp32 weights, fp32 accumulation, every product formed from TWO fp16 planes per operand (x s = h + l,
// three v_mfma_f32_32x32x16_f16 per tile and 16 entries; the scheme of csrc/gemm_bf16x3.hip's gemm_f16x2_kernel).
//
//   forward   X[b,c,k,p,a]  = sum_n  F[b,c,idx[b,p,n],a]  * w(p,a,k,n)          (vgtk/vgtk/so3conv/functional.py:L1112-1261)
//   backward  Z[b,o,k,r,a]  = sum_{(p,n)->q_r} dY[b,o,p,a] * w(p,a,k,n)         (its autograd transpose over inverse lists)
//   w(p,a,k,n) = relu(1 - |g(p,n) - A_a kappa_k|^2 / sigma)
//
// Scales: the feature operand by a power of two per CHANNEL ROW (from the row's largest magnitude, eap_absmax_rows_f32: the
// row's maximum lands in [2^14, 2^15)), the weights (in [0, 1]) by 2^13; both come off at the row end.  |x - h - l| <= 2^-23 |x|
// down to 2^-17 of the row's maximum (weights: down to 2^-15), an absolute 2^-40 of it below.
// Geometry as tools/experiments/kernels/so3_inter_lists3.hip (the 3 x bf16 attempt of round 3, vector-bound by its 11-instruction
// splits): workgroup = 8 waves = (row run, 64 channels, 16 anchors); wave = 2 anchors x 2 channel tiles; K = 16 entries per
// MFMA; LDS image of a stage [16 entries][64 channel rows][4 slots of 16 bytes] (64 KB, two stages); one workgroup per CU.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int CT = 2;         // channel tiles (MFMA M tiles) per wave
constexpr int CB = 32 * CT;   // channels per block
constexpr int NBK = 16;       // entries per LDS stage = K of one bf16 MFMA
constexpr int APW = 2;        // anchors per wave
constexpr int NWV = 8;
constexpr int TM = 64 * NWV;
constexpr int SL = 4;         // 16-byte slots per LDS row (= pieces of a 16-anchor group)
constexpr int GSZ = 4 * SL;   // anchors per workgroup
constexpr int PITCH = 4 * SL; // floats per LDS row
constexpr int NSTD = NBK * CB * SL / TM;   // DMA instructions per thread and stage (8): instruction u carries entries 2u, 2u+1
constexpr unsigned BUF_BYTES = NBK * CB * PITCH * 4;      // 64 KB
static_assert(2 * CB * SL == TM, "one DMA instruction per entry pair: thread t <-> (entry parity t >> 8, row (t >> 2) & 63, slot t & 3)");

__device__ inline unsigned lds_addr(const void *ptr) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)ptr;
}
__device__ inline void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
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

__device__ __forceinline__ void split_pair_h(float x0, float x1, unsigned &h, unsigned &l) {      // round to nearest, both planes
    const f16x2 hh = __builtin_convertvector((f32x2){x0, x1}, f16x2);
    const f16x2 ll = __builtin_convertvector((f32x2){x0 - (float)hh.x, x1 - (float)hh.y}, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}
// 2^(14 - e) for v in [2^e, 2^(e+1)); 1 for 0, inf, nan (as csrc/gemm_bf16x3.hip)
__device__ __forceinline__ float pow2_scale(float v) {
    const unsigned b = __float_as_uint(v) & 0x7fffffffu;
    const int e = (int)(b >> 23) - 127;
    if (b == 0u || e == 128) return 1.0f;
    const int se = max(-120, min(120, 14 - max(e, -126)));
    return __uint_as_float((unsigned)(se + 127) << 23);
}
constexpr float W_SCALE = 8192.0f;        // weights in [0, 1] -> [0, 2^13]
struct Planes { u32x4 h, l; };            // 8 values along K as two planes of 8 fp16

// LISTS = true : rows / off / cnt describe variable-length entry lists (backward);
// LISTS = false: row r of cloud b owns entries [ (b*R + r)*nn, +nn ) (forward: its neighbours).
// LAYOUT of the output: 0 = [b,c,k,row,a] (reference), 2 = transposed [row*na+a][c*ks+k]
template <bool LISTS, int LAYOUT>
__global__ __launch_bounds__(TM, 2) void so3_group_listsh_kernel(
    int C, int PF, int na, int fpitch, int ks, int R, int nn, int ent_stride, int AG, int RPB, float inv_sigma,
    const float *__restrict__ F, const int32_t *__restrict__ rows, const int32_t *__restrict__ off,
    const int32_t *__restrict__ cnt, const int32_t *__restrict__ ent_p, const float4 *__restrict__ ent_gx,
    const float *__restrict__ rk, const int32_t *__restrict__ nonident, float *__restrict__ out, const unsigned *__restrict__ row_abs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- block -> (row run, anchor group, channel slice, cloud), as csrc/so3_inter_lists2.hip ----
    const int nrun = (R + RPB - 1) / RPB;
    const int ny = gridDim.y, nsl = ny * gridDim.z, per_slice = nrun * AG;
    int qd = blockIdx.x, sl = blockIdx.y + ny * blockIdx.z;
    if ((nsl & 7) == 0) {
        const unsigned lin = blockIdx.x + (unsigned)per_slice * (blockIdx.y + (unsigned)ny * blockIdx.z);
        const unsigned j = lin >> 3;
        sl = (int)((lin & 7u) + 8u * (j / (unsigned)per_slice));
        qd = (int)(j % (unsigned)per_slice);
    } else {
        qd = xcd_point(blockIdx.x, per_slice);
    }
    const int run = qd / AG, ag = qd - run * AG;
    const int r_begin = run * RPB, rows_blk = min(RPB, R - r_begin);
    const int cy = sl % ny, bi = sl / ny, c0 = cy * CB;
    if (nonident != nullptr && __builtin_amdgcn_readfirstlane(nonident[bi]) != 0) return;   // permuted cloud: not ours

    const int t = threadIdx.x, lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lk = lane & 31, lh = lane >> 5;
    const int a0 = ag * GSZ, gcount = min(GSZ, na - a0);      // anchors [a0, a0 + gcount) of this block
    const int npg = gcount >> 2;                               // 16-byte pieces per feature row that exist
    const int al_beg = wave_u * APW;                           // this wave's anchors: half (wave_u & 1) of piece wave_u >> 1
    const bool active = al_beg < gcount;                       // wave-uniform

    float *s_f = reinterpret_cast<float *>(smem);                           // [2][NBK][CB][PITCH]
    float4 *s_g = reinterpret_cast<float4 *>(s_f + 2 * NBK * CB * PITCH);   // [3][NBK] ring
    int *s_p = reinterpret_cast<int *>(s_g + 3 * NBK);                      // [3][NBK] ring

    int n_ent, nchunk_row;
    size_t e0;
    if (LISTS) {
        const int q = rows[(size_t)bi * R + r_begin];
        n_ent = q >= 0 ? cnt[(size_t)bi * R + r_begin] : 0;
        e0 = (size_t)bi * ent_stride + (q >= 0 ? off[(size_t)bi * R + r_begin] : 0);
        nchunk_row = (n_ent + NBK - 1) / NBK;
    } else {
        n_ent = rows_blk * nn;
        e0 = ((size_t)bi * R + r_begin) * nn;
        nchunk_row = (nn + NBK - 1) / NBK;
    }
    const int nchunk = LISTS ? nchunk_row : rows_blk * nchunk_row;

    // ---- per-lane weight constants of this wave's two anchors (k = lane & 31) ----
    f32x2 kxp, kyp, kzp, kcp;
#pragma unroll
    for (int ai = 0; ai < APW; ++ai) {
        const int a = a0 + min(al_beg + ai, gcount - 1);
        const float *r3 = rk + ((size_t)a * ks + min(lk, ks - 1)) * 3;
        const float x = r3[0], y = r3[1], z = r3[2];
        kxp[ai] = 2.f * inv_sigma * x;
        kyp[ai] = 2.f * inv_sigma * y;
        kzp[ai] = 2.f * inv_sigma * z;
        kcp[ai] = lk < ks ? -inv_sigma * (x * x + y * y + z * z) : -1e30f;
    }
    // operand read: the wave's two anchors are 8 bytes of piece wave_u >> 1 of a channel row; tile 1 = rows + 32
    const float2 *fa_lane = reinterpret_cast<const float2 *>(s_f + (size_t)(8 * lh * CB + lk) * PITCH + 4 * (((wave_u >> 1) + (lk >> 2)) & 3) + 2 * (wave_u & 1));
    constexpr int ENT_F2 = CB * PITCH / 2;             // float2s between consecutive entries
    constexpr int TILE_F2 = 32 * PITCH / 2;            // float2s between the two channel tiles of an entry
    constexpr int BUF_F2 = NBK * CB * PITCH / 2;

    f32x16 acc[CT][APW];
    // operand scale of this lane's channel row per tile, and what takes both scales off the accumulators at the row end
    // (register r of a tile = channel (r & 3) + 8 (r >> 2) + 4 lh)
    float scl[CT], uns[CT][16];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        scl[ct] = pow2_scale(__uint_as_float(row_abs[(size_t)bi * C + min(c0 + 32 * ct + lk, C - 1)]));
#pragma unroll
        for (int r = 0; r < 16; ++r)
            uns[ct][r] = (1.0f / W_SCALE) / pow2_scale(__uint_as_float(row_abs[(size_t)bi * C + min(c0 + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lh, C - 1)]));
    }

    // ---- DMA: instruction u of a stage carries entries 2u and 2u+1; thread t <-> entry parity t >> 8, channel row
    //      (t >> 2) & 63, slot t & 3.  Pieces a smaller anchor group does not have are requested clamped (a duplicate of its
    //      last piece lands in a slot nobody reads): no branch inside the loop ----
    const float *fb = F + ((size_t)bi * C + c0) * PF * fpitch;   // fpitch: floats between consecutive feature rows (>= na)
    const unsigned lds_f = lds_addr(s_f);
    const unsigned row_bytes = (unsigned)fpitch * 4u;
    const int d_row = (t >> 2) & 63, d_piece = ((t & 3) - (d_row >> 2)) & 3, d_par = t >> 8;
    const unsigned dma_off = ((unsigned)(min(c0 + d_row, C - 1) - c0) * (unsigned)PF * (unsigned)fpitch + (unsigned)(a0 + 4 * min(d_piece, npg - 1))) * 4u;
    const unsigned lds_g = lds_addr(s_g), lds_p = lds_addr(s_p);
    auto issue_idx = [&](int j0, int slot) {
        if (wave_u == 0 && lane < NBK) {
            const size_t e = e0 + min(j0 + lane, max(n_ent - 1, 0));
            glds4(ent_p + e, __builtin_amdgcn_readfirstlane(lds_p + (unsigned)slot * NBK * 4u));
            glds16(ent_gx + e, __builtin_amdgcn_readfirstlane(lds_g + (unsigned)slot * NBK * 16u));
        }
    };
    unsigned src_off[NSTD];
    auto prep_rows = [&](int slot) {
#pragma unroll
        for (int u = 0; u < NSTD; ++u) {
            int pe = s_p[slot * NBK + 2 * u + d_par];
            if (!LISTS) pe = (unsigned)pe < (unsigned)PF ? pe : 0;     // shadow row: any valid row, weight 0
            src_off[u] = dma_off + __umul24((unsigned)pe, row_bytes);
        }
    };
    auto issue = [&](int u, int buf) {
        glds16s(fb, src_off[u], __builtin_amdgcn_readfirstlane(lds_f + (unsigned)buf * BUF_BYTES + (unsigned)(u * TM + wave_u * 64) * 16u));
    };

    if (nchunk > 0) {
        issue_idx(0, 0);
        issue_idx(NBK, 1);
        dma_wait();
        __syncthreads();
        prep_rows(0);
#pragma unroll
        for (int u = 0; u < NSTD; ++u) issue(u, 0);
        dma_wait();
    }
    __syncthreads();

    // per stage: lane e (mod 16) evaluates the per-entry term 1 - |g_e|^2/s of the stage's entry e, or a dead value for
    // entries past the end of the list and for the forward's shadow rows; every lane fetches its eight by bpermute
    auto chunk_bases = [&](int ch, int gslot) {
        const int e = lane & (NBK - 1);
        const float4 g = s_g[gslot * NBK + e];
        float b = 1.0f - inv_sigma * (g.x * g.x + g.y * g.y + g.z * g.z);
        bool dead = ch * NBK + e >= n_ent;
        if (!LISTS) dead = dead || (unsigned)s_p[gslot * NBK + e] >= (unsigned)PF;
        return __float_as_int(dead ? -1e30f : b);
    };

    // Row end: accumulators straight to global memory.  D[i = channel][j = kernel point] sits as col = lane&31,
    // row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const size_t o_ks = (size_t)R * na, o_cs = (size_t)ks * R * na;
    float *ob = out + (size_t)bi * C * o_cs + (size_t)c0 * o_cs + a0;
    const unsigned lane_off = (unsigned)((size_t)(4 * lh) * o_cs + (size_t)min(lk, ks - 1) * o_ks) + (unsigned)al_beg;
    float *obb = out + (size_t)bi * C * o_cs;
    auto store_row = [&](int row) {
        if (active && lk < ks) {
            if (LAYOUT == 2) {
                const size_t CK = (size_t)C * ks;
                float *rb = obb + ((size_t)row * na + a0 + al_beg) * CK + (size_t)c0 * ks;      // uniform
                const unsigned lo_b = (unsigned)((4 * lh) * ks + lk) * 4u;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (c0 + 32 * ct + 32 <= C || c0 + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lh < C) {
#pragma unroll
                            for (int ai = 0; ai < APW; ++ai)
                                *reinterpret_cast<float *>(reinterpret_cast<char *>(rb + (size_t)ai * CK + (size_t)(32 * ct + (r & 3) + 8 * (r >> 2)) * ks) + lo_b) = acc[ct][ai][r] * uns[ct][r];
                        }
            } else {
                float *rb = ob + (size_t)row * na;             // uniform
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (c0 + 32 * ct + 32 <= C || c0 + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lh < C)
                            *reinterpret_cast<float2 *>(rb + (size_t)(32 * ct + (r & 3) + 8 * (r >> 2)) * o_cs + lane_off) = make_float2(acc[ct][0][r] * uns[ct][r], acc[ct][1][r] * uns[ct][r]);
            }
        }
    };

    int g0 = 0, g1 = 1, g2 = 2;                           // ring slots of stages ch, ch+1, ch+2
    int ch_row = 0, row = r_begin;
    if (active) {
        for (int ch = 0; ch < nchunk; ++ch) {
            const int buf = ch & 1, nb = buf ^ 1;
            const float2 *fbuf = fa_lane + buf * BUF_F2;
            const int bases = chunk_bases(ch, g0);
            prep_rows(g1);
            issue_idx((ch + 2) * NBK, g2);
            // ---- B operand: the weights of this lane's kernel point for its 8 entries and the wave's two anchors ----
            f32x2 wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 g = s_g[g0 * NBK + 8 * lh + i];
                const float bk = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (8 * lh + i), bases));
                f32x2 x = __builtin_elementwise_fma((f32x2){g.x, g.x}, kxp, kcp + (f32x2){bk, bk});
                x = __builtin_elementwise_fma((f32x2){g.y, g.y}, kyp, x);
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp" : "=v"(wv[i]) : "v"((f32x2){g.z, g.w}), "v"(kzp), "v"(x));
            }
            Planes B[APW];
#pragma unroll
            for (int ai = 0; ai < APW; ++ai) {
                unsigned h[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split_pair_h(wv[2 * j][ai] * W_SCALE, wv[2 * j + 1][ai] * W_SCALE, h[j], l[j]);
                B[ai].h = (u32x4){h[0], h[1], h[2], h[3]};
                B[ai].l = (u32x4){l[0], l[1], l[2], l[3]};
            }
            issue(0, nb);
            issue(1, nb);
            // ---- per channel tile: A operand (8 entries of this lane's channel row, two anchors), then 12 MFMAs ----
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float2 fv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) fv[i] = fbuf[i * ENT_F2 + ct * TILE_F2];
                Planes A[APW];
#pragma unroll
                for (int ai = 0; ai < APW; ++ai) {
                    unsigned h[4], l[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        split_pair_h((ai ? fv[2 * j].y : fv[2 * j].x) * scl[ct], (ai ? fv[2 * j + 1].y : fv[2 * j + 1].x) * scl[ct], h[j], l[j]);
                    A[ai].h = (u32x4){h[0], h[1], h[2], h[3]};
                    A[ai].l = (u32x4){l[0], l[1], l[2], l[3]};
                }
                issue(2 + 3 * ct, nb);
                issue(3 + 3 * ct, nb);
                issue(4 + 3 * ct, nb);
#define L3_MFMA(AP, BP, CIN) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[ai].AP), __builtin_bit_cast(f16x8, B[ai].BP), CIN, 0, 0, 0)
                if (ch_row == 0) {        // block-uniform: the first stage of a row starts from C = 0 (accumulators are never zeroed)
                    const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ai = 0; ai < APW; ++ai) acc[ct][ai] = L3_MFMA(h, l, zc);
                } else {
#pragma unroll
                    for (int ai = 0; ai < APW; ++ai) acc[ct][ai] = L3_MFMA(h, l, acc[ct][ai]);
                }
#pragma unroll
                for (int ai = 0; ai < APW; ++ai) acc[ct][ai] = L3_MFMA(l, h, acc[ct][ai]);
#pragma unroll
                for (int ai = 0; ai < APW; ++ai) acc[ct][ai] = L3_MFMA(h, h, acc[ct][ai]);
#undef L3_MFMA
            }
            dma_wait();
            if (++ch_row == nchunk_row) {
                store_row(row);
                ch_row = 0;
                ++row;
            }
            __syncthreads();
            const int gt = g0; g0 = g1; g1 = g2; g2 = gt;
        }
        if (nchunk == 0) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int ai = 0; ai < APW; ++ai)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ct][ai][r] = 0.f;
            store_row(r_begin);
        }
    } else {
        for (int ch = 0; ch < nchunk; ++ch) {
            prep_rows(g1);
#pragma unroll
            for (int u = 0; u < NSTD; ++u) issue(u, (ch & 1) ^ 1);
            dma_wait();
            __syncthreads();
            const int gt = g0; g0 = g1; g1 = g2; g2 = gt;
        }
    }
}

constexpr size_t SHMEM = 2 * BUF_BYTES + 16 * 3 * NBK + 16 * NBK;
int g_h2 = 1;                       // eap_so3_group_lists_f16x2

template <bool LISTS>
int launch_h(int layout, int b, int C, int PF, int na, int fpitch, int ks, int R, int nn, int ent_stride, float sigma, const float *F,
            const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p, const float *ent_gx,
            const float *rk, const int32_t *nonident, float *out, hipStream_t s, const char *what) {
    if (fpitch < na || (fpitch & 3) != 0) return eap::bad_arg("so3_group_lists_h2: the feature row pitch must be a multiple of 4, at least the anchor count");
    if ((long long)CB * PF * fpitch * 4 >= (1ll << 32) || PF >= (1 << 24) || fpitch * 4 >= (1 << 24))
        return eap::bad_arg("so3_group_lists_h2: 64 feature rows of a cloud exceed the 32-bit request offsets");
    if (((long long)ks * R * na * 4 + 64ll * R * na + 64) * 4 >= (1ll << 31) || (long long)CB * ks * 4 >= (1ll << 31))
        return eap::bad_arg("so3_group_lists_h2: output rows too far apart for 32-bit store offsets");
    auto kern = layout == 2 ? so3_group_listsh_kernel<LISTS, LISTS ? 0 : 2> : so3_group_listsh_kernel<LISTS, 0>;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SHMEM), what);
    if (e) return e;
    // the operand's largest magnitude per channel row (stream-ordered scratch, released after the launch)
    void *row_abs = nullptr;
    if ((e = eap::hip_fail(hipMallocAsync(&row_abs, (size_t)b * C * 4, s), what))) return e;
    struct Release { void *p; hipStream_t s; ~Release() { (void)hipFreeAsync(p, s); } } release{row_abs, s};
    if ((e = eap_absmax_rows_f32(F, 1, b * C, PF * fpitch, (int64_t)PF * fpitch, 0, reinterpret_cast<int32_t *>(row_abs), s))) return e;
    const int AG = (na + GSZ - 1) / GSZ;
    const int RPB = LISTS ? 1 : ((nn % NBK) == 0 ? 8 : 1);
    dim3 grid((R + RPB - 1) / RPB * AG, (C + CB - 1) / CB, b);
    hipLaunchKernelGGL(kern, grid, dim3(TM), SHMEM, s, C, PF, na, fpitch, ks, R, nn, ent_stride, AG, RPB, 1.0f / sigma, F,
                       rows, off, cnt, ent_p, reinterpret_cast<const float4 *>(ent_gx), rk, nonident, out, reinterpret_cast<const unsigned *>(row_abs));
    eap::set_kernel(LISTS ? "so3_group_listsh_kernel<true, 0>" : layout == 2 ? "so3_group_listsh_kernel<false, 2>" : "so3_group_listsh_kernel<false, 0>");
    return eap::check_launch(what);
}

}  // namespace

namespace eap {

int group_listsh_fwd(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                     const int32_t *idx, const float *gx, const float *rk, const int32_t *nonident, int layout, float *out,
                     hipStream_t s) {
    return launch_h<false>(layout, b, c, n, na, na, ks, p, nn, 0, sigma, feats, nullptr, nullptr, nullptr, idx, gx, rk, nonident, out, s,
                          "so3_inter_group_fwd (lists, 2 x fp16 planes)");
}

int group_listsh_inv(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap, float sigma, const float *gy,
                     const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p,
                     const float *ent_gx, const float *rk, float *z, hipStream_t s) {
    return launch_h<true>(0, b, o, p, na, gy_pitch, ks, rcap, nn, p * nn, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, nullptr, z, s,
                         "so3_inter_group_inv (lists, 2 x fp16 planes)");
}

}  // namespace eap
