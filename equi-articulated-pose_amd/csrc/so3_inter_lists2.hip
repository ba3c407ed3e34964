// csrc/so3_inter_lists2.hip -- SO(3) grouping over entry lists on the matrix cores, second generation: the matrix
// waves never touch global memory.
//
// Same operation and interface as csrc/so3_inter_lists.hip (forward X[b,c,k,p,a] = sum_n F[b,c,idx[p,n],a] w(p,a,k,n),
// vgtk/vgtk/so3conv/functional.py:L1112-1261; backward Z over inverse lists), same MFMA mapping (M = 32 channels,
// N = kernel points padded to 32, K = entries, v_mfma_f32_32x32x2_f32, weights generated in registers), same LDS
// image of a chunk ([8 entries][32 channels][anchor pieces], rotated by the channel row).
//
// What the counters said about the first generation (profiles/r02_a_*): matrix pipe busy 60 % (backward) / 49 %
// (forward), waves parked at a barrier or a DMA drain 34 % of their cycles -- 2 x 8 waves per CU, every one of them
// computes AND issues global -> LDS DMA, and a wave that sits in the issue of a DMA instruction (60-180 cycles) feeds
// no MFMA.  The GEMM work of this round (csrc/gemm_dma_f32.hip) showed the way out: give the matrix pipe of a SIMD to
// ONE wave and keep everything that can stall away from it.  Here a workgroup is 8 waves, two per SIMD:
//   * waves 0-3, the matrix waves, one per SIMD: 8 anchors each (128 accumulator registers), they only read LDS,
//     evaluate the kernel weights on the VALU and issue MFMAs -- no vector-memory instruction inside the loop;
//   * waves 4-7, the loaders: all global -> LDS DMA of the feature rows (and of the entry ring), three chunks ahead
//     through a four-stage ring, counted waits;
//   * one raw s_barrier per 8-entry chunk.  The barrier of chunk c certifies chunk c + 1, so the operands of the
//     next chunk's first MFMA steps are read before the current chunk's last MFMAs are issued.
// 32 + 28 anchors in two workgroups as before (one workgroup per CU now: 131 KB of LDS).
#include "common.h"

#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CB = 32;        // channels per block (one MFMA M tile)
constexpr int NBK = 8;        // entries per chunk (4 MFMA k-steps)
constexpr int APW = 8;        // anchors per matrix wave
constexpr int NMW = 4, NLW = 4, TM = 64 * (NMW + NLW);
constexpr int NST = 4;        // chunk stages in LDS
constexpr int RING = 8;       // entry-ring slots (chunks)
constexpr int NPL = 8;        // DMA pieces per loader thread and chunk: NBK * CB * (pieces <= 8) / (64 * NLW)

__device__ inline unsigned lds_addr(const void *ptr) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)ptr;
}
__device__ inline void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// counted wait with a run-time (wave-uniform) count: s_waitcnt takes an immediate
__device__ inline void wait_vmcnt(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    }
}
__device__ inline void glds4(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// LISTS = true : rows / off / cnt describe variable-length entry lists (backward);
// LISTS = false: row r of cloud b owns entries [ (b*R + r)*nn, +nn ) (forward: its neighbours).
// LAYOUT of the output: 0 = [b,c,k,row,a] (reference), 1 = blocked by anchor quads, 2 = transposed [row*na+a][c*ks+k]
template <bool LISTS, int LAYOUT>
__global__ __launch_bounds__(TM, 2) void so3_group_lists2_kernel(
    int C, int PF, int na, int fpitch, int ks, int R, int nn, int ent_stride, int AG, int gsz, int RPB, float inv_sigma,
    const float *__restrict__ F, const int32_t *__restrict__ rows, const int32_t *__restrict__ off,
    const int32_t *__restrict__ cnt, const int32_t *__restrict__ ent_p, const float4 *__restrict__ ent_gx,
    const float *__restrict__ rk, const int32_t *__restrict__ nonident, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- block -> (row run, anchor group, channel slice, cloud); XCDs get whole (slice, cloud) pairs ----
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
    const bool loader = wave_u >= NMW;                         // wave-uniform
    const int lk = lane & 31, lh = lane >> 5;
    const int a0 = ag * gsz, gcount = min(gsz, na - a0);       // anchors [a0, a0 + gcount) of this block
    const int npg = gcount >> 2, pitch = gcount;               // 16-byte pieces per row, LDS row pitch (floats)
    const bool rotate = (npg & 1) == 0;

    float *s_f = reinterpret_cast<float *>(smem);                              // [NST][NBK][CB][pitch]
    float4 *s_g = reinterpret_cast<float4 *>(s_f + NST * NBK * CB * pitch);    // [RING][NBK]
    int *s_p = reinterpret_cast<int *>(s_g + RING * NBK);                      // [RING][NBK]

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
    const unsigned buf_floats = (unsigned)(NBK * CB) * (unsigned)pitch;

    if (loader) {
        // =========================================== loader waves ===========================================
        if (nchunk == 0) return;
        const int lw = wave_u - NMW, tl = t - 64 * NMW;          // loader wave 0..3, loader thread 0..255
        const float *fb = F + (size_t)bi * C * PF * fpitch;
        const int total4 = NBK * CB * npg;                        // a multiple of 64
        const unsigned lds_f = lds_addr(s_f), lds_g = lds_addr(s_g), lds_p = lds_addr(s_p);
        unsigned dma_off[NPL], nl_pack = 0;
        int npieces = 0;                                          // DMA instructions of this wave per chunk (uniform)
#pragma unroll
        for (int u = 0; u < NPL; ++u) {
            const int f = min(u * 64 * NLW + tl, total4 - 1);
            const int row = f / npg, slot = f - row * npg;
            const int nl = row / CB, cl = row - nl * CB;
            const int piece = rotate ? (slot + npg - cl % npg) % npg : slot;
            dma_off[u] = (unsigned)min(c0 + cl, C - 1) * (unsigned)PF * (unsigned)fpitch + (unsigned)(a0 + 4 * piece);
            nl_pack |= (unsigned)nl << (3 * u);
            if (u * 64 * NLW + lw * 64 < total4) ++npieces;
        }
        auto issue_idx = [&](int ch) {                            // entries of chunk ch -> ring slot ch % RING (wave 4 only)
            if (lw == 0 && lane < NBK) {
                const size_t e = e0 + min(ch * NBK + lane, max(n_ent - 1, 0));
                const unsigned slot = (unsigned)(ch & (RING - 1));
                glds4(ent_p + e, __builtin_amdgcn_readfirstlane(lds_p + slot * NBK * 4u));
                glds16(ent_gx + e, __builtin_amdgcn_readfirstlane(lds_g + slot * NBK * 16u));
            }
        };
        auto issue_rows = [&](int ch) {                           // feature rows of chunk ch -> stage ch % NST
            const int rs = (ch & (RING - 1)) * NBK;
            const unsigned sb = lds_f + (unsigned)(ch & (NST - 1)) * buf_floats * 4u;
#pragma unroll
            for (int u = 0; u < NPL; ++u) {
                const int f0 = u * 64 * NLW + lw * 64;            // wave-uniform
                if (f0 < total4) {
                    int pe = s_p[rs + ((nl_pack >> (3 * u)) & 7)];
                    if (!LISTS) pe = (unsigned)pe < (unsigned)PF ? pe : 0;     // shadow row: any valid row, weight 0
                    glds16(fb + dma_off[u] + __umul24((unsigned)pe, (unsigned)fpitch), __builtin_amdgcn_readfirstlane(sb + (unsigned)f0 * 16u));
                }
            }
        };
        // entry ring six chunks ahead, feature rows three chunks ahead (past the end the last entry repeats: harmless)
#pragma unroll
        for (int ch = 0; ch < 6; ++ch) issue_idx(ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // P1: entries of chunks 0..5 visible to every loader
        asm volatile("" ::: "memory");
        issue_rows(0);
        issue_rows(1);
        issue_rows(2);
        const int per_trip = __builtin_amdgcn_readfirstlane(npieces + (lw == 0 ? 2 : 0));
        for (int ch = 0; ch < nchunk; ++ch) {
            // my rows of chunk ch + 1 have landed (only what this wave issued in the previous trip may be in flight):
            // one trip = [2 entry-ring DMAs (wave 4 only)] + npieces row DMAs
            wait_vmcnt(ch < 2 ? 0 : per_trip);
            __builtin_amdgcn_s_barrier();                        // chunk ch (and ch + 1) certified; chunk ch - 1 released
            asm volatile("" ::: "memory");
            issue_idx(ch + 6);
            issue_rows(ch + 3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing may still be writing LDS when the block ends
        return;
    }

    // ============================================= matrix waves =============================================
    const int al_beg = wave_u * APW;                              // first local anchor of this wave
    const bool active = al_beg < gcount;                          // wave-uniform
    const int nanch = min(APW, max(gcount - al_beg, 0));          // anchors this wave owns (8, or 4 in the 28-anchor group)

    float kx[APW], ky[APW], kz[APW], kc[APW];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai) {
        const int a = a0 + min(al_beg + ai, gcount - 1);
        const float *r3 = rk + ((size_t)a * ks + min(lk, ks - 1)) * 3;
        const float x = r3[0], y = r3[1], z = r3[2];
        kx[ai] = 2.f * inv_sigma * x; ky[ai] = 2.f * inv_sigma * y; kz[ai] = 2.f * inv_sigma * z;
        kc[ai] = lk < ks ? -inv_sigma * (x * x + y * y + z * z) : -1e30f;
    }
    // operand read offsets of the two anchor quads (rotation by the channel row)
    int roff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int al = min(al_beg + 4 * j, gcount - 4);
        const int piece = al >> 2, slot = rotate ? (piece + lk) % npg : piece;
        roff[j] = lk * pitch + 4 * slot;
    }

    f32x16 acc[APW];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ai][r] = 0.f;

    // operands of MFMA k-step `q` (two entries) counted over the whole block: chunk q / 4, step q % 4
    struct Ops { float4 f0, f1, g; int pe; };
    auto gather = [&](int q, Ops &o) __attribute__((always_inline)) {
        const int ch = q >> 2, s = q & 3;
        const float *base = s_f + (size_t)(ch & (NST - 1)) * buf_floats + (size_t)(2 * s + lh) * CB * pitch;
        o.f0 = *reinterpret_cast<const float4 *>(base + roff[0]);
        o.f1 = *reinterpret_cast<const float4 *>(base + roff[1]);
        const int ei = (ch & (RING - 1)) * NBK + 2 * s + lh;
        o.g = s_g[ei];
        o.pe = LISTS ? 0 : s_p[ei];
    };
    auto step = [&](int q, const Ops &o) __attribute__((always_inline)) {
        const int je = 2 * q + lh;                               // entry index within the block
        const float4 g = o.g;
        float base = 1.0f - inv_sigma * (g.x * g.x + g.y * g.y + g.z * g.z);
        if (je >= n_ent || (!LISTS && (unsigned)o.pe >= (unsigned)PF)) base = -1e30f;    // dead entry: weight 0
        float wv[APW];
#pragma unroll
        for (int ai = 0; ai < APW; ++ai) {
            float x = fmaf(g.x, kx[ai], kc[ai]);
            x = fmaf(g.y, ky[ai], x);
            x = fmaf(g.z, kz[ai], x);
            wv[ai] = fmaxf(x + base, 0.0f);
        }
        const float fa[APW] = {o.f0.x, o.f0.y, o.f0.z, o.f0.w, o.f1.x, o.f1.y, o.f1.z, o.f1.w};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ai = 0; ai < APW; ++ai) acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ai], wv[ai], acc[ai], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- row end (see csrc/so3_inter_lists.hip for the layouts); 8 anchors = two anchor quads per register ----
    const size_t o_ks = (size_t)R * na, o_cs = (size_t)ks * R * na;
    float *ob = out + (size_t)bi * C * o_cs + (size_t)c0 * o_cs + a0;
    const unsigned lane_off = (unsigned)((size_t)(4 * lh) * o_cs + (size_t)min(lk, ks - 1) * o_ks) + (unsigned)al_beg;
    const bool full_c = c0 + CB <= C;
    const bool quad2 = al_beg + 4 < gcount;                       // wave-uniform: the second anchor quad exists
    const int npq = na >> 2, aq0 = (a0 + al_beg) >> 2;
    float *obb = out + (size_t)bi * C * o_cs;
    const unsigned lane_off_b = (unsigned)((4 * lh) * ks + min(lk, ks - 1)) * 4u;
    auto store_row = [&](int row) {
        if (active && lk < ks) {
            if (LAYOUT == 2) {
                const size_t CK = (size_t)C * ks;
                float *rb = obb + ((size_t)row * na + a0 + al_beg) * CK + (size_t)c0 * ks;      // uniform
                const unsigned lo = (unsigned)((4 * lh) * ks + lk);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (full_c || c0 + (r & 3) + 8 * (r >> 2) + 4 * lh < C) {
#pragma unroll
                        for (int ai = 0; ai < APW; ++ai)
                            if (ai < 4 || quad2) rb[(size_t)ai * CK + (size_t)((r & 3) + 8 * (r >> 2)) * ks + lo] = acc[ai][r];
                    }
            } else if (LAYOUT == 1) {
                float *rb = obb + (((size_t)row * npq + aq0) * C + c0) * ks * 4;     // uniform
                const size_t q2 = (size_t)C * ks * 4;                               // next anchor quad
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (full_c || c0 + (r & 3) + 8 * (r >> 2) + 4 * lh < C) {
                        float *d = rb + (size_t)((r & 3) + 8 * (r >> 2)) * ks * 4 + lane_off_b;
                        *reinterpret_cast<float4 *>(d) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                        if (quad2) *reinterpret_cast<float4 *>(d + q2) = make_float4(acc[4][r], acc[5][r], acc[6][r], acc[7][r]);
                    }
            } else {
                float *rb = ob + (size_t)row * na;             // uniform
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (full_c || c0 + (r & 3) + 8 * (r >> 2) + 4 * lh < C) {
                        float *d = rb + (size_t)((r & 3) + 8 * (r >> 2)) * o_cs + lane_off;
                        *reinterpret_cast<float4 *>(d) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                        if (quad2) *reinterpret_cast<float4 *>(d + 4) = make_float4(acc[4][r], acc[5][r], acc[6][r], acc[7][r]);
                    }
            }
        }
#pragma unroll
        for (int ai = 0; ai < APW; ++ai)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ai][r] = 0.f;
    };
    (void)nanch;

    if (nchunk == 0) { store_row(r_begin); return; }              // unreferenced row: zeros

    __builtin_amdgcn_s_barrier();                                // P1 (the loaders' entry ring)
    asm volatile("" ::: "memory");
    int ch_row = 0, row = r_begin;
    Ops oa, ob_;
    __builtin_amdgcn_s_barrier();                                // chunk 0 (and 1) certified
    asm volatile("" ::: "memory");
    if (active) { gather(0, oa); gather(1, ob_); }
    for (int ch = 0; ch < nchunk; ++ch) {
        if (ch > 0) {
            __builtin_amdgcn_s_barrier();                        // chunk ch + 1 certified, chunk ch - 1 released
            asm volatile("" ::: "memory");
        }
        if (active) {
            const int q = 4 * ch;
            // operands run two k-steps ahead; steps 2 and 3 already read the next chunk (certified by this barrier)
            step(q, oa);
            gather(q + 2, oa);
            step(q + 1, ob_);
            gather(q + 3, ob_);
            step(q + 2, oa);
            gather(q + 4, oa);
            step(q + 3, ob_);
            gather(q + 5, ob_);
        }
        if (++ch_row == nchunk_row) {                             // block-uniform
            store_row(row);
            ch_row = 0;
            ++row;
        }
    }
}

struct Geometry { int AG, gsz; size_t shmem; };

bool geometry(int na, int ks, Geometry &g) {
    if (na <= 0 || (na & 3) != 0 || na > 64 || ks <= 0 || ks > 32) return false;
    g.AG = na > 32 ? 2 : 1;
    g.gsz = g.AG == 1 ? na : ((na / 2 + 3) & ~3);
    g.shmem = sizeof(float) * NST * NBK * CB * g.gsz + 16 * RING * NBK + 4 * RING * NBK;
    return g.shmem <= 160 * 1024;
}

template <bool LISTS>
int launch(int blocked, int b, int C, int PF, int na, int fpitch, int ks, int R, int nn, int ent_stride, float sigma, const float *F,
           const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p, const float *ent_gx,
           const float *rk, const int32_t *nonident, float *out, hipStream_t s, const char *what) {
    Geometry g;
    if (!geometry(na, ks, g)) return eap::bad_arg("so3_group_lists2: unsupported anchor / kernel-point count");
    if (fpitch < na || (fpitch & 3) != 0) return eap::bad_arg("so3_group_lists2: the feature row pitch must be a multiple of 4, at least the anchor count");
    if ((long long)C * PF * fpitch >= (1ll << 31)) return eap::bad_arg("so3_group_lists2: one cloud's features exceed 2^31 elements");
    if (((long long)ks * R * na * 4 + 32ll * R * na + 64) * 4 >= (1ll << 31)) return eap::bad_arg("so3_group_lists2: output rows too far apart for 32-bit store offsets");
    auto kern = blocked == 2 ? so3_group_lists2_kernel<LISTS, LISTS ? 0 : 2> : blocked == 1 ? so3_group_lists2_kernel<LISTS, LISTS ? 0 : 1>
                                                                                           : so3_group_lists2_kernel<LISTS, 0>;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.shmem), what);
    if (e) return e;
    const int RPB = LISTS ? 1 : ((nn % NBK) == 0 ? 8 : 1);
    dim3 grid((R + RPB - 1) / RPB * g.AG, (C + CB - 1) / CB, b);
    hipLaunchKernelGGL(kern, grid, dim3(TM), g.shmem, s, C, PF, na, fpitch, ks, R, nn, ent_stride, g.AG, g.gsz, RPB, 1.0f / sigma, F,
                       rows, off, cnt, ent_p, reinterpret_cast<const float4 *>(ent_gx), rk, nonident, out);
    return eap::check_launch(what);
}

}  // namespace

namespace eap {

// EAP_LISTS_V2=0 keeps the first-generation kernel (csrc/so3_inter_lists.hip) for A/B runs
bool group_lists2_enabled(int na, int ks) {
    static const int on = getenv("EAP_LISTS_V2") ? atoi(getenv("EAP_LISTS_V2")) : 1;
    Geometry g;
    return on && geometry(na, ks, g);
}

int group_lists2_fwd(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                     const int32_t *idx, const float *gx, const float *rk, const int32_t *nonident, int blocked, float *out,
                     hipStream_t s) {
    return launch<false>(blocked, b, c, n, na, na, ks, p, nn, 0, sigma, feats, nullptr, nullptr, nullptr, idx, gx, rk, nonident, out, s,
                         "so3_inter_group_fwd (lists2)");
}

int group_lists2_inv(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap, float sigma, const float *gy,
                     const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p,
                     const float *ent_gx, const float *rk, float *z, hipStream_t s) {
    return launch<true>(0, b, o, p, na, gy_pitch, ks, rcap, nn, p * nn, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, nullptr, z, s,
                        "so3_inter_group_inv (lists2)");
}

}  // namespace eap
