// csrc/so3_inter_inv.hip -- feature gradient of the SO(3) inter convolution by re-association,
// on the matrix cores.
//
// Forward:  Y[o,p,a] = sum_{c,k} W[o,(c,k)] * sum_n F[c,q_n,perm_n(a)] * w(p,a,k,n)
//           (vgtk/vgtk/so3conv/functional.py:L1221-1261 + modules.py:L48-55).
// The textbook backward first forms dX = W^T dY ([C*K, P*A], as large as X) and then scatters it
// through the transposed grouping.  Re-associated:
//           dF[c,q,a'] = sum_{o,k} W[o,(c,k)] * Z[o,k,q,a']
//           Z[o,k,q,a'] = sum_{(p,n): idx[p,n]=q} dY[o,p,a] * w(p,a,k,n),   a = perm_n^{-1}(a')
// Z is the SAME operation as the forward grouping -- features = dY (O channels), "neighbours" of
// row q = the (point, slot) pairs that reference it (inverse neighbour lists, sorted on the host
// side of the C ABI by a stable sort) -- and it only has as many rows as there are REFERENCED
// support points.  With the reference's first-nsample-in-index-order ball query and the large
// radii of the deeper layers that is ~100-300 rows instead of 4096, so Z is 10-30x smaller than
// dX, the [C*K x O] x [O x P*A] GEMM and the whole scatter (atomics or slabs) disappear, and
// what is left is one small GEMM  dF = W2[C, O*K] x Z[O*K, R*A].
//
// Kernel = csrc/so3_inter_mfma.hip with variable-length entry lists: A operand = dY rows staged
// through LDS, B operand = kernel weights generated in registers, accumulators of all anchors in
// VGPRs, 8 waves (2 per SIMD).  With an anchor permutation the weight's anchor changes per entry,
// so the rotated kernel points come from an LDS table instead of registers.
#include "common.h"
#include <algorithm>

#ifdef EAP_INV_TRACE
__device__ unsigned long long eap_inv_trace[8 * 16];
#define TR(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tr[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define TR(i)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CB = 32;        // dY channels per block (one MFMA M tile)
constexpr int NBK = 8;        // entries per LDS stage (4 MFMA k-steps)
constexpr int FPMAX = 68;     // LDS pitch of one staged row: 60 floats for <= 60 anchors, 68 for 61..64 (16-byte
                              // aligned, never a multiple of 32 banks)
constexpr int APW = 8;        // max anchors per wave
constexpr int NWV = 8;
constexpr int TM = 64 * NWV;

// LDS byte address of a __shared__ object (what M0 takes for the LDS-DMA loads)
__device__ inline unsigned lds_addr(const void *ptr) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)ptr;
}

// One wave-wide 16-byte-per-lane global -> LDS DMA: lane l's 16 bytes land at lds_dst + 16*l.
// Inline asm on purpose: hipcc tracks the builtin as an LDS store and drains vmcnt before every
// later ds_read, which would serialise the next chunk's rows behind this chunk's operand reads;
// an asm load is invisible to its bookkeeping and the kernel waits for it explicitly (dma_wait)
// before the chunk barrier.  M0 is saved and restored in the same statement.
__device__ inline void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ inline void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// DMA = true: dY rows go global -> LDS directly (row pitch = na, no padding possible), issued
// between the MFMA steps of the previous chunk; DMA = false: register-staged rows with a padded
// pitch (anchor counts whose pitch would be a multiple of 32 banks).
// FWD = true: the same kernel as the FORWARD grouping of clouds with per-neighbour anchor permutations (round 3; it
// replaces csrc/so3_inter_mfma.hip's register-staged kernel there): row ri = query point ri, its entries = its nn
// neighbours (ent_p = idx, shadow indices >= p carry weight 0), the table is `mult`, the offset vectors are NOT rotated
// (the weight of output anchor a uses a itself), clouds whose flag says "all identity" are left to the entry-list kernel,
// and the output is X in the reference layout (blocked = 0) or transposed [row*na + a][c*ks + k] (blocked = 2).
// COSET = true (HAS_MULT, backward): the anchor axis of the operand is in COSET-MAJOR order -- in LDS and, with DMA, already in
// global memory (the caller re-orders gy once, eap_anchor_reorder_f32: a 1.6 ms pass for a 4 GB gradient) --
// (`order`: blocks of 4 = left cosets a.H of a Klein four-group H of the anchor group, vgtk/so3conv/functional.py
// _coset_tables).  The permutation of an entry is a left multiplication: it moves whole blocks and XORs the position inside,
// `code[r][block] = sigma | x << 4`.  A wave owns two blocks; the four permuted anchors of a block are ONE aligned 16-byte LDS
// read of block sigma + 8 selects, instead of 4 byte-table lookups + 4 dword reads at conflicting banks.
template <bool HAS_MULT, bool DMA, bool FWD = false, bool COSET = false>
__global__ __launch_bounds__(TM, 2) void so3_inter_group_inv_kernel(
    int o, int p, int nn, int na, int ks, int rcap, float inv_sigma, int identity_anchor,
    const float *__restrict__ gy,
    const int32_t *__restrict__ rows, const int32_t *__restrict__ off, const int32_t *__restrict__ cnt,
    const int32_t *__restrict__ ent_p, const float4 *__restrict__ ent_gx, const float *__restrict__ rk,
    const uint8_t *__restrict__ multinv, const float *__restrict__ anchors, float *__restrict__ out,
    int blocked, const int32_t *__restrict__ nonident, const uint8_t *__restrict__ order = nullptr) {
    static_assert(!COSET || (HAS_MULT && !FWD), "coset-major operand: permuted backward");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int FP = DMA ? na : (na <= 60 ? 60 : FPMAX), FP_ = FP;
    float *s_f = reinterpret_cast<float *>(smem);                           // [2][NBK][CB][FP]
    float4 *s_g = reinterpret_cast<float4 *>(s_f + 2 * NBK * CB * FP_);       // [3][NBK] (ring; [2][NBK] used without DMA)
    int *s_p = reinterpret_cast<int *>(s_g + 3 * NBK);                      // [3][NBK] entry -> query point (DMA), 16-byte multiple
    float4 *s_A = reinterpret_cast<float4 *>(s_p + 4 * NBK);                // [na][3] rows of the anchor rotations (HAS_MULT)
    uint8_t *s_mult = reinterpret_cast<uint8_t *>(s_A + (HAS_MULT ? 3 * na : 0));

    // Block -> (row, channel slice, cloud).  Every entry list walks the query points in ascending
    // order, so blocks that work on the SAME channel slice of the SAME cloud at the same time read
    // the same window of dY.  Workgroups go to the 8 XCDs round-robin by linear id: hand each XCD
    // whole slices (its 32 CUs then share one window in their own L2) instead of one row in eight
    // of every slice.  Rows arrive sorted by entry count, so neighbours advance at the same pace.
    int ri = blockIdx.x, cy = blockIdx.y, bi = blockIdx.z;
    {
        const int ny = gridDim.y, nsl = ny * gridDim.z;
        if ((nsl & 7) == 0) {
            const unsigned lin = blockIdx.x + (unsigned)rcap * (blockIdx.y + (unsigned)ny * blockIdx.z);
            const unsigned j = lin >> 3, sl = (lin & 7u) + 8u * (j / (unsigned)rcap);
            ri = (int)(j % (unsigned)rcap);
            cy = (int)(sl % (unsigned)ny);
            bi = (int)(sl / (unsigned)ny);
        }
    }
    const int c0 = cy * CB;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (FWD && nonident != nullptr && __builtin_amdgcn_readfirstlane(nonident[bi]) == 0) return;   // identity cloud: the entry-list kernel's
    const int q = FWD ? ri : rows[(size_t)bi * rcap + ri];
    const int n_ent = FWD ? nn : (q >= 0 ? cnt[(size_t)bi * rcap + ri] : 0);
    const size_t e0 = FWD ? ((size_t)bi * rcap + ri) * nn : (size_t)bi * p * nn + (q >= 0 ? off[(size_t)bi * rcap + ri] : 0);

    // COSET: s_mult holds code [na][16] (the caller passes that table as `multinv`), then pos [64] = position of every anchor in
    // the coset-major order
    uint8_t *s_pos = s_mult + na * 16;
    if (HAS_MULT) {
        const int words = COSET ? (na * 16) >> 2 : (na * na) >> 2;
        for (int i = t; i < words; i += TM)
            reinterpret_cast<uint32_t *>(s_mult)[i] = reinterpret_cast<const uint32_t *>(multinv)[i];
        if (COSET && t < na) s_pos[order[t]] = (uint8_t)t;
        if (!FWD) for (int i = t; i < 3 * na; i += TM) s_A[i] = make_float4(anchors[3 * i], anchors[3 * i + 1], anchors[3 * i + 2], 0.f);
        __syncthreads();
    }
    // Anchor permutation.  An entry with relative-rotation anchor r pairs the accumulator's anchor a' with the dY
    // column a = multinv[r][a'], and its weight belongs to anchor a: w = relu(1 - |g - A_a kappa|^2 / s).  The
    // anchors are a group and mult[r][a] = a' means A_a' = A_r A_a, so |g - A_a kappa| = |A_r g - A_a' kappa|:
    // with the entry's offset vector rotated ONCE by A_r when it enters the LDS ring, the weights are those of the
    // wave's OWN anchors -- per-lane register constants, as without permutation (round 1 looked the constants of
    // every (entry, anchor) pair up in an LDS table: a 16-byte LDS read and an address computation per weight).
    auto rotate_entry = [&](float4 g) {
        if (HAS_MULT && !FWD) {
            const int r = __float_as_int(g.w);
            if (r != identity_anchor && (unsigned)r < (unsigned)na && g.x < 1e17f) {
                const float4 r0 = s_A[3 * r], r1 = s_A[3 * r + 1], r2 = s_A[3 * r + 2];
                g = make_float4(r0.x * g.x + r0.y * g.y + r0.z * g.z, r1.x * g.x + r1.y * g.y + r1.z * g.z,
                                r2.x * g.x + r2.y * g.y + r2.z * g.z, g.w);
            }
        }
        return g;
    };

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // 8 waves, contiguous anchor ranges that all start at an EVEN anchor (8,8,8,8,8,8,6,6 at na = 60)
    // so that a pair of anchors is one aligned 8-byte LDS read; waves w and w+4 share a SIMD
    const int per = COSET ? 8 : min(APW, (((na + NWV - 1) / NWV) + 1) & ~1);
    const int a_beg = min(wave_u * per, na);                          // COSET: first POSITION in the coset-major order
    const int a_cnt = max(0, min(per, na - a_beg));
    // memory index of the wave's ai-th anchor (COSET: two blocks of the order; otherwise a contiguous range)
    int am[APW];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai)
        am[ai] = COSET ? __builtin_amdgcn_readfirstlane((int)order[min(a_beg + ai, na - 1)]) : min(a_beg + ai, na - 1);
    const int lk = lane & 31, lh = lane >> 5;
    const int lkc = min(lk, ks - 1);
    // kernel weight  w = relu(1 - |g - k|^2 / sigma) = relu(base_e + kc + g . k'),
    //   base_e = 1 - |g|^2/sigma (once per entry),  k' = 2k/sigma,  kc = -|k|^2/sigma (constants):
    // 3 FMAs + add + max per weight instead of 9 operations; unused kernel-point columns carry
    // kc = -1e30 so their weight is 0
    // evaluated with as few VALU instructions as it takes (they cost fp32-MFMA time on this part, csrc/so3_inter_lists.hip):
    // two anchors per packed instruction, the relu as the clamp modifier of the last FMA
    f32x2 kxp[APW / 2], kyp[APW / 2], kzp[APW / 2], kcp[APW / 2];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai) {
        const float *r3 = rk + ((size_t)am[ai] * ks + lkc) * 3;
        const float x = r3[0], y = r3[1], z = r3[2];
        kxp[ai >> 1][ai & 1] = 2.f * inv_sigma * x; kyp[ai >> 1][ai & 1] = 2.f * inv_sigma * y; kzp[ai >> 1][ai & 1] = 2.f * inv_sigma * z;
        kcp[ai >> 1][ai & 1] = lk < ks ? -inv_sigma * (x * x + y * y + z * z) : -1e30f;
    }

    f32x16 acc[APW];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ai][r] = 0.f;

    const int piece = t & 15, rgrp = t >> 4;
    const int npiece = na >> 2;                          // na % 4 == 0 (checked by the launcher)
    const int pc = min(piece, npiece - 1);
    const float *fb = gy + (size_t)bi * o * p * na;
    constexpr int NST = NBK * CB * 16 / TM;
    float4 stage[NST];
    int stage_p[NST], next_p[NST];
    float4 gtmp = make_float4(1e18f, 1e18f, 1e18f, 0.f);
    // entry -> query point indices are requested ONE CHUNK AHEAD of the feature rows that depend
    // on them, so a stage never waits for two dependent memory round trips
    auto fetch_index = [&](int j0) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int nl = (u * (TM / 16) + rgrp) / CB;
            next_p[u] = ent_p[e0 + min(j0 + nl, max(n_ent - 1, 0))];
            if (j0 + nl >= n_ent) next_p[u] = -1;
            if (FWD && (unsigned)next_p[u] >= (unsigned)p) next_p[u] = -1;       // shadow neighbour: zero features
        }
    };
    unsigned row_off[NST];                            // 32-bit element offsets, channel part precomputed
#pragma unroll
    for (int u = 0; u < NST; ++u) {
        const int cl = (u * (TM / 16) + rgrp) % CB;
        row_off[u] = (unsigned)min(c0 + cl, o - 1) * (unsigned)p * (unsigned)na + 4u * (unsigned)pc;
    }
    int cpos[4] = {0, 0, 0, 0};                        // COSET: where this thread's piece (anchors 4 pc .. 4 pc + 3) lands in a row
    if (COSET) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) cpos[jj] = s_pos[4 * pc + jj];
    }
    auto fetch = [&](int j0) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int pe = next_p[u];
            stage_p[u] = pe;
            stage[u] = *reinterpret_cast<const float4 *>(fb + (row_off[u] + __umul24((unsigned)max(pe, 0), (unsigned)na)));
        }
        if (t < NBK) gtmp = ent_gx[e0 + min(j0 + t, max(n_ent - 1, 0))];
        if (t < NBK && j0 + t >= n_ent) gtmp = make_float4(1e18f, 1e18f, 1e18f, 0.f);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int row = u * (TM / 16) + rgrp;
            const int cl = row % CB;
            const bool live = stage_p[u] >= 0 && c0 + cl < o && piece < npiece;
            if (COSET) {          // the piece's four anchors go to their positions in the coset-major order (fixed per thread)
                if (piece < npiece) {
                    float *dst = s_f + ((size_t)buf * NBK * CB + row) * FP;
                    const float4 v = live ? stage[u] : make_float4(0.f, 0.f, 0.f, 0.f);
                    dst[cpos[0]] = v.x; dst[cpos[1]] = v.y; dst[cpos[2]] = v.z; dst[cpos[3]] = v.w;
                }
            } else if (piece < npiece)   // the row pitch has no slack for the idle 16th lane
                *reinterpret_cast<float4 *>(s_f + ((size_t)buf * NBK * CB + row) * FP + 4 * piece) =
                    live ? stage[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (t < NBK) s_g[buf * NBK + t] = rotate_entry(gtmp);
    };

    const int nchunk = (n_ent + NBK - 1) / NBK;

    // ---- DMA path state: every thread owns NSTD fixed 16-byte pieces of a chunk's LDS image
    // (piece f = u*TM + t of [NBK][CB][na/4]); only the entry's query point changes per chunk.
    constexpr int NSTD = 8;                               // NBK*CB*(na/4)/TM <= 8 for na <= 64
    const int total4 = NBK * CB * npiece;                 // a multiple of 64: whole waves are in or out
    const unsigned lds_f = lds_addr(s_f);
    const unsigned buf_bytes = (unsigned)(NBK * CB) * (unsigned)FP * 4u;
    unsigned dma_off[NSTD], src_off[NSTD];
    int dma_nl[NSTD];
    int idx_p = 0;
    float4 idx_g = make_float4(1e18f, 1e18f, 1e18f, 0.f);
    if constexpr (DMA) {
#pragma unroll
        for (int u = 0; u < NSTD; ++u) {
            const int f = min(u * TM + t, total4 - 1);
            const int nl = f / (CB * npiece), rem = f - nl * (CB * npiece);
            const int cl = rem / npiece, pcs = rem - cl * npiece;
            dma_off[u] = (unsigned)min(c0 + cl, o - 1) * (unsigned)p * (unsigned)na + 4u * (unsigned)pcs;
            dma_nl[u] = nl;
        }
    }
    // wave 0 keeps the ring of entry -> (query point, offset vector) two chunks ahead of the MFMAs;
    // entries past the end of the list repeat the last one with a dead offset vector (weight 0)
    auto load_idx = [&](int j0) {
        if (wave_u == 0) {
            const int j = j0 + (lane & (NBK - 1));
            const size_t e = e0 + min(j, max(n_ent - 1, 0));
            idx_p = ent_p[e];
            idx_g = ent_gx[e];
            if (j >= n_ent) idx_g = make_float4(1e18f, 1e18f, 1e18f, 0.f);
            if (FWD && (unsigned)idx_p >= (unsigned)p) {                           // shadow neighbour: any valid row, weight 0
                idx_p = 0;
                idx_g = make_float4(1e18f, 1e18f, 1e18f, 0.f);
            }
        }
    };
    auto store_idx = [&](int slot) {
        if (wave_u == 0 && lane < NBK) {
            s_p[slot * NBK + lane] = idx_p;
            s_g[slot * NBK + lane] = rotate_entry(idx_g);
        }
    };
    auto prep_rows = [&](int slot) {
#pragma unroll
        for (int u = 0; u < NSTD; ++u)
            src_off[u] = dma_off[u] + __umul24((unsigned)s_p[slot * NBK + dma_nl[u]], (unsigned)na);
    };
    auto issue = [&](int u, int buf) {
        const int f0 = u * TM + wave_u * 64;              // wave-uniform
        if (f0 < total4)
            glds16(fb + src_off[u], __builtin_amdgcn_readfirstlane(lds_f + (unsigned)buf * buf_bytes + (unsigned)f0 * 16u));
    };

    if constexpr (DMA) {
        if (nchunk > 0) {
            load_idx(0);
            store_idx(0);
            load_idx(NBK);
            store_idx(1);
            __syncthreads();
            prep_rows(0);
#pragma unroll
            for (int u = 0; u < NSTD; ++u) issue(u, 0);
            dma_wait();
        }
    } else if (nchunk > 0) {
        fetch_index(0);
        fetch(0);
        fetch_index(NBK);
        stash(0);
    }
    __syncthreads();

    // HAS_MULT = false (no permutation, or the caller found every relative rotation to be the
    // identity): anchors map to themselves and the weight constants are per-lane registers.
    // HAS_MULT = true: the permuted anchor of every (entry, anchor) pair comes from the LDS
    // table, and so do its weight constants (read 4 at a time, one wait per group).
    auto gather = [&](const float *fbuf, int gb, int s, float (&fa)[APW]) {
        const int nl = 2 * s + lh;
        const float *frow = fbuf + ((size_t)nl * CB + lk) * FP;
        if (!HAS_MULT) {   // anchors map to themselves: pairs of anchors = one aligned 8-byte read
#pragma unroll
            for (int ai = 0; ai < APW; ai += 2) {
                const float2 v = *reinterpret_cast<const float2 *>(frow + min(a_beg + ai, na - 2));
                fa[ai] = v.x; fa[ai + 1] = v.y;
            }
        } else if (COSET) {
            // the wave's two blocks: source block sigma, inner XOR x -- out[j] = in[j ^ x]
            const int r = __float_as_int(s_g[gb * NBK + nl].w);
            const unsigned codes = *reinterpret_cast<const unsigned short *>(s_mult + r * 16 + 2 * wave_u);
#pragma unroll
            for (int bl = 0; bl < 2; ++bl) {
                const unsigned code = (codes >> (8 * bl)) & 0xffu;
                const float4 v = *reinterpret_cast<const float4 *>(frow + 4 * (code & 15u));
                const bool x0 = (code & 16u) != 0, x1 = (code & 32u) != 0;
                const float a0 = x0 ? v.y : v.x, a1 = x0 ? v.x : v.y, a2 = x0 ? v.w : v.z, a3 = x0 ? v.z : v.w;
                fa[4 * bl] = x1 ? a2 : a0; fa[4 * bl + 1] = x1 ? a3 : a1; fa[4 * bl + 2] = x1 ? a0 : a2; fa[4 * bl + 3] = x1 ? a1 : a3;
            }
        } else {
            const int r = __float_as_int(s_g[gb * NBK + nl].w);
#pragma unroll
            for (int ai = 0; ai < APW; ++ai) {
                fa[ai] = frow[(int)s_mult[r * na + min(a_beg + ai, na - 1)]];
            }
        }
    };
    auto nothing = [] {};
    // mid() / end() run after the first / second half of the step's MFMAs (the DMA path requests
    // the next chunk's rows there, a few at a time, so the memory pipe never backs up into a wave)
    auto step_g = [&](const float4 g, const float (&fa)[APW], auto mid, auto end) {
        const float base = 1.0f - inv_sigma * (g.x * g.x + g.y * g.y + g.z * g.z);
        // all weights of the step first (independent chains), then the MFMAs back to back
        f32x2 wv[APW / 2];
#pragma unroll
        for (int j = 0; j < APW / 2; ++j) {
            f32x2 x = __builtin_elementwise_fma((f32x2){g.x, g.x}, kxp[j], kcp[j] + (f32x2){base, base});
            x = __builtin_elementwise_fma((f32x2){g.y, g.y}, kyp[j], x);
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp\n\ts_nop 1" : "=v"(wv[j]) : "v"((f32x2){g.z, g.w}), "v"(kzp[j]), "v"(x));
        }
        // no per-anchor guard: a wave with fewer than APW anchors repeats its last one into
        // accumulators the epilogue never stores (its SIMD partner owns a full set anyway)
#pragma unroll
        for (int ai = 0; ai < APW / 2; ++ai)
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ai], wv[ai >> 1][ai & 1], acc[ai], 0, 0, 0);
        mid();
#pragma unroll
        for (int ai = APW / 2; ai < APW; ++ai)
            acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ai], wv[ai >> 1][ai & 1], acc[ai], 0, 0, 0);
        end();
    };

    auto step = [&](int gb, int s, const float (&fa)[APW], auto mid, auto end) {
        step_g(s_g[gb * NBK + 2 * s + lh], fa, mid, end);
    };

#ifdef EAP_INV_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    if constexpr (DMA && !HAS_MULT) {
        // No permutation: every LDS read of the chunk (operands of all four steps, offset vectors,
        // the next chunk's query points) is issued right after the barrier -- reads queued behind
        // an in-flight LDS DMA of the same wave wait for it -- and the next chunk's rows are then
        // requested one at a time, after every fourth MFMA, so the memory pipe never backs up into
        // a wave that still has matrix work (tools/microbench/glds.hip has the measurements).
        int g0 = 0, g1 = 1, g2 = 2;                       // ring slots of chunks ch, ch+1, ch+2
        for (int ch = 0; ch < nchunk; ++ch) {
            const int buf = ch & 1, nb = buf ^ 1;
            TR(0);
            const float *fbuf = s_f + (size_t)buf * NBK * CB * FP;
            float fa[4][APW];
            float4 gv[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                gv[st] = s_g[g0 * NBK + 2 * st + lh];
                gather(fbuf, g0, st, fa[st]);
            }
            prep_rows(g1);
            load_idx((ch + 2) * NBK);
            __builtin_amdgcn_sched_barrier(0);
            TR(1);
            step_g(gv[0], fa[0], [&] { issue(0, nb); }, [&] { issue(1, nb); });
            __builtin_amdgcn_sched_barrier(0);
            step_g(gv[1], fa[1], [&] { issue(2, nb); }, [&] { issue(3, nb); });
            __builtin_amdgcn_sched_barrier(0);
            step_g(gv[2], fa[2], [&] { issue(4, nb); }, [&] { issue(5, nb); });
            __builtin_amdgcn_sched_barrier(0);
            step_g(gv[3], fa[3], [&] { issue(6, nb); }, [&] { issue(7, nb); });
            TR(2);
            store_idx(g2);
            dma_wait();
            TR(3);
            __syncthreads();
            TR(4);
            const int gt = g0; g0 = g1; g1 = g2; g2 = gt;
        }
    } else if constexpr (DMA) {
        int g0 = 0, g1 = 1, g2 = 2;                       // ring slots of chunks ch, ch+1, ch+2
        for (int ch = 0; ch < nchunk; ++ch) {
            const int buf = ch & 1, nb = buf ^ 1;
            TR(0);
            const float *fbuf = s_f + (size_t)buf * NBK * CB * FP;
            float fa0[APW], fa1[APW];
            // operands of the first two steps before anything else, so the matrix pipe restarts
            // right after the barrier; the next chunk's row addresses follow in their shadow.
            // The rows themselves are requested unconditionally (past the end of the list the
            // ring repeats the last entry: a harmless reload into the idle buffer).
            gather(fbuf, g0, 0, fa0);
            __builtin_amdgcn_sched_barrier(0);
            load_idx((ch + 2) * NBK);
            gather(fbuf, g0, 1, fa1);
            __builtin_amdgcn_sched_barrier(0);
            prep_rows(g1);
            __builtin_amdgcn_sched_barrier(0);
            TR(1);
            step(g0, 0, fa0, [&] { issue(0, nb); }, [&] { issue(1, nb); issue(2, nb); });
            __builtin_amdgcn_sched_barrier(0);
            gather(fbuf, g0, 2, fa0);
            __builtin_amdgcn_sched_barrier(0);
            step(g0, 1, fa1, [&] { issue(3, nb); }, [&] { issue(4, nb); issue(5, nb); });
            __builtin_amdgcn_sched_barrier(0);
            gather(fbuf, g0, 3, fa1);
            __builtin_amdgcn_sched_barrier(0);
            step(g0, 2, fa0, [&] { issue(6, nb); }, [&] { issue(7, nb); });
            __builtin_amdgcn_sched_barrier(0);
            step(g0, 3, fa1, nothing, nothing);
            TR(2);
            store_idx(g2);
            dma_wait();
            TR(3);
            __syncthreads();
            TR(4);
            const int gt = g0; g0 = g1; g1 = g2; g2 = gt;
        }
    } else
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        TR(0);
        if (ch + 1 < nchunk) {
            fetch((ch + 1) * NBK);
            fetch_index((ch + 2) * NBK);
        }
        TR(1);
        const float *fbuf = s_f + (size_t)buf * NBK * CB * FP;
        float fa0[APW], fa1[APW];
        gather(fbuf, buf, 0, fa0);
        __builtin_amdgcn_sched_barrier(0);
        gather(fbuf, buf, 1, fa1);
        __builtin_amdgcn_sched_barrier(0);
        step(buf, 0, fa0, nothing, nothing);
        __builtin_amdgcn_sched_barrier(0);
        gather(fbuf, buf, 2, fa0);
        __builtin_amdgcn_sched_barrier(0);
        step(buf, 1, fa1, nothing, nothing);
        __builtin_amdgcn_sched_barrier(0);
        gather(fbuf, buf, 3, fa1);
        __builtin_amdgcn_sched_barrier(0);
        step(buf, 2, fa0, nothing, nothing);
        __builtin_amdgcn_sched_barrier(0);
        step(buf, 3, fa1, nothing, nothing);
        TR(2);
        if (ch + 1 < nchunk) stash(buf ^ 1);
        TR(3);
        __syncthreads();
        TR(4);
    }
#ifdef EAP_INV_TRACE
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && lane == 0) {
        for (int i = 0; i < 8; ++i) eap_inv_trace[wave * 16 + i] = tr[i];
        eap_inv_trace[wave * 16 + 8] = nchunk;
    }
#endif

    // ---- epilogue: Z[b, o, k, ri, a'] (rows that are not referenced write zeros) ----------------
    float *s_o = s_f;
    float *ob = out + (size_t)bi * o * ks * rcap * na + (size_t)ri * na;
    const size_t o_ks = (size_t)rcap * na, o_cs = (size_t)ks * rcap * na;
    if (FWD && blocked == 2) {
        // transposed output [row*na + a][c*ks + k]: LDS tile [anchor][8 channels x ks], an anchor's 8 * ks values of a pass leave
        // as one contiguous run of its output row (as csrc/so3_inter_mfma.hip; the launcher checks ks % 4 == 0, o % 8 == 0)
        __syncthreads();                                   // the last chunk's operand reads are done before the tile is overwritten
        const int run = 8 * ks, pitch = run + 4;
        const size_t CK = (size_t)o * ks;
        float *obt = out + (size_t)bi * o * o_cs + (size_t)ri * na * CK + (size_t)c0 * ks;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            if (lk < ks) {
#pragma unroll
                for (int ai = 0; ai < APW; ++ai) {
                    if (ai < a_cnt) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            s_o[(size_t)(a_beg + ai) * pitch + (rr + 4 * lh) * ks + lk] = acc[ai][ps * 4 + rr];
                    }
                }
            }
            __syncthreads();
            if (c0 + ps * 8 < o) {
                const int q4 = run >> 2;                                 // float4 per anchor
                const int da = TM / q4, dj = TM - da * q4;               // (anchor, piece) advance incrementally: no division in the loop
                int a = t / q4, j = t - a * q4;
                while (a < na) {
                    *reinterpret_cast<float4 *>(obt + (size_t)a * CK + (size_t)ps * run + 4 * j) =
                        *reinterpret_cast<const float4 *>(s_o + (size_t)a * pitch + 4 * j);
                    a += da; j += dj;
                    if (j >= q4) { j -= q4; ++a; }
                }
            }
            __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        if (lk < ks) {
#pragma unroll
            for (int ai = 0; ai < APW; ++ai) {
                if (ai < a_cnt) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int cl8 = rr + 4 * lh;
                        s_o[((size_t)cl8 * ks + lk) * na + am[ai]] = acc[ai][ps * 4 + rr];
                    }
                }
            }
        }
        __syncthreads();
        {
            const int nrows = 8 * ks;
            int cl8 = rgrp / ks, k = rgrp - cl8 * ks;
            for (int row = rgrp; row < nrows; row += TM / 16) {
                const int ci = c0 + ps * 8 + cl8;
                if (ci < o && piece < npiece)
                    *reinterpret_cast<float4 *>(ob + (size_t)ci * o_cs + (size_t)k * o_ks + 4 * pc) =
                        *reinterpret_cast<const float4 *>(s_o + (size_t)row * na + 4 * pc);
                k += TM / 16;
                while (k >= ks) { k -= ks; ++cl8; }
            }
        }
        __syncthreads();
    }
}

}  // namespace

// forward grouping of the clouds WITH per-neighbour anchor permutations on the kernel above (FWD = true); identity clouds
// (nonident[b] == 0) are skipped.  feats [b,c,n,na], idx [b,p,nn], gx [b,p,nn,4]; out: X [b,c,ks,p,na] (blocked = 0) or
// transposed (blocked = 2).  -1: this shape is not taken (the caller falls back to csrc/so3_inter_mfma.hip).
static int g_perm_fwd_lists = 1;      // eap_so3_group_perm_fwd(0): the round-1 register-staged kernel (csrc/so3_inter_mfma.hip), for A/B runs

extern "C" int eap_so3_group_perm_fwd(int on) {
    const int was = g_perm_fwd_lists;
    if (on == 0 || on == 1) g_perm_fwd_lists = on;
    return was;
}

int eap::group_fwd_perm_lists(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats, const int32_t *idx,
                              const float *gx, const float *rk, const uint8_t *mult, const int32_t *nonident, int blocked, float *out,
                              hipStream_t s) {
    if (!g_perm_fwd_lists) return -1;
    if (!mult || !nonident || (blocked != 0 && blocked != 2) || (na & 7) != 4 || na > 64 || ks > 32) return -1;
    if (blocked == 2 && ((ks & 3) != 0 || (c & 7) != 0)) return -1;
    if ((long long)c * n * na >= (1ll << 31)) return -1;
    const size_t stage_b = sizeof(float) * 2 * NBK * CB * na;
    if (sizeof(float) * (8 * (size_t)ks + 4) * na > stage_b) return -1;
    const size_t shmem = stage_b + 16 * 3 * NBK + 16 * NBK + 16 * 3 * (size_t)na + (size_t)na * na;
    if (shmem > 160 * 1024) return -1;
    auto kern = so3_inter_group_inv_kernel<true, true, true>;
    if (int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem), "so3_inter_group_fwd (permuted) shared memory")) return e;
    hipLaunchKernelGGL(kern, dim3(p, (c + CB - 1) / CB, b), dim3(TM), shmem, s, c, n, nn, na, ks, p, 1.0f / sigma, -1, feats,
                       (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr, idx, reinterpret_cast<const float4 *>(gx), rk,
                       mult, (const float *)nullptr, out, blocked, nonident, (const uint8_t *)nullptr);
    return eap::check_launch("so3_inter_group_fwd (permuted clouds, entry-list kernel)");
}

// gy rows padded to `gy_pitch` floats (a multiple of 4, >= na; e.g. 64: every row starts a 256-byte line):
// same result, only the entry-list kernel (no anchor permutation) takes it
extern "C" int eap_so3_inter_group_inv_pitch_f32(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap,
                                                 float sigma, const float *gy, const int32_t *rows,
                                                 const int32_t *off, const int32_t *cnt,
                                                 const int32_t *ent_p, const float *ent_gx, const float *rk,
                                                 float *z, eap_stream_t stream) {
    if (b <= 0 || o <= 0 || rcap <= 0 || na <= 0 || ks <= 0) return 0;
    if (!eap::group_lists_supported(na, ks)) return eap::bad_arg("so3_inter_group_inv_pitch: unsupported anchor / kernel-point count");
#ifdef EAP_EXPERIMENTS
    if (eap::group_listsh_preferred(o, na, ks, 0))
        return eap::group_listsh_inv(b, o, p, nn, na, gy_pitch, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, z, eap::S(stream));
    if (eap::group_lists3_preferred(o, na, ks, 0))
        return eap::group_lists3_inv(b, o, p, nn, na, gy_pitch, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, z, eap::S(stream));
#endif
    if (eap::group_lists2_preferred(o, na, ks, 0))
        return eap::group_lists2_inv(b, o, p, nn, na, gy_pitch, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, z, eap::S(stream));
    return eap::group_lists_inv(b, o, p, nn, na, gy_pitch, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, z, eap::S(stream));
}

static int group_inv(int b, int o, int p, int nn, int na, int ks, int rcap, float sigma, const float *gy, const int32_t *rows,
                     const int32_t *off, const int32_t *cnt, const int32_t *ent_p, const float *ent_gx, const float *rk,
                     const uint8_t *multinv, const float *anchors, int identity_anchor, const uint8_t *coset_order,
                     const uint8_t *coset_code, float *z, eap_stream_t stream);

namespace {
// dst[row, i] = src[row, order[i]]: the anchor axis of a [rows, na] tensor re-ordered (thread = one 16-byte word of dst)
__global__ __launch_bounds__(256) void anchor_reorder_kernel(long long words, int npiece, int na, const float *__restrict__ src,
                                                             const uint8_t *__restrict__ order, float4 *__restrict__ dst,
                                                             long long words_per_cloud = 0, const int32_t *__restrict__ nonident = nullptr) {
    // grid-stride: a launch over clouds that are all skipped costs a few thousand workgroups, not one per 256 words
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < words; i += (long long)gridDim.x * 256) {
        if (nonident != nullptr && nonident[i / words_per_cloud] == 0) continue;
        const long long row = i / npiece;
        const int pc = (int)(i - row * npiece);
        const float *r = src + row * na;
        const uchar4 o4 = *reinterpret_cast<const uchar4 *>(order + 4 * pc);
        dst[i] = make_float4(r[o4.x], r[o4.y], r[o4.z], r[o4.w]);
    }
}
}  // namespace

// dst [rows, na] = src with its anchor axis re-ordered: dst[., i] = src[., order[i]] (order uint8 [>= na], 4-byte aligned; na a
// multiple of 4).  What eap_so3_inter_group_inv_coset_f32 expects of gy when na = 4 mod 8 (its rows then travel by DMA).
extern "C" int eap_anchor_reorder_f32(int64_t rows, int na, const float *src, const uint8_t *order, float *dst, eap_stream_t stream) {
    if (rows <= 0 || na <= 0) return 0;
    if ((na & 3) != 0 || (reinterpret_cast<uintptr_t>(order) & 3) || (reinterpret_cast<uintptr_t>(dst) & 15))
        return eap::bad_arg("anchor_reorder: na must be a multiple of 4, order 4-byte and dst 16-byte aligned");
    const long long words = rows * (na >> 2);
    hipLaunchKernelGGL(anchor_reorder_kernel, dim3((unsigned)std::min<long long>((words + 255) / 256, 16384)), dim3(256), 0, eap::S(stream), words, na >> 2, na, src, order,
                       reinterpret_cast<float4 *>(dst), 0ll, (const int32_t *)nullptr);
    return eap::check_launch("anchor_reorder");
}

// the same for [b, rows_per_cloud, na] with a per-cloud flag: clouds whose nonident[b] is 0 are skipped (dst left unwritten)
extern "C" int eap_anchor_reorder_clouds_f32(int b, int64_t rows_per_cloud, int na, const float *src, const uint8_t *order,
                                             const int32_t *nonident, float *dst, eap_stream_t stream) {
    if (b <= 0 || rows_per_cloud <= 0 || na <= 0) return 0;
    if ((na & 3) != 0 || (reinterpret_cast<uintptr_t>(order) & 3) || (reinterpret_cast<uintptr_t>(dst) & 15))
        return eap::bad_arg("anchor_reorder_clouds: na must be a multiple of 4, order 4-byte and dst 16-byte aligned");
    const long long per = rows_per_cloud * (na >> 2), words = per * b;
    hipLaunchKernelGGL(anchor_reorder_kernel, dim3((unsigned)std::min<long long>((words + 255) / 256, 16384)), dim3(256), 0, eap::S(stream), words, na >> 2, na, src, order,
                       reinterpret_cast<float4 *>(dst), per, nonident);
    return eap::check_launch("anchor_reorder_clouds");
}

extern "C" int eap_so3_inter_group_inv_f32(int b, int o, int p, int nn, int na, int ks, int rcap,
                                           float sigma, const float *gy, const int32_t *rows,
                                           const int32_t *off, const int32_t *cnt,
                                           const int32_t *ent_p, const float *ent_gx, const float *rk,
                                           const uint8_t *multinv, const float *anchors, int identity_anchor, float *z,
                                           eap_stream_t stream) {
    return group_inv(b, o, p, nn, na, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, multinv, anchors, identity_anchor, nullptr, nullptr, z, stream);
}

// the same with the coset tables of `multinv` (vgtk/so3conv/functional.py _coset_tables: order uint8 [64], code uint8 [na,16]):
// the permuted clouds' operand is coset-major in LDS (kernel template COSET).  For na = 4 mod 8 (the 60 anchors) the rows
// travel by DMA and `gy` must ALREADY have its anchor axis in that order (eap_anchor_reorder_f32); z comes back in memory order.
extern "C" int eap_so3_inter_group_inv_coset_f32(int b, int o, int p, int nn, int na, int ks, int rcap,
                                                 float sigma, const float *gy, const int32_t *rows,
                                                 const int32_t *off, const int32_t *cnt,
                                                 const int32_t *ent_p, const float *ent_gx, const float *rk,
                                                 const uint8_t *multinv, const float *anchors, int identity_anchor,
                                                 const uint8_t *coset_order, const uint8_t *coset_code, float *z, eap_stream_t stream) {
    if (!multinv || !coset_order || !coset_code || (na & 3) != 0 || na > 64 || (reinterpret_cast<uintptr_t>(coset_code) & 3))
        return eap::bad_arg("so3_inter_group_inv_coset: permutation and coset tables are required (4-byte aligned), na a multiple of 4 up to 64");
    return group_inv(b, o, p, nn, na, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, multinv, anchors, identity_anchor, coset_order, coset_code, z, stream);
}

static int group_inv(int b, int o, int p, int nn, int na, int ks, int rcap, float sigma, const float *gy, const int32_t *rows,
                     const int32_t *off, const int32_t *cnt, const int32_t *ent_p, const float *ent_gx, const float *rk,
                     const uint8_t *multinv, const float *anchors, int identity_anchor, const uint8_t *coset_order,
                     const uint8_t *coset_code, float *z, eap_stream_t stream) {
    if (b <= 0 || o <= 0 || rcap <= 0 || na <= 0 || ks <= 0) return 0;
    if (na > 64 || (na & 3) != 0) return eap::bad_arg("so3_inter_group_inv: the anchor count must be a multiple of 4, at most 64");
    if (ks > 32) return eap::bad_arg("so3_inter_group_inv: at most 32 kernel points");
    if (multinv && !anchors) return eap::bad_arg("so3_inter_group_inv: the anchor rotations are needed with a permutation table");
    if ((long long)o * p * na >= (1ll << 31)) return eap::bad_arg("so3_inter_group_inv: one cloud's gradient exceeds 2^31 elements");
    hipStream_t s = eap::S(stream);
    // no anchor permutation: the two-workgroups-per-CU kernel of csrc/so3_inter_lists.hip
    if (!multinv && eap::group_lists_supported(na, ks)) {
#ifdef EAP_EXPERIMENTS
        if (eap::group_listsh_preferred(o, na, ks, 0))
            return eap::group_listsh_inv(b, o, p, nn, na, na, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, z, s);
        if (eap::group_lists3_preferred(o, na, ks, 0))
            return eap::group_lists3_inv(b, o, p, nn, na, na, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, z, s);
#endif
        if (eap::group_lists2_preferred(o, na, ks, 0))
            return eap::group_lists2_inv(b, o, p, nn, na, na, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, z, s);
        return eap::group_lists_inv(b, o, p, nn, na, na, ks, rcap, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, z, s);
    }
    // row pitch = na (direct global -> LDS rows) when that pitch spreads the 32 channel lanes of
    // an operand read over 16 bank pairs (na = 4 mod 8, e.g. the 60 icosahedral anchors);
    // other anchor counts take the register-staged variant with a padded pitch
    const bool dma = (na & 7) == 4;
    const int FP_ = dma ? na : (na <= 60 ? 60 : FPMAX);
    const size_t stage_b = sizeof(float) * 2 * NBK * CB * FP_;
    if (sizeof(float) * 8 * (size_t)ks * na > stage_b) return eap::bad_arg("so3_inter_group_inv: epilogue tile too large");
    size_t shmem = stage_b + 16 * 3 * NBK + 16 * NBK + (multinv ? 16 * 3 * (size_t)na + (size_t)na * na : 0);
    if (shmem > 160 * 1024) return eap::bad_arg("so3_inter_group_inv: LDS budget exceeded");
    dim3 grid(rcap, (o + CB - 1) / CB, b);
    const float4 *g4 = reinterpret_cast<const float4 *>(ent_gx);
    auto launch = [&](auto kern) {
        int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem), "so3_inter_group_inv shared memory");
        if (e) return e;
        hipLaunchKernelGGL(kern, grid, dim3(TM), shmem, s, o, p, nn, na, ks, rcap, 1.0f / sigma, identity_anchor, gy, rows, off, cnt, ent_p, g4, rk, multinv, anchors, z,
                           0, (const int32_t *)nullptr, (const uint8_t *)nullptr);
        return 0;
    };
    int e;
    if (multinv && coset_order) {
        // (the code table travels in the `multinv` argument slot; with the DMA loader gy's anchor axis is ALREADY coset-major)
        auto go = [&](auto kern) {
            int e2 = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem), "so3_inter_group_inv shared memory");
            if (e2) return e2;
            hipLaunchKernelGGL(kern, grid, dim3(TM), shmem, s, o, p, nn, na, ks, rcap, 1.0f / sigma, identity_anchor, gy, rows, off, cnt, ent_p, g4, rk, coset_code,
                               anchors, z, 0, (const int32_t *)nullptr, coset_order);
            return 0;
        };
        e = dma ? go(so3_inter_group_inv_kernel<true, true, false, true>) : go(so3_inter_group_inv_kernel<true, false, false, true>);
    } else if (multinv) e = dma ? launch(so3_inter_group_inv_kernel<true, true>) : launch(so3_inter_group_inv_kernel<true, false>);
    else e = dma ? launch(so3_inter_group_inv_kernel<false, true>) : launch(so3_inter_group_inv_kernel<false, false>);
    if (e) return e;
#ifdef EAP_INV_TRACE
    {
        unsigned long long h[8 * 16];
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h, HIP_SYMBOL(eap_inv_trace), sizeof(h));
        for (int w = 0; w < 8 && h[w * 16 + 8] > 0; ++w)
            fprintf(stderr, "inv trace wave %d: chunks %llu  fetch %llu  compute %llu  stash %llu  barrier %llu  (cycles/chunk)\n", w,
                    h[w * 16 + 8], h[w * 16 + 1] / h[w * 16 + 8], h[w * 16 + 2] / h[w * 16 + 8], h[w * 16 + 3] / h[w * 16 + 8],
                    h[w * 16 + 4] / h[w * 16 + 8]);
    }
#endif
    return eap::check_launch("so3_inter_group_inv");
}
