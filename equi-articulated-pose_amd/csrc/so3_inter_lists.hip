// csrc/so3_inter_lists.hip -- SO(3) grouping over entry lists on the matrix cores, two
// workgroups per CU.  One kernel serves both directions of the inter convolution when no anchor
// permutation is in play (no pose, or every relative rotation of the cloud is the identity):
//
//   forward   X[b,c,k,p,a]  = sum_n  F[b,c,idx[b,p,n],a]  * w(p,a,k,n)      entries of row p = its neighbours
//             (vgtk/vgtk/so3conv/functional.py:L1112-1261, einsum 'bcpna,bpakn->bckpa' at L1261)
//   backward  Z[b,o,k,r,a]  = sum_{(p,n)->q_r} dY[b,o,p,a] * w(p,a,k,n)      entries of row r = the (p,n) pairs
//             that reference support point q_r (inverse neighbour lists, csrc/so3_inter_inv.hip)
//   w(p,a,k,n) = relu(1 - |g(p,n) - A_a kappa_k|^2 / sigma)
//
// Per (row, 32 channels, anchor group) this is a [32 x E] x [E x 24] product for every anchor,
// mapped on v_mfma_f32_32x32x2_f32 (M = channels, N = kernel points padded to 32, K = entries):
//   * A operand: the 16-byte pieces of the entries' feature rows go global -> LDS by DMA
//     (global_load_lds_dwordx4, wave-uniform base in SGPRs + one 32-bit offset per lane), one
//     wave-instruction after every second MFMA; no staging registers, no ds_write pass.  The LDS image of
//     a chunk is [8 entries][32 channels][8 slots of 16 bytes]; the piece of anchors 4w..4w+3 of channel
//     row r sits in slot (w + r) mod 8 (the rotation is applied on the global-address side, for free), so
//     the 32 channel lanes of an operand read land on different banks and ONE ds_read_b128 brings the
//     wave's four anchors.
//   * B operand: never in memory.  Lane (k = l&31, e = l>>5) evaluates
//     w = clamp(1 - |g|^2/s - |kappa_k|^2/s + g.(2 A_a kappa_k / s)) in registers.  The fp32 MFMA and the
//     vector ALU share their multipliers on this part (tools/microbench/mfma_waves.hip: every VALU
//     instruction between MFMAs costs its 4 cycles of matrix time at ANY occupancy; the round-1 loop
//     spent 150 VALU instructions per 16 MFMAs and sat at 95 % of the resulting 63 % ceiling), so the
//     loop is written for VALU count: weights of two anchors per packed instruction (v_pk_add/v_pk_fma,
//     relu = the clamp modifier of the last FMA: 2 instructions per weight instead of 5), the per-entry
//     term evaluated once per chunk by 8 lanes and handed out by ds_bpermute, every LDS address an
//     immediate offset of one register, no 64-bit address arithmetic: ~48 VALU per 16 MFMAs.
//   * The anchors of a row are split into two groups (32 + 28 at na = 60) handled by different
//     workgroups: 4 anchors per wave = 64 accumulator VGPRs, under 128 VGPRs in total and 64 KB of
//     LDS, so TWO workgroups share a CU.  A workgroup is alone for its chunk barrier, its DMA
//     drain, its prologue and its epilogue (20-30 % of its life in the one-per-CU version:
//     tools/microbench/glds.hip, profiles/r01_*); with a second one resident the matrix pipe has
//     other MFMAs to run in those gaps.
//   * Row end: accumulators straight to global (16-byte stores, no LDS transposition, no barrier);
//     in the forward a workgroup streams through 8 consecutive rows without draining its pipeline.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CB = 32;        // channels per block (one MFMA M tile)
constexpr int NBK = 8;        // entries per LDS stage (4 MFMA k-steps)
constexpr int APW = 4;        // anchors per wave
constexpr int NWV = 8;
constexpr int TM = 64 * NWV;
constexpr int NSTD = 4;       // DMA instructions per thread and chunk: NBK*CB*8 pieces / TM
constexpr int PITCH = 32;     // floats per LDS row = 8 pieces, whatever the group's anchor count (see the DMA mapping)
constexpr unsigned BUF_BYTES = NBK * CB * PITCH * 4;

__device__ inline unsigned lds_addr(const void *ptr) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)ptr;
}
// wave-wide 16-byte-per-lane global -> LDS DMA, invisible to hipcc's waitcnt bookkeeping on
// purpose (see csrc/so3_inter_inv.hip); the kernel waits with dma_wait() before the chunk barrier
__device__ inline void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same with a wave-uniform 64-bit base in SGPRs and a 32-bit byte offset per lane: no 64-bit VALU add per request
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

// LISTS = true : rows / off / cnt describe variable-length entry lists (backward);
// LISTS = false: row r of cloud b owns entries [ (b*R + r)*nn, +nn ) (forward: its neighbours).
// LAYOUT of the output: 0 = [b,c,k,row,a] (reference), 1 = blocked by anchor quads, 2 = transposed [row*na+a][c*ks+k]
template <bool LISTS, int LAYOUT>
__global__ __launch_bounds__(TM, 4) void so3_group_lists_kernel(
    int C, int PF, int na, int fpitch, int ks, int R, int nn, int ent_stride, int AG, int gsz, int RPB, float inv_sigma,
    const float *__restrict__ F, const int32_t *__restrict__ rows, const int32_t *__restrict__ off,
    const int32_t *__restrict__ cnt, const int32_t *__restrict__ ent_p, const float4 *__restrict__ ent_gx,
    const float *__restrict__ rk, const int32_t *__restrict__ nonident, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- block -> (row, anchor group, channel slice, cloud); XCDs get whole (slice, cloud) pairs ----
    const int nrun = (R + RPB - 1) / RPB;                 // runs of RPB consecutive rows (RPB = 1 with lists)
    const int ny = gridDim.y, nsl = ny * gridDim.z, per_slice = nrun * AG;
    int qd = blockIdx.x, sl = blockIdx.y + ny * blockIdx.z;
    if ((nsl & 7) == 0) {
        const unsigned lin = blockIdx.x + (unsigned)per_slice * (blockIdx.y + (unsigned)ny * blockIdx.z);
        const unsigned j = lin >> 3;
        sl = (int)((lin & 7u) + 8u * (j / (unsigned)per_slice));
        qd = (int)(j % (unsigned)per_slice);
    } else {
        qd = xcd_point(blockIdx.x, per_slice);           // contiguous rows per XCD (whole output lines in one L2)
    }
    const int run = qd / AG, ag = qd - run * AG;
    const int r_begin = run * RPB, rows_blk = min(RPB, R - r_begin);
    const int cy = sl % ny, bi = sl / ny, c0 = cy * CB;
    if (nonident != nullptr && __builtin_amdgcn_readfirstlane(nonident[bi]) != 0) return;   // permuted cloud: not ours

    const int t = threadIdx.x, lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lk = lane & 31, lh = lane >> 5;
    const int a0 = ag * gsz, gcount = min(gsz, na - a0);      // anchors [a0, a0 + gcount) of this block
    const int npg = gcount >> 2;                               // 16-byte pieces per feature row that exist
    const int al_beg = wave_u * APW;                           // first local anchor of this wave = piece wave_u
    const bool active = al_beg < gcount;                       // wave-uniform

    float *s_f = reinterpret_cast<float *>(smem);                           // [2][NBK][CB][PITCH]
    float4 *s_g = reinterpret_cast<float4 *>(s_f + 2 * NBK * CB * PITCH);   // [3][NBK] ring
    int *s_p = reinterpret_cast<int *>(s_g + 3 * NBK);                      // [3][NBK] ring

    // entries of the block: one list (backward), or the neighbours of rows_blk consecutive rows
    // (forward; they are contiguous in idx / gx, and with nn a multiple of the chunk the flat chunk
    // sequence never straddles two rows)
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

    // ---- per-lane weight constants of this wave's anchors (k = lane & 31) -------------------------
    // On this part the fp32 MFMA and the vector ALU share their multipliers (vector fp32 peak = matrix fp32 peak;
    // tools/microbench/mfma_waves.hip: every VALU instruction between MFMAs costs its 4 cycles of matrix time at any
    // occupancy), so the weight w = relu(1 - |g|^2/s - |kappa|^2/s + 2 g.kappa'/s) is evaluated with as few VALU
    // instructions as it takes: everything goes through packed operations on two anchors at a time -- the
    // per-entry term joins -|kappa|^2/s in one packed add, then three packed FMAs, and the relu is the clamp
    // modifier of the last one (weights never exceed 1): 2 instructions per weight instead of 5.
    f32x2 kxp[APW / 2], kyp[APW / 2], kzp[APW / 2], kcp[APW / 2];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai) {
        const int a = a0 + min(al_beg + ai, gcount - 1);
        const float *r3 = rk + ((size_t)a * ks + min(lk, ks - 1)) * 3;
        const float x = r3[0], y = r3[1], z = r3[2];
        kxp[ai >> 1][ai & 1] = 2.f * inv_sigma * x;
        kyp[ai >> 1][ai & 1] = 2.f * inv_sigma * y;
        kzp[ai >> 1][ai & 1] = 2.f * inv_sigma * z;
        kcp[ai >> 1][ai & 1] = lk < ks ? -inv_sigma * (x * x + y * y + z * z) : -1e30f;
    }
    // Kernel points rotated by the anchors all have the norm of the unrotated point, so -|kappa|^2/s is normally the same
    // for the wave's four anchors (to rounding): it then joins the per-entry term with ONE plain add per k-step instead
    // of two packed ones (a packed instruction costs the matrix pipe twice a plain one, tools/microbench/mfma_riders.hip).
    // Arbitrary rk tables (norms that differ) keep the general form; the choice is wave-uniform.
    const float kcl = kcp[0][0];
    const bool kc_uniform = __all(fabsf(kcp[0][1] - kcl) <= 1e-6f * fabsf(kcl) && fabsf(kcp[1][0] - kcl) <= 1e-6f * fabsf(kcl) &&
                                  fabsf(kcp[1][1] - kcl) <= 1e-6f * fabsf(kcl)) != 0;
    // operand read: the wave's four anchors are ONE 16-byte piece (piece wave_u) of channel row lk, stored at slot
    // (piece + row) mod 8 so that the 32 channel lanes of a read spread over the banks
    const float4 *fa_lane = reinterpret_cast<const float4 *>(s_f + (size_t)(lh * CB + lk) * PITCH + 4 * ((wave_u + lk) & 7));

    f32x16 acc[APW];

    // ---- DMA: thread -> NSTD 16-byte pieces of a chunk's LDS image ---------------------------------
    // A DMA instruction writes its 64 lanes' pieces to consecutive LDS addresses, so the image is
    // [8 entries][32 channel rows][8 slots] with thread t of instruction u at flat piece u*512 + t: always the same
    // channel row (t>>3)&31 and slot t&7, entry 2u + (t>>8) -- one request offset and one ring address serve the
    // four instructions.  Slot -> piece by the row's rotation; pieces a smaller anchor group does not have are
    // masked out (their slots stay unwritten and are never read).
    // request address = slice base (SGPRs) + 32-bit byte offset per lane (the launcher bounds 32 rows of a cloud)
    const float *fb = F + ((size_t)bi * C + c0) * PF * fpitch;   // fpitch: floats between consecutive feature rows (>= na)
    const unsigned lds_f = lds_addr(s_f);
    const unsigned row_bytes = (unsigned)fpitch * 4u;
    const int d_cl = (t >> 3) & 31, d_piece = ((t & 7) - d_cl) & 7, d_nl = t >> 8;
    const bool d_valid = d_piece < npg;
    const unsigned dma_off = ((unsigned)(min(c0 + d_cl, C - 1) - c0) * (unsigned)PF * (unsigned)fpitch + (unsigned)(a0 + 4 * min(d_piece, npg - 1))) * 4u;
    // The ring of entry -> (feature row, offset vector) runs two chunks ahead of the MFMAs and is
    // filled by DMA as well (8 lanes of wave 0).  Entries past the end of the list repeat the last
    // one; they -- and the forward's shadow rows -- are given a dead offset vector (weight 0)
    // when the weights are evaluated.
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
            int pe = s_p[slot * NBK + 2 * u + d_nl];
            if (!LISTS) pe = (unsigned)pe < (unsigned)PF ? pe : 0;     // shadow row: any valid row, weight 0
            src_off[u] = dma_off + __umul24((unsigned)pe, row_bytes);
        }
    };
    auto issue = [&](int u, int buf) {
        if (d_valid)
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

    // per chunk: lane e (mod 8) evaluates the per-entry term 1 - |g_e|^2/s of the chunk's entry e, or a dead value
    // for entries past the end of the list and for the forward's shadow rows; the k-steps fetch theirs by bpermute
    auto chunk_bases = [&](int ch, int gslot) {
        const int e = lane & (NBK - 1);
        const float4 g = s_g[gslot * NBK + e];
        float b = 1.0f - inv_sigma * (g.x * g.x + g.y * g.y + g.z * g.z);
        bool dead = ch * NBK + e >= n_ent;
        if (!LISTS) dead = dead || (unsigned)s_p[gslot * NBK + e] >= (unsigned)PF;
        return __float_as_int(dead ? -1e30f : b);
    };
    // operands of one MFMA k-step (2 entries): the anchor pairs of this lane's channel row, the entry's offset
    // vector and its per-entry term
    auto gather = [&](const float4 *fab, int gslot, int bases, int s, float4 &fa, float4 &g, float &bk) {
        fa = fab[s * (2 * CB * PITCH / 4)];
        g = s_g[gslot * NBK + 2 * s + lh];
        bk = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (2 * s + lh), bases));
    };
    auto nothing = [] {};
    // first (block-uniform): the k-step that opens a row starts from the constant C = 0 -- the accumulators are never
    // zeroed by vector instructions (64 per row and wave, i.e. matrix time)
    auto step = [&](auto kcu, const float4 g, float bk, const float4 fv, auto mid, auto end, bool first = false) {
        const float fa[APW] = {fv.x, fv.y, fv.z, fv.w};
        f32x2 wv[APW / 2];
        const float bkc = bk + kcl;
#pragma unroll
        for (int j = 0; j < APW / 2; ++j) {
            f32x2 x = __builtin_elementwise_fma((f32x2){g.x, g.x}, kxp[j], decltype(kcu)::value ? (f32x2){bkc, bkc} : kcp[j] + (f32x2){bk, bk});
            x = __builtin_elementwise_fma((f32x2){g.y, g.y}, kyp[j], x);
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp\n\ts_nop 1" : "=v"(wv[j]) : "v"((f32x2){g.z, g.w}), "v"(kzp[j]), "v"(x));
        }
        // unguarded: a wave whose last anchors fall off the group repeats its last one into
        // accumulators the epilogue never stores
        __builtin_amdgcn_s_setprio(3);                      // a wave with MFMAs ready goes first (-2.5 % on one box, A/B)
        if (first) {
            const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ai = 0; ai < APW / 2; ++ai)
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ai], wv[ai >> 1][ai & 1], zc, 0, 0, 0);
            mid();
#pragma unroll
            for (int ai = APW / 2; ai < APW; ++ai)
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ai], wv[ai >> 1][ai & 1], zc, 0, 0, 0);
        } else {
#pragma unroll
            for (int ai = 0; ai < APW / 2; ++ai)
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ai], wv[ai >> 1][ai & 1], acc[ai], 0, 0, 0);
            mid();
#pragma unroll
            for (int ai = APW / 2; ai < APW; ++ai)
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ai], wv[ai >> 1][ai & 1], acc[ai], 0, 0, 0);
        }
        if (wave_u >= NWV / 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        end();
    };

    // Row end: the accumulators go straight to global memory.  D[i = channel][j = kernel point]
    // sits as col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5); register r of this wave's four
    // anchors is one 16-byte store into out[b, c0+i, k, row, a0 + al_beg ..].  64 partial lines per
    // wave-instruction, completed in L2 by the other waves of the two anchor groups; no LDS
    // transposition, no barrier, and nobody waits for the stores (they are issued right after
    // the chunk's DMA drain, so the next drain finds them long retired).
    // Addresses: wave-uniform 64-bit base per store + ONE 32-bit per-lane offset (the launcher
    // bounds it), so the 16 stores cost no address registers inside the MFMA loop.
    const size_t o_ks = (size_t)R * na, o_cs = (size_t)ks * R * na;
    float *ob = out + (size_t)bi * C * o_cs + (size_t)c0 * o_cs + a0;
    const unsigned lane_off = (unsigned)((size_t)(4 * lh) * o_cs + (size_t)min(lk, ks - 1) * o_ks) + (unsigned)al_beg;
    const bool full_c = c0 + CB <= C;                       // block-uniform
    // blocked output (the forward's X when the contraction GEMM reads it with eap_gemm_f32_xb):
    // out[b][row][anchor quad][c][k][4] -- for one register r the 24 kernel-point lanes of a
    // channel write 384 contiguous bytes, 6 cache lines per wave-instruction instead of 64
    const int npq = na >> 2, aq0 = (a0 + al_beg) >> 2;
    float *obb = out + (size_t)bi * C * o_cs;
    const unsigned lane_off_b = (unsigned)((4 * lh) * ks + min(lk, ks - 1)) * 4u;
    auto store_row = [&](int row) {
        if (active && lk < ks) {
            if (LAYOUT == 2) {
                // transposed output out[b][row*na + a][c*ks + k] (the plain [P*A, C*K] matrix the contraction GEMM reads
                // as B^T): for one register r and one anchor the 24 kernel-point lanes of a channel write 96 contiguous
                // bytes, consecutive channels follow -- dword stores.  Wave-uniform base (SALU arithmetic) + one 32-bit
                // byte offset per lane: the 64 stores of a row end cost no vector address arithmetic (they did: ~150
                // VALU instructions per row end, i.e. matrix time)
                const size_t CK = (size_t)C * ks;
                float *rb = obb + ((size_t)row * na + a0 + al_beg) * CK + (size_t)c0 * ks;      // uniform
                const unsigned lo_b = (unsigned)((4 * lh) * ks + lk) * 4u;
                if (full_c) {
                    // (inline asm: hipcc does not pick the SGPR-base form here; the accumulators were last written by MFMAs
                    // the asm cannot declare a dependency on, hence the wait states first)
                    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int ai = 0; ai < APW; ++ai)
                            asm volatile("global_store_dword %0, %1, %2" : : "v"(lo_b), "v"(acc[ai][r]),
                                         "s"(rb + (size_t)ai * CK + (size_t)((r & 3) + 8 * (r >> 2)) * ks) : "memory");
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (c0 + (r & 3) + 8 * (r >> 2) + 4 * lh < C) {
#pragma unroll
                            for (int ai = 0; ai < APW; ++ai)
                                *reinterpret_cast<float *>(reinterpret_cast<char *>(rb + (size_t)ai * CK + (size_t)((r & 3) + 8 * (r >> 2)) * ks) + lo_b) = acc[ai][r];
                        }
                }
            } else if (LAYOUT == 1) {
                float *rb = obb + (((size_t)row * npq + aq0) * C + c0) * ks * 4;     // uniform
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (full_c || c0 + (r & 3) + 8 * (r >> 2) + 4 * lh < C)
                        *reinterpret_cast<float4 *>(rb + (size_t)((r & 3) + 8 * (r >> 2)) * ks * 4 + lane_off_b) =
                            make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            } else {
            float *rb = ob + (size_t)row * na;             // uniform
            if (full_c) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    *reinterpret_cast<float4 *>(rb + (size_t)((r & 3) + 8 * (r >> 2)) * o_cs + lane_off) =
                        make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (c0 + (r & 3) + 8 * (r >> 2) + 4 * lh < C)
                        *reinterpret_cast<float4 *>(rb + (size_t)((r & 3) + 8 * (r >> 2)) * o_cs + lane_off) =
                            make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
            }
            }
        }
    };

    // the hardware favours the older waves of a SIMD; the younger half would otherwise reach every chunk
    // barrier last
    if (wave_u >= NWV / 2) __builtin_amdgcn_s_setprio(2);
    int g0 = 0, g1 = 1, g2 = 2;                           // ring slots of chunks ch, ch+1, ch+2
    int ch_row = 0, row = r_begin;
    // two loops, one per kind of wave (a wave without anchors only feeds the DMA): the accumulators of the working
    // waves then never meet a control-flow join, which the register allocator answered with a second copy of them
    if (active) {
        // (the whole loop, row ends included, once per form of the weight evaluation -- a macro: behind a generic lambda the
        // accumulators of one instantiation went to scratch; each copy is self-contained, so they meet no control-flow join)
#define EAP_LISTS_LOOP(kcu)                                                            \
        for (int ch = 0; ch < nchunk; ++ch) {                                                                         \
            const int buf = ch & 1, nb = buf ^ 1;                                                                     \
            const float4 *fbuf = fa_lane + buf * (NBK * CB * PITCH / 4);                                              \
            float4 fa0, fa1, ga, gb;                                                                                  \
            float ba, bb;                                                                                             \
            const int bases = chunk_bases(ch, g0);                                                                    \
            gather(fbuf, g0, bases, 0, fa0, ga, ba);                                                                  \
            gather(fbuf, g0, bases, 1, fa1, gb, bb);                                                                  \
            prep_rows(g1);                                                                                            \
            issue_idx((ch + 2) * NBK, g2);                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            step(kcu, ga, ba, fa0, [&] { issue(0, nb); }, [&] { issue(1, nb); }, ch_row == 0);                        \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            gather(fbuf, g0, bases, 2, fa0, ga, ba);                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            step(kcu, gb, bb, fa1, [&] { issue(2, nb); }, [&] { issue(3, nb); });                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            gather(fbuf, g0, bases, 3, fa1, gb, bb);                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            step(kcu, ga, ba, fa0, nothing, nothing);                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            step(kcu, gb, bb, fa1, nothing, nothing);                                                                 \
            dma_wait();                                                                                               \
            if (++ch_row == nchunk_row) {                                                                             \
                store_row(row);                                                                                       \
                ch_row = 0;                                                                                           \
                ++row;                                                                                                \
            }                                                                                                         \
            __syncthreads();                                                                                          \
            const int gt = g0; g0 = g1; g1 = g2; g2 = gt;                                                             \
        }                                                                                                             \
        if (nchunk == 0) {                                                                                            \
            _Pragma("unroll") for (int ai = 0; ai < APW; ++ai)                                                        \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[ai][r] = 0.f;                                      \
            store_row(r_begin);                                                                                       \
        }                                                                                                             \

        if (kc_uniform) { EAP_LISTS_LOOP(std::true_type{}); } else { EAP_LISTS_LOOP(std::false_type{}); }
#undef EAP_LISTS_LOOP
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

struct Geometry { int AG, gsz; size_t shmem; };

// anchor groups: one up to 32 anchors, two above (the first a multiple of 4: 32 + 28 at na = 60)
bool geometry(int na, int ks, Geometry &g) {
    if (na <= 0 || (na & 3) != 0 || na > 64 || ks <= 0 || ks > 32) return false;
    g.AG = na > 32 ? 2 : 1;
    g.gsz = g.AG == 1 ? na : ((na / 2 + 3) & ~3);
    g.shmem = 2 * BUF_BYTES + 16 * 3 * NBK + 16 * NBK;
    return g.shmem <= 80 * 1024;
}

template <bool LISTS>
int launch(int blocked, int b, int C, int PF, int na, int fpitch, int ks, int R, int nn, int ent_stride, float sigma, const float *F,
           const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p, const float *ent_gx,
           const float *rk, const int32_t *nonident, float *out, hipStream_t s, const char *what) {
    Geometry g;
    if (!geometry(na, ks, g)) return eap::bad_arg("so3_group_lists: unsupported anchor / kernel-point count");
    if (fpitch < na || (fpitch & 3) != 0) return eap::bad_arg("so3_group_lists: the feature row pitch must be a multiple of 4, at least the anchor count");
    if ((long long)CB * PF * fpitch * 4 >= (1ll << 32) || PF >= (1 << 24) || fpitch * 4 >= (1 << 24))
        return eap::bad_arg("so3_group_lists: 32 feature rows of a cloud exceed the 32-bit request offsets");
    if (((long long)ks * R * na * 4 + 32ll * R * na + 64) * 4 >= (1ll << 31)) return eap::bad_arg("so3_group_lists: output rows too far apart for 32-bit store offsets");
    auto kern = blocked == 2 ? so3_group_lists_kernel<LISTS, LISTS ? 0 : 2> : blocked == 1 ? so3_group_lists_kernel<LISTS, LISTS ? 0 : 1>
                                                                                          : so3_group_lists_kernel<LISTS, 0>;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.shmem), what);
    if (e) return e;
    // forward: a workgroup streams through a run of consecutive rows (the next row's entries and
    // first chunk are in flight during the current row's last chunk)
    const int RPB = LISTS ? 1 : ((nn % NBK) == 0 ? 8 : 1);
    dim3 grid((R + RPB - 1) / RPB * g.AG, (C + CB - 1) / CB, b);
    hipLaunchKernelGGL(kern, grid, dim3(TM), g.shmem, s, C, PF, na, fpitch, ks, R, nn, ent_stride, g.AG, g.gsz, RPB, 1.0f / sigma, F,
                       rows, off, cnt, ent_p, reinterpret_cast<const float4 *>(ent_gx), rk, nonident, out);
    eap::set_kernel(LISTS ? "so3_group_lists_kernel<true, 0>" : blocked == 2 ? "so3_group_lists_kernel<false, 2>" : blocked == 1 ? "so3_group_lists_kernel<false, 1>" : "so3_group_lists_kernel<false, 0>");
    return eap::check_launch(what);
}

}  // namespace

namespace eap {

bool group_lists_supported(int na, int ks) {
    Geometry g;
    return geometry(na, ks, g);
}

// forward over the clouds whose relative rotations are all the identity (nonident[b] == 0, or
// nonident == nullptr for every cloud); other clouds are left untouched
int group_lists_fwd(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                    const int32_t *idx, const float *gx, const float *rk, const int32_t *nonident, int blocked, float *out,
                    hipStream_t s) {
    return launch<false>(blocked, b, c, n, na, na, ks, p, nn, 0, sigma, feats, nullptr, nullptr, nullptr, idx, gx, rk, nonident, out, s,
                         "so3_inter_group_fwd (lists)");
}

int group_lists_inv(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap, float sigma, const float *gy,
                    const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p,
                    const float *ent_gx, const float *rk, float *z, hipStream_t s) {
    return launch<true>(0, b, o, p, na, gy_pitch, ks, rcap, nn, p * nn, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, nullptr, z, s,
                        "so3_inter_group_inv (lists)");
}

}  // namespace eap
