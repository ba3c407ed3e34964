// csrc/so3_inter_lists2.hip -- the entry-list grouping kernel of csrc/so3_inter_lists.hip with TWO channel tiles per
// wave sharing ONE weight evaluation (round 3).
//
//   forward   X[b,c,k,p,a]  = sum_n  F[b,c,idx[b,p,n],a]  * w(p,a,k,n)
//             (vgtk/vgtk/so3conv/functional.py:L1112-1261, einsum 'bcpna,bpakn->bckpa' at L1261)
//   backward  Z[b,o,k,r,a]  = sum_{(p,n)->q_r} dY[b,o,p,a] * w(p,a,k,n)      (its autograd transpose over inverse lists)
//   w(p,a,k,n) = relu(1 - |g(p,n) - A_a kappa_k|^2 / sigma)
//
// Why: on this part every vector-ALU instruction between two fp32 MFMAs costs its issue cycles of matrix time
// (tools/microbench/mfma_waves.hip, mfma_riders.hip), and the weights -- the MFMA's B operand --
// are evaluated in registers, ~3 plain-instruction equivalents each.  A weight depends on (entry, anchor, kernel point)
// but NOT on the channel, so the one lever left is channels per weight: a wave now feeds each weight register to two
// v_mfma_f32_32x32x2_f32 (channel rows 0-31 and 32-63 of a 64-channel block) instead of one, halving the vector work per
// matrix instruction; per-entry terms, ring reads and chunk barriers are shared the same way.
//
// Geometry: workgroup = 4 waves (one per SIMD) = (row run, 64 channels, 16 anchors); wave = 4 anchors x 2 channel
// tiles = 128 accumulator registers; 64 KB of LDS -> two workgroups per CU (two waves per SIMD: the measured matrix-pipe
// utilisation is the same for 2 and 4 resident waves, mfma_waves.hip).  na = 60 -> anchor groups 16+16+16+12.
// LDS image of a chunk: [8 entries][64 channel rows][4 slots of 16 bytes]; the piece of anchors 4w..4w+3 of row r sits in
// slot (w + (r >> 2)) & 3: the four 16-lane groups a ds_read_b128 is serviced in ({0-3,12-15,20-27}, ...,
// MI355X_MICROARCH.md section LDS) then touch 16 different 16-byte slots of the 256-byte bank row -- conflict-free at a
// 64-byte row pitch, and still one contiguous 1 KB destination per global->LDS DMA instruction.
#include "common.h"
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CT = 2;         // channel tiles (MFMA M tiles) per wave
constexpr int CB = 32 * CT;   // channels per block
constexpr int NBK = 8;        // entries per LDS stage (4 MFMA k-steps)
constexpr int APW = 4;        // anchors per wave
constexpr int NWV = 4;
constexpr int TM = 64 * NWV;
constexpr int SL = 4;         // 16-byte slots per LDS row (= pieces of a 16-anchor group)
constexpr int GSZ = 4 * SL;   // anchors per workgroup
constexpr int PITCH = 4 * SL; // floats per LDS row
constexpr int NSTD = NBK * CB * SL / TM;   // DMA instructions per thread and chunk (8): instruction u carries entry u
constexpr unsigned BUF_BYTES = NBK * CB * PITCH * 4;
static_assert(CB * SL == TM, "one DMA instruction per entry: thread t <-> (row t >> 2, slot t & 3)");

__device__ inline unsigned lds_addr(const void *ptr) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)ptr;
}
// wave-wide 16-byte-per-lane global -> LDS DMA, invisible to hipcc's waitcnt bookkeeping on purpose (the kernel waits
// with dma_wait() before the chunk barrier); same idiom as csrc/so3_inter_lists.hip
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

// Timing ablations (WRONG RESULTS), compiled only with `make ABLATION=1` and selected by EAP_LISTS2_DEBUG (bit mask):
// 1 no feature DMA after the prologue, 2 constant weights (no weight evaluation), 4 no row-end stores, 8 no chunk barrier
// (only together with 1), 16 no per-k-step LDS operand reads; PERM: 32 no block move (DMA pieces from the thread's own
// block), 64 no in-block XOR (selects); 512 the feature DMA always fetches rows 0..7 (all requests hit in cache: the issue cost
// without the misses), 1024 the forward's row-end stores with the lanes in address order (the same bytes in 1 KB runs, wrong
// places: what a transposed row end would issue).  tools/lists2_ablation.py, tools/lists2_perm_ablation.py
#ifdef EAP_ABLATION
#define ABL(bit) ((dbg & (bit)) != 0)
#else
#define ABL(bit) false
#endif

// LISTS = true : rows / off / cnt describe variable-length entry lists (backward);
// LISTS = false: row r of cloud b owns entries [ (b*R + r)*nn, +nn ) (forward: its neighbours).
// LAYOUT of the output: 0 = [b,c,k,row,a] (reference), 2 = transposed [row*na+a][c*ks+k], 3 = the same transposed matrix
// computed with the MFMA operands exchanged (D = kernel points x channels: a lane then holds FOUR CONSECUTIVE kernel
// points of one channel per accumulator quad, i.e. 16 contiguous bytes of an output row, so the row end is 24 dwordx4
// stores per wave instead of 128 dword stores; both operands of v_mfma_f32_32x32x2_f32 use the same lane mapping, so the
// k-loop is unchanged).  Needs ks % 4 == 0.
// 4 = LAYOUT 3 with the COLUMNS of a 32-channel tile in the order the lanes hold them: piece (q, lh, lk) -- kernel points
// 8 q + 4 lh .. + 3 of channel lk -- goes to bytes [16 (64 q + 32 lh + lk), + 16) of the tile's 128 ks bytes, so a store
// instruction writes one contiguous 1 KB run instead of 64 pieces 4 ks bytes apart (12.6 -> 11.6 ms at C = 128).  The
// contraction that reads this matrix sums over the columns, so it only needs W's columns in the same order
// (eap_so3_group_fwd_tp_columns).  Needs ks % 8 == 0 and whole 64-channel blocks.
#define EAP_MM(f, w, c, x, y, z) (LAYOUT >= 3 ? __builtin_amdgcn_mfma_f32_32x32x2f32(w, f, c, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x2f32(f, w, c, 0, 0, 0))
//
// PERM = true: clouds WITH per-entry anchor permutations (articulated input: relative rotations between neighbours), round 4.
// The permutation of an entry is a left multiplication in the anchor group; with the operand's anchor axis in COSET-MAJOR
// order (blocks of 4 = left cosets of a Klein four-group, vgtk/so3conv/functional.py _coset_tables; the caller re-orders F
// once, eap_anchor_reorder_f32) it moves whole 16-byte blocks and XORs the position inside: code[r][block] = sigma | x << 4.
// The BLOCK move costs nothing here: a lane of the global -> LDS DMA names its own source address, so the piece of block
// beta of entry e is simply fetched from block sigma_r(beta) of the row -- the LDS image of a workgroup holds its own 16
// anchors only, exactly as without permutation (csrc/so3_inter_inv.hip had to keep whole 60-anchor rows on chip: one
// workgroup per CU, one channel tile per weight).  The XOR inside the block is 8 selects per 16-byte operand read.
// Everything that depends on the entry's rotation r alone is prepared per ENTRY by eap_so3_perm_entries_f32 (below), not
// per (entry, workgroup, chunk) in here -- table lookups inside the chunk loop cost 12.6 of 71 ms when they were
// (profiles/r04_lists2_perm_ablation.txt):
//   ent_p  uint32 [entries][4 anchor groups][4 pieces]: byte offset of the piece's source inside a channel row of the cloud
//                              (point * row bytes + 16 * sigma_r(block)); a group's four words are one 16-byte ring entry
//   ent_gx [entries][4]:       offset vector, rotated by A_r in the backward (the weights are then those of the wave's OWN
//                              anchors, see so3_inter_inv.hip), dead (1e18) for shadow neighbours; w = the x bits of all 15
//                              blocks, 2 per block
// The backward's output Z keeps the coset-major anchor order (contiguous 16-byte stores; the caller un-permutes the small
// tensors that follow); the forward's transposed output has one row per anchor and comes back in memory order.
template <bool LISTS, int LAYOUT, bool PERM = false>
__global__ __launch_bounds__(TM, 2) void so3_group_lists2_kernel(
    int C, int PF, int na, int fpitch, int ks, int R, int nn, int ent_stride, int AG, int RPB, int ag_major, int dbg, float inv_sigma,
    const float *__restrict__ F, const int32_t *__restrict__ rows, const int32_t *__restrict__ off,
    const int32_t *__restrict__ cnt, const int32_t *__restrict__ ent_p, const float4 *__restrict__ ent_gx,
    const float *__restrict__ rk, const int32_t *__restrict__ nonident, float *__restrict__ out,
    const uint8_t *__restrict__ order = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- block -> (row run, anchor group, channel slice, cloud); an XCD gets whole (slice, cloud) pairs when their
    //      number allows it, else a contiguous range of rows (whole output lines in one L2) ----
    const int nrun = (R + RPB - 1) / RPB;
    const int ny = gridDim.y, nsl = ny * gridDim.z, per_slice = nrun * AG;
    int qd = blockIdx.x, sl = blockIdx.y + ny * blockIdx.z;
    int run, ag;
    if (ag_major && ((nsl * AG) & 7) == 0) {
        // an XCD owns whole (slice, cloud, ANCHOR GROUP) triples: the rows of a cloud that its resident workgroups walk
        // at the same time are then 64-byte pieces, a quarter of the (slice, cloud) working set per point -- four times
        // as many points of the walk stay in the 4 MB L2, and all 64 resident workgroups (not 16 x 4) share them
        const unsigned lin = blockIdx.x + (unsigned)per_slice * (blockIdx.y + (unsigned)ny * blockIdx.z);
        const unsigned j = lin >> 3;
        const unsigned sl2 = (lin & 7u) + 8u * (j / (unsigned)nrun);
        run = (int)(j % (unsigned)nrun);
        ag = (int)(sl2 % (unsigned)AG);
        sl = (int)(sl2 / (unsigned)AG);
    } else {
        if ((nsl & 7) == 0) {
            const unsigned lin = blockIdx.x + (unsigned)per_slice * (blockIdx.y + (unsigned)ny * blockIdx.z);
            const unsigned j = lin >> 3;
            sl = (int)((lin & 7u) + 8u * (j / (unsigned)per_slice));
            qd = (int)(j % (unsigned)per_slice);
        } else {
            qd = xcd_point(blockIdx.x, per_slice);
        }
        run = qd / AG;
        ag = qd - run * AG;
    }
    const int r_begin = run * RPB, rows_blk = min(RPB, R - r_begin);
    const int cy = sl % ny, bi = sl / ny, c0 = cy * CB;
    if (nonident != nullptr && (__builtin_amdgcn_readfirstlane(nonident[bi]) != 0) != PERM) return;   // permuted clouds: the PERM kernel's, the others: not

    const int t = threadIdx.x, lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lk = lane & 31, lh = lane >> 5;
    const int a0 = ag * GSZ, gcount = min(GSZ, na - a0);      // anchors [a0, a0 + gcount) of this block
    const int npg = gcount >> 2;                               // 16-byte pieces per feature row that exist
    const int al_beg = wave_u * APW;                           // first local anchor of this wave = piece wave_u
    const bool active = al_beg < gcount;                       // wave-uniform

    float *s_f = reinterpret_cast<float *>(smem);                           // [2][NBK][CB][PITCH]
    float4 *s_g = reinterpret_cast<float4 *>(s_f + 2 * NBK * CB * PITCH);   // [3][NBK] ring
    int *s_p = reinterpret_cast<int *>(s_g + 3 * NBK);                      // [3][NBK] ring (PERM: [3][NBK][4], byte offsets of the four pieces)

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

    // memory index of the wave's ai-th anchor (PERM: position a0 + al_beg + ai of the coset-major order)
    int am[APW];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai) {
        const int pos = a0 + min(al_beg + ai, gcount - 1);
        am[ai] = PERM ? __builtin_amdgcn_readfirstlane((int)order[pos]) : pos;
    }
    const int blk_w = (a0 >> 2) + wave_u;                          // PERM: the wave's block of the order
    // ---- per-lane weight constants of this wave's anchors (k = lane & 31): see csrc/so3_inter_lists.hip ----
    f32x2 kxp[APW / 2], kyp[APW / 2], kzp[APW / 2], kcp[APW / 2];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai) {
        const int a = am[ai];
        const float *r3 = rk + ((size_t)a * ks + min(lk, ks - 1)) * 3;
        const float x = r3[0], y = r3[1], z = r3[2];
        kxp[ai >> 1][ai & 1] = 2.f * inv_sigma * x;
        kyp[ai >> 1][ai & 1] = 2.f * inv_sigma * y;
        kzp[ai >> 1][ai & 1] = 2.f * inv_sigma * z;
        kcp[ai >> 1][ai & 1] = lk < ks ? -inv_sigma * (x * x + y * y + z * z) : -1e30f;
    }
    const float kcl = kcp[0][0];
    const bool kc_uniform = __all(fabsf(kcp[0][1] - kcl) <= 1e-6f * fabsf(kcl) && fabsf(kcp[1][0] - kcl) <= 1e-6f * fabsf(kcl) &&
                                  fabsf(kcp[1][1] - kcl) <= 1e-6f * fabsf(kcl)) != 0;
    // operand read: the wave's four anchors are ONE 16-byte piece (piece wave_u) of a channel row; tile 1 = rows + 32
    // (same slot: (32 >> 2) & 3 == 0)
    const float4 *fa_lane = reinterpret_cast<const float4 *>(s_f + (size_t)(lh * CB + lk) * PITCH + 4 * ((wave_u + (lk >> 2)) & 3));
    constexpr int TILE_F4 = 32 * PITCH / 4;            // float4s between the two channel tiles of an entry
    constexpr int STEP_F4 = 2 * CB * PITCH / 4;        // float4s per MFMA k-step (2 entries)
    constexpr int BUF_F4 = NBK * CB * PITCH / 4;

    f32x16 acc[CT][APW];

    // ---- DMA: instruction u of a chunk carries entry u; thread t <-> channel row t >> 2, slot t & 3 ----
    const float *fb = F + ((size_t)bi * C + c0) * PF * fpitch;   // fpitch: floats between consecutive feature rows (>= na)
    const unsigned lds_f = lds_addr(s_f);
    const unsigned row_bytes = (unsigned)fpitch * 4u;
    const int d_row = t >> 2, d_piece = ((t & 3) - (d_row >> 2)) & 3;
    const bool d_valid = d_piece < npg;
    const unsigned dma_off = ((unsigned)(min(c0 + d_row, C - 1) - c0) * (unsigned)PF * (unsigned)fpitch + (PERM ? 0u : (unsigned)(a0 + 4 * min(d_piece, npg - 1)))) * 4u;
    const int d_pc = min(d_piece, npg - 1), blk_t = (a0 >> 2) + d_pc;   // PERM: the block this thread's DMA piece belongs to
    const unsigned lds_g = lds_addr(s_g), lds_p = lds_addr(s_p);
    auto issue_idx = [&](int j0, int slot) {
        if (wave_u == 0 && lane < NBK) {
            const size_t e = e0 + min(j0 + lane, max(n_ent - 1, 0));
            if (PERM) glds16(ent_p + 4 * (4 * e + ag), __builtin_amdgcn_readfirstlane(lds_p + (unsigned)slot * NBK * 16u));
            else glds4(ent_p + e, __builtin_amdgcn_readfirstlane(lds_p + (unsigned)slot * NBK * 4u));
            glds16(ent_gx + e, __builtin_amdgcn_readfirstlane(lds_g + (unsigned)slot * NBK * 16u));
        }
    };
    unsigned src_off[NSTD];
    auto prep_rows = [&](int slot) {
#pragma unroll
        for (int u = 0; u < NSTD; ++u) {
            if (PERM) {                // byte offset of the piece inside the cloud's channel row: point row + its SOURCE block
                src_off[u] = dma_off + (ABL(32) ? (unsigned)s_p[(slot * NBK + u) * 4] + 16u * (unsigned)blk_t : (unsigned)s_p[(slot * NBK + u) * 4 + d_pc]);
                continue;
            }
            int pe = s_p[slot * NBK + u];
            if (ABL(512)) pe = u;                                      // every chunk fetches the same eight rows: cache hits only
            if (!LISTS) pe = (unsigned)pe < (unsigned)PF ? pe : 0;     // shadow row: any valid row, weight 0
            src_off[u] = dma_off + __umul24((unsigned)pe, row_bytes);
        }
    };
    auto issue = [&](int u, int buf) {
        if (d_valid && !ABL(1))
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

    auto chunk_bases = [&](int ch, int gslot) {
        const int e = lane & (NBK - 1);
        const float4 g = s_g[gslot * NBK + e];
        float b = 1.0f - inv_sigma * (g.x * g.x + g.y * g.y + g.z * g.z);
        bool dead = ch * NBK + e >= n_ent;
        if (!LISTS && !PERM) dead = dead || (unsigned)s_p[gslot * NBK + e] >= (unsigned)PF;     // (PERM: shadow entries carry a dead offset vector)
        return __float_as_int(dead ? -1e30f : b);
    };
    // PERM: position j of the wave's block takes the value at position j ^ x of the source block
    auto unxor = [&](float4 &v, int x) {
        const bool x0 = (x & 1) != 0, x1 = (x & 2) != 0;
        const float a0_ = x0 ? v.y : v.x, a1_ = x0 ? v.x : v.y, a2_ = x0 ? v.w : v.z, a3_ = x0 ? v.z : v.w;
        v = make_float4(x1 ? a2_ : a0_, x1 ? a3_ : a1_, x1 ? a0_ : a2_, x1 ? a1_ : a3_);
    };
    auto gather = [&](const float4 *fab, int gslot, int bases, int s, float4 &fa, float4 &fb1, float4 &g, float &bk) {
        if (ABL(16)) {
            fa = make_float4(1.f, 2.f, 3.f, 4.f); fb1 = fa; g = make_float4(0.01f * (float)s, 0.02f, 0.03f, 0.f); bk = 0.5f;
            return;
        }
        fa = fab[s * STEP_F4];
        fb1 = fab[s * STEP_F4 + TILE_F4];
        g = s_g[gslot * NBK + 2 * s + lh];
        bk = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (2 * s + lh), bases));
        if (PERM && !ABL(64)) {
            const int x = __float_as_int(g.w) >> (2 * blk_w);        // the entry's x bits, 2 per block
            unxor(fa, x);
            unxor(fb1, x);
        }
    };
    auto nothing = [] {};
    // one MFMA k-step (2 entries): 4 weights per lane, 8 matrix instructions; q0..q3 run after MFMA pairs 1..4
    auto step = [&](auto kcu, const float4 g, float bk, const float4 fv0, const float4 fv1, auto q0, auto q1, auto q2, auto q3,
                    bool first = false) {
        const float fa0[APW] = {fv0.x, fv0.y, fv0.z, fv0.w};
        const float fa1[APW] = {fv1.x, fv1.y, fv1.z, fv1.w};
        f32x2 wv[APW / 2];
        const float bkc = bk + kcl;
        if (ABL(2)) {
            wv[0] = kxp[0];
            wv[1] = kxp[1];
        } else {
#pragma unroll
            for (int j = 0; j < APW / 2; ++j) {
                f32x2 x = __builtin_elementwise_fma((f32x2){g.x, g.x}, kxp[j], decltype(kcu)::value ? (f32x2){bkc, bkc} : kcp[j] + (f32x2){bk, bk});
                x = __builtin_elementwise_fma((f32x2){g.y, g.y}, kyp[j], x);
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp\n\ts_nop 1" : "=v"(wv[j]) : "v"((f32x2){g.z, g.w}), "v"(kzp[j]), "v"(x));
            }
        }
        __builtin_amdgcn_s_setprio(3);
        if (first) {
            const f32x16 zc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[0][0] = EAP_MM(fa0[0], wv[0][0], zc, 0, 0, 0);
            acc[0][1] = EAP_MM(fa0[1], wv[0][1], zc, 0, 0, 0);
            q0();
            acc[0][2] = EAP_MM(fa0[2], wv[1][0], zc, 0, 0, 0);
            acc[0][3] = EAP_MM(fa0[3], wv[1][1], zc, 0, 0, 0);
            q1();
            acc[1][0] = EAP_MM(fa1[0], wv[0][0], zc, 0, 0, 0);
            acc[1][1] = EAP_MM(fa1[1], wv[0][1], zc, 0, 0, 0);
            q2();
            acc[1][2] = EAP_MM(fa1[2], wv[1][0], zc, 0, 0, 0);
            acc[1][3] = EAP_MM(fa1[3], wv[1][1], zc, 0, 0, 0);
        } else {
            acc[0][0] = EAP_MM(fa0[0], wv[0][0], acc[0][0], 0, 0, 0);
            acc[0][1] = EAP_MM(fa0[1], wv[0][1], acc[0][1], 0, 0, 0);
            q0();
            acc[0][2] = EAP_MM(fa0[2], wv[1][0], acc[0][2], 0, 0, 0);
            acc[0][3] = EAP_MM(fa0[3], wv[1][1], acc[0][3], 0, 0, 0);
            q1();
            acc[1][0] = EAP_MM(fa1[0], wv[0][0], acc[1][0], 0, 0, 0);
            acc[1][1] = EAP_MM(fa1[1], wv[0][1], acc[1][1], 0, 0, 0);
            q2();
            acc[1][2] = EAP_MM(fa1[2], wv[1][0], acc[1][2], 0, 0, 0);
            acc[1][3] = EAP_MM(fa1[3], wv[1][1], acc[1][3], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        q3();
    };

    // Row end: accumulators straight to global memory.  D[i = channel][j = kernel point] sits as col = lane&31,
    // row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Wave-uniform 64-bit base per store + ONE 32-bit per-lane offset.
    const size_t o_ks = (size_t)R * na, o_cs = (size_t)ks * R * na;
    float *ob = out + (size_t)bi * C * o_cs + (size_t)c0 * o_cs + a0;
    const unsigned lane_off = (unsigned)((size_t)(4 * lh) * o_cs + (size_t)min(lk, ks - 1) * o_ks) + (unsigned)al_beg;
    float *obb = out + (size_t)bi * C * o_cs;
    auto store_row = [&](int row) {
        if (LAYOUT >= 3) {
            // D[i = kernel point][j = channel]: lane column = channel lk of the tile, accumulator quad q holds kernel points
            // 8 q + 4 lh .. + 3 -- 16 contiguous bytes of out[b][row*na + a][c*ks + k].
            // (Every instruction is 64 separate 16-byte requests, 96 bytes apart; the same bytes written in address order take
            // 11.6 instead of 12.6 ms (ABL 1024).  Bringing them into address order through a wave-private LDS scratch -- 24
            // ds_write_b128 + 24 ds_read_b128 per row end, pipelined over the (tile, anchor) pairs -- measured 12.8 ms: the wave
            // waits out the LDS round trips instead of issuing matrix instructions.  profiles/r04_row_order_experiment.txt)
            if (active && !ABL(4)) {
                const size_t CK = (size_t)C * ks;
                float *rb = obb + ((size_t)row * na + (PERM ? 0 : a0 + al_beg)) * CK + (size_t)c0 * ks;      // uniform
                const unsigned lo_b = (unsigned)(lk * ks + 4 * lh) * 4u;
                asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");       // the accumulators were last written by MFMAs the asm cannot see
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const bool whole = c0 + 32 * ct + 32 <= C;            // block-uniform
                    if (whole || c0 + 32 * ct + lk < C) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (8 * q >= ks || 8 * q + 4 * lh >= ks) continue;       // (first test: uniform, drops the padding quad)
#pragma unroll
                            for (int ai = 0; ai < APW; ++ai) {
                                const f32x4 v = {acc[ct][ai][4 * q], acc[ct][ai][4 * q + 1], acc[ct][ai][4 * q + 2], acc[ct][ai][4 * q + 3]};
                                if (LAYOUT == 4 || ABL(1024))     // lanes in address order: 1 KB runs (LAYOUT 3 under ABL 1024: wrong places)
                                    asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"((unsigned)lane * 16u), "v"(v),
                                                 "s"(rb + (size_t)(PERM ? am[ai] : ai) * CK + (size_t)(32 * ct) * ks + 256 * q) : "memory");
                                else
                                asm volatile("global_store_dwordx4 %0, %1, %2" : : "v"(lo_b), "v"(v),
                                             "s"(rb + (size_t)(PERM ? am[ai] : ai) * CK + (size_t)(32 * ct) * ks + 8 * q) : "memory");
                            }
                        }
                    }
                }
            }
            return;
        }
        if (active && lk < ks && !ABL(4)) {
            if (LAYOUT == 2) {
                // transposed output out[b][row*na + a][c*ks + k] (the plain [P*A, C*K] matrix the contraction GEMM reads)
                const size_t CK = (size_t)C * ks;
                float *rb = obb + ((size_t)row * na + (PERM ? 0 : a0 + al_beg)) * CK + (size_t)c0 * ks;      // uniform
                const unsigned lo_b = (unsigned)((4 * lh) * ks + lk) * 4u;
                asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");       // the accumulators were last written by MFMAs the asm cannot see
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (c0 + 32 * ct + 32 <= C) {            // block-uniform
#pragma unroll
                        for (int r = 0; r < 16; ++r)
#pragma unroll
                            for (int ai = 0; ai < APW; ++ai)
                                asm volatile("global_store_dword %0, %1, %2" : : "v"(lo_b), "v"(acc[ct][ai][r]),
                                             "s"(rb + (size_t)(PERM ? am[ai] : ai) * CK + (size_t)(32 * ct + (r & 3) + 8 * (r >> 2)) * ks) : "memory");
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (c0 + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lh < C) {
#pragma unroll
                                for (int ai = 0; ai < APW; ++ai)
                                    *reinterpret_cast<float *>(reinterpret_cast<char *>(rb + (size_t)(PERM ? am[ai] : ai) * CK + (size_t)(32 * ct + (r & 3) + 8 * (r >> 2)) * ks) + lo_b) = acc[ct][ai][r];
                            }
                    }
                }
            } else {
                float *rb = ob + (size_t)row * na;             // uniform
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (c0 + 32 * ct + 32 <= C) {
                        // (sc1 write-through stores measured slower: 43.3 vs 42.2 ms, profiles/r03_store_flavour_experiment.txt)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            *reinterpret_cast<float4 *>(rb + (size_t)(32 * ct + (r & 3) + 8 * (r >> 2)) * o_cs + lane_off) =
                                make_float4(acc[ct][0][r], acc[ct][1][r], acc[ct][2][r], acc[ct][3][r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (c0 + 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lh < C)
                                *reinterpret_cast<float4 *>(rb + (size_t)(32 * ct + (r & 3) + 8 * (r >> 2)) * o_cs + lane_off) =
                                    make_float4(acc[ct][0][r], acc[ct][1][r], acc[ct][2][r], acc[ct][3][r]);
                    }
                }
            }
        }
    };

    int g0 = 0, g1 = 1, g2 = 2;                           // ring slots of chunks ch, ch+1, ch+2
    int ch_row = 0, row = r_begin;
    if (active) {
#define EAP_LISTS2_LOOP(kcu)                                                                                          \
        for (int ch = 0; ch < nchunk; ++ch) {                                                                         \
            const int buf = ch & 1, nb = buf ^ 1;                                                                     \
            const float4 *fbuf = fa_lane + buf * BUF_F4;                                                              \
            float4 fa0, fb0, fa1, fb1, ga, gb;                                                                        \
            float ba, bb;                                                                                             \
            const int bases = chunk_bases(ch, g0);                                                                    \
            gather(fbuf, g0, bases, 0, fa0, fb0, ga, ba);                                                             \
            gather(fbuf, g0, bases, 1, fa1, fb1, gb, bb);                                                             \
            prep_rows(g1);                                                                                            \
            issue_idx((ch + 2) * NBK, g2);                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            step(kcu, ga, ba, fa0, fb0, [&] { issue(0, nb); }, [&] { issue(1, nb); }, [&] { issue(2, nb); }, nothing, ch_row == 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            gather(fbuf, g0, bases, 2, fa0, fb0, ga, ba);                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            step(kcu, gb, bb, fa1, fb1, [&] { issue(3, nb); }, [&] { issue(4, nb); }, [&] { issue(5, nb); }, nothing); \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            gather(fbuf, g0, bases, 3, fa1, fb1, gb, bb);                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            step(kcu, ga, ba, fa0, fb0, [&] { issue(6, nb); }, [&] { issue(7, nb); }, nothing, nothing);              \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            step(kcu, gb, bb, fa1, fb1, nothing, nothing, nothing, nothing);                                          \
            dma_wait();                                                                                               \
            if (++ch_row == nchunk_row) {                                                                             \
                store_row(row);                                                                                       \
                ch_row = 0;                                                                                           \
                ++row;                                                                                                \
            }                                                                                                         \
            if (!ABL(8)) __syncthreads();                                                                             \
            const int gt = g0; g0 = g1; g1 = g2; g2 = gt;                                                             \
        }                                                                                                             \
        if (nchunk == 0) {                                                                                            \
            _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                                         \
                _Pragma("unroll") for (int ai = 0; ai < APW; ++ai)                                                    \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[ct][ai][r] = 0.f;                              \
            store_row(r_begin);                                                                                       \
        }

        if (kc_uniform) { EAP_LISTS2_LOOP(std::true_type{}); } else { EAP_LISTS2_LOOP(std::false_type{}); }
#undef EAP_LISTS2_LOOP
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
constexpr size_t SHMEM_PERM = 2 * BUF_BYTES + 16 * 3 * NBK + 16 * 3 * NBK;

int g_xcd_map_fwd = 1, g_xcd_map_inv = 1;       // eap_so3_group_lists_xcd_map
int g_store16 = 1;                              // eap_so3_group_lists_store16

int g_perm_lists2 = 1;                          // eap_so3_group_perm_lists2

template <bool LISTS>
int launch2(int layout, int b, int C, int PF, int na, int fpitch, int ks, int R, int nn, int ent_stride, float sigma, const float *F,
            const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p, const float *ent_gx,
            const float *rk, const int32_t *nonident, float *out, hipStream_t s, const char *what, const uint8_t *order = nullptr) {
    if (fpitch < na || (fpitch & 3) != 0) return eap::bad_arg("so3_group_lists2: the feature row pitch must be a multiple of 4, at least the anchor count");
    if ((long long)CB * PF * fpitch * 4 >= (1ll << 32) || PF >= (1 << 24) || fpitch * 4 >= (1 << 24))
        return eap::bad_arg("so3_group_lists2: 64 feature rows of a cloud exceed the 32-bit request offsets");
    if (((long long)ks * R * na * 4 + 64ll * R * na + 64) * 4 >= (1ll << 31) || (long long)CB * ks * 4 >= (1ll << 31))
        return eap::bad_arg("so3_group_lists2: output rows too far apart for 32-bit store offsets");
    const bool perm = order != nullptr;          // ent_p / ent_gx are then the per-entry words of eap_so3_perm_entries_f32
    if (perm && ((na & 3) != 0 || fpitch != na)) return eap::bad_arg("so3_group_lists2: the permuted variant takes unpadded rows of a multiple of 4 anchors");
    const bool lane_order = !LISTS && layout == 4;
    if (lane_order && ((ks & 7) != 0 || (C % CB) != 0)) return eap::bad_arg("so3_group_lists2: the store-order columns need ks % 8 == 0 and whole 64-channel blocks");
    const bool wide = !LISTS && layout == 2 && g_store16 != 0 && (ks & 3) == 0;
    auto kern = perm ? (lane_order ? so3_group_lists2_kernel<LISTS, LISTS ? 0 : 4, true> : wide ? so3_group_lists2_kernel<LISTS, LISTS ? 0 : 3, true> : layout == 2 ? so3_group_lists2_kernel<LISTS, LISTS ? 0 : 2, true> : so3_group_lists2_kernel<LISTS, 0, true>)
                     : (lane_order ? so3_group_lists2_kernel<LISTS, LISTS ? 0 : 4> : wide ? so3_group_lists2_kernel<LISTS, LISTS ? 0 : 3> : layout == 2 ? so3_group_lists2_kernel<LISTS, LISTS ? 0 : 2> : so3_group_lists2_kernel<LISTS, 0>);
    const size_t shmem = perm ? SHMEM_PERM : SHMEM;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem), what);
    if (e) return e;
    const int AG = (na + GSZ - 1) / GSZ;
    const int RPB = LISTS ? 1 : ((nn % NBK) == 0 ? 8 : 1);
    dim3 grid((R + RPB - 1) / RPB * AG, (C + CB - 1) / CB, b);
#ifdef EAP_ABLATION
    const int dbg = getenv("EAP_LISTS2_DEBUG") ? atoi(getenv("EAP_LISTS2_DEBUG")) : 0;
#else
    const int dbg = 0;
#endif
    hipLaunchKernelGGL(kern, grid, dim3(TM), shmem, s, C, PF, na, fpitch, ks, R, nn, ent_stride, AG, RPB, (LISTS ? g_xcd_map_inv : g_xcd_map_fwd) == 2, dbg, 1.0f / sigma, F,
                       rows, off, cnt, ent_p, reinterpret_cast<const float4 *>(ent_gx), rk, nonident, out, order);
    if (lane_order) eap::set_kernel(perm ? "so3_group_lists2_kernel<false, 4, true>" : "so3_group_lists2_kernel<false, 4>");
    else
    eap::set_kernel(perm ? (LISTS ? "so3_group_lists2_kernel<true, 0, true>" : wide ? "so3_group_lists2_kernel<false, 3, true>" : layout == 2 ? "so3_group_lists2_kernel<false, 2, true>" : "so3_group_lists2_kernel<false, 0, true>")
                         : (LISTS ? "so3_group_lists2_kernel<true, 0>" : wide ? "so3_group_lists2_kernel<false, 3>" : layout == 2 ? "so3_group_lists2_kernel<false, 2>" : "so3_group_lists2_kernel<false, 0>"));
    return eap::check_launch(what);
}

}  // namespace

static int g_tiles = 2;       // eap_so3_group_lists_tiles

// A/B and test switch: 1 = always the one-tile kernel of csrc/so3_inter_lists.hip, 2 = two tiles where they pay (default),
// (3 = the 3 x bf16 split kernel of tools/experiments/kernels/so3_inter_lists3.hip, in `make EXPERIMENTS=1` builds only: measured
// slower on real neighbour lists, the grouping is bound by the L2 -> LDS gather, not by the matrix pipe); 0 = query.  Returns the value in force.  Process-wide, not thread-safe (set it before launching).
extern "C" int eap_so3_group_lists_tiles(int tiles) {
#ifdef EAP_EXPERIMENTS
    if (tiles >= 1 && tiles <= 4) g_tiles = tiles;
#else
    if (tiles >= 1 && tiles <= 2) g_tiles = tiles;
#endif
    return g_tiles;
}

// Row end of the forward kernel writing the transposed intermediate: 1 (default) = exchanged MFMA operands, 16-byte stores
// (LAYOUT 3); 0 = dword stores in 96-byte runs (LAYOUT 2).  Bit-identical results (the same products summed in the same
// order).  Returns the previous setting; other values only query.
extern "C" int eap_so3_group_lists_store16(int on) {
    const int was = g_store16;
    if (on == 0 || on == 1) g_store16 = on;
    return was;
}

// Clouds with anchor permutations on the two-tile kernel (PERM; default 1) or on csrc/so3_inter_inv.hip's whole-row kernel (0),
// for A/B runs.  Returns the previous setting; other values only query.
extern "C" int eap_so3_group_perm_lists2(int on) {
    const int was = g_perm_lists2;
    if (on == 0 || on == 1) g_perm_lists2 = on;
    return was;
}

// Which unit of work an XCD (one L2) owns in the two-tile kernel: 1 = whole (channel slice, cloud) pairs, 2 = whole
// (channel slice, cloud, anchor group) triples.  `which` 0 = forward, 1 = backward (inverse lists); mode 0 = query.
extern "C" int eap_so3_group_lists_xcd_map(int which, int mode) {
    int &m = which ? g_xcd_map_inv : g_xcd_map_fwd;
    if (mode == 1 || mode == 2) m = mode;
    return m;
}

namespace eap {

// two channel tiles per wave pay when the channel count fills (most of) the 64-channel blocks; layout 1 (blocked by
// anchor quads) stays with the one-tile kernel
bool group_lists2_preferred(int c, int na, int ks, int layout) {
    if (g_tiles < 2 || na <= 0 || (na & 3) != 0 || na > 64 || ks <= 0 || ks > 32 || layout == 1) return false;
    const int rem = c % CB;
    return c >= CB && (rem == 0 || rem > 32);
}

#ifdef EAP_EXPERIMENTS
// mode 3 (eap_so3_group_lists_tiles, `make EXPERIMENTS=1` builds only): the 3 x bf16 split kernel of
// tools/experiments/kernels/so3_inter_lists3.hip wherever the two-tile kernel would run
bool group_lists3_preferred(int c, int na, int ks, int layout) { return g_tiles == 3 && group_lists2_preferred(c, na, ks, layout); }
// mode 4: the two-fp16-plane kernel of tools/experiments/kernels/so3_inter_lists_h2.hip
bool group_listsh_preferred(int c, int na, int ks, int layout) { return g_tiles == 4 && group_lists2_preferred(c, na, ks, layout); }
#endif

int group_lists2_fwd(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                     const int32_t *idx, const float *gx, const float *rk, const int32_t *nonident, int layout, float *out,
                     hipStream_t s) {
    return launch2<false>(layout, b, c, n, na, na, ks, p, nn, 0, sigma, feats, nullptr, nullptr, nullptr, idx, gx, rk, nonident, out, s,
                          "so3_inter_group_fwd (lists, two channel tiles)");
}

int group_lists2_inv(int b, int o, int p, int nn, int na, int gy_pitch, int ks, int rcap, float sigma, const float *gy,
                     const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_p,
                     const float *ent_gx, const float *rk, float *z, hipStream_t s) {
    return launch2<true>(0, b, o, p, na, gy_pitch, ks, rcap, nn, p * nn, sigma, gy, rows, off, cnt, ent_p, ent_gx, rk, nullptr, z, s,
                         "so3_inter_group_inv (lists, two channel tiles)");
}

}  // namespace eap

// ---- clouds with anchor permutations on the two-tile kernel (PERM) ---------------------------------------------------
namespace {
// one thread per entry: everything the PERM kernel needs that depends on the entry's rotation r alone (see the kernel's header)
__global__ __launch_bounds__(256) void perm_entries_kernel(long long total, int per_cloud, const int32_t *__restrict__ nonident, int na, int PF, int ident, int rotate,
                                                           const int32_t *__restrict__ ent_p, const float4 *__restrict__ ent_gx,
                                                           const uint8_t *__restrict__ code, const float *__restrict__ anchors,
                                                           uint4 *__restrict__ ent_off, float4 *__restrict__ ent_gx2) {
    __shared__ uint32_t s_code[64 * 4];
    __shared__ float s_A[64 * 9];
    for (int i = threadIdx.x; i < na * 4; i += 256) s_code[i] = reinterpret_cast<const uint32_t *>(code)[i];
    if (rotate) for (int i = threadIdx.x; i < na * 9; i += 256) s_A[i] = anchors[i];
    __syncthreads();
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total || (nonident != nullptr && nonident[i / per_cloud] == 0)) return;      // (clouds without permutations: not needed)
    float4 g = ent_gx[i];
    const int p = ent_p[i];
    int r = __float_as_int(g.w);
    r = (unsigned)r < (unsigned)na ? r : ident;
    if (rotate && r != ident && g.x < 1e17f) {
        const float *A = s_A + 9 * r;
        g = make_float4(A[0] * g.x + A[1] * g.y + A[2] * g.z, A[3] * g.x + A[4] * g.y + A[5] * g.z, A[6] * g.x + A[7] * g.y + A[8] * g.z, 0.f);
    }
    const bool shadow = (unsigned)p >= (unsigned)PF;                 // forward: a neighbour slot without a point
    const unsigned row = shadow ? 0u : (unsigned)p * (unsigned)na * 4u;
    if (shadow) g = make_float4(1e18f, 1e18f, 1e18f, 0.f);
    unsigned xb = 0;
#pragma unroll
    for (int ag = 0; ag < 4; ++ag) {
        const uint32_t w = s_code[4 * r + ag];                       // code bytes of blocks 4 ag .. 4 ag + 3: sigma | x << 4
        ent_off[4 * i + ag] = make_uint4(row + 16u * (w & 15u), row + 16u * ((w >> 8) & 15u), row + 16u * ((w >> 16) & 15u), row + 16u * ((w >> 24) & 15u));
        xb |= (((w >> 4) & 3u) | ((w >> 12) & 3u) << 2 | ((w >> 20) & 3u) << 4 | ((w >> 28) & 3u) << 6) << (8 * ag);
    }
    g.w = __uint_as_float(xb);
    ent_gx2[i] = g;
}
}  // namespace

// 1 if clouds with anchor permutations and this shape go to the two-tile kernel (the caller then prepares the per-entry words
// and the coset-major operand), 0: csrc/so3_inter_inv.hip's whole-row kernel takes them
extern "C" int eap_so3_group_perm_lists2_takes(int channels, int na, int ks, int n_support) {
    return g_perm_lists2 && (na & 3) == 0 && na <= 60 && (long long)n_support * na * 4 < (1ll << 31) && eap::group_lists_supported(na, ks) &&
           eap::group_lists2_preferred(channels, na, ks, 0);
}

// per-entry words of the PERM kernel.  ent_p int32 [b*per_cloud] (inverse lists: query point of the entry; forward: idx),
// ent_gx [b*per_cloud,4] (offset vector, w = bits of the relative-rotation anchor r), code uint8 [na,16] = coset code table
// of the permutation table in force (row-wise inverse of the multiplication table in the backward), anchors [na,3,3] or NULL:
// offset vectors rotated by A_r (backward); nonident int32 [b] or NULL: clouds whose flag is 0 are skipped (their words stay
// unwritten).  -> ent_pc uint32 [b*per_cloud,4,4], ent_gx2 [b*per_cloud,4].
extern "C" int eap_so3_perm_entries_f32(int b, int per_cloud, int na, int n_support, const int32_t *ent_p, const float *ent_gx,
                                        const uint8_t *code, const float *anchors, int identity_anchor, const int32_t *nonident,
                                        int32_t *ent_pc, float *ent_gx2, eap_stream_t stream) {
    if (b <= 0 || per_cloud <= 0) return 0;
    if ((na & 3) != 0 || na > 60 || (long long)n_support * na * 4 >= (1ll << 31) || (reinterpret_cast<uintptr_t>(code) & 3) != 0)
        return eap::bad_arg("so3_perm_entries: na a multiple of 4 up to 60, point rows within 2^31 bytes, code table 4-byte aligned");
    const long long total = (long long)b * per_cloud;
    hipLaunchKernelGGL(perm_entries_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, eap::S(stream), total, per_cloud, nonident, na, n_support, identity_anchor,
                       anchors != nullptr, ent_p, reinterpret_cast<const float4 *>(ent_gx), code, anchors, reinterpret_cast<uint4 *>(ent_pc),
                       reinterpret_cast<float4 *>(ent_gx2));
    return eap::check_launch("so3_perm_entries");
}

// Z of the re-associated backward (csrc/so3_inter_inv.hip) for clouds WITH anchor permutations, on the two-tile kernel.
// gy [b,o,p,na] with its anchor axis COSET-MAJOR (eap_anchor_reorder_f32 with `order`), inverse lists as for
// eap_so3_inter_group_inv_f32 with ent_pc / ent_gx2 from eap_so3_perm_entries_f32 (code table of the row-wise inverse of the
// multiplication table, offset vectors rotated); z [b,o,ks,rcap,na] comes back with its anchor axis COSET-MAJOR too.
extern "C" int eap_so3_inter_group_inv_perm2_f32(int b, int o, int p, int nn, int na, int ks, int rcap, float sigma, const float *gy,
                                                 const int32_t *rows, const int32_t *off, const int32_t *cnt, const int32_t *ent_pc,
                                                 const float *ent_gx2, const float *rk, const uint8_t *order, float *z, eap_stream_t stream) {
    if (b <= 0 || o <= 0 || rcap <= 0) return 0;
    if (!order || !eap_so3_group_perm_lists2_takes(o, na, ks, p)) return eap::bad_arg("so3_inter_group_inv_perm2: shape not taken (ask eap_so3_group_perm_lists2_takes)");
    return launch2<true>(0, b, o, p, na, na, ks, rcap, nn, p * nn, sigma, gy, rows, off, cnt, ent_pc, ent_gx2, rk, nullptr, z, eap::S(stream),
                         "so3_inter_group_inv (permuted clouds, two channel tiles)", order);
}

// ---- transposed output with the columns in store order (LAYOUT 4) -------------------------------------------------------
// 1 if eap_so3_inter_group_fwd_tp_f32 takes this shape
extern "C" int eap_so3_group_fwd_tp_takes(int c, int na, int ks) {
    return g_store16 != 0 && (ks & 7) == 0 && ks <= 32 && (c % CB) == 0 && eap::group_lists_supported(na, ks) && eap::group_lists2_preferred(c, na, ks, 2);
}

// the column order: perm[j] (host array, c*ks entries) = the column c_*ks + k of the plain transposed matrix that sits at
// position j of a row -- per 32-channel tile t: j = t*32*ks + ((q*2 + lh)*32 + lk)*4 + i  <->  channel 32 t + lk, kernel
// point 8 q + 4 lh + i.  A contraction reads W[:, perm] beside this matrix.
extern "C" int eap_so3_group_fwd_tp_columns(int c, int ks, int32_t *perm) {
    if (c <= 0 || ks <= 0 || (ks & 7) != 0 || (c & 31) != 0 || !perm) return eap::bad_arg("so3_group_fwd_tp_columns: ks % 8 == 0, c % 32 == 0");
    for (int t = 0; t < c / 32; ++t)
        for (int q = 0; q < ks / 8; ++q)
            for (int lh = 0; lh < 2; ++lh)
                for (int lk = 0; lk < 32; ++lk)
                    for (int i = 0; i < 4; ++i)
                        perm[(size_t)t * 32 * ks + ((q * 2 + lh) * 32 + lk) * 4 + i] = (32 * t + lk) * ks + 8 * q + 4 * lh + i;
    return 0;
}

// eap_so3_inter_group_fwd_t_f32 for clouds WITHOUT anchor permutations, columns in store order (above): out [b][p*na + a][c*ks]
extern "C" int eap_so3_inter_group_fwd_tp_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                                              const int32_t *idx, const float *gx, const float *rk, float *out, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || p <= 0) return 0;
    if (nn <= 0 || !eap_so3_group_fwd_tp_takes(c, na, ks) || (long long)c * n * na >= (1ll << 31))
        return eap::bad_arg("so3_inter_group_fwd_tp: shape not taken (ask eap_so3_group_fwd_tp_takes)");
    return eap::group_lists2_fwd(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, nullptr, 4, out, eap::S(stream));
}

// Forward grouping, transposed output X^T [b][p*na + a][c*ks + k] (eap_so3_inter_group_fwd_t_f32), with the clouds WITH
// anchor permutations (nonident[b] != 0) on the two-tile kernel: feats_c = feats with a coset-major anchor axis
// (eap_anchor_reorder_clouds_f32 with the `order` of the multiplication table's coset tables), ent_pc / ent_gx2 from
// eap_so3_perm_entries_f32 over (idx, gx) with that table's code and anchors = NULL.  Clouds without permutations take the
// plain kernel on feats / idx / gx as before.  The output is in memory order (one row per anchor); store_order_columns != 0:
// its columns as eap_so3_inter_group_fwd_tp_f32 writes them.
extern "C" int eap_so3_inter_group_fwd_perm2_t_f32(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                                                   const float *feats_c, const int32_t *idx, const float *gx, const int32_t *ent_pc,
                                                   const float *ent_gx2, const float *rk, const uint8_t *order, const int32_t *nonident,
                                                   int store_order_columns, float *out, eap_stream_t stream) {
    if (b <= 0 || c <= 0 || p <= 0) return 0;
    if (store_order_columns && !eap_so3_group_fwd_tp_takes(c, na, ks)) return eap::bad_arg("so3_inter_group_fwd_perm2_t: store-order columns not available for this shape");
    const int lay = store_order_columns ? 4 : 2;
    if (!order || !nonident || nn <= 0 || !eap_so3_group_perm_lists2_takes(c, na, ks, n) || (long long)c * n * na >= (1ll << 31))
        return eap::bad_arg("so3_inter_group_fwd_perm2_t: shape not taken (ask eap_so3_group_perm_lists2_takes; flags and order are required)");
    int e = eap::group_lists2_fwd(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, nonident, lay, out, eap::S(stream));
    if (e) return e;
    e = launch2<false>(lay, b, c, n, na, na, ks, p, nn, 0, sigma, feats_c, nullptr, nullptr, nullptr, ent_pc, ent_gx2, rk, nonident, out, eap::S(stream),
                       "so3_inter_group_fwd (permuted clouds, two channel tiles)", order);
    // (which of the two launches did the work is only known on the device: the per-cloud flags)
    eap::set_kernel(store_order_columns ? "so3_group_lists2_kernel<false, 4> | <false, 4, true>"
                    : g_store16 != 0 && (ks & 3) == 0 ? "so3_group_lists2_kernel<false, 3> | <false, 3, true>" : "so3_group_lists2_kernel<false, 2> | <false, 2, true>");
    return e;
}
