// csrc/zpconv_bwd_hot.hip -- the native inter "zpconv" backward (zpconv_cuda.cpp:L58-75, kernel
// zpconv_cuda_kernel.cu:L77-116) with the scatter target held ON CHIP: no per-(point, neighbour) intermediate.
//
//   gfeats[b,c,q,a] = sum over (p,n) with idx[b,p,.,.,n] == q of  sum_k w[b,p,a,k,n] * grad[b,c,k,p,a]
//
// csrc/zpconv_bwd.hip forms the products T[b,p,a,n,c] = sum_k grad w in forward order (everything of a point read once),
// WRITES them (4 GB per cloud at 4096 points, C = 64) and sums them over device-built inverse lists: 3.65 x the op's
// algorithmic bytes.  The reference's neighbour lists (first nsample hits in index order inside a ball,
// grouping_cuda_kernel.cu:L68-113) reference few support rows when the ball is large -- ~280 of 4096 at the second
// layer's radius -- so the scatter target of a cloud, [rows x 60 anchors x C] floats (4.3 MB at 280 rows, C = 64), fits the
// chip's LDS when it is spread over 30 workgroups:
//
//   workgroup = (cloud, point range, anchor QUAD, 32 channels), one per CU (all 160 KB of LDS), 8 waves = (anchor of the
//              quad, half of the point's 64 neighbours), two per SIMD;
//   LDS      = acc[row slot][4 anchors][32 channels] (512 B per referenced row, at most 290 rows + a dump row for
//              out-of-range indices) + one operand stage (12.25 KB), two when the rows leave room (<= 266); 2 KB of the CU's 160 stay free
//              for the index check's workgroups, which stream beside this kernel;
//   per point and wave: T[32 n, 32 c] = sum_k w[p,a,k,n] grad[c,k,p,a] as 12 v_mfma_f32_32x32x2_f32 (M = neighbours,
//              N = channels, K = kernel points: no padding), then acc[slot(idx[p,n])][a][c] += T as read - add - write;
//              lanes run along the channels (conflict-free).  A word receives at most ONE contribution per point -- a list
//              names a support row once; checked on the device, a cloud whose lists repeat a row (the ball query pads short
//              lists with their first hit) is reported and left to the other path -- and the workgroup meets at a barrier
//              between points, so every word sums its points in order: bit-reproducible, no atomics anywhere.  (LDS float
//              atomics were the first version: ds_add_f32 costs ~800 cycles per wave-instruction on this part, 59 ms
//              against 9.5; profiles/r04_zpconv_bwd_hot_ablation.txt.)
//   w        streams straight into registers (4 bytes per lane and kernel point, the wave's 32 neighbours), two points
//              ahead; grad[c,k,p,4 anchors] arrives as 16-byte pieces (one piece serves the quad's four anchors), two points
//              ahead in registers, then through the LDS stage [anchor][k][c];
//   end      the workgroup writes its rows to gfeats (16-byte stores along the anchors), or, when a cloud's points
//              are split over several workgroups to fill the chip (batches below 8 clouds), to a partial buffer that a
//              second kernel sums in a fixed order.
// w is read by the two channel halves of a quad (2 x), everything else once: 6.1 GB per cloud instead of 16.8.  The 5-D index
// check (the op's 12 GB index read) streams on a side stream beside these kernels.
// A cloud whose referenced rows do not fit, whose lists repeat a row, or whose 5-D index is not one list per point is
// reported in `status` and left to csrc/zpconv_bwd.hip.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int AQ = 4;                         // anchors per workgroup = waves
constexpr int CH = 32;                        // channels per workgroup = MFMA N
constexpr int KS = 24, NN = 64, KS2 = KS / 2;
constexpr int NT = 2;                         // neighbour halves = MFMA M tiles of a point: one wave each
constexpr int TM = 64 * AQ * NT;              // 8 waves: (anchor, neighbour half), two per SIMD
constexpr int ROWB = AQ * CH * 4;             // bytes of accumulators per referenced row
constexpr int STAGE_G = AQ * KS * CH * 4;     // grad stage [anchor][k][c]
constexpr int STAGE_S = NN * 4;               // byte offsets of the accumulator rows of the point's 64 neighbours
constexpr int LDS_BYTES = 158 * 1024;      // 2 KB stay free on every CU: the index check's workgroups run BESIDE this kernel
constexpr int RCAP = (LDS_BYTES - STAGE_G - STAGE_S) / ROWB - 1;      // referenced rows a cloud may have (+ one dump row)

// Timing ablations (WRONG RESULTS), compiled only with `make ABLATION=1` and selected by EAP_ZPHOT_DEBUG (bit mask): 1 no LDS
// accumulation, 2 no grad requests after the prologue, 4 no weight requests after the prologue, 8 no matrix instructions,
// 16 no flush, 32 no point loop, 64 no barriers in the loop, 128 no staging writes
#ifdef EAP_ABLATION
#define ABL(bit) ((dbg & (bit)) != 0)
#else
#define ABL(bit) false
#endif

template <typename V>
__device__ __forceinline__ V ld_off(const float *ubase, unsigned voff) {
    return *reinterpret_cast<const V *>(reinterpret_cast<const char *>(ubase) + voff);
}

// One workgroup per cloud: the cloud's referenced support rows from its per-point lists idx0[b, p, :] -- a bit per row in LDS
// (read before the atomic: after the first few hundred entries every bit is set), then ranks by a scan over the words:
// rows[b, r] = q in ascending order, slot_of[b, q] = r, n_rows[b], status[b] = 1 when the cloud cannot take this path.
// (csrc/inv_lists.hip's rows -- a histogram, then a bitonic sort by list length for the grouping kernels -- cost 1.9 ms here.)
__global__ __launch_bounds__(1024) void zp_hot_rows_kernel(int np, int nq, const int32_t *__restrict__ idx0, int32_t *__restrict__ rows,
                                                           int32_t *__restrict__ n_rows, int32_t *__restrict__ slot_of,
                                                           int32_t *__restrict__ status) {
    __shared__ unsigned s_bits[512];                                         // nq <= 16384
    __shared__ int s_w[16];
    const int bi = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t < 512) s_bits[t] = 0u;
    __syncthreads();
    const int4 *src = reinterpret_cast<const int4 *>(idx0 + (size_t)bi * np * NN);
    const int n4 = np * (NN / 4);
    for (int e = t; e < n4; e += 1024) {
        const int4 v = src[e];
        const int q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((unsigned)q[u] < (unsigned)nq) {
                const unsigned m = 1u << (q[u] & 31);
                if (!(s_bits[q[u] >> 5] & m)) atomicOr(&s_bits[q[u] >> 5], m);
            }
    }
    __syncthreads();
    const unsigned word = t < 512 ? s_bits[t] : 0u;
    const int cnt = __popc(word);
    int inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(inc, d, 64);
        if (lane >= d) inc += v;
    }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int v = s_w[i];
        total += v;
        if (i < wave) base += v;
    }
    const bool ok = total <= RCAP;
    if (t == 0) {
        n_rows[bi] = total;
        status[bi] = ok ? 0 : 1;
    }
    if (!ok) return;
    int r = base + inc - cnt;
    for (unsigned m = word; m != 0u; m &= m - 1u) {
        const int q = 32 * t + __ffs((int)m) - 1;
        slot_of[(size_t)bi * nq + q] = r;
        rows[(size_t)bi * nq + r] = q;
        ++r;
    }
}

// the verdict of the index check (side stream) joins the status: a cloud whose index is not one list per point was computed
// from its first rows only -- reported, and recomputed by the caller's other path
__global__ void zp_hot_status_kernel(int b, const int32_t *__restrict__ flag, int32_t *__restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < b && flag[i] != 0) status[i] = 1;
}

// slot_off[b, p, n] = byte offset of the accumulator row of neighbour n of point p (tile t = n >> 5, row m = n & 31)
// A list that names the same support row twice (the ball query pads short lists with their first hit) would make two lanes
// of one accumulation step update the same word: such a cloud is reported (status 1) and left to the product pipeline.
__global__ __launch_bounds__(256) void zp_hot_slot_off_kernel(int np, int nq, const int32_t *__restrict__ idx0, const int32_t *__restrict__ slot_of,
                                                              const int32_t *__restrict__ n_rows, int32_t *__restrict__ status,
                                                              int32_t *__restrict__ slot_off) {
    __shared__ int s_q[4][NN];
    const int bi = blockIdx.y;
    if (status[bi] != 0) return;                                             // block-uniform enough: a late flag only costs work
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;          // (p, t, m)
    const bool in = e < (long long)np * NN;
    const int p = (int)(e >> 6), tm = (int)(e & 63), n = tm;               // tile t = n >> 5, row m = n & 31
    const int q = in ? idx0[((size_t)bi * np + p) * NN + n] : -1;
    s_q[threadIdx.x >> 6][n] = q;
    __syncthreads();
    if (!in) return;
    const bool valid = (unsigned)q < (unsigned)nq;
    bool dup = false;
    if (valid)
        for (int j = 0; j < NN; ++j) dup = dup || (j != n && s_q[threadIdx.x >> 6][j] == q);
    if (dup) atomicOr(status + bi, 1);
    const int r = valid ? slot_of[(size_t)bi * nq + q] : n_rows[bi];        // out of range: the dump row (the one after the cloud's last)
    slot_off[((size_t)bi * np + p) * NN + tm] = r * ROWB;
}

// grid: 8 * members * ceil(groups / 8) blocks, members = (na / 4) * (C / 32), group = (cloud, point range).  Block
// 8 * (round * members + member) + x belongs to group 8 * round + x: the 30 workgroups that stream the same points of the
// same cloud run on ONE XCD (block % 8), so a 128-byte line of grad[c,k,p,:] -- shared by 8 anchor quads -- and the
// weight rows -- shared by the two channel halves -- reach that XCD's L2 once.
__global__ __launch_bounds__(TM, 2) void zp_hot_kernel(int nb, int S, int np, int nq, int na, int C, const float *__restrict__ grad,
                                                       const float *__restrict__ w, const int32_t *__restrict__ slot_off,
                                                       const int32_t *__restrict__ rows, const int32_t *__restrict__ n_rows,
                                                       const int32_t *__restrict__ status, float *__restrict__ gfeats,
                                                       float *__restrict__ partial, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int naq = na >> 2, members = naq * (C / CH);
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, member = j % members, group = (j / members) * 8 + x;
    if (group >= nb * S) return;
    const int bi = group / S, sp = group - bi * S;
    if (status[bi] != 0) return;
    const int aq = member % naq, c0 = (member / naq) * CH;
    const int R = n_rows[bi];
    const int p0 = (int)((long long)np * sp / S), p1 = (int)((long long)np * (sp + 1) / S);

    // LDS: acc[R + 1][AQ][CH] (row R = the dump row of out-of-range indices), then one or -- when the rows leave room,
    // R <= 266 -- two operand stages: with two, a point's pieces are staged while the previous point's matrix work runs
    // and the workgroup meets at ONE barrier per point
    float *acc_f = reinterpret_cast<float *>(smem);
    const unsigned stage0 = (unsigned)(R + 1) * ROWB;
    const bool two_stages = stage0 + 2 * (STAGE_G + STAGE_S) <= (unsigned)LDS_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_u & (AQ - 1), wt = wave_u >> 2;                               // anchor of the quad, neighbour half (M tile)

    // zero the accumulators (and the dump row)
    for (int i = tid; i < (R + 1) * ROWB / 16; i += TM) reinterpret_cast<f32x4 *>(acc_f)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane byte offsets off wave-uniform bases
    const unsigned offW = (unsigned)(((KS2 * lh) * NN + 32 * wt + li) * 4);             // w[p, a, 12 lh + j, 32 wt + li]
    const size_t g_k = (size_t)np * na;                                                  // floats between kernel points of grad
    // grad pieces of a point: (c = tid & 31, k = tid >> 5) for k < 16, and (c, 16 + (tid >> 5)) from the first half of the threads
    const unsigned offG0 = (unsigned)((((size_t)(tid & 31) * KS + (tid >> 5)) * g_k) * 4);
    const unsigned offG1 = (unsigned)((((size_t)(tid & 31) * KS + 16 + ((tid >> 5) & 7)) * g_k) * 4);
    const bool second = tid < 256;                                                       // wave-uniform
    const float *gbase = grad + ((size_t)bi * C + c0) * KS * g_k + 4 * aq;               // + p * na
    const float *wbase = w + (((size_t)bi * np) * na + 4 * aq + wave) * (size_t)(KS * NN);   // + p * na * KS * NN
    const int32_t *sbase = slot_off + (size_t)bi * np * NN;
    char *acc_w = smem + (wave * CH + li) * 4;                                           // this lane's word inside row 0

    // Operands travel two points ahead in two register sets (X, Y): a set's grad pieces are re-requested right after they
    // have been written to a stage, its weights right after its matrix instructions -- every request has more than a whole
    // point's matrix work to land, and no register copy forces a wait before the data is needed.
    struct GSet { f32x4 g0, g1; int s; };
    GSet GX{}, GY{};
    float AX[KS2], AY[KS2];
    auto request_g = [&](GSet &G, int p) {
        const float *gp = gbase + (size_t)p * na;
        G.g0 = ld_off<f32x4>(gp, offG0);
        if (second) G.g1 = ld_off<f32x4>(gp, offG1);
        if (tid < NN) G.s = sbase[(size_t)p * NN + tid];
    };
    auto request_w = [&](float (&A)[KS2], int p) {
        const float *wp = wbase + (size_t)p * na * (KS * NN);
#pragma unroll
        for (int s = 0; s < KS2; ++s) A[s] = ld_off<float>(wp + s * NN, offW);
    };
    // a point's grad pieces -> stage [anchor][k][c], its slot offsets -> [n]
    auto put = [&](unsigned stage, const GSet &G) {
        float *sg = reinterpret_cast<float *>(smem + stage);
        const int c = tid & 31, k0 = tid >> 5;
#pragma unroll
        for (int a = 0; a < AQ; ++a) sg[(a * KS + k0) * CH + c] = G.g0[a];
        if (second) {
#pragma unroll
            for (int a = 0; a < AQ; ++a) sg[(a * KS + 16 + k0) * CH + c] = G.g1[a];
        }
        if (tid < NN) reinterpret_cast<int32_t *>(smem + stage + STAGE_G)[tid] = G.s;
    };
    // the matrix work of one point from a filled stage: T[n, c] for this wave's 32 neighbours and its accumulation into their
    // rows.  An accumulator word receives at most ONE contribution per point (a list names a row once: checked on the device)
    // and the workgroup meets at a barrier between points, so the update is a plain read - add - write and every word sums its
    // points in order: bit-reproducible, and 30 x faster than ds_add_f32 (measured: an LDS float atomic costs ~800 cycles
    // per wave-instruction on this part, profiles/r04_zpconv_bwd_hot_ablation.txt).
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto work = [&](unsigned stage, float (&A)[KS2], int pn) {
        const float *sg = reinterpret_cast<const float *>(smem + stage);
        const int32_t *ss = reinterpret_cast<const int32_t *>(smem + stage + STAGE_G);
        float Bf[KS2];
#pragma unroll
        for (int s = 0; s < KS2; ++s) Bf[s] = sg[(wave * KS + KS2 * lh + s) * CH + li];
        // D[m][c]: lane column c = li, register i <-> row m = 8 (i >> 2) + 4 lh + (i & 3), neighbour n = 32 wt + m
        int so[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int4 v = *reinterpret_cast<const int4 *>(ss + 32 * wt + 8 * q + 4 * lh);
            so[4 * q] = v.x; so[4 * q + 1] = v.y; so[4 * q + 2] = v.z; so[4 * q + 3] = v.w;
        }
        // the old values are requested BEFORE the matrix instructions and have landed when those finish (left to itself the
        // compiler sinks these reads below the MFMAs: two LDS round trips on the critical path of every point)
        float old[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) old[i] = ABL(1) && i ? 0.f : *reinterpret_cast<const float *>(acc_w + so[i]);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc = zero16;
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            if (ABL(8)) { acc[s] += A[s] * Bf[s]; continue; }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], Bf[s], acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!ABL(4)) request_w(A, pn);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (!ABL(1) || i == 0) *reinterpret_cast<float *>(acc_w + so[i]) = old[i] + acc[i];
    };

    const int plast = p1 - 1;
    request_g(GX, p0);
    request_w(AX, p0);
    request_g(GY, min(p0 + 1, plast));
    request_w(AY, min(p0 + 1, plast));
    __syncthreads();                                   // the accumulators are zero
    if (ABL(32)) {
    } else if (two_stages) {
        // stage (p - p0) & 1 holds point p.  Step p: matrix work from its stage, then point p + 1 (waiting in the OTHER
        // register set) goes into the other stage -- last read during step p - 1, which every wave left through that
        // step's barrier -- and the set is re-requested for p + 3.
        const unsigned st[2] = {stage0, stage0 + STAGE_G + STAGE_S};
        put(st[0], GX);
        if (!ABL(2)) request_g(GX, min(p0 + 2, plast));
        __syncthreads();
        auto step = [&](int cur, float (&A)[KS2], GSet &Gn, int p) {
            work(st[cur], A, min(p + 2, plast));
            if (!ABL(128)) put(st[cur ^ 1], Gn);
            if (!ABL(2)) request_g(Gn, min(p + 3, plast));
            if (!ABL(64)) __syncthreads();
        };
        for (int p = p0; p < p1; p += 2) {
            step(0, AX, GY, p);
            if (p + 1 < p1) step(1, AY, GX, p + 1);
        }
    } else {
        auto step = [&](GSet &G, float (&A)[KS2], int p) {
            if (!ABL(128)) put(stage0, G);
            if (!ABL(2)) request_g(G, min(p + 2, plast));
            if (!ABL(64)) __syncthreads();
            // (the stage is read into registers at the top of work(); the second barrier lets the next put() overwrite it, and
            // keeps the accumulation steps of consecutive points apart)
            work(stage0, A, min(p + 2, plast));
            if (!ABL(64)) __syncthreads();
        };
        for (int p = p0; p < p1; p += 2) {
            step(GX, AX, p);
            if (p + 1 < p1) step(GY, AY, p + 1);
        }
    }
    __syncthreads();

    // ---- flush: thread <-> (row, channel): the row's four anchors as one 16-byte word
    if (ABL(16)) {
    } else if (S == 1) {
        for (int e = tid; e < R * CH; e += TM) {
            const int r = e / CH, c = e - r * CH;
            const int q = rows[(size_t)bi * nq + r];
            const float *src = acc_f + (size_t)r * (AQ * CH) + c;
            *reinterpret_cast<f32x4 *>(gfeats + (((size_t)bi * C + c0 + c) * nq + q) * na + 4 * aq) =
                f32x4{src[0], src[CH], src[2 * CH], src[3 * CH]};
        }
    } else {
        // partial[b][s][member][row][anchor][c]: the accumulator image as it is
        float *dst = partial + (((size_t)bi * S + sp) * members + member) * (size_t)(RCAP * AQ * CH);
        for (int i = tid; i < R * (AQ * CH) / 4; i += TM) reinterpret_cast<f32x4 *>(dst)[i] = reinterpret_cast<const f32x4 *>(acc_f)[i];
    }
}

// gfeats[b, c0 + c, rows[r], 4 aq ..] = sum over the point ranges s (in order) of partial[b][s][member][r][.][c]
__global__ __launch_bounds__(256) void zp_hot_reduce_kernel(int S, int nq, int na, int C, const float *__restrict__ partial,
                                                            const int32_t *__restrict__ rows, const int32_t *__restrict__ n_rows,
                                                            const int32_t *__restrict__ status, float *__restrict__ gfeats) {
    const int member = blockIdx.x, bi = blockIdx.y;
    if (status[bi] != 0) return;
    const int naq = na >> 2, members = naq * (C / CH), aq = member % naq, c0 = (member / naq) * CH;
    const int R = n_rows[bi];
    for (int e = threadIdx.x; e < R * CH; e += 256) {
        const int r = e / CH, c = e - r * CH;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {
            const float *src = partial + (((size_t)bi * S + s) * members + member) * (size_t)(RCAP * AQ * CH) + (size_t)r * (AQ * CH) + c;
            sum[0] += src[0]; sum[1] += src[CH]; sum[2] += src[2 * CH]; sum[3] += src[3 * CH];
        }
        const int q = rows[(size_t)bi * nq + r];
        *reinterpret_cast<f32x4 *>(gfeats + (((size_t)bi * C + c0 + c) * nq + q) * na + 4 * aq) = sum;
    }
}

// Workspace layout, shared by the size query and the launcher (every chunk on a 256-byte boundary):
//   flag [b] | n_rows [b] | idx0 [b,np,64] | rows, slot_of [b,nq] | slot_off [b,np,64] |
//   partial [b, S, members, RCAP, 4, 32] (S > 1 only)
struct HotWorkspace {
    int64_t flag, n_rows, idx0, rows, slot_of, slot_off, partial, total;
    int S;
    HotWorkspace(int b, int np, int nq, int na, int c) {
        S = b >= 8 ? 1 : (8 + b - 1) / b;                       // point ranges per cloud: at least 8 groups of 30 workgroups
        if (S > np) S = np;
        const int64_t fl = 4 * 64 * (((int64_t)b + 63) / 64), ent = 4ll * b * np * NN, rq = 4ll * b * nq;
        int64_t at = 0;
        auto take = [&](int64_t bytes) { const int64_t r = at; at += (bytes + 255) / 256 * 256; return r; };
        flag = take(fl); n_rows = take(fl);
        idx0 = take(ent);
        rows = take(rq); slot_of = take(rq);
        slot_off = take(ent);
        partial = take(S > 1 ? 4ll * b * S * (na / 4) * (c / CH) * RCAP * AQ * CH : 0);
        total = at;
    }
};

bool hot_supported(int np, int nq, int na, int ks, int ann, int c) {
    return ks == KS && ann == NN && na > 0 && na <= 64 && (na & 3) == 0 && c >= CH && c % CH == 0 && nq > 0 && nq <= 16384 && np > 0 &&
           (long long)c * KS * np * na * 4 < (1ll << 32);          // 32-bit byte offsets inside a cloud of grad
}

}  // namespace

extern "C" int eap_inter_zpconv_bwd_hot_rows(void) { return RCAP; }

extern "C" int64_t eap_inter_zpconv_bwd_hot_workspace(int b, int np, int nq, int na, int ks, int ann, int c) {
    if (b <= 0 || !hot_supported(np, nq, na, ks, ann, c)) return 0;
    return HotWorkspace(b, np, nq, na, c).total;
}

extern "C" int eap_inter_zpconv_bwd_hot_f32(int b, int np, int nq, int na, int ks, int ann, int c, const int32_t *idx, const float *w,
                                            const float *grad, float *gfeats, void *workspace, int32_t *status, eap_stream_t stream) {
    if (b <= 0) return 0;
    if (!workspace || !status || !hot_supported(np, nq, na, ks, ann, c) ||
        ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(gfeats) |
          reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
        return eap::bad_arg("inter_zpconv_backward (on-chip rows): shape or alignment not taken (query eap_inter_zpconv_bwd_hot_workspace first)");
    hipStream_t s = eap::S(stream);
    const HotWorkspace L(b, np, nq, na, c);
    char *wsb = reinterpret_cast<char *>(workspace);
    int32_t *flag = reinterpret_cast<int32_t *>(wsb + L.flag);
    int32_t *n_rows = reinterpret_cast<int32_t *>(wsb + L.n_rows);
    int32_t *idx0 = reinterpret_cast<int32_t *>(wsb + L.idx0);
    int32_t *rows = reinterpret_cast<int32_t *>(wsb + L.rows);
    int32_t *slot_of = reinterpret_cast<int32_t *>(wsb + L.slot_of);
    int32_t *slot_off = reinterpret_cast<int32_t *>(wsb + L.slot_off);
    float *partial = reinterpret_cast<float *>(wsb + L.partial);

    int e = eap::hip_fail(hipMemsetAsync(flag, 0, sizeof(int32_t) * b, s), "inter_zpconv_backward (on-chip rows) flags");
    if (e) return e;
    // idx0 = every point's first (a,k) row (what the kernels below walk), the cloud's referenced rows and the slot of every list entry
    e = eap::zpconv_first_rows(b, np, na * ks * ann, ann, idx, idx0, s);
    if (e) return e;
    hipLaunchKernelGGL(zp_hot_rows_kernel, dim3(b), dim3(1024), 0, s, np, nq, idx0, rows, n_rows, slot_of, status);
    hipLaunchKernelGGL(zp_hot_slot_off_kernel, dim3(eap::cdiv((long long)np * NN, 256), b), dim3(256), 0, s, np, nq, idx0, slot_of, n_rows, status,
                       slot_off);
    e = eap::check_launch("inter_zpconv_backward (on-chip rows) slots");
    if (e) return e;
    // rows nobody references receive no gradient (clouds left to the other path are zeroed again there)
    e = eap::hip_fail(hipMemsetAsync(gfeats, 0, sizeof(float) * (size_t)b * c * nq * na, s), "inter_zpconv_backward (on-chip rows) memset");
    if (e) return e;
    e = eap::hip_fail(hipFuncSetAttribute((const void *)zp_hot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES),
                      "inter_zpconv_backward (on-chip rows) shared memory");
    if (e) return e;
    // The comparison of every other (a,k) row with the first -- the op's 12 GB index read -- streams on the side stream BESIDE
    // the matrix kernel (forked here: behind the short kernels above, which its 32768 workgroups would starve of wave slots;
    // the matrix kernel leaves 2 KB of LDS per CU for them)
    hipStream_t side;
    e = eap::side_fork(s, &side);
    if (e) return e;
    eap::SideJoin joiner(s);              // (also on the error returns below)
    const int members = (na / 4) * (c / CH), groups = b * L.S;
    const long long blocks = 8ll * members * ((groups + 7) / 8);
    if (blocks >= (1ll << 31)) return eap::bad_arg("inter_zpconv_backward (on-chip rows): too many workgroups");
#ifdef EAP_ABLATION
    const int dbg = getenv("EAP_ZPHOT_DEBUG") ? atoi(getenv("EAP_ZPHOT_DEBUG")) : 0;
#else
    const int dbg = 0;
#endif
    hipLaunchKernelGGL(zp_hot_kernel, dim3((unsigned)blocks), dim3(TM), LDS_BYTES, s, b, L.S, np, nq, na, c, grad, w, slot_off, rows, n_rows,
                       status, gfeats, partial, dbg);
    e = eap::check_launch("inter_zpconv_backward (on-chip rows)");
    if (e) return e;
    eap::set_kernel("zp_hot_kernel");
    // (submitted behind the matrix kernel, whose workgroups take their CUs first)
    e = eap::zpconv_index_check(b, np, na * ks * ann, ann, idx, nullptr, nullptr, flag, side);
    if (e) return e;
    if (L.S > 1) {
        hipLaunchKernelGGL(zp_hot_reduce_kernel, dim3(members, b), dim3(256), 0, s, L.S, nq, na, c, partial, rows, n_rows, status, gfeats);
        e = eap::check_launch("inter_zpconv_backward (on-chip rows) reduce");
        if (e) return e;
    }
    e = joiner.join();
    if (e) return e;
    hipLaunchKernelGGL(zp_hot_status_kernel, dim3(eap::cdiv(b, 256)), dim3(256), 0, s, b, flag, status);
    return eap::check_launch("inter_zpconv_backward (on-chip rows) status");
}
