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
//   workgroup = (cloud, point range, anchor QUAD, 32 channels), one per CU, 4 waves = the quad's anchors;
//   LDS      = acc[row slot][4 anchors][32 channels] (512 B per referenced row; 294 rows + a dump row) + one point's
//              operand stage (12.25 KB);
//   per point and wave: T[n, c] = sum_k w[p,a,k,n] grad[c,k,p,a] as 2 x 12 v_mfma_f32_32x32x2_f32 (M = neighbours,
//              N = channels, K = kernel points: no padding), then 32 ds_add_f32 of the accumulator registers into
//              acc[slot(idx[p,n])][a][c] -- lanes run along the channels (conflict-free), and an accumulator word is
//              only ever touched by ONE wave, in program order: the sums are bit-reproducible, no global atomics;
//   w        is streamed straight into registers (8 bytes per lane: neighbours 2m, 2m+1 = the two M tiles), one point
//              ahead; grad[c,k,p,4 anchors] arrives as 16-byte pieces (one piece serves the four waves), one point
//              ahead in registers, then through the LDS stage [anchor][k][c];
//   end      the workgroup writes its rows to gfeats (16-byte stores along the anchors), or, when a cloud's points
//              are split over several workgroups to fill the chip (small batches), to a partial buffer that a second
//              kernel sums in a fixed order.
// w is read by the two channel halves of a quad (2 x), everything else once: 6.1 GB per cloud instead of 16.8.
// A cloud whose referenced rows do not fit (or whose 5-D index is not one list per point) is reported in `status` and
// left to csrc/zpconv_bwd.hip.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int AQ = 4;                         // anchors per workgroup = waves
constexpr int CH = 32;                        // channels per workgroup = MFMA N
constexpr int KS = 24, NN = 64, KS2 = KS / 2;
constexpr int TM = 64 * AQ;
constexpr int ROWB = AQ * CH * 4;             // bytes of accumulators per referenced row
constexpr int STAGE_G = AQ * KS * CH * 4;     // grad stage [anchor][k][c]
constexpr int STAGE_S = NN * 4;               // slot byte offsets of the point's 64 neighbours, [tile t][row m]: n = 2 m + t
constexpr int LDS_BYTES = 160 * 1024;
constexpr int RCAP = (LDS_BYTES - STAGE_G - STAGE_S) / ROWB - 1;      // 294 rows + one dump row for out-of-range indices

template <typename V>
__device__ __forceinline__ V ld_off(const float *ubase, unsigned voff) {
    return *reinterpret_cast<const V *>(reinterpret_cast<const char *>(ubase) + voff);
}

// slot_of[b, q] = position of support row q among the cloud's referenced rows (rows[b, r] = q), status[b] = 1 when the
// cloud cannot take this path
__global__ __launch_bounds__(1024) void zp_hot_slot_of_kernel(int nq, const int32_t *__restrict__ rows, const int32_t *__restrict__ n_rows,
                                                              const int32_t *__restrict__ flag, int32_t *__restrict__ slot_of,
                                                              int32_t *__restrict__ status) {
    const int bi = blockIdx.x, t = threadIdx.x;
    const int R = n_rows[bi];
    const bool ok = flag[bi] == 0 && R <= RCAP;
    if (t == 0) status[bi] = ok ? 0 : 1;
    if (!ok) return;
    for (int r = t; r < R; r += 1024) {
        const int q = rows[(size_t)bi * nq + r];
        if (q >= 0) slot_of[(size_t)bi * nq + q] = r;
    }
}

// slot_off[b, p, t, m] = byte offset of the accumulator row of neighbour n = 2 m + t of point p
__global__ __launch_bounds__(256) void zp_hot_slot_off_kernel(int np, int nq, const int32_t *__restrict__ idx0, const int32_t *__restrict__ slot_of,
                                                              const int32_t *__restrict__ status, int32_t *__restrict__ slot_off) {
    const int bi = blockIdx.y;
    if (status[bi] != 0) return;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;          // (p, t, m)
    if (e >= (long long)np * NN) return;
    const int p = (int)(e >> 6), tm = (int)(e & 63), t = tm >> 5, m = tm & 31;
    const int q = idx0[((size_t)bi * np + p) * NN + 2 * m + t];
    const int r = (unsigned)q < (unsigned)nq ? slot_of[(size_t)bi * nq + q] : RCAP;       // out of range: the dump row
    slot_off[((size_t)bi * np + p) * NN + tm] = r * ROWB;
}

// grid: 8 * members * ceil(groups / 8) blocks, members = (na / 4) * (C / 32), group = (cloud, point range).  Block
// 8 * (round * members + member) + x belongs to group 8 * round + x: the 30 workgroups that stream the same points of the
// same cloud run on ONE XCD (block % 8), so a 128-byte line of grad[c,k,p,:] -- shared by 8 anchor quads -- and the
// weight rows -- shared by the two channel halves -- reach that XCD's L2 once.
__global__ __launch_bounds__(TM, 1) void zp_hot_kernel(int nb, int S, int np, int nq, int na, int C, const float *__restrict__ grad,
                                                       const float *__restrict__ w, const int32_t *__restrict__ slot_off,
                                                       const int32_t *__restrict__ rows, const int32_t *__restrict__ n_rows,
                                                       const int32_t *__restrict__ status, float *__restrict__ gfeats,
                                                       float *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *acc_f = reinterpret_cast<float *>(smem);                                   // [RCAP + 1][AQ][CH]
    float *stage_g = reinterpret_cast<float *>(smem + (RCAP + 1) * ROWB);              // [AQ][KS][CH]
    int32_t *stage_s = reinterpret_cast<int32_t *>(smem + (RCAP + 1) * ROWB + STAGE_G);  // [2][32]

    const int naq = na >> 2, members = naq * (C / CH);
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, member = j % members, group = (j / members) * 8 + x;
    if (group >= nb * S) return;
    const int bi = group / S, sp = group - bi * S;
    if (status[bi] != 0) return;
    const int aq = member % naq, c0 = (member / naq) * CH;
    const int R = n_rows[bi];
    const int p0 = (int)((long long)np * sp / S), p1 = (int)((long long)np * (sp + 1) / S);

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                          // = anchor of the quad

    // zero the accumulators (and the dump row)
    for (int i = tid; i < (RCAP + 1) * ROWB / 16; i += TM) reinterpret_cast<f32x4 *>(acc_f)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane byte offsets off wave-uniform bases
    const unsigned offW = (unsigned)(((KS2 * lh) * NN + 2 * li) * 4);                   // w[p, a, 12 lh + j, 2 li .. 2 li + 1]
    const size_t g_k = (size_t)np * na;                                                  // floats between kernel points of grad
    unsigned offG[3];                                                                    // grad pieces (c = tid & 31, k = (tid >> 5) + 8 u)
#pragma unroll
    for (int u = 0; u < 3; ++u) offG[u] = (unsigned)((((size_t)(tid & 31) * KS + (tid >> 5) + 8 * u) * g_k) * 4);
    const float *gbase = grad + ((size_t)bi * C + c0) * KS * g_k + 4 * aq;               // + p * na
    const float *wbase = w + (((size_t)bi * np) * na + 4 * aq + wave) * (size_t)(KS * NN);   // + p * na * KS * NN
    const int32_t *sbase = slot_off + (size_t)bi * np * NN;
    const unsigned acc_lane = (unsigned)((wave * CH + li) * 4);                          // this lane's word inside a row
    float *acc_w = acc_f + (acc_lane >> 2);

    // Operands travel two points ahead in two register sets (X: even steps, Y: odd steps): a set's grad pieces are
    // re-requested right after they have been written to the stage, its weights right after its matrix instructions --
    // every request has more than a whole point's matrix work (~1800 cycles) to land, and no register copy forces a
    // wait before the data is needed.
    f32x4 GX[3], GY[3];
    f32x2 AX[KS2], AY[KS2];
    int sX = 0, sY = 0;
    auto request_g = [&](f32x4 (&G)[3], int &sreg, int p) {
        const float *gp = gbase + (size_t)p * na;
#pragma unroll
        for (int u = 0; u < 3; ++u) G[u] = ld_off<f32x4>(gp, offG[u]);
        if (tid < NN) sreg = sbase[(size_t)p * NN + tid];
    };
    auto request_w = [&](f32x2 (&A)[KS2], int p) {
        const float *wp = wbase + (size_t)p * na * (KS * NN);
#pragma unroll
        for (int s = 0; s < KS2; ++s) A[s] = ld_off<f32x2>(wp + s * NN, offW);
    };
    const int plast = p1 - 1;
    request_g(GX, sX, p0);
    request_w(AX, p0);
    request_g(GY, sY, min(p0 + 1, plast));
    request_w(AY, min(p0 + 1, plast));
    __syncthreads();

    auto step = [&](f32x4 (&G)[3], f32x2 (&A)[KS2], int &sreg, int p) {
        // ---- stage the point's grad pieces [anchor][k][c] and slot offsets; re-request the set for point p + 2
        {
            const int c = tid & 31;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int k = (tid >> 5) + 8 * u;
#pragma unroll
                for (int a = 0; a < AQ; ++a) stage_g[(a * KS + k) * CH + c] = G[u][a];
            }
            if (tid < NN) stage_s[tid] = sreg;
        }
        const int pn = min(p + 2, plast);
        request_g(G, sreg, pn);
        __syncthreads();
        // ---- this wave's B operand (grad of its anchor) and the accumulator rows of the 64 neighbours
        float Bf[KS2];
#pragma unroll
        for (int s = 0; s < KS2; ++s) Bf[s] = stage_g[(wave * KS + KS2 * lh + s) * CH + li];
        int so[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int4 v = *reinterpret_cast<const int4 *>(stage_s + t * 32 + 8 * q + 4 * lh);
                so[t][4 * q] = v.x; so[t][4 * q + 1] = v.y; so[t][4 * q + 2] = v.z; so[t][4 * q + 3] = v.w;
            }
        __syncthreads();                               // the stage may be overwritten from here on
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][0], Bf[s], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s][1], Bf[s], acc[1], 0, 0, 0);
        }
        request_w(A, pn);
        // D[m][c]: lane column c = li, register i <-> row m = 8 (i >> 2) + 4 lh + (i & 3), neighbour n = 2 m + t
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                __hip_atomic_fetch_add(reinterpret_cast<float *>(reinterpret_cast<char *>(acc_w) + so[t][i]), acc[t][i], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    for (int p = p0; p < p1; p += 2) {
        step(GX, AX, sX, p);
        if (p + 1 < p1) step(GY, AY, sY, p + 1);
    }
    __syncthreads();

    // ---- flush: thread <-> (row, channel): the row's four anchors as one 16-byte word
    if (S == 1) {
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
//   flag [b] | status [b] | n_rows [b] | idx0 [b,np,64] | counts, rows, off, cnt, slot_of [b,nq] | slot_off [b,np,64] |
//   partial [b, S, members, RCAP, 4, 32] (S > 1 only)
struct HotWorkspace {
    int64_t flag, status, n_rows, idx0, counts, rows, off, cnt, slot_of, slot_off, partial, total;
    int S;
    HotWorkspace(int b, int np, int nq, int na, int c) {
        S = b >= 8 ? 1 : (8 + b - 1) / b;                       // point ranges per cloud: at least 8 groups of 30 workgroups
        if (S > np) S = np;
        const int64_t fl = 4 * 64 * (((int64_t)b + 63) / 64), ent = 4ll * b * np * NN, rq = 4ll * b * nq;
        int64_t at = 0;
        auto take = [&](int64_t bytes) { const int64_t r = at; at += (bytes + 255) / 256 * 256; return r; };
        flag = take(fl); status = take(fl); n_rows = take(fl);
        idx0 = take(ent);
        counts = take(rq); rows = take(rq); off = take(rq); cnt = take(rq); slot_of = take(rq);
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
    int32_t *counts = reinterpret_cast<int32_t *>(wsb + L.counts);
    int32_t *rows = reinterpret_cast<int32_t *>(wsb + L.rows);
    int32_t *off = reinterpret_cast<int32_t *>(wsb + L.off);
    int32_t *cnt = reinterpret_cast<int32_t *>(wsb + L.cnt);
    int32_t *slot_of = reinterpret_cast<int32_t *>(wsb + L.slot_of);
    int32_t *slot_off = reinterpret_cast<int32_t *>(wsb + L.slot_off);
    float *partial = reinterpret_cast<float *>(wsb + L.partial);

    int e = eap::hip_fail(hipMemsetAsync(flag, 0, sizeof(int32_t) * b, s), "inter_zpconv_backward (on-chip rows) flags");
    if (e) return e;
    e = eap::zpconv_index_check(b, np, na * ks * ann, ann, idx, idx0, nullptr, flag, s);
    if (e) return e;
    e = eap_inv_lists_rows(b, np, nq, ann, idx0, counts, rows, off, cnt, n_rows, stream);
    if (e) return e;
    hipLaunchKernelGGL(zp_hot_slot_of_kernel, dim3(b), dim3(1024), 0, s, nq, rows, n_rows, flag, slot_of, status);
    hipLaunchKernelGGL(zp_hot_slot_off_kernel, dim3(eap::cdiv((long long)np * NN, 256), b), dim3(256), 0, s, np, nq, idx0, slot_of, status,
                       slot_off);
    e = eap::check_launch("inter_zpconv_backward (on-chip rows) slots");
    if (e) return e;
    // rows nobody references receive no gradient (clouds left to the other path are zeroed again there)
    e = eap::hip_fail(hipMemsetAsync(gfeats, 0, sizeof(float) * (size_t)b * c * nq * na, s), "inter_zpconv_backward (on-chip rows) memset");
    if (e) return e;
    e = eap::hip_fail(hipFuncSetAttribute((const void *)zp_hot_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES),
                      "inter_zpconv_backward (on-chip rows) shared memory");
    if (e) return e;
    const int members = (na / 4) * (c / CH), groups = b * L.S;
    const long long blocks = 8ll * members * ((groups + 7) / 8);
    if (blocks >= (1ll << 31)) return eap::bad_arg("inter_zpconv_backward (on-chip rows): too many workgroups");
    hipLaunchKernelGGL(zp_hot_kernel, dim3((unsigned)blocks), dim3(TM), LDS_BYTES, s, b, L.S, np, nq, na, c, grad, w, slot_off, rows, n_rows,
                       status, gfeats, partial);
    e = eap::check_launch("inter_zpconv_backward (on-chip rows)");
    if (e) return e;
    eap::set_kernel("zp_hot_kernel");
    if (L.S > 1) {
        hipLaunchKernelGGL(zp_hot_reduce_kernel, dim3(members, b), dim3(256), 0, s, L.S, nq, na, c, partial, rows, n_rows, status, gfeats);
        e = eap::check_launch("inter_zpconv_backward (on-chip rows) reduce");
    }
    return e;
}
