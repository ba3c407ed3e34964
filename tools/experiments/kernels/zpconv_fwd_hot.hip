// tools/experiments/kernels/zpconv_fwd_hot.hip -- EXPERIMENT, not in the production library (`make EXPERIMENTS=1` compiles it; its
// three entries are declared here, not in include/eap_hip.h).  MEASURED SLOWER than csrc/zpconv_mfma.hip: 19-20 ms against 11.0
// at 8 x 4096, C = 64 (profiles/r05_zpconv_fwd_hot_experiment.txt).  Correct (2e-7 of the output scale against float64 on six
// shapes), and everything but the output path works as designed -- the weights stream once per channel half, no barrier in the
// loop -- but a workgroup owns FOUR anchors, so every output line out[c,k,p,:] (240 B) is written as fifteen 16-byte pieces by
// fifteen workgroups at different times: they do not merge in the XCD's L2 (one step of the 30 workgroups has 2.9 MB of output
// lines open beside 2.9 MB of weights, in 4 MB) and reach memory as partial lines: +10 ms.  With the same stores pointed at
// whole lines the kernel takes 9 ms, without stores 6.6.  The backward (csrc/zpconv_bwd_hot.hip) READS grad in exactly this pattern
// and is fine: a partially read line costs nothing.  Kept for the record and for its pieces (counted waits around inline-asm
// memory instructions, the store-data hazard note, the rows kernel that the backward now uses).
//
// The native inter "zpconv" forward (zpconv_cuda.cpp:L41-56, kernel zpconv_cuda_kernel.cu:L33-73) with the referenced feature
// rows held ON CHIP (round 5; the mirror image of csrc/zpconv_bwd_hot.hip).
//
//   out[b,c,k,p,a] = sum_n w[b,p,a,k,n] * feats[b,c,idx[b,p,.,.,n],a]
//
// csrc/zpconv_mfma.hip fetches the 64 feature rows of every point again (34 GB through the L2s, 21 GB of them through the
// fabric) and meets at a barrier per 8 neighbours.  The reference's neighbour lists (first nsample hits in index order inside a
// ball, grouping_cuda_kernel.cu:L68-113) name few support rows when the ball is large -- ~280 of 4096 at the second layer's
// radius -- so a cloud's referenced rows, [rows x 60 anchors x C] floats, fit the chip's LDS spread over 30 workgroups:
//
//   workgroup = (cloud, point range, anchor QUAD, 32 channels), one per CU; 8 waves, each with its OWN points
//              (p0 + wave, + 8, ...): after the feature image is loaded nothing is shared and NO barrier is left in the loop;
//   LDS      = feat[row slot][4 anchors][32 channels] (512 B per referenced row, at most 312 rows; 4 KB stay free so that the
//              index check's workgroups find room beside this kernel on every CU);
//   per point and wave: for each of the quad's anchors  T[k, c] = sum_n w[p,a,k,n] feat[slot(idx[p,n])][a][c]  as 32
//              v_mfma_f32_32x32x2_f32 (M = kernel points 24 -> 32, N = channels, K = neighbours): lane (k, h) owns
//              neighbours 32 h .. 32 h + 31 of row w[p,a,k,:] -- 128 contiguous bytes, eight 16-byte loads straight into
//              registers one anchor ahead -- and the feature operand is one LDS word per lane and k-step (lanes along the
//              channels: conflict-free) at the row slot of that neighbour (16-bit slots, two per register);
//   end      a lane ends a point with out[c, k, p, a0..a0+3] of 12 (c, k) pairs in registers: 16-byte stores, no exchange.
// w is read by the two channel halves of a quad (the second one from the XCD's L2: the 30 workgroups of a cloud run on one
// XCD), everything else once; the 16-byte pieces of an output line are written by the 15 quads of that XCD and merge in
// its L2.  The 5-D index check (the op's 12 GB index read) streams on the side stream beside this kernel.
// A cloud that references more rows than fit is left to csrc/zpconv_mfma.hip (decided on the device: `mat_skip`), one whose
// 5-D index is not one list per point is recomputed by csrc/zpconv_rows.hip as before.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int AQ = 4;                         // anchors per workgroup
constexpr int CH = 32;                        // channels per workgroup = MFMA N
constexpr int NN = 64;                        // neighbours per list
constexpr int NW = 8;                         // waves = points in flight per workgroup
constexpr int TM = 64 * NW;
constexpr int ROWB = AQ * CH * 4;             // bytes of features per referenced row
constexpr int LDS_BYTES = 156 * 1024;
constexpr int RCAP = LDS_BYTES / ROWB;        // referenced rows a cloud may have: 312

// Timing ablations (WRONG RESULTS), compiled only with `make ABLATION=1` and selected by EAP_ZPFHOT_DEBUG (bit mask): 1 no
// output stores, 2 no weight requests after the prologue, 4 no matrix instructions, 8 no LDS operand reads, 16 every counted wait preceded by a full one, 32 no index check beside the kernel, 64 output pieces stored as whole lines (elsewhere)
#ifdef EAP_ABLATION
#define ABL(bit) ((dbg & (bit)) != 0)
#else
#define ABL(bit) false
#endif

template <typename V>
__device__ __forceinline__ V ld_off(const void *ubase, unsigned voff) {
    return *reinterpret_cast<const V *>(reinterpret_cast<const char *>(ubase) + voff);
}

// marks[b, q] = 1 when a list of the cloud names support row q (out-of-range indices clamped as csrc/zpconv_mfma.hip clamps them)
__global__ __launch_bounds__(256) void zpf_mark_kernel(long long n, int per_cloud, int nq, const int32_t *__restrict__ idx0,
                                                       int32_t *__restrict__ marks) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int bi = (int)(e / per_cloud);
    const unsigned q = min((unsigned)idx0[e], (unsigned)nq - 1u);
    marks[(size_t)bi * nq + q] = 1;
}

// one workgroup per cloud: slot_of[b, q] = rank of q among the cloud's marked rows (ascending), rows[b, r] = q, n_rows[b];
// hot_skip[b] = 1 / mat_skip[b] = 0 when the rows do not fit the LDS (the cloud goes to the matrix kernel of zpconv_mfma.hip)
__global__ __launch_bounds__(1024) void zpf_scan_kernel(int nq, const int32_t *__restrict__ marks, int32_t *__restrict__ slot_of,
                                                        int32_t *__restrict__ rows, int32_t *__restrict__ n_rows,
                                                        int32_t *__restrict__ hot_skip, int32_t *__restrict__ mat_skip) {
    __shared__ int s_w[16];
    const int bi = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (nq + 1023) >> 10, q0 = t * per;
    const int32_t *mk = marks + (size_t)bi * nq;
    int cnt = 0;
    for (int j = 0; j < per; ++j) cnt += (q0 + j < nq && mk[q0 + j] != 0) ? 1 : 0;
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
    const bool hot = total <= RCAP;
    if (t == 0) {
        n_rows[bi] = total;
        hot_skip[bi] = hot ? 0 : 1;
        mat_skip[bi] = hot ? 1 : 0;
    }
    if (!hot) return;
    int r = base + inc - cnt;
    for (int j = 0; j < per; ++j) {
        const int q = q0 + j;
        if (q < nq && mk[q] != 0) {
            slot_of[(size_t)bi * nq + q] = r;
            rows[(size_t)bi * nq + r] = q;
            ++r;
        }
    }
}

// slot16[b, p, n] = row slot of neighbour n of point p
__global__ __launch_bounds__(256) void zpf_slots_kernel(long long n, int per_cloud, int nq, const int32_t *__restrict__ idx0,
                                                        const int32_t *__restrict__ slot_of, const int32_t *__restrict__ hot_skip,
                                                        uint16_t *__restrict__ slot16) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int bi = (int)(e / per_cloud);
    if (hot_skip[bi] != 0) return;
    const unsigned q = min((unsigned)idx0[e], (unsigned)nq - 1u);
    slot16[e] = (uint16_t)slot_of[(size_t)bi * nq + q];
}

// grid: 8 * members * ceil(groups / 8) blocks, members = (na / 4) * (C / 32), group = (cloud, point range).  Block
// 8 * (round * members + member) + x belongs to group 8 * round + x: the 30 workgroups that stream the same points of the
// same cloud run on ONE XCD (block % 8), so the weight rows -- shared by the two channel halves -- reach that XCD's L2 once
// and the 16-byte pieces of an output line -- written by 15 anchor quads -- meet there.
// NI = accumulator registers of a lane that can hold a kernel point k < ks: 12 (ks <= 24) or 16
template <int NI, bool FULL>
__global__ __launch_bounds__(TM, 2) void zpf_hot_kernel(int nb, int S, int np, int nq, int na, int ks, int C, const float *__restrict__ F,
                                                        const float *__restrict__ w, const uint16_t *__restrict__ slot16,
                                                        const int32_t *__restrict__ rows, const int32_t *__restrict__ n_rows,
                                                        const int32_t *__restrict__ hot_skip, float *__restrict__ out, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int naq = na >> 2, members = naq * (C / CH);
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, member = j % members, group = (j / members) * 8 + x;
    if (group >= nb * S) return;
    const int bi = group / S, sp = group - bi * S;
    if (hot_skip[bi] != 0) return;
    const int aq = member % naq, c0 = (member / naq) * CH;
    const int R = n_rows[bi];
    const int p0 = (int)((long long)np * sp / S), p1 = (int)((long long)np * (sp + 1) / S);

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- the feature image feat[r][a][c]: thread <-> (row, channel), the row's four anchors as one 16-byte word
    {
        float *feat = reinterpret_cast<float *>(smem);
        const float *fb = F + ((size_t)bi * C + c0) * nq * na + 4 * aq;
        for (int e = tid; e < R * CH; e += TM) {
            const int r = e >> 5, c = e & 31;
            const int q = rows[(size_t)bi * nq + r];
            const f32x4 v = *reinterpret_cast<const f32x4 *>(fb + ((size_t)c * nq + q) * na);
            float *d = feat + r * (AQ * CH) + c;
            d[0] = v[0]; d[CH] = v[1]; d[2 * CH] = v[2]; d[3 * CH] = v[3];
        }
    }

    // ---- per-lane byte offsets off wave-uniform bases
    const unsigned offW = (unsigned)((min(li, ks - 1) * NN + 32 * lh) * 4);             // w[p, a, k = li, 32 lh ..]
    const unsigned offS = (unsigned)(64 * lh);                                           // slot16[p, 32 lh ..]
    const size_t PA = (size_t)np * na;
    const unsigned offO = (unsigned)((((size_t)li * ks + 4 * lh) * PA) * 4);            // out[c = li, k = 4 lh + ..]
    const float *wbase = w + (((size_t)bi * np) * na + 4 * aq) * (size_t)(ks * NN);      // + (p * na + a) * ks * NN
    const uint16_t *sbase = slot16 + (size_t)bi * np * NN;                               // + p * NN
    float *obase = out + ((size_t)bi * C + c0) * ks * PA + 4 * aq;                       // + (8 (i >> 2) + (i & 3)) * PA + p * na
    const char *feat_lane = smem + li * 4;                                               // + slot * ROWB + a * 128

    // Every vector-memory instruction of the loop is inline asm with its own counted waits.  Left to hipcc, the waits at the
    // loop head are merged with the state of the loop ENTRY (no stores in flight yet), so every point would wait for the
    // previous point's stores and for all the weights just requested.  Issue order of one point (16-byte instructions):
    //   S(next point) 4 | WY(p,1) 8 | anchor 0 | WX(p,2) 8 | anchor 1 | WY(p,3) 8 | anchor 2 | WX(next,0) 8 | anchor 3 | stores NI
    f32x4 WX[8], WY[8];                                                                  // weights of two anchors in flight
    i32x4 SN[4], SC[4];                                                                  // 32 row slots (16 bits each) of this lane half: next / current point
    auto request_w = [&](f32x4 (&W)[8], int p, int a) {
        // (lanes past the last kernel point repeat its row)
        // (64-bit per-lane addresses, not scalar base + lane offset: the scalar base would come out of the SALU right in front of
        // the asm statement, and hipcc does not see the 5 wait states a vector-memory instruction needs behind that)
        const char *wp = reinterpret_cast<const char *>(wbase + ((size_t)p * na + a) * (size_t)(ks * NN)) + offW;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(W[0]) : "v"(wp));
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(W[1]) : "v"(wp));
        asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(W[2]) : "v"(wp));
        asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(W[3]) : "v"(wp));
        asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(W[4]) : "v"(wp));
        asm volatile("global_load_dwordx4 %0, %1, off offset:80" : "=v"(W[5]) : "v"(wp));
        asm volatile("global_load_dwordx4 %0, %1, off offset:96" : "=v"(W[6]) : "v"(wp));
        asm volatile("global_load_dwordx4 %0, %1, off offset:112" : "=v"(W[7]) : "v"(wp));
    };
    auto request_s = [&](int p) {
        const char *sp16 = reinterpret_cast<const char *>(sbase + (size_t)p * NN) + offS;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(SN[0]) : "v"(sp16));
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(SN[1]) : "v"(sp16));
        asm volatile("global_load_dwordx4 %0, %1, off offset:32" : "=v"(SN[2]) : "v"(sp16));
        asm volatile("global_load_dwordx4 %0, %1, off offset:48" : "=v"(SN[3]) : "v"(sp16));
    };
    // at most N vector-memory instructions still in flight; the registers named are the ones that have landed by then
#define ZPF_LANDED_W(N, W)                                                                                                    \
    if (ABL(16)) asm volatile("s_waitcnt vmcnt(0)");                                                                          \
    asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]), "+v"(W[6]), "+v"(W[7]))
#define ZPF_LANDED_S(N)                                                                                                       \
    if (ABL(16)) asm volatile("s_waitcnt vmcnt(0)");                                                                          \
    asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(SN[0]), "+v"(SN[1]), "+v"(SN[2]), "+v"(SN[3]))
    // one anchor of one point: T[k, c] over the 64 neighbours (k-step s: neighbour 32 lh + s)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto anchor = [&](const f32x4 (&W)[8], int a) -> f32x16 {
        f32x16 acc = zero16;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const unsigned pair = (unsigned)SC[s >> 3][(s >> 1) & 3];
            const unsigned slot = (s & 1) ? (pair >> 16) : (pair & 0xffffu);
            const float f = ABL(8) ? 1.f : *reinterpret_cast<const float *>(feat_lane + slot * ROWB + a * (CH * 4));
            if (ABL(4)) { acc[s & 15] += W[s >> 2][s & 3] * f; continue; }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(W[s >> 2][s & 3], f, acc, 0, 0, 0);
        }
        return acc;
    };

    const int pfirst = p0 + wave_u, plast = p1 - 1;
    if (pfirst < p1) {
        request_s(pfirst);
        ZPF_LANDED_S(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) SC[q] = SN[q];
        request_w(WX, pfirst, 0);
        // (landed before the loop: its first counted wait assumes the stores of a previous point behind these requests.  ONE wait
        // statement per use inside the loop -- two alternatives would meet in copies of registers that have not landed)
        ZPF_LANDED_W(0, WX);
    }
    __syncthreads();                                   // the feature image is complete; nothing is shared after this

    for (int p = pfirst; p < p1; p += NW) {
        const int pn = min(p + NW, plast);             // (the last point again past the end: no branch around the requests)
        f32x4 o[NI];                                   // o[i] = (anchors a0 .. a0 + 3) of out[c = li, k = 8 (i >> 2) + 4 lh + (i & 3)]
        f32x16 t;
        request_s(pn);
        request_w(WY, p, 1);
        if constexpr (NI == 12) {
            ZPF_LANDED_W(24, WX);                      // newer: the previous point's stores, S, WY
        } else {
            ZPF_LANDED_W(28, WX);
        }
        __builtin_amdgcn_sched_barrier(0);
        t = anchor(WX, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) o[i][0] = t[i];
        __builtin_amdgcn_sched_barrier(0);
        if (!ABL(2)) request_w(WX, p, 2);
        ZPF_LANDED_W(8, WY);
        __builtin_amdgcn_sched_barrier(0);
        t = anchor(WY, 1);
#pragma unroll
        for (int i = 0; i < NI; ++i) o[i][1] = t[i];
        __builtin_amdgcn_sched_barrier(0);
        if (!ABL(2)) request_w(WY, p, 3);
        ZPF_LANDED_W(8, WX);
        __builtin_amdgcn_sched_barrier(0);
        t = anchor(WX, 2);
#pragma unroll
        for (int i = 0; i < NI; ++i) o[i][2] = t[i];
        __builtin_amdgcn_sched_barrier(0);
        if (!ABL(2)) request_w(WX, pn, 0);
        ZPF_LANDED_W(8, WY);
        __builtin_amdgcn_sched_barrier(0);
        t = anchor(WY, 3);
#pragma unroll
        for (int i = 0; i < NI; ++i) o[i][3] = t[i];
        __builtin_amdgcn_sched_barrier(0);
        {
            float *op = obase + (size_t)p * na;        // uniform
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int ku = 8 * (i >> 2) + (i & 3);                                   // + 4 lh
                char *row = reinterpret_cast<char *>(op + (size_t)ku * PA) + offO;
                if (ABL(64))                           // (timing only) the same stores as whole lines: 1 KB per wave and instruction
                    row = reinterpret_cast<char *>(out) + ((((size_t)blockIdx.x * np + p) * NI + i) * 64 + lane) * 16 % ((size_t)nb * C * ks * PA * 4);
                if (FULL) {
                    // (s_nop: a VALU write to the data registers needs 2 wait states behind a 16-byte store; hipcc cannot see the store)
                    if (!ABL(1)) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(row), "v"(o[i]) : "memory");
                } else {
                    // exec-masked, never branched around: the counted waits above assume NI stores per point
                    const unsigned long long keep = __builtin_amdgcn_read_exec();
                    const unsigned long long on = __ballot(ku + 4 * lh < ks);
                    asm volatile("s_nop 4\n\ts_mov_b64 exec, %2\n\tglobal_store_dwordx4 %0, %1, off\n\ts_mov_b64 exec, %3\n\ts_nop 1"
                                 : : "v"(row), "v"(o[i]), "s"(on), "s"(keep) : "memory");
                }
            }
        }
        // S(next point) -- requested first in this iteration: 32 weight requests and NI stores are newer -- becomes current
        if constexpr (NI == 12) { ZPF_LANDED_S(44); } else { ZPF_LANDED_S(48); }
#pragma unroll
        for (int q = 0; q < 4; ++q) SC[q] = SN[q];
    }
#undef ZPF_LANDED_W
#undef ZPF_LANDED_S
}

// Workspace layout (every chunk on a 256-byte boundary):
//   flag [b] | mat_skip [b] | hot_skip [b] | n_rows [b] | idx0 [b,np,64] | marks, slot_of, rows [b,nq] | slot16 [b,np,64] (16-bit)
struct FwdHotWorkspace {
    int64_t flag, mat_skip, hot_skip, n_rows, idx0, marks, slot_of, rows, slot16, total;
    int S;
    FwdHotWorkspace(int b, int np, int nq) {
        S = b >= 8 ? 1 : (8 + b - 1) / b;                       // point ranges per cloud: at least 8 groups of 30 workgroups
        if (S > (np + NW - 1) / NW) S = (np + NW - 1) / NW;
        if (S < 1) S = 1;
        const int64_t fl = 4 * 64 * (((int64_t)b + 63) / 64), ent = (int64_t)b * np * NN, rq = 4ll * b * nq;
        int64_t at = 0;
        auto take = [&](int64_t bytes) { const int64_t r = at; at += (bytes + 255) / 256 * 256; return r; };
        flag = take(fl); mat_skip = take(fl); hot_skip = take(fl); n_rows = take(fl);
        idx0 = take(4 * ent);
        marks = take(rq); slot_of = take(rq); rows = take(rq);
        slot16 = take(2 * ent);
        total = at;
    }
};

bool fwd_hot_supported(int np, int nq, int na, int ks, int ann, int c) {
    return ks > 0 && ks <= 32 && ann == NN && na > 0 && na <= 64 && (na & 3) == 0 && c >= CH && c % CH == 0 && nq > 0 && nq <= 16384 &&
           np > 0 && (long long)c * ks * np * na * 4 < (1ll << 32) &&                   // 32-bit byte offsets inside a cloud of out
           eap::inter_zpconv_mfma_supported(np, nq, na, ks, ann, c) && eap::inter_zpconv_rows_supported(np, nq, na, ks, ann, c) &&
           (long long)na * ks * ann < (1ll << 31);
}

}  // namespace

extern "C" int eap_inter_zpconv_fwd_hot_rows(void) { return RCAP; }

extern "C" int64_t eap_inter_zpconv_fwd_hot_workspace(int b, int np, int nq, int na, int ks, int ann, int c) {
    if (b <= 0 || !fwd_hot_supported(np, nq, na, ks, ann, c)) return 0;
    return FwdHotWorkspace(b, np, nq).total;
}

extern "C" int eap_inter_zpconv_fwd_hot_f32(int b, int np, int nq, int na, int ks, int ann, int c, const int32_t *idx, const float *w,
                                            const float *src, float *dst, void *workspace, eap_stream_t stream) {
    if (b <= 0) return 0;
    if (!workspace || !fwd_hot_supported(np, nq, na, ks, ann, c) ||
        ((reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) |
          reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
        return eap::bad_arg("inter_zpconv_forward (on-chip rows): shape or alignment not taken (query eap_inter_zpconv_fwd_hot_workspace first)");
    hipStream_t s = eap::S(stream);
    const FwdHotWorkspace L(b, np, nq);
    char *wsb = reinterpret_cast<char *>(workspace);
    int32_t *flag = reinterpret_cast<int32_t *>(wsb + L.flag);
    int32_t *mat_skip = reinterpret_cast<int32_t *>(wsb + L.mat_skip);
    int32_t *hot_skip = reinterpret_cast<int32_t *>(wsb + L.hot_skip);
    int32_t *n_rows = reinterpret_cast<int32_t *>(wsb + L.n_rows);
    int32_t *idx0 = reinterpret_cast<int32_t *>(wsb + L.idx0);
    int32_t *marks = reinterpret_cast<int32_t *>(wsb + L.marks);
    int32_t *slot_of = reinterpret_cast<int32_t *>(wsb + L.slot_of);
    int32_t *rows = reinterpret_cast<int32_t *>(wsb + L.rows);
    uint16_t *slot16 = reinterpret_cast<uint16_t *>(wsb + L.slot16);

    int e = eap::hip_fail(hipMemsetAsync(flag, 0, (size_t)(L.idx0 - L.flag), s), "inter_zpconv_forward (on-chip rows) flags");
    if (e) return e;
    e = eap::hip_fail(hipMemsetAsync(marks, 0, sizeof(int32_t) * (size_t)b * nq, s), "inter_zpconv_forward (on-chip rows) marks");
    if (e) return e;
    // The side stream waits for the flags only; its kernel -- the comparison of every (a,k) row of the 5-D index with the
    // point's first row: the op's 12 GB index read -- is submitted AFTER the matrix kernel so that this one's workgroups get
    // their CUs first, and streams beside it.
    hipStream_t side;
    e = eap::side_fork(s, &side);
    if (e) return e;
    eap::SideJoin joiner(s);              // (also on the error returns below)
    e = eap::zpconv_first_rows(b, np, na * ks * ann, ann, idx, idx0, s);
    if (e) return e;
    const long long ent = (long long)b * np * NN;
    hipLaunchKernelGGL(zpf_mark_kernel, dim3(eap::cdiv(ent, 256)), dim3(256), 0, s, ent, np * NN, nq, idx0, marks);
    hipLaunchKernelGGL(zpf_scan_kernel, dim3(b), dim3(1024), 0, s, nq, marks, slot_of, rows, n_rows, hot_skip, mat_skip);
    hipLaunchKernelGGL(zpf_slots_kernel, dim3(eap::cdiv(ent, 256)), dim3(256), 0, s, ent, np * NN, nq, idx0, slot_of, hot_skip, slot16);
    e = eap::check_launch("inter_zpconv_forward (on-chip rows) slots");
    if (e) return e;
    const int members = (na / 4) * (c / CH), groups = b * L.S;
    const long long blocks = 8ll * members * ((groups + 7) / 8);
    if (blocks >= (1ll << 31)) return eap::bad_arg("inter_zpconv_forward (on-chip rows): too many workgroups");
#ifdef EAP_ABLATION
    const int dbg = getenv("EAP_ZPFHOT_DEBUG") ? atoi(getenv("EAP_ZPFHOT_DEBUG")) : 0;
#else
    const int dbg = 0;
#endif
    auto launch = [&](auto kernel) -> int {
        int er = eap::hip_fail(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES),
                               "inter_zpconv_forward (on-chip rows) shared memory");
        if (er) return er;
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(TM), LDS_BYTES, s, b, L.S, np, nq, na, ks, c, src, w, slot16, rows, n_rows,
                           hot_skip, dst, dbg);
        return 0;
    };
    e = ks == 24 ? launch(zpf_hot_kernel<12, true>) : ks < 24 ? launch(zpf_hot_kernel<12, false>) :
        ks == 32 ? launch(zpf_hot_kernel<16, true>) : launch(zpf_hot_kernel<16, false>);
    if (e) return e;
    e = eap::check_launch("inter_zpconv_forward (on-chip rows)");
    if (e) return e;
    eap::set_kernel("zpf_hot_kernel");
    if (!(dbg & 32)) e = eap::zpconv_index_check(b, np, na * ks * ann, ann, idx, nullptr, nullptr, flag, side);
    if (e) return e;
    // clouds with more referenced rows than the LDS holds: the matrix kernel (its workgroups leave at once for the others)
    e = eap::inter_zpconv_mfma_fwd(b, np, nq, na, ks, ann, c, idx0, w, src, mat_skip, dst, s);
    if (e) return e;
    e = joiner.join();
    if (e) return e;
    // clouds whose 5-D index is not one list per point: recomputed by the arbitrary-index kernel
    return eap::inter_zpconv_rows_fwd(b, np, nq, na, ks, ann, c, idx, w, src, dst, flag, s);
}
