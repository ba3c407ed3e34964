// csrc/so3_inter_mfma.hip -- fused SO(3) grouping on the matrix cores.
//
//   X[b,c,k,p,a] = sum_n feats[b,c,idx[b,p,n],perm_n(a)] * w(p,a,k,n)
//   w(p,a,k,n)   = relu(1 - |g(p,n) - A_a kappa_k|^2 / sigma)
// (vgtk/vgtk/so3conv/functional.py:L1112-1261; the einsum 'bcpna,bpakn->bckpa' at L1261 is the
//  same contraction as the reference's zpconv forward, zpconv_cuda_kernel.cu:L33-73.)
//
// For one (point, anchor) pair this is a [C x NN] x [NN x K] product.  It is mapped onto
// v_mfma_f32_32x32x2_f32 with   M = 32 channels,  N = 32 kernel points (24 used),  K = neighbours:
//   * the A operand (features) is staged through LDS: every (channel, neighbour) feature row is
//     one coalesced 240-byte read (15 lanes x 16 B), each row is fetched once per block and then
//     serves all 60 anchors;
//   * the B operand (kernel weights) never exists in memory: lane (k = l&31, n = l>>5) evaluates
//     its own w(p,a,k,n) in registers right before the MFMA that consumes it (about 10 VALU
//     instructions against a 64-cycle MFMA, on a different pipe);
//   * a block owns one point and 32 channels; its 4 waves own 15 anchors each and keep the
//     [32 x 32] accumulators of all of them in registers (240 VGPRs) while the neighbour loop
//     streams features through a double-buffered LDS tile;
//   * the epilogue transposes accumulators through LDS so X is written as full 240-byte anchor rows.
// HBM traffic is the algorithmic minimum (X written once, feats from L2); the kernel is bound by
// fp32 MFMA issue: 2*32*32*NN*A flops per (point, 32 channels) = 4/3 of the useful flops (K padded
// 24 -> 32).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CB = 32;        // channels per block (one MFMA M tile)
constexpr int NBK = 8;        // neighbours per LDS stage (4 MFMA k-steps; the pipeline below is written for 4)
constexpr int FPMAX = 68;     // LDS pitch of one staged row: 60 floats for <= 60 anchors, 68 for 61..64 (16-byte
                              // aligned, never a multiple of 32 banks)
constexpr int NWV = 8;        // waves per block: 2 per SIMD, so one wave's weight VALU overlaps the other's MFMAs
constexpr int TM = 64 * NWV;

// APW = max anchors per wave (8 waves: a quarter of the anchors per SIMD, split 8 + 7 at na = 60);
// EXACT = the anchor count is a multiple of 4 (float4 feature rows)
template <int APW, bool EXACT, bool HAS_MULT>
__global__ __launch_bounds__(TM, 2) void so3_inter_group_fwd_mfma_kernel(
    int c, int p, int n_sup, int nn, int na, int ks, float inv_sigma,
    const float *__restrict__ feats, const int32_t *__restrict__ idx, const float4 *__restrict__ gx,
    const float *__restrict__ rk, const uint8_t *__restrict__ mult, const int32_t *__restrict__ nonident,
    int skip_plain, int blocked, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // clouds without any non-identity relative rotation were served by csrc/so3_inter_lists.hip
    if (skip_plain && __builtin_amdgcn_readfirstlane(nonident[blockIdx.z]) == 0) return;
    const int FP = na <= 60 ? 60 : FPMAX, FP_ = FP;
    float *s_f = reinterpret_cast<float *>(smem);                           // [2][NBK][CB][FP]
    float4 *s_g = reinterpret_cast<float4 *>(s_f + 2 * NBK * CB * FP_);       // [nn_pad]
    const int nn_pad = (nn + NBK - 1) / NBK * NBK;
    int32_t *s_q = reinterpret_cast<int32_t *>(s_g + nn_pad);               // [nn_pad]
    uint8_t *s_mult = reinterpret_cast<uint8_t *>(s_q + nn_pad);            // [na*na]

    const int pi = xcd_point(blockIdx.x, p), c0 = blockIdx.y * CB, bi = blockIdx.z;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t pn = ((size_t)bi * p + pi) * nn;
    for (int i = t; i < nn_pad; i += TM) {
        if (i < nn) {
            s_g[i] = gx[pn + i];
            const int q = idx[pn + i];
            s_q[i] = q < n_sup ? q : -1;    // shadow row (zeros in the reference) -> zero features
        } else {
            s_g[i] = make_float4(1e18f, 1e18f, 1e18f, 0.f);   // padding neighbour: weight 0
            s_q[i] = -1;
        }
    }
    if (HAS_MULT) {   // 32-bit copies (torch allocations are >= 16-byte aligned); byte tail
        const int words = (na * na) >> 2;
        for (int i = t; i < words; i += TM)
            reinterpret_cast<uint32_t *>(s_mult)[i] = reinterpret_cast<const uint32_t *>(mult)[i];
        for (int i = (words << 2) + t; i < na * na; i += TM) s_mult[i] = mult[i];
    }

    // anchors of this wave and this lane's rotated kernel point (k = lane & 31) for each of them
    // wave-uniform by construction; readfirstlane tells the compiler so (no exec-mask branches
    // around the MFMAs)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // a quarter of the anchors per SIMD, split between the two waves that share it (w, w + 4)
    // 8 waves, contiguous anchor ranges that all start at an EVEN anchor (8,8,8,8,8,8,6,6 at na = 60)
    // so that a pair of anchors is one aligned 8-byte LDS read; waves w and w+4 share a SIMD
    const int per = min(APW, (((na + NWV - 1) / NWV) + 1) & ~1);
    const int a_beg = min(wave_u * per, na);
    const int a_cnt = max(0, min(per, na - a_beg));
    const int lk = lane & 31, lh = lane >> 5;
    // kernel weight  w = relu(1 - |g - k|^2 / sigma) = relu(base_n + kc + g . k'),
    //   base_n = 1 - |g|^2/sigma (once per neighbour),  k' = 2k/sigma,  kc = -|k|^2/sigma:
    // 3 FMAs + add + max per weight; unused kernel-point columns carry kc = -1e30 (weight 0)
    // packed pairs: the weights are evaluated two anchors per VALU instruction (csrc/so3_inter_lists.hip has the why)
    f32x2 kxp[APW / 2], kyp[APW / 2], kzp[APW / 2], kcp[APW / 2];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai) {
        const bool ok = ai < a_cnt && lk < ks;
        const float *r3 = rk + ((size_t)min(a_beg + ai, na - 1) * ks + min(lk, ks - 1)) * 3;   // clamped, always valid
        const float x = r3[0], y = r3[1], z = r3[2];
        kxp[ai >> 1][ai & 1] = 2.f * inv_sigma * x; kyp[ai >> 1][ai & 1] = 2.f * inv_sigma * y; kzp[ai >> 1][ai & 1] = 2.f * inv_sigma * z;
        kcp[ai >> 1][ai & 1] = ok ? -inv_sigma * (x * x + y * y + z * z) : -1e30f;
    }
    // identity relative rotations everywhere in this cloud (flag from so3_prep): no table lookups
    const bool plain = !HAS_MULT || (nonident != nullptr && __builtin_amdgcn_readfirstlane(nonident[bi]) == 0);

    f32x16 acc[APW];
#pragma unroll
    for (int ai = 0; ai < APW; ++ai)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ai][r] = 0.f;

    // ---- feature staging: thread -> (row group, 16-byte piece); 16 rows per pass, 16 passes -------
    const int piece = t & 15, rgrp = t >> 4;            // piece < 15 active for na = 60
    const int npiece = (na + 3) >> 2;
    const float *fb = feats + (size_t)bi * c * n_sup * na;
    float4 stage[NBK * CB * 16 / TM];
    // all 16 loads of a stage are issued back to back from clamped (always valid) addresses and
    // masked afterwards -- a predicated load per row would serialise 16 L2 round trips per stage
    constexpr bool vec_ok = EXACT;                      // anchor count is a multiple of 4
    const int pc = min(piece, npiece - 1);
    // fetch(): raw, UNCONDITIONAL loads from clamped (always valid) addresses -- nothing the
    // compiler could turn into 16 predicated, individually awaited loads; masking of shadow
    // rows / channel tail / idle lanes happens in stash(), right before the LDS write
    // 32-bit element offsets (the launcher guarantees c*n_sup*na < 2^31): per staged row the
    // channel part of the address is a per-thread constant, the neighbour part one 24-bit multiply
    unsigned row_off[NBK * CB * 16 / TM];
#pragma unroll
    for (int u = 0; u < NBK * CB * 16 / TM; ++u) {
        const int cl = (u * (TM / 16) + rgrp) % CB;
        row_off[u] = (unsigned)min(c0 + cl, c - 1) * (unsigned)n_sup * (unsigned)na + 4u * (unsigned)pc;
    }
    auto fetch = [&](int n0) {
#pragma unroll
        for (int u = 0; u < NBK * CB * 16 / TM; ++u) {
            const int nl = (u * (TM / 16) + rgrp) / CB;
            const int q = s_q[n0 + nl];
            const float *src = fb + (row_off[u] + __umul24((unsigned)max(q, 0), (unsigned)na));
            if constexpr (vec_ok) {
                stage[u] = *reinterpret_cast<const float4 *>(src);
            } else {
                stage[u].x = src[0];
                stage[u].y = src[min(1, na - 1 - 4 * pc)];
                stage[u].z = src[min(2, na - 1 - 4 * pc)];
                stage[u].w = src[min(3, na - 1 - 4 * pc)];
            }
        }
    };
    auto stash = [&](int buf, int n0) {
#pragma unroll
        for (int u = 0; u < NBK * CB * 16 / TM; ++u) {
            const int row = u * (TM / 16) + rgrp;
            const int nl = row / CB, cl = row - nl * CB;
            float4 v = stage[u];
            if (!vec_ok) {
                if (4 * pc + 1 >= na) v.y = 0.f;
                if (4 * pc + 2 >= na) v.z = 0.f;
                if (4 * pc + 3 >= na) v.w = 0.f;
            }
            const bool live = s_q[n0 + nl] >= 0 && c0 + cl < c && piece < npiece;
            if (piece < npiece)   // the row pitch has no slack for the idle 16th lane
                *reinterpret_cast<float4 *>(s_f + ((size_t)buf * NBK * CB + row) * FP + 4 * piece) =
                    live ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    __syncthreads();
    const int nchunk = nn_pad / NBK;
    fetch(0);
    stash(0, 0);
    __syncthreads();
    // A-operand gather for one MFMA k-step (2 neighbours): the 15 feature reads of a half-wave
    // are issued together, one k-step AHEAD of the MFMAs that consume them, so the LDS latency
    // hides behind 15 x 64 cycles of matrix work instead of stalling every MFMA
    auto gather = [&](const float *fbuf, int n0, int s, float (&fa)[APW]) {
        const int nl = 2 * s + lh;
        const float *frow = fbuf + ((size_t)nl * CB + lk) * FP;
        if (plain) {                                     // wave-uniform; anchor pairs = one aligned 8-byte read
#pragma unroll
            for (int ai = 0; ai < APW; ai += 2) {
                const float2 v = *reinterpret_cast<const float2 *>(frow + min(a_beg + ai, na - 2));
                fa[ai] = v.x; fa[ai + 1] = v.y;
            }
        } else {
            const int r = __float_as_int(s_g[n0 + nl].w);
#pragma unroll
            for (int ai = 0; ai < APW; ++ai) fa[ai] = frow[(int)s_mult[r * na + min(a_beg + ai, na - 1)]];
        }
    };
    auto step = [&](int n0, int s, const float (&fa)[APW]) {
        const float4 g = s_g[n0 + 2 * s + lh];
        const float base = 1.0f - inv_sigma * (g.x * g.x + g.y * g.y + g.z * g.z);
        f32x2 wv[APW / 2];
#pragma unroll
        for (int j = 0; j < APW / 2; ++j) {
            f32x2 x = __builtin_elementwise_fma((f32x2){g.x, g.x}, kxp[j], kcp[j] + (f32x2){base, base});
            x = __builtin_elementwise_fma((f32x2){g.y, g.y}, kyp[j], x);
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] clamp\n\ts_nop 1" : "=v"(wv[j]) : "v"((f32x2){g.z, g.w}), "v"(kzp[j]), "v"(x));
        }
#pragma unroll
        for (int ai = 0; ai < APW; ++ai)
            if (ai < a_cnt)                              // wave-uniform
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ai], wv[ai >> 1][ai & 1], acc[ai], 0, 0, 0);
    };

    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1, n0 = ch * NBK;
        if (ch + 1 < nchunk) fetch(n0 + NBK);
        const float *fbuf = s_f + (size_t)buf * NBK * CB * FP;
        float fa0[APW], fa1[APW];
        gather(fbuf, n0, 0, fa0);
        __builtin_amdgcn_sched_barrier(0);
        gather(fbuf, n0, 1, fa1);
        __builtin_amdgcn_sched_barrier(0);
        step(n0, 0, fa0);
        __builtin_amdgcn_sched_barrier(0);
        gather(fbuf, n0, 2, fa0);
        __builtin_amdgcn_sched_barrier(0);
        step(n0, 1, fa1);
        __builtin_amdgcn_sched_barrier(0);
        gather(fbuf, n0, 3, fa1);
        __builtin_amdgcn_sched_barrier(0);
        step(n0, 2, fa0);
        __builtin_amdgcn_sched_barrier(0);
        step(n0, 3, fa1);
        if (ch + 1 < nchunk) stash(buf ^ 1, n0 + NBK);
        __syncthreads();
    }

    // ---- epilogue: D[i = channel][j = kernel point]; col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // 4 passes of 8 channels through LDS ([8][ks][na] floats), then 240-byte rows to global
    float *s_o = s_f;
    float *ob = out + (size_t)bi * c * ks * p * na + (size_t)pi * na;
    const size_t o_ks = (size_t)p * na, o_cs = (size_t)ks * p * na;
    // transposed output [point*na + a][c*ks + k] (what the contraction reads k-contiguous): the LDS tile is laid out
    // [anchor][8 channels x ks] so that an anchor's 8 * ks values of a pass leave as ONE contiguous run of its output row
    // (768 bytes at ks = 24; round 2 wrote them as 4-byte stores 12 KB apart)
    if (blocked == 2 && (ks & 3) == 0 && (c & 7) == 0) {
        const int run = 8 * ks, pitch = run + 4;                        // floats; 16-byte aligned rows, off the bank period
        const size_t CK = (size_t)c * ks;
        float *obt = out + (size_t)bi * c * o_cs + (size_t)pi * na * CK + (size_t)c0 * ks;
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
            if (c0 + ps * 8 < c) {                                       // block-uniform (c is a multiple of 8)
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
                        const int cl8 = rr + 4 * lh;     // channel within this pass
                        s_o[((size_t)cl8 * ks + lk) * na + a_beg + ai] = acc[ai][ps * 4 + rr];
                    }
                }
            }
        }
        __syncthreads();
        // copy-out: 16 lanes per (channel, k) row (15 float4 at na = 60), 16 rows per sweep; the
        // (channel, k) pair advances incrementally -- no integer divisions in the loop
        {
            const int rows = 8 * ks;
            int cl8 = rgrp / ks, k = rgrp - cl8 * ks;
            for (int row = rgrp; row < rows; row += TM / 16) {
                const int ci = c0 + ps * 8 + cl8;
                if (ci < c && piece < npiece) {
                    const float *src = s_o + (size_t)row * na + 4 * pc;
                    float *dst = ob + (size_t)ci * o_cs + (size_t)k * o_ks + 4 * pc;
                    // blocked output [b][point][anchor quad][c][k][4] (see csrc/so3_inter_lists.hip)
                    if (blocked == 1) dst = out + (size_t)bi * c * o_cs + ((((size_t)pi * npiece + pc) * c + ci) * ks + k) * 4;
                    if (blocked == 2) {   // transposed output [point*na + a][c*ks + k]
                        float *dt = out + (size_t)bi * c * o_cs + ((size_t)pi * na + 4 * pc) * ((size_t)c * ks) + (size_t)ci * ks + k;
                        for (int j = 0; j < 4 && 4 * pc + j < na; ++j) dt[(size_t)j * c * ks] = src[j];
                    } else if (vec_ok) {
                        *reinterpret_cast<float4 *>(dst) = *reinterpret_cast<const float4 *>(src);
                    } else {
                        for (int j = 0; j < 4 && 4 * pc + j < na; ++j) dst[j] = src[j];
                    }
                }
                k += TM / 16;
                while (k >= ks) { k -= ks; ++cl8; }
            }
        }
        __syncthreads();
    }
}

}  // namespace

int eap::group_fwd_mfma(int b, int c, int p, int n, int nn, int na, int ks, float sigma, const float *feats,
                        const int32_t *idx, const float *gx, const float *rk, const uint8_t *mult,
                        const int32_t *nonident, int skip_plain, int blocked, float *out, hipStream_t s) {
    if (b <= 0 || c <= 0 || p <= 0 || na <= 0 || ks <= 0) return 0;
    if (na > 64) return eap::bad_arg("so3_inter_group_fwd_mfma: at most 64 anchors");
    if (ks > 32) return eap::bad_arg("so3_inter_group_fwd_mfma: at most 32 kernel points");
    if ((long long)c * n * na >= (1ll << 31)) return eap::bad_arg("so3_inter_group_fwd_mfma: one cloud's features exceed 2^31 elements");
    if (!nonident) skip_plain = 0;
    const int FP_ = na <= 60 ? 60 : FPMAX;
    if (nn <= 0)
        return eap::hip_fail(hipMemsetAsync(out, 0, sizeof(float) * (size_t)b * c * ks * p * na, s), "so3_inter_group_fwd memset");
    const int nn_pad = (nn + NBK - 1) / NBK * NBK;
    size_t shmem = sizeof(float) * 2 * NBK * CB * FP_ + 20 * (size_t)nn_pad + (mult ? (size_t)na * na : 0);
    const size_t epi = sizeof(float) * (8 * (size_t)ks + 4) * na;      // the transposed epilogue pads its rows by 4 floats
    if (epi > sizeof(float) * 2 * NBK * CB * FP_) return eap::bad_arg("so3_inter_group_fwd_mfma: epilogue tile too large");
    dim3 grid(p, (c + CB - 1) / CB, b);
    const float4 *g4 = reinterpret_cast<const float4 *>(gx);
    const float inv_sigma = 1.0f / sigma;
    int e = 0;
#define EAP_MFMA_LAUNCH(APW_, EXACT_, MULT_)                                                                  \
    do {                                                                                                      \
        auto kern = so3_inter_group_fwd_mfma_kernel<APW_, EXACT_, MULT_>;                                     \
        e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                              (int)shmem), "so3_inter_group_fwd_mfma shared memory");         \
        if (e) return e;                                                                                      \
        hipLaunchKernelGGL(kern, grid, dim3(TM), shmem, s, c, p, n, nn, na, ks, inv_sigma, feats, idx, g4,    \
                           rk, mult, nonident, skip_plain, blocked, out);                                     \
    } while (0)
    if ((na & 3) == 0) { if (mult) EAP_MFMA_LAUNCH(8, true, true); else EAP_MFMA_LAUNCH(8, true, false); }
    else if (mult) EAP_MFMA_LAUNCH(8, false, true);
    else EAP_MFMA_LAUNCH(8, false, false);
#undef EAP_MFMA_LAUNCH
    return eap::check_launch("so3_inter_group_fwd_mfma");
}

extern "C" int eap_so3_inter_group_fwd_mfma_f32(int b, int c, int p, int n, int nn, int na, int ks,
                                                float sigma, const float *feats, const int32_t *idx,
                                                const float *gx, const float *rk, const uint8_t *mult,
                                                const int32_t *nonident, float *out, eap_stream_t stream) {
    return eap::group_fwd_mfma(b, c, p, n, nn, na, ks, sigma, feats, idx, gx, rk, mult, nonident, 0, 0, out, eap::S(stream));
}
