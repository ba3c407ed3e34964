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

#ifdef EAP_INV_TRACE
__device__ unsigned long long eap_inv_trace[8 * 16];
#define TR(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); tr[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define TR(i)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CB = 32;        // dY channels per block (one MFMA M tile)
constexpr int NBK = 8;        // entries per LDS stage (4 MFMA k-steps)
constexpr int FPMAX = 68;     // LDS pitch of one staged row: 60 floats for <= 60 anchors, 68 for 61..64 (16-byte
                              // aligned, never a multiple of 32 banks)
constexpr int APW = 8;        // max anchors per wave
constexpr int NWV = 8;
constexpr int TM = 64 * NWV;

template <bool HAS_MULT>
__global__ __launch_bounds__(TM, 2) void so3_inter_group_inv_kernel(
    int o, int p, int nn, int na, int ks, int rcap, float inv_sigma, int identity_anchor,
    const float *__restrict__ gy,
    const int32_t *__restrict__ rows, const int32_t *__restrict__ off, const int32_t *__restrict__ cnt,
    const int32_t *__restrict__ ent_p, const float4 *__restrict__ ent_gx, const float *__restrict__ rk,
    const uint8_t *__restrict__ multinv, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int FP = na <= 60 ? 60 : FPMAX, FP_ = FP;
    float *s_f = reinterpret_cast<float *>(smem);                           // [2][NBK][CB][FP]
    float4 *s_g = reinterpret_cast<float4 *>(s_f + 2 * NBK * CB * FP_);       // [2][NBK]
    float4 *s_rk = s_g + 2 * NBK;                                           // [na][ks] scaled kernel points (HAS_MULT)
    uint8_t *s_mult = reinterpret_cast<uint8_t *>(s_rk + (HAS_MULT ? na * ks : 0));

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
    const int q = rows[(size_t)bi * rcap + ri];
    const int n_ent = q >= 0 ? cnt[(size_t)bi * rcap + ri] : 0;
    const size_t e0 = (size_t)bi * p * nn + (q >= 0 ? off[(size_t)bi * rcap + ri] : 0);

    if (HAS_MULT) {
        const int words = (na * na) >> 2;
        for (int i = t; i < words; i += TM)
            reinterpret_cast<uint32_t *>(s_mult)[i] = reinterpret_cast<const uint32_t *>(multinv)[i];
        for (int i = t; i < na * ks; i += TM) {       // w = max(0, base_e + kc + g . k')  (see step())
            const float x = rk[i * 3], y = rk[i * 3 + 1], z = rk[i * 3 + 2];
            s_rk[i] = make_float4(2.f * inv_sigma * x, 2.f * inv_sigma * y, 2.f * inv_sigma * z,
                                  -inv_sigma * (x * x + y * y + z * z));
        }
    }

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // 8 waves, contiguous anchor ranges that all start at an EVEN anchor (8,8,8,8,8,8,6,6 at na = 60)
    // so that a pair of anchors is one aligned 8-byte LDS read; waves w and w+4 share a SIMD
    const int per = min(APW, (((na + NWV - 1) / NWV) + 1) & ~1);
    const int a_beg = min(wave_u * per, na);
    const int a_cnt = max(0, min(per, na - a_beg));
    const int lk = lane & 31, lh = lane >> 5;
    const int lkc = min(lk, ks - 1);
    // kernel weight  w = relu(1 - |g - k|^2 / sigma) = relu(base_e + kc + g . k'),
    //   base_e = 1 - |g|^2/sigma (once per entry),  k' = 2k/sigma,  kc = -|k|^2/sigma (constants):
    // 3 FMAs + add + max per weight instead of 9 operations; unused kernel-point columns carry
    // kc = -1e30 so their weight is 0
    float kx[APW], ky[APW], kz[APW], kc[APW];
    if (!HAS_MULT)
#pragma unroll
    for (int ai = 0; ai < APW; ++ai) {   // register constants (no permutation: anchors map to themselves)
        const float *r3 = rk + ((size_t)min(a_beg + ai, na - 1) * ks + lkc) * 3;
        const float x = r3[0], y = r3[1], z = r3[2];
        kx[ai] = 2.f * inv_sigma * x; ky[ai] = 2.f * inv_sigma * y; kz[ai] = 2.f * inv_sigma * z;
        kc[ai] = lk < ks ? -inv_sigma * (x * x + y * y + z * z) : -1e30f;
    }
    const float kdead = lk < ks ? 0.f : -1e30f;          // permuted path: added to the LDS constant

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
        }
    };
    unsigned row_off[NST];                            // 32-bit element offsets, channel part precomputed
#pragma unroll
    for (int u = 0; u < NST; ++u) {
        const int cl = (u * (TM / 16) + rgrp) % CB;
        row_off[u] = (unsigned)min(c0 + cl, o - 1) * (unsigned)p * (unsigned)na + 4u * (unsigned)pc;
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
            if (piece < npiece)   // the row pitch has no slack for the idle 16th lane
                *reinterpret_cast<float4 *>(s_f + ((size_t)buf * NBK * CB + row) * FP + 4 * piece) =
                    live ? stage[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (t < NBK) s_g[buf * NBK + t] = gtmp;
    };

    const int nchunk = (n_ent + NBK - 1) / NBK;
    if (nchunk > 0) {
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
    auto gather = [&](const float *fbuf, int buf, int s, float (&fa)[APW], int (&aw)[APW]) {
        const int nl = 2 * s + lh;
        const float *frow = fbuf + ((size_t)nl * CB + lk) * FP;
        if (!HAS_MULT) {   // anchors map to themselves: pairs of anchors = one aligned 8-byte read
#pragma unroll
            for (int ai = 0; ai < APW; ai += 2) {
                const float2 v = *reinterpret_cast<const float2 *>(frow + min(a_beg + ai, na - 2));
                fa[ai] = v.x; fa[ai + 1] = v.y;
            }
        } else {
            const int r = __float_as_int(s_g[buf * NBK + nl].w);
#pragma unroll
            for (int ai = 0; ai < APW; ++ai) {
                const int a = (int)s_mult[r * na + min(a_beg + ai, na - 1)];
                aw[ai] = a * ks + lkc;
                fa[ai] = frow[a];
            }
        }
    };
    auto step = [&](int buf, int s, const float (&fa)[APW], const int (&aw)[APW]) {
        const float4 g = s_g[buf * NBK + 2 * s + lh];
        const float base = 1.0f - inv_sigma * (g.x * g.x + g.y * g.y + g.z * g.z);
        if (!HAS_MULT) {
#pragma unroll
            for (int ai = 0; ai < APW; ++ai) {
                float t = fmaf(g.x, kx[ai], kc[ai]);
                t = fmaf(g.y, ky[ai], t);
                t = fmaf(g.z, kz[ai], t);
                const float wv = fmaxf(t + base, 0.0f);
                if (ai < a_cnt)                          // wave-uniform
                    acc[ai] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ai], wv, acc[ai], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int h = 0; h < APW; h += 4) {
                float4 k4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) k4[j] = s_rk[aw[h + j]];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = fmaf(g.x, k4[j].x, k4[j].w + kdead);
                    t = fmaf(g.y, k4[j].y, t);
                    t = fmaf(g.z, k4[j].z, t);
                    const float wv = fmaxf(t + base, 0.0f);
                    if (h + j < a_cnt)
                        acc[h + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[h + j], wv, acc[h + j], 0, 0, 0);
                }
            }
        }
    };

#ifdef EAP_INV_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
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
        int aw0[APW], aw1[APW];
        gather(fbuf, buf, 0, fa0, aw0);
        __builtin_amdgcn_sched_barrier(0);
        gather(fbuf, buf, 1, fa1, aw1);
        __builtin_amdgcn_sched_barrier(0);
        step(buf, 0, fa0, aw0);
        __builtin_amdgcn_sched_barrier(0);
        gather(fbuf, buf, 2, fa0, aw0);
        __builtin_amdgcn_sched_barrier(0);
        step(buf, 1, fa1, aw1);
        __builtin_amdgcn_sched_barrier(0);
        gather(fbuf, buf, 3, fa1, aw1);
        __builtin_amdgcn_sched_barrier(0);
        step(buf, 2, fa0, aw0);
        __builtin_amdgcn_sched_barrier(0);
        step(buf, 3, fa1, aw1);
        TR(2);
        if (ch + 1 < nchunk) stash(buf ^ 1);
        TR(3);
        __syncthreads();
        TR(4);
    }
#ifdef EAP_INV_TRACE
    if (blockIdx.x == 3 && blockIdx.y == 1 && blockIdx.z == 0 && lane == 0) {
        for (int i = 0; i < 8; ++i) eap_inv_trace[wave * 16 + i] = tr[i];
        eap_inv_trace[wave * 16 + 8] = nchunk;
    }
#endif

    // ---- epilogue: Z[b, o, k, ri, a'] (rows that are not referenced write zeros) ----------------
    float *s_o = s_f;
    float *ob = out + (size_t)bi * o * ks * rcap * na + (size_t)ri * na;
    const size_t o_ks = (size_t)rcap * na, o_cs = (size_t)ks * rcap * na;
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        if (lk < ks) {
#pragma unroll
            for (int ai = 0; ai < APW; ++ai) {
                if (ai < a_cnt) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int cl8 = rr + 4 * lh;
                        s_o[((size_t)cl8 * ks + lk) * na + a_beg + ai] = acc[ai][ps * 4 + rr];
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

extern "C" int eap_so3_inter_group_inv_f32(int b, int o, int p, int nn, int na, int ks, int rcap,
                                           float sigma, const float *gy, const int32_t *rows,
                                           const int32_t *off, const int32_t *cnt,
                                           const int32_t *ent_p, const float *ent_gx, const float *rk,
                                           const uint8_t *multinv, int identity_anchor, float *z,
                                           eap_stream_t stream) {
    if (b <= 0 || o <= 0 || rcap <= 0 || na <= 0 || ks <= 0) return 0;
    if (na > 64 || (na & 3) != 0) return eap::bad_arg("so3_inter_group_inv: the anchor count must be a multiple of 4, at most 64");
    if (ks > 32) return eap::bad_arg("so3_inter_group_inv: at most 32 kernel points");
    if ((long long)o * p * na >= (1ll << 31)) return eap::bad_arg("so3_inter_group_inv: one cloud's gradient exceeds 2^31 elements");
    hipStream_t s = eap::S(stream);
    const int FP_ = na <= 60 ? 60 : FPMAX;
    const size_t stage_b = sizeof(float) * 2 * NBK * CB * FP_;
    if (sizeof(float) * 8 * (size_t)ks * na > stage_b) return eap::bad_arg("so3_inter_group_inv: epilogue tile too large");
    size_t shmem = stage_b + 16 * 2 * NBK + (multinv ? 16 * (size_t)na * ks + (size_t)na * na : 0);
    if (shmem > 160 * 1024) return eap::bad_arg("so3_inter_group_inv: LDS budget exceeded");
    dim3 grid(rcap, (o + CB - 1) / CB, b);
    const float4 *g4 = reinterpret_cast<const float4 *>(ent_gx);
    int e;
    if (multinv) {
        auto kern = so3_inter_group_inv_kernel<true>;
        e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem), "so3_inter_group_inv shared memory");
        if (e) return e;
        hipLaunchKernelGGL(kern, grid, dim3(TM), shmem, s, o, p, nn, na, ks, rcap, 1.0f / sigma, identity_anchor, gy, rows, off, cnt, ent_p, g4, rk, multinv, z);
    } else {
        auto kern = so3_inter_group_inv_kernel<false>;
        e = eap::hip_fail(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem), "so3_inter_group_inv shared memory");
        if (e) return e;
        hipLaunchKernelGGL(kern, grid, dim3(TM), shmem, s, o, p, nn, na, ks, rcap, 1.0f / sigma, identity_anchor, gy, rows, off, cnt, ent_p, g4, rk, multinv, z);
    }
#ifdef EAP_INV_TRACE
    {
        unsigned long long h[8 * 16];
        hipDeviceSynchronize();
        hipMemcpyFromSymbol(h, HIP_SYMBOL(eap_inv_trace), sizeof(h));
        for (int w = 0; w < 8; ++w)
            fprintf(stderr, "inv trace wave %d: chunks %llu  fetch %llu  compute %llu  stash %llu  barrier %llu  (cycles/chunk)\n", w,
                    h[w * 16 + 8], h[w * 16 + 1] / h[w * 16 + 8], h[w * 16 + 2] / h[w * 16 + 8], h[w * 16 + 3] / h[w * 16 + 8],
                    h[w * 16 + 4] / h[w * 16 + 8]);
    }
#endif
    return eap::check_launch("so3_inter_group_inv");
}
