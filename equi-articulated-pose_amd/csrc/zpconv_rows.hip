// csrc/zpconv_rows.hip -- the native inter "zpconv" forward (zpconv_cuda.cpp:L41-56, kernel
// zpconv_cuda_kernel.cu:L33-73) near HBM speed for the index pattern the Python layer produces.
//
//   out[b,c,k,p,a] = sum_n w[b,p,a,k,n] * feats[b,c,idx[b,p,a,k,n],a]
//
// The signature carries a 5-D index, but every caller in the reference builds it by broadcasting
// ONE neighbour list per point over (a,k) (spconv/functional.py:L232-249,
// so3conv/functional.py:L2508-2549).  The op is then HBM-bound: idx + w read once, out written
// once = 4.59 GB per 4096-point cloud at C = 64 against 4.8e10 flop (~10 flop/B).  The generic
// kernel (csrc/zpconv.hip) re-gathers a feature row per (a,k,n) tuple -- 96 GB of L2 traffic per
// cloud -- and sits at 3-5 % of the HBM roofline.
//
// Here a workgroup owns (point, 8 kernel points, 64 channels), lanes run along anchor quads:
//   1. it streams its idx slab [A][8][NN] once (coalesced 16-byte reads), comparing every row with
//      one reference row -- that IS the index read the op is charged for -- and its w slab once,
//      transposed into LDS as [(n&3, k, n>>2)][anchor] (139 KB) so that a lane reads the weights of
//      its own 4 anchors as one 16-byte word;
//   2. rows agree: each wave takes 8 channels and walks the NN neighbours once; a lane owns 4
//      anchors of 2 channels, so one 16-byte load per lane brings 4 feature rows (240 B each) per
//      wave-instruction, and 8 x 2 x 4 packed FMAs per lane follow against 8 weight words read from
//      LDS one neighbour ahead; rows of 4 neighbours in flight;
//   3. rows differ (arbitrary 5-D index): same sums with a per-anchor, per-kernel-point index read
//      from global memory -- same result, the reference's speed class;
//   4. every output row is a full 240-byte anchor row (no partial lines).
// Summation order over the neighbours is sequential, as in the reference kernel.
//
// Measured (tools/zpconv_roofline.py, B = 2 x 4096 points, C = 64): 28-32 % of the 8 TB/s HBM
// roofline (3-5 % for the generic kernel).  What holds it there: the slab phase runs at the per-CU
// streaming rate (8.8 B/clk/CU = 4.7 TB/s over the chip) but does not overlap the multiply phase
// (the 139 KB tile leaves room for one workgroup per CU), and the feature rows of a point are
// re-read for each of its three kernel-point tiles (983 KB each at C = 64; served by L2 only when
// the neighbour rows are hot).  Variants tried and kept under tools/experiments/: MFMA with 8/16
// anchors per workgroup (32-byte row segments: TA-bound), 4-kernel-point tiles with two
// workgroups per CU or with dedicated loader waves (row re-reads double: MALL-bound), and
// half-neighbour tiles with 4 loader + 8 compute waves persistent over 4 points (same speed).
#include "common.h"

namespace {

constexpr int KT = 8;          // kernel points per workgroup
constexpr int CT = 8;          // channels per wave: 4 row groups of lanes x 2 channels
constexpr int NWV = 8;
constexpr int TM = 64 * NWV;

// LDS pitch of one (k, n) weight row over the anchors: 16-byte aligned rows, 4 mod 32 banks
__host__ __device__ inline int w_pitch(int na) { return ((na + 3) & ~3) + ((((na + 3) >> 2) & 7) == 1 ? 0 : 4) + 4; }

__global__ __launch_bounds__(TM, 2) void inter_zpconv_rows_kernel(
    int np, int nq, int na, int ks, int nn, int c, int nkt, const int32_t *__restrict__ idx,
    const float *__restrict__ w, const float *__restrict__ feats, const int32_t *__restrict__ only_flagged,
    float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // behind the matrix path (csrc/zpconv_mfma.hip) this kernel serves only the clouds whose index is irregular
    if (only_flagged != nullptr && __builtin_amdgcn_readfirstlane(only_flagged[blockIdx.z]) == 0) return;
    const int PW = w_pitch(na);
    float *s_w = reinterpret_cast<float *>(smem);                  // [(n&3, k, n>>2)][PW]
    int *s_q = reinterpret_cast<int *>(s_w + KT * nn * PW);        // [nn]

    // consecutive workgroups = the kernel-point tiles of one point, then the next point; XCDs
    // get contiguous ranges (feature rows of a point are fetched into one L2, output lines that
    // straddle two points meet in one L2)
    const int unit = xcd_point(blockIdx.x, np * nkt);
    const int p = unit / nkt, kt = unit - p * nkt, bi = blockIdx.z;
    const int k0 = kt * KT, kcnt = min(KT, ks - k0);
    const int t = threadIdx.x, lane = t & 63;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    // lane -> (row group rg: which pair of the wave's 8 channels, anchor quad aq): one 16-byte
    // load per lane fetches 4 feature rows (240 B each at 60 anchors) per wave-instruction
    const int npiece = na >> 2;
    const int rg_raw = lane / npiece, rg = min(rg_raw, 3), aq = min(lane - rg_raw * npiece, npiece - 1);
    const bool lane_on = rg_raw < 4;
    const int c0 = blockIdx.y * (NWV * CT) + wave_u * CT + 2 * rg;  // first of this lane's two channels

    // ---- 1. slabs: idx compared against one reference row, w transposed into LDS ----------------
    const size_t pbase = ((size_t)bi * np + p) * na;              // (cloud, point) -> first anchor row
    const int qpr = nn >> 2, per_a = kcnt * qpr, nquad = na * per_a;
    const int4 *ref4 = reinterpret_cast<const int4 *>(idx + (pbase * ks + k0) * nn);
    // a thread always meets the same 16-byte column of the reference row when the row length
    // divides the stride of its pieces: one reference load instead of one per piece
    const bool ref_fixed = (TM % qpr) == 0;
    const int4 ref_mine = ref4[t % qpr];
    int mismatch = 0;
    // software-pipelined: the loads of batch i+1 (4 pieces = 8 x 16 bytes per thread) are in flight
    // while batch i is compared and scattered into LDS, so the HBM stream never pauses
    auto load_batch = [&](int f0, int4 (&iv)[4], float4 (&wv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = min(f0 + j * TM, nquad - 1);
            const int a = f / per_a, rem = f - a * per_a;        // rem = kl*qpr + q4: contiguous in memory
            const size_t g = (((pbase + a) * ks + k0) * nn >> 2) + rem;
            iv[j] = reinterpret_cast<const int4 *>(idx)[g];
            wv[j] = reinterpret_cast<const float4 *>(w)[g];
        }
    };
    auto store_batch = [&](int f0, const int4 (&iv)[4], const float4 (&wv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = f0 + j * TM;
            if (f < nquad) {
                const int a = f / per_a, rem = f - a * per_a;
                const int4 ir = ref_fixed ? ref_mine : ref4[rem % qpr];
                mismatch |= (iv[j].x ^ ir.x) | (iv[j].y ^ ir.y) | (iv[j].z ^ ir.z) | (iv[j].w ^ ir.w);
                // row(kl, n) = (n & 3) * KT*qpr + kl*qpr + (n >> 2): consecutive lanes (consecutive rem)
                // write consecutive rows
                float *dst = s_w + (size_t)rem * PW + a;
                const size_t js = (size_t)KT * qpr * PW;
                dst[0] = wv[j].x; dst[js] = wv[j].y; dst[2 * js] = wv[j].z; dst[3 * js] = wv[j].w;
            }
        }
    };
    {
        int4 iva[4], ivb[4];
        float4 wva[4], wvb[4];
        load_batch(t, iva, wva);
        for (int f0 = t; f0 < nquad; f0 += 8 * TM) {
            load_batch(f0 + 4 * TM, ivb, wvb);                   // clamped past the end (harmless reload)
            store_batch(f0, iva, wva);
            load_batch(f0 + 8 * TM, iva, wva);
            store_batch(f0 + 4 * TM, ivb, wvb);
        }
    }
    for (int n = t; n < nn; n += TM) s_q[n] = idx[(pbase * ks + k0) * nn + n];
    const bool irregular = __syncthreads_or(mismatch != 0) != 0;
    if (blockIdx.y * (NWV * CT) + wave_u * CT >= c) return;       // wave-uniform

    float4 acc[KT][2];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[k][j] = make_float4(0.f, 0.f, 0.f, 0.f);

    const size_t f_cs = (size_t)nq * na;
    const float *fb = feats + (size_t)bi * c * f_cs + 4 * aq;
    unsigned coff[2];                                              // channel offsets (clamped: never stored past c)
#pragma unroll
    for (int j = 0; j < 2; ++j) coff[j] = (unsigned)(min(c0 + j, c - 1) * f_cs);
    const float *wl = s_w + 4 * aq;                                // this lane's 4 anchors in every weight row

    if (!irregular) {
        // ---- 2. one neighbour list for the whole point -------------------------------------------
        auto fetch = [&](int n, float4 (&fv)[2]) {
            const unsigned q = (unsigned)__builtin_amdgcn_readfirstlane(s_q[min(n, nn - 1)]) * (unsigned)na;
#pragma unroll
            for (int j = 0; j < 2; ++j) fv[j] = *reinterpret_cast<const float4 *>(fb + coff[j] + q);
        };
        auto wread = [&](int n, float4 (&wk)[KT]) {
            const int nc = min(n, nn - 1);
            const float *wr = wl + ((size_t)(nc & 3) * KT * qpr + (nc >> 2)) * PW;
#pragma unroll
            for (int k = 0; k < KT; ++k) wk[k] = *reinterpret_cast<const float4 *>(wr + (size_t)k * qpr * PW);
        };
        auto fma_all = [&](const float4 (&wk)[KT], const float4 (&fv)[2]) {
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[k][j].x = fmaf(fv[j].x, wk[k].x, acc[k][j].x); acc[k][j].y = fmaf(fv[j].y, wk[k].y, acc[k][j].y);
                    acc[k][j].z = fmaf(fv[j].z, wk[k].z, acc[k][j].z); acc[k][j].w = fmaf(fv[j].w, wk[k].w, acc[k][j].w);
                }
        };
        float4 fv[4][2];                                           // rows of 4 neighbours in flight
#pragma unroll
        for (int d = 0; d < 4; ++d) fetch(d, fv[d]);
        float4 wa[KT], wb[KT];                                     // weight words one neighbour ahead of the FMAs
        wread(0, wa);
        for (int n = 0; n < nn; n += 4) {
            float4 gv[4][2];
#pragma unroll
            for (int d = 0; d < 4; ++d) fetch(n + 4 + d, gv[d]);   // clamped past the end (harmless reload)
            wread(n + 1, wb);
            fma_all(wa, fv[0]);
            wread(n + 2, wa);
            if (n + 1 < nn) fma_all(wb, fv[1]);
            wread(n + 3, wb);
            if (n + 2 < nn) fma_all(wa, fv[2]);
            wread(n + 4, wa);
            if (n + 3 < nn) fma_all(wb, fv[3]);
#pragma unroll
            for (int d = 0; d < 4; ++d) { fv[d][0] = gv[d][0]; fv[d][1] = gv[d][1]; }
        }
    } else {
        // ---- 3. arbitrary index: every (anchor, kernel point) has its own neighbour ---------------
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            if (k < kcnt) {
                for (int n = 0; n < nn; ++n) {
                    const float4 wk = *reinterpret_cast<const float4 *>(wl + ((size_t)(n & 3) * KT * qpr + (size_t)k * qpr + (n >> 2)) * PW);
                    const int32_t *ip = idx + ((pbase + 4 * aq) * ks + k0 + k) * nn + n;
                    const size_t as = (size_t)ks * nn;           // anchor stride in idx
                    const unsigned q0 = (unsigned)ip[0] * (unsigned)na, q1 = (unsigned)ip[as] * (unsigned)na;
                    const unsigned q2 = (unsigned)ip[2 * as] * (unsigned)na, q3 = (unsigned)ip[3 * as] * (unsigned)na;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float *fr = fb + coff[j];
                        acc[k][j].x = fmaf(fr[q0], wk.x, acc[k][j].x); acc[k][j].y = fmaf(fr[q1 + 1], wk.y, acc[k][j].y);
                        acc[k][j].z = fmaf(fr[q2 + 2], wk.z, acc[k][j].z); acc[k][j].w = fmaf(fr[q3 + 3], wk.w, acc[k][j].w);
                    }
                }
            }
        }
    }

    // ---- 4. out[b, c, k, p, :]: full anchor rows --------------------------------------------------
    if (lane_on) {
        const size_t o_ks = (size_t)np * na, o_cs = (size_t)ks * np * na;
        float *ob = out + (size_t)bi * c * o_cs + (size_t)p * na + 4 * aq;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (c0 + j < c)
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k < kcnt) *reinterpret_cast<float4 *>(ob + (size_t)(c0 + j) * o_cs + (size_t)(k0 + k) * o_ks) = acc[k][j];
    }
}

}  // namespace

namespace eap {

// true if the row kernel can serve these sizes (otherwise the caller uses csrc/zpconv.hip)
bool inter_zpconv_rows_supported(int np, int nq, int na, int ks, int nn, int c) {
    if (na <= 0 || na > 64 || (na & 3) != 0 || ks <= 0 || nn <= 0 || (nn & 3) != 0) return false;
    if (sizeof(float) * KT * (size_t)nn * w_pitch(na) + 4 * (size_t)nn > 160 * 1024) return false;
    if ((long long)c * nq * na >= (1ll << 31)) return false;
    return c >= 8;
}

int inter_zpconv_rows_fwd(int b, int np, int nq, int na, int ks, int nn, int c, const int32_t *idx, const float *w,
                          const float *feats, float *out, const int32_t *only_flagged, hipStream_t s) {
    const int nkt = (ks + KT - 1) / KT;
    const size_t shmem = sizeof(float) * KT * (size_t)nn * w_pitch(na) + 4 * (size_t)nn;
    int e = eap::hip_fail(hipFuncSetAttribute((const void *)inter_zpconv_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                          "inter_zpconv_forward shared memory");
    if (e) return e;
    dim3 grid(np * nkt, (c + NWV * CT - 1) / (NWV * CT), b);
    hipLaunchKernelGGL(inter_zpconv_rows_kernel, grid, dim3(TM), shmem, s, np, nq, na, ks, nn, c, nkt, idx, w, feats, only_flagged, out);
    return eap::check_launch("inter_zpconv_forward (rows)");
}

}  // namespace eap
