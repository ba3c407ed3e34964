// csrc/so3_dense.hip -- the fused inter conv RE-ASSOCIATED OVER ITS REFERENCED SUPPORT ROWS as a dense product on the
// fp16 matrix cores (round 5).
//
//   reference (vgtk/vgtk/so3conv/functional.py:L1112-1261 + so3conv/modules.py:L48-55), identity poses:
//       Y[o,p,a] = sum_{c,k} W[o,c,k] sum_n F[c,idx[p,n],a] w(p,a,k,n),   w = relu(1 - |x_idx[p,n] - x_p - A_a kappa_k|^2 / sigma)
//
// The reference's ball query keeps the FIRST nsample hits in index order (grouping_cuda_kernel.cu:L68-113); with the large
// radii of the deep layers a cloud's lists name only R ~ 76..280 of its support rows, and every point's nsample = 64
// neighbours are 64 of those R.  Written over the referenced rows r = 0..R-1 the operator is a DENSE matrix with a mask,
//       Wd[p,(k,r),a] = m[p,r] relu(1 - |x_r - x_p - A_a kappa_k|^2 / sigma),      m[p,r] = 1 iff row r is in p's list,
// and both directions are plain GEMMs per (cloud, anchor) whose second operand is GENERATED IN REGISTERS:
//       backward  Z[o,(k,r)]  = sum_p      dY[o,p]     Wd[p,(k,r)]        M = O, N = K R, contraction over the P points
//       forward   Y[o,p]      = sum_(k,r)  G[o,(k,r)]  Wd[p,(k,r)]        G[o,(k,r),a] = sum_c W[o,c,k] F[c,r,a]  (a small GEMM)
// (dF, dW follow from Z exactly as from the inverse-list kernel's, vgtk/so3conv/functional.py _InterConv.backward).  That is
// 2.1 x the flops of the sparse form at R = 136 -- on a pipe that is 16 x faster than the fp32 MFMA the sparse kernels
// (csrc/so3_inter_lists2.hip) are tied to, with every dY row fetched ~13 x instead of 64 x.
//
// Arithmetic: fp32-accurate products from TWO fp16 PLANES per operand, exactly as csrc/gemm_bf16x3.hip (PL = 2): the stored
// operand is scaled per row by a power of two and split h = fp16(x), l = fp16(x - h) once (dense_split_kernel); the
// generated operand w in [0, 1] is split in registers; a product is h h' + h l' + l h' -- three
// v_mfma_f32_32x32x16_f16 -- with fp32 accumulation, |error| <= 2^-21 sum |a||b| + 2^-25 sum |a|.
//
// The generated operand.  With x~ = x - centre (the cloud's centroid) and u = x~_r - A_a kappa_k, one float4 table per side
// (kr[a][(k,r)] and pt[p], both evaluated in float64 and rounded once), two forms (template FORM, eap_so3_dense_form):
//   1 (default)  w = clamp(1 - |u - x~_p|^2 / sigma): 3 subtractions, 3 multiply-adds for the square, one fma with the [0, 1]
//                clamp as its output modifier -- the reference's own order of operations up to the association of
//                x_r - x_p - A_a kappa_k, absolute error ~2e-7 on a weight (what the reference's fp32 evaluation has itself);
//   0            the square expanded: w = clamp([1 - |u|^2/sigma] + [-|x~_p|^2/sigma] + u . [2 x~_p / sigma]): 1 add + 3 fma, but
//                partial sums up to ~6.5 at the bench radii -> ~7e-7 absolute (bar on the weights: 2e-6, tests/test_gpu_dense.py).
// The weights are NOT scaled: l = fp16(w - fp16(w)) is a subnormal for w < 0.12 and then carries an absolute error <= 2^-25,
// below the evaluation error above.  The mask is one bit per generated weight -- a dword per lane and k-step, riding in the
// operand ring -- and enters as the ADDEND of the weight's last fma (1.0 for a list member, else 0.0: the clamp then returns 0).
// (First version: 64-bit lane masks through the scalar cache + v_cndmask; every use made the wave wait for all its outstanding
// LDS reads as well -- scalar loads return out of order -- 2.2 of 16.5 ms, profiles/r05_dense_ablation.txt.)
//
// Geometry (kc_gemm_kernel): 4 waves, one per SIMD; block tile 256 x 256, wave tile 256 x 64 = 256 accumulator registers;
// k-step 32 (two MFMA k-blocks).  The stored operand lives in HBM already in FRAGMENT ORDER ([k-block][row tile][plane][lane]
// 16-byte pieces, written by dense_split_kernel), so a k-step of a block is 32 contiguous KB moved by 32 global->LDS DMA
// instructions and read back with conflict-free ds_read_b128; the side of the generated operand that runs along k (16 float4
// per k-block) rides in the same ring; four stages, one barrier per k-step.  Every wave generates the operand of ITS 64
// columns for all 256 rows: 16 weights per lane and k-block against 48 matrix instructions, ~2.5 vector instructions per
// matrix instruction, issued in their shadow.
#include <atomic>
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
typedef const __attribute__((address_space(4))) unsigned long long *cu64p;      // constant address space: uniform loads go through the scalar cache

constexpr int KC_BK = 32;                 // contraction elements per k-step (two MFMA k-blocks of 16)
constexpr int KC_LIST_BYTES = 2048;       // LDS behind the operand ring for a column block's k-step list: at most 512 k-steps with a list

// The DENSE INDEX d of a (kernel point k, row slot r) pair: d = (r / 16) 16 ks + 16 k + r % 16 -- groups of 16 row slots outermost,
// then the kernel point, then the 16 slots.  Rows are sorted longest list first and every cloud has its own count R <= rp, so a
// cloud uses the PREFIX d < ceil16(R) ks of the index range: the product kernel ends its column blocks (backward) / its k-steps
// (forward) there instead of multiplying by the empty slots of the batch-wide rp (16 % of the work at the bench clouds:
// 86..136 rows per cloud).  16 consecutive d = 16 rows of one kernel point; rp % 16 == 0.
__host__ __device__ __forceinline__ void dense_kr(int d, int ks, int &k, int &r) {
    const int g = d / (16 * ks), rem = d - g * 16 * ks;
    k = rem >> 4;
    r = 16 * g + (rem & 15);
}

__device__ inline unsigned lds_addr(const void *ptr) {
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)ptr;
}
// wave-wide global -> LDS DMA (16 / 4 bytes per lane, destination = lds_dst + 16 / 4 * lane), invisible to hipcc's waitcnt
// bookkeeping on purpose: the k-loop waits with its own counted s_waitcnt before the step's barrier
__device__ inline void glds16s(const void *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ inline void glds4s(const void *sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// the scale 2^(14 - e) of a row whose largest magnitude is v in [2^e, 2^(e+1)); 1 for 0, inf, nan (as csrc/gemm_bf16x3.hip)
__device__ __forceinline__ float pow2_scale(float v) {
    const unsigned b = __float_as_uint(v) & 0x7fffffffu;
    const int e = (int)(b >> 23) - 127;
    if (b == 0u || e == 128) return 1.0f;
    const int se = max(-120, min(120, 14 - max(e, -126)));
    return __uint_as_float((unsigned)(se + 127) << 23);
}

// x (scaled) = h + l + e, h = fp16(x), l = fp16(x - h), both to nearest even; two values -> the packed h word and l word.
// v_fma_mix{lo,hi}_f16 form fp16(-1 * h + x) in one instruction each, reading h's half straight from the packed word.
__device__ __forceinline__ void split2(float x0, float x1, unsigned &h, unsigned &l) {
#ifdef EAP_DENSE_PLAIN_SPLIT
    const f16x2 hh = __builtin_convertvector((f32x2){x0, x1}, f16x2);
    const f16x2 ll = __builtin_convertvector((f32x2){x0 - (float)hh.x, x1 - (float)hh.y}, f16x2);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
#else
    // (one statement: hipcc pads every asm statement whose result the next instruction reads with a wait state)
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(h), "=&v"(l) : "v"(x0), "v"(x1));
#endif
}

// ------------------------------------------------------------------------------------------------------------------------
// membership of the referenced rows in every point's neighbour list
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_slots_kernel(int n_sup, int rp, int rows_ld, const int32_t *__restrict__ rows,
                                                          const int32_t *__restrict__ n_rows, int32_t *__restrict__ slot_of) {
    const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= min(n_rows[b], rp)) return;
    const int r = rows[(size_t)b * rows_ld + j];
    if ((unsigned)r < (unsigned)n_sup) slot_of[(size_t)b * n_sup + r] = j;
}

// memb[b][p][w] (w < MEMB_WORDS = 16 words per point, whatever rp) bit i = row slot 32 w + i is named by p's list; flags[b] |= 1 when a list names a row twice (the ball
// query pads short lists with their first hit, grouping_cuda_kernel.cu:L98-107: such clouds stay on the list kernels),
// |= 2 when a list names a row without a slot (more referenced rows than `rp`)
constexpr int MEMB_WORDS = 16;           // rp <= 512
__global__ __launch_bounds__(256) void dense_member_kernel(int p, int n_sup, int nn, int rp, const int32_t *__restrict__ idx,
                                                           const int32_t *__restrict__ slot_of, unsigned *__restrict__ memb,
                                                           int32_t *__restrict__ flags) {
    __shared__ unsigned sw[256][MEMB_WORDS + 1];
    const int b = blockIdx.y, pt = blockIdx.x * 256 + threadIdx.x;
    if (pt >= p) return;
    constexpr int W = MEMB_WORDS;
    unsigned *w = sw[threadIdx.x];
    for (int k = 0; k < W; ++k) w[k] = 0u;
    const int32_t *list = idx + ((size_t)b * p + pt) * nn;
    int valid = 0, bad = 0;
    for (int n = 0; n < nn; ++n) {
        const int i = list[n];
        if ((unsigned)i >= (unsigned)n_sup) continue;          // shadow entry: a zero feature row
        const int s = slot_of[(size_t)b * n_sup + i];
        if (s < 0 || s >= rp) { bad |= 2; continue; }
        ++valid;
        w[s >> 5] |= 1u << (s & 31);
    }
    int bits = 0;
    for (int k = 0; k < W; ++k) { bits += __popc(w[k]); memb[((size_t)b * p + pt) * W + k] = w[k]; }
    if (bits != valid) bad |= 1;
    if (bad) atomicOr(&flags[b], bad);
}

// bits[b][wave tile wt][step][lane]: bit 16 t + 8 j + e of the lane's dword  <->  column n = 64 wt + 32 j + (lane & 31), contraction index
// kk = 32 step + 16 t + 8 (lane >> 5) + e -- the 32 weights the lane generates in a k-step of the product kernel
//   dir 0 (backward): kk = point, n = dense index of (k, r), n < ks rp        dir 1 (forward): kk = dense index of (k, r), n = point
__global__ __launch_bounds__(256) void dense_mask_kernel(int p, int ks, int rp, int dir, int wtiles, int steps,
                                                         const unsigned *__restrict__ memb, unsigned *__restrict__ bits) {
    constexpr int W = MEMB_WORDS;
    const int b = blockIdx.y, lane = threadIdx.x & 63, li = lane & 31, kg = lane >> 5;
    const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= (long long)wtiles * steps) return;
    const int wt = (int)(wv / steps), step = (int)(wv - (long long)wt * steps);
    const int nkr = ks * rp;
    unsigned word = 0;
    // The dense index keeps a group of 16 row slots together (d = (r / 16) 16 ks + 16 k + r % 16), so a wave's 32 weights per lane meet
    // few membership words: (first version: 32 dependent word reads per lane, 0.25 ms per launch)
    if (dir == 0 && (ks & 3) == 0) {
        // backward: the wave tile's 64 columns are 4 kernel points x the 16 rows of ONE group g; the row of a column is 16 g + (lane & 15)
        // for both column halves j; the k-step's 32 points need one word each -- lane l holds the word of point 32 step + (l & 31)
        const int g = (64 * wt) / (16 * ks), pt_own = 32 * step + li;
        const unsigned own = (pt_own < p && 2 * (g >> 1) < 2 * W) ? memb[((size_t)b * p + pt_own) * W + (g >> 1)] : 0u;
        const int sh = 16 * (g & 1) + (li & 15);
        const bool col0 = 64 * wt + li < nkr, col1 = 64 * wt + 32 + li < nkr;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const unsigned wd = (unsigned)__shfl((int)own, 16 * tt + 8 * kg + e);
                const unsigned bit = (wd >> sh) & 1u;
                word |= (col0 ? bit : 0u) << (16 * tt + e);
                word |= (col1 ? bit : 0u) << (16 * tt + 8 + e);
            }
    } else if (dir == 1) {
        // forward: the k-step's 32 dense indices are 2 kernel points x the 16 rows of ONE group g (16 ks is a multiple of 32): the lane's
        // rows are 16 g + 8 kg + e for both k-blocks tt; two points per lane (the column halves j), one word each
        const int d0 = 32 * step, g = d0 / (16 * ks);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pt = 64 * wt + 32 * j + li;
            const unsigned wd = (pt < p && (g >> 1) < W) ? memb[((size_t)b * p + pt) * W + (g >> 1)] : 0u;
            const unsigned byte = (wd >> (16 * (g & 1) + 8 * kg)) & 0xffu;                      // rows 16 g + 8 kg .. + 7
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                if (d0 + 16 * tt < nkr) word |= byte << (16 * tt + 8 * j);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int tt = i >> 4, j = (i >> 3) & 1, e = i & 7;
            const int n = 64 * wt + 32 * j + (lane & 31);
            const int kk = 32 * step + 16 * tt + 8 * (lane >> 5) + e;
            const int pt = dir ? n : kk, kr = dir ? kk : n;
            if (pt < p && kr < nkr) {
                int k_, r;
                dense_kr(kr, ks, k_, r);
                word |= ((memb[((size_t)b * p + pt) * W + (r >> 5)] >> (r & 31)) & 1u) << i;
            }
        }
    }
    bits[(((size_t)b * wtiles + wt) * steps + step) * 64 + lane] = word;
}

// keys[b][p]: bit g = point p's list names a row of the 16-row group g (row slots 16 g .. 16 g + 15; rp <= 512: 32 groups).  Sorting a
// cloud's points by this key (any total order: equal keys become neighbours, and keys that share their high groups stay close) makes
// the 0/1 mask of the dense product BLOCK-sparse: the first-nsample-by-index ball query (grouping_cuda_kernel.cu:L68-113) gives
// every point 64 of the R referenced rows, i.e. rows of only 0.45-0.7 of the 16-row groups, and points with the same groups meet
// in the same 32-point k-steps (backward) / 256-point column blocks (forward) -- dense_steps_kernel lists the non-empty ones.
__global__ __launch_bounds__(256) void dense_keys_kernel(long long total, const unsigned *__restrict__ memb, int32_t *__restrict__ keys) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;            // (b, p)
    if (i >= total) return;
    const u32x4 *w = reinterpret_cast<const u32x4 *>(memb + i * MEMB_WORDS);
    unsigned key = 0;
#pragma unroll
    for (int q = 0; q < MEMB_WORDS / 4; ++q) {
        const u32x4 v = w[q];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = 2 * (4 * q + e);
            key |= ((v[e] & 0xffffu) ? 1u : 0u) << g;
            key |= ((v[e] >> 16) ? 1u : 0u) << (g + 1);
        }
    }
    keys[i] = (int32_t)key;
}

// steps[(b blocks_n + bn) (KS + 1)]: [0] = count >= 1, [1 ..] = the k-steps (ascending) of column block bn of cloud b in which at least
// one of the block's 4 x 64 lanes generates a weight that is not masked out -- every other k-step multiplies the stored operand by
// exact zeros and is skipped by the product kernel.  skip = 0: every k-step (dir 1 with n_rows: of the cloud's own prefix).  A block
// without any listed step gets step 0 (it still has to write its zeros).
__global__ __launch_bounds__(256) void dense_steps_kernel(int KS, int blocks_n, int wtiles, int dir, int ks, int row_slots, int skip,
                                                          const int32_t *__restrict__ n_rows, const unsigned *__restrict__ bits,
                                                          int32_t *__restrict__ steps) {
    extern __shared__ int nz[];
    const int b = blockIdx.y, bn = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < KS; i += 256) nz[i] = 0;
    __syncthreads();
    int limit = KS;
    if (dir == 1 && n_rows != nullptr) limit = max(min(KS, ((min(n_rows[b], row_slots) + 15) & ~15) * ks / KC_BK), 1);
    if (skip) {
        const int wt = min(4 * bn + wave, wtiles - 1);
        const unsigned *src = bits + ((size_t)b * wtiles + wt) * (size_t)KS * 64 + lane;
        for (int s = 0; s < limit; s += 4) {
            unsigned v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (s + u < limit) ? src[(size_t)(s + u) * 64] : 0u;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (__builtin_amdgcn_ballot_w64(v[u] != 0u) != 0ull && lane == 0 && s + u < limit) nz[s + u] = 1;      // (all writers store 1)
        }
    } else {
        for (int i = t; i < limit; i += 256) nz[i] = 1;
    }
    __syncthreads();
    if (t == 0) {
        int32_t *dst = steps + ((size_t)b * blocks_n + bn) * (size_t)(KS + 1);
        int n = 0;
        for (int i = 0; i < KS; ++i)
            if (nz[i]) dst[1 + n++] = i;
        if (n == 0) dst[1 + n++] = 0;
        dst[0] = n;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// the two float4 tables of the generated operand (evaluated in float64, rounded once)
// ------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dense_centre_kernel(int n, const float *__restrict__ xyz, float *__restrict__ centre) {
    __shared__ double s[3][256];
    const int b = blockIdx.x;
    double a[3] = {0, 0, 0};
    for (int i = threadIdx.x; i < n; i += 256)
        for (int d = 0; d < 3; ++d) a[d] += xyz[((size_t)b * 3 + d) * n + i];
    for (int d = 0; d < 3; ++d) s[d][threadIdx.x] = a[d];
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (threadIdx.x < h)
            for (int d = 0; d < 3; ++d) s[d][threadIdx.x] += s[d][threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x < 3) centre[b * 4 + threadIdx.x] = (float)(s[threadIdx.x][0] / n);
    if (threadIdx.x == 3) centre[b * 4 + 3] = 0.f;
}

// pt[b][i] for the query points: FORM 1 (x~, 0); FORM 0 (2 x~/sigma, -|x~|^2/sigma); entries i >= p: zeros
__global__ __launch_bounds__(256) void dense_points_kernel(int p, int p_pad, int form, float inv_sigma, const float *__restrict__ q_xyz,
                                                           const float *__restrict__ centre, f32x4 *__restrict__ pt) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p_pad) return;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (i < p) {
        const double is = (double)inv_sigma;
        const double x = (double)q_xyz[((size_t)b * 3 + 0) * p + i] - (double)centre[b * 4 + 0];
        const double y = (double)q_xyz[((size_t)b * 3 + 1) * p + i] - (double)centre[b * 4 + 1];
        const double z = (double)q_xyz[((size_t)b * 3 + 2) * p + i] - (double)centre[b * 4 + 2];
        if (form) v = (f32x4){(float)x, (float)y, (float)z, 0.f};
        else v = (f32x4){(float)(2.0 * is * x), (float)(2.0 * is * y), (float)(2.0 * is * z), (float)(-is * (x * x + y * y + z * z))};
    }
    pt[(size_t)b * p_pad + i] = v;
}

// kr[b][a][dense index of (k, r)], u = x~_row(r) - rk[a][k]: FORM 1 (u, 0); FORM 0 (u, 1 - |u|^2/sigma); entries past ks rp and empty
// slots: far away (FORM 1) / zeros (FORM 0) -- weight 0 either way, and their mask bits are 0.
// row_rot (may be null) float [b][rows_ld][9]: a rotation M per row slot, u = x~_row(r) - M rk[a][k] -- clouds whose points carry ONE
// pose rotation per rigid part: the reference rotates the offset by R_rel = R_p R_r^T (so3conv/functional.py:L1112-1160) and
// |R_rel (x_r - x_p) - A kappa| = |x_r - x_p - R_rel^T A kappa|, so for the query points of one part M = R_rel^T depends on the row only
__global__ __launch_bounds__(256) void dense_rows_kernel(int n_sup, int na, int ks, int rp, int kd_pad, int rows_ld, int form, float inv_sigma,
                                                         const float *__restrict__ s_xyz, const float *__restrict__ centre,
                                                         const int32_t *__restrict__ rows, const float *__restrict__ rk,
                                                         const float *__restrict__ row_rot, f32x4 *__restrict__ kr) {
    const int b = blockIdx.z, a = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kd_pad) return;
    f32x4 v = form ? (f32x4){1e4f, 1e4f, 1e4f, 0.f} : (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < ks * rp) {
        int k, r;
        dense_kr(i, ks, k, r);
        const int row = rows[(size_t)b * rows_ld + r];
        if ((unsigned)row < (unsigned)n_sup) {
            const double is = (double)inv_sigma;
            const float *kp = rk + ((size_t)a * ks + k) * 3;
            double k0 = (double)kp[0], k1 = (double)kp[1], k2 = (double)kp[2];
            if (row_rot != nullptr) {
                const float *M = row_rot + ((size_t)b * rows_ld + r) * 9;
                const double t0 = M[0] * k0 + M[1] * k1 + M[2] * k2, t1 = M[3] * k0 + M[4] * k1 + M[5] * k2, t2 = M[6] * k0 + M[7] * k1 + M[8] * k2;
                k0 = t0; k1 = t1; k2 = t2;
            }
            const double x = (double)s_xyz[((size_t)b * 3 + 0) * n_sup + row] - (double)centre[b * 4 + 0] - k0;
            const double y = (double)s_xyz[((size_t)b * 3 + 1) * n_sup + row] - (double)centre[b * 4 + 1] - k1;
            const double z = (double)s_xyz[((size_t)b * 3 + 2) * n_sup + row] - (double)centre[b * 4 + 2] - k2;
            v = (f32x4){(float)x, (float)y, (float)z, form ? 0.f : (float)(1.0 - is * (x * x + y * y + z * z))};
        }
    }
    kr[((size_t)b * na + a) * kd_pad + i] = v;
}

// ------------------------------------------------------------------------------------------------------------------------
// the stored operand: T[b][m][l][na] (dY [b,o,p,a] or G [b,o,(k,r),a]) -> per (cloud, anchor) the row scales and the two
// fp16 planes in fragment order
// ------------------------------------------------------------------------------------------------------------------------
// scale[(b na + a) m + row] = scale2[(b m + row) na + a] = 2^(14 - e) for max_l |T[b][row][l][a]| in [2^e, 2^(e+1)).  One block per (b, row): thread
// (anchor quad aq, lane group pg) walks the row's float4 pieces aq + nq (pg + G j): consecutive threads, consecutive pieces
// (a row's l elements may come in l / seg segments of seg elements, seg_pitch4 float4 apart: the rows of G as the small GEMM
// leaves them, [o][k][pitch >= rp na])
// colmap (may be null) int32 [b][l]: element i of a row is element colmap[b][i] of a source row of seg_pitch4 float4 x nq (one segment);
// negative: a zero
__global__ __launch_bounds__(256) void dense_rowmax_kernel(int m, int l, int na, int seg, long long seg_pitch4, const int32_t *__restrict__ colmap,
                                                           const f32x4 *__restrict__ T, float *__restrict__ scale, float *__restrict__ scale2) {
    __shared__ unsigned s[256][4];
    const int nq = na >> 2, G = 256 / nq, b = blockIdx.y, row = blockIdx.x, t = threadIdx.x;
    const int aq = t % nq, pg = t / nq;
    unsigned v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    if (pg < G) {
        const f32x4 *src = T + ((size_t)b * m + row) * (colmap ? (size_t)1 : (size_t)(l / seg)) * seg_pitch4;
        for (int i0 = pg; i0 < l; i0 += 8 * G) {                   // eight independent 16-byte loads in flight per thread
            f32x4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u * G, l - 1), sg = i / seg;      // (past the end: the last element again -- a maximum does not mind)
                if (colmap != nullptr) {
                    const int ci = colmap[(size_t)b * l + i];
                    q[u] = ci >= 0 ? src[(size_t)ci * nq + aq] : (f32x4){0.f, 0.f, 0.f, 0.f};
                } else {
                    q[u] = src[(size_t)sg * seg_pitch4 + (size_t)(i - sg * seg) * nq + aq];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v0 = max(v0, __float_as_uint(q[u].x) & 0x7fffffffu);
                v1 = max(v1, __float_as_uint(q[u].y) & 0x7fffffffu);
                v2 = max(v2, __float_as_uint(q[u].z) & 0x7fffffffu);
                v3 = max(v3, __float_as_uint(q[u].w) & 0x7fffffffu);
            }
        }
    }
    s[t][0] = v0; s[t][1] = v1; s[t][2] = v2; s[t][3] = v3;
    __syncthreads();
    if (t < na) {
        unsigned v = 0;
        for (int g = 0; g < G; ++g) v = max(v, s[g * nq + (t >> 2)][t & 3]);
        const float sc = pow2_scale(__uint_as_float(v));
        scale[((size_t)b * na + t) * m + row] = sc;            // [b][a][row]: the product's epilogue
        scale2[((size_t)b * m + row) * na + t] = sc;           // [b][row][a]: the split
    }
}

// the same two scale arrays from row maxima somebody else already has (rowmax [b][m][na], float bit patterns: the BatchNorm
// backward that produced the gradient leaves them, eap_bn_act_bwd_apply_rowmax_f32)
__global__ __launch_bounds__(256) void dense_scale_kernel(long long total, int m, int na, const unsigned *__restrict__ rowmax,
                                                          float *__restrict__ scale, float *__restrict__ scale2) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;          // (b, row, a)
    if (i >= total) return;
    const int a = (int)(i % na);
    const long long br = i / na;
    const int row = (int)(br % m);
    const long long b = br / m;
    const float sc = pow2_scale(__uint_as_float(rowmax[i]));
    scale[(b * na + a) * m + row] = sc;
    scale2[i] = sc;
}

// planes[z = b na + a][k-block kb][row tile mt][plane][lane] (16 bytes): lane (i = lane & 31, kg = lane >> 5) holds the 8
// contraction elements 16 kb + 8 kg .. + 7 of row 32 mt + i -- what a lane of v_mfma_f32_32x32x16_f16 takes as its A operand.
// One block per (b, mt, kb): thread (anchor quad aq = t % nq, row group t / nq) takes the 8 elements of (row, kg) for its four
// anchors -- 8 float4 loads 4 na bytes apart, the nq threads of a row covering each element's 4 na contiguous bytes -- and
// writes its 8 pieces; the 64 pieces of a 1 KB run come from one block within a few hundred cycles.
// mapped (the forward's G, seg = rp, l = ks rp): element l of a row is the pair (k, r) with DENSE INDEX l, found at segment k,
// position r; k-blocks past the cloud's own prefix (n_rows) are not written -- the product never reads them.
// colmap: as dense_rowmax_kernel's (the columns of dY that are one rigid part's query points)
// BN (template): the operand is the gradient BEHIND a training-mode BatchNorm + leaky_relu (csrc/bn_act.hip), formed here from the
// gradient in front of it (T = dL/dy') and the activated output Y2 = y' itself instead of being written out and read back:
//     pre = y' > 0 ? y' : y' / slope,  xhat = (pre - beta) / gamma,  g = y' > 0 ? dy' : slope dy',  gx = k1 g - k2 - k3 xhat
// per row (= channel) coefficients bn[0..4][m] = k1, k2, k3, beta, 1 / gamma (k1 = gamma invstd, k2 = k1 mean(g), k3 = k1 mean(g xhat)).
// The pre-activation is recovered from the OUTPUT (leaky_relu with a positive slope is invertible), so the layer keeps no copy of
// the conv output for its backward.  SPConvNets/utils/base_so3poseconv.py:L214-221.
struct SplitBn { const f32x4 *Y2; const float *coef; float slope, inv_slope; };
template <bool BN>
__global__ __launch_bounds__(256) void dense_split_kernel(int nb, int m, int l, int na, int kb_total, int seg, long long seg_pitch4, int mapped,
                                                          const int32_t *__restrict__ n_rows, const int32_t *__restrict__ colmap,
                                                          const f32x4 *__restrict__ T, const float *__restrict__ scale2,
                                                          u32x4 *__restrict__ planes, SplitBn bn) {
    // block -> (k-block kb fastest, row tile mt, cloud b).  (Measured and dropped: all k-blocks of one (mt, b) on one XCD, so that the
    // partial lines a column map's scattered 240-byte rows share meet in one L2 -- 3.16 -> 3.56 ms unmapped, 3.35 -> 3.72 mapped.)
    const int t = threadIdx.x;
    const int mts = m >> 5;
    const int kb = (int)(blockIdx.x % (unsigned)kb_total), grp_ = (int)(blockIdx.x / (unsigned)kb_total), mt = grp_ % mts, b = grp_ / mts;
    (void)nb;
    const int nq = na >> 2, RG = 256 / nq;                      // rows per pass
    const int aq = t % nq, rr = t / nq;
    if (rr >= RG) return;
    const int MT = m >> 5, nseg = l / seg;
    if (mapped && n_rows != nullptr && 16 * kb >= ((min(n_rows[b], seg) + 15) & ~15) * nseg) return;
    // item = (row i, k half kg); two items per round: sixteen independent 16-byte loads in flight per thread
    for (int it0 = rr; it0 < 64; it0 += 2 * RG) {
        f32x4 q[2][8];
        f32x4 q2[BN ? 2 : 1][BN ? 8 : 1];
        int rowv[2], lanef[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = min(it0 + u * RG, 63);
            const int i = it & 31, kg = it >> 5, row = 32 * mt + i, l0 = 16 * kb + 8 * kg;
            rowv[u] = row; lanef[u] = i + 32 * kg;
            const size_t row_off = ((size_t)b * m + row) * (colmap ? (size_t)1 : (size_t)nseg) * seg_pitch4 + aq;
            const f32x4 *src = T + row_off;
            int sg = l0 / seg, sr = l0 - sg * seg;                // segment and position of element l0 + e
            if (mapped) dense_kr(l0, nseg, sg, sr);               // (8 consecutive dense indices: one kernel point, 8 consecutive slots)
            if (colmap != nullptr) {
                // (l is a multiple of 8 here -- the launcher checks -- so the 8 map entries are two aligned 16-byte words)
                int ci[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
                if (l0 < l) {
                    const int4 c0 = *reinterpret_cast<const int4 *>(colmap + (size_t)b * l + l0), c1 = *reinterpret_cast<const int4 *>(colmap + (size_t)b * l + l0 + 4);
                    ci[0] = c0.x; ci[1] = c0.y; ci[2] = c0.z; ci[3] = c0.w; ci[4] = c1.x; ci[5] = c1.y; ci[6] = c1.z; ci[7] = c1.w;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) q[u][e] = ci[e] >= 0 ? src[(size_t)ci[e] * nq] : (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (BN) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) q2[u][e] = ci[e] >= 0 ? bn.Y2[row_off + (size_t)ci[e] * nq] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bool in = l0 + e < l;
                    const size_t at = (size_t)sg * seg_pitch4 + (size_t)sr * nq;
                    q[u][e] = in ? src[at] : (f32x4){0.f, 0.f, 0.f, 0.f};
                    if constexpr (BN) q2[u][e] = in ? bn.Y2[row_off + at] : (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (++sr == seg) { sr = 0; ++sg; }
                }
            }
        }
        if constexpr (BN) {
            // columns that do not exist (map < 0, past l) must stay exact zeros: their y' reads as 0 above, which alone would give -k2
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int it = min(it0 + u * RG, 63);
                const int l0 = 16 * kb + 8 * (it >> 5);
                const float k1 = bn.coef[rowv[u]], k2 = bn.coef[m + rowv[u]], k3 = bn.coef[2 * m + rowv[u]], be = bn.coef[3 * m + rowv[u]], ig = bn.coef[4 * m + rowv[u]];
                int live = 0;                                   // bit e: element e of this item exists
                if (colmap != nullptr) {
                    if (l0 < l) {
                        const int4 c0 = *reinterpret_cast<const int4 *>(colmap + (size_t)b * l + l0), c1 = *reinterpret_cast<const int4 *>(colmap + (size_t)b * l + l0 + 4);
                        live = (c0.x >= 0) | (c0.y >= 0) << 1 | (c0.z >= 0) << 2 | (c0.w >= 0) << 3 | (c1.x >= 0) << 4 | (c1.y >= 0) << 5 | (c1.z >= 0) << 6 | (c1.w >= 0) << 7;
                    }
                } else {
                    for (int e = 0; e < 8; ++e) live |= (l0 + e < l) << e;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float yv = q2[u][e][j], gv = q[u][e][j];
                        const bool pos = yv > 0.f;
                        const float gg = pos ? gv : gv * bn.slope, pre = pos ? yv : yv * bn.inv_slope;
                        const float gx = fmaf(gg, k1, -k2) - ((pre - be) * ig) * k3;
                        q[u][e][j] = ((live >> e) & 1) ? gx : 0.f;
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (it0 + u * RG >= 64) break;
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale2 + ((size_t)b * m + rowv[u]) * na + 4 * aq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned h[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2(q[u][2 * e][j] * sc[j], q[u][2 * e + 1][j] * sc[j], h[e], lo[e]);
                u32x4 *dst = planes + ((((size_t)b * na + 4 * aq + j) * kb_total + kb) * MT + mt) * 128 + lanef[u];
                dst[0] = (u32x4){h[0], h[1], h[2], h[3]};
                dst[64] = (u32x4){lo[0], lo[1], lo[2], lo[3]};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// the forward's stored operand made in ONE kernel (round 6): G[o,(k,r),a] = sum_c W[o,c,k] F[c,r,a] over the referenced rows, written
// straight as the product's two fp16 planes in fragment order.  (Before: a GEMM wrote G in fp32 -- 3.4 GB at 8 x 4096, O = 512 -- and
// dense_split_kernel read it back and wrote the planes: 2.3 + 2.3 ms for 0.2 TFLOP of work.)
//   block = (4 row tiles of 32 output channels, one per wave; z = (cloud, anchor)); the anchor's feature rows Ft[b][a][r][c] (c contiguous)
//   sit in LDS as two fp16 planes after one power-of-two scale per (cloud, anchor); a wave walks the kernel points k: its 32 rows (o, k) of
//   W3[(o,k)][c] go to registers as two fp16 planes (own power-of-two scale per row), then per tile of 32 rows r: D^T[r][o] = sum_c F[r][c]
//   W[o][c] (three v_mfma_f32_32x32x16_f16 per 16 channels: h h' + h l' + l h'), brought to the plane scale of the output row (from a BOUND
//   on max_{k,r} |G|: a bound that is too large costs dynamic range only, see dense_split_kernel<true>), split, and the 8-row pieces --
//   4 values from a lane, 4 from its partner 32 lanes away -- stored 1 KB per (k-block, plane) and wave.
//   Rows past the cloud's own prefix (ceil16(n_rows)) are neither computed nor written.
template <int CSTEPS>                                    // channels / 16
__global__ __launch_bounds__(256) void dense_gplanes_kernel(int m, int na, int ks, int rp, int rch, int kb_total, const float *__restrict__ W3,
                                                            const float *__restrict__ Ft, const int32_t *__restrict__ n_rows,
                                                            const float *__restrict__ bound, float *__restrict__ scale, u32x4 *__restrict__ planes) {
    constexpr int C = 16 * CSTEPS, PITCH = 2 * C + 16;                      // bytes per LDS row of one plane (+16: rows 272 / 144 bytes apart)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_red[4];
    const int z = blockIdx.y, b = z / na;
    const int t = threadIdx.x, lane = t & 63, li = lane & 31, kg = lane >> 5, wave = t >> 6;
    const int rows_b = n_rows ? min((min(n_rows[b], rp) + 15) & ~15, rp) : rp;          // this cloud's row slots (whole groups of 16)
    unsigned char *Fh = smem, *Fl = smem + (size_t)rch * PITCH;
    const int MT = m >> 5, mt = blockIdx.x * 4 + wave;
    const bool active = mt < MT;                                             // (wave-uniform; an idle wave still meets the barriers below)
    const int o = 32 * min(mt, MT - 1) + li;
    // the output row's plane scale, from the bound; also published for the product's epilogue (scale[z][o]: one lane per row)
    const float s_out = pow2_scale(bound[(size_t)z * m + o]);
    if (active && kg == 0) scale[(size_t)z * m + o] = s_out;
    // The anchor's rows pass through LDS in chunks of rch rows (a multiple of 32; all of them at once where they fit: the bench layers);
    // a chunk has its own power-of-two scale, which the output scale below takes off again.
    for (int rc0 = 0; rc0 < rows_b; rc0 += rch) {
        const int nrows = min(rch, rows_b - rc0);
        if (rc0 > 0) __syncthreads();                                        // everybody is done with the previous chunk's planes
        // ---- the chunk's feature rows -> LDS planes ----
        const f32x4 *src = reinterpret_cast<const f32x4 *>(Ft + ((size_t)z * rp + rc0) * C);
        const int n4 = nrows * (C / 4);
        unsigned mx = 0;
        for (int i = t; i < n4; i += 256) {
            const f32x4 v = src[i];
            mx = max(mx, max(max(__float_as_uint(v.x) & 0x7fffffffu, __float_as_uint(v.y) & 0x7fffffffu),
                             max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
        }
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o_));
        if (lane == 0) s_red[wave] = mx;
        __syncthreads();
        const float sF = pow2_scale(__uint_as_float(max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]))));
        for (int i = t; i < n4; i += 256) {
            const f32x4 v = src[i];                                          // (second read: L2)
            const int r = i / (C / 4), c4 = i - r * (C / 4);
            unsigned h0, l0, h1, l1;
            split2(v.x * sF, v.y * sF, h0, l0);
            split2(v.z * sF, v.w * sF, h1, l1);
            *reinterpret_cast<uint2 *>(Fh + (size_t)r * PITCH + c4 * 8) = make_uint2(h0, h1);
            *reinterpret_cast<uint2 *>(Fl + (size_t)r * PITCH + c4 * 8) = make_uint2(l0, l1);
        }
        __syncthreads();
        if (!active) continue;
        const int rtiles = (nrows + 31) >> 5;
        // this lane's half of row (o, k) of W3: channels 16 s + 8 kg .. + 7 of every 16-channel step; the next kernel point's row is requested
        // before this one's tiles are computed (one wave per SIMD and block: nobody else hides the L2 round trip)
        f32x4 wn[CSTEPS][2];
        auto load_w = [&](int k) __attribute__((always_inline)) {
            const f32x4 *wrow = reinterpret_cast<const f32x4 *>(W3 + ((size_t)o * ks + k) * C) + 2 * kg;
#pragma unroll
            for (int s_ = 0; s_ < CSTEPS; ++s_) { wn[s_][0] = wrow[4 * s_]; wn[s_][1] = wrow[4 * s_ + 1]; }
        };
        load_w(0);
        for (int k = 0; k < ks; ++k) {
            f32x4 w[CSTEPS][2];
            unsigned wm = 0;
#pragma unroll
            for (int s_ = 0; s_ < CSTEPS; ++s_) {
                w[s_][0] = wn[s_][0]; w[s_][1] = wn[s_][1];
#pragma unroll
                for (int e = 0; e < 4; ++e) wm = max(wm, max(__float_as_uint(w[s_][0][e]) & 0x7fffffffu, __float_as_uint(w[s_][1][e]) & 0x7fffffffu));
            }
            if (k + 1 < ks) load_w(k + 1);
            wm = max(wm, (unsigned)__shfl_xor((int)wm, 32));
            const float sW = pow2_scale(__uint_as_float(wm));
            u32x4 Bh[CSTEPS], Bl[CSTEPS];
#pragma unroll
            for (int s_ = 0; s_ < CSTEPS; ++s_) {
                unsigned h[4], l[4];
                split2(w[s_][0].x * sW, w[s_][0].y * sW, h[0], l[0]);
                split2(w[s_][0].z * sW, w[s_][0].w * sW, h[1], l[1]);
                split2(w[s_][1].x * sW, w[s_][1].y * sW, h[2], l[2]);
                split2(w[s_][1].z * sW, w[s_][1].w * sW, h[3], l[3]);
                Bh[s_] = (u32x4){h[0], h[1], h[2], h[3]};
                Bl[s_] = (u32x4){l[0], l[1], l[2], l[3]};
            }
            const float tsc = s_out / (sF * sW);                            // (powers of two: exact)
            for (int rt = 0; rt < rtiles; ++rt) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                const int r = min(32 * rt + li, rch - 1);                    // (rows past the chunk: a valid LDS address, their pieces are not stored)
                const unsigned char *ah = Fh + (size_t)r * PITCH + 16 * kg, *al = Fl + (size_t)r * PITCH + 16 * kg;
#pragma unroll
                for (int s_ = 0; s_ < CSTEPS; ++s_) {
                    const u32x4 fh = *reinterpret_cast<const u32x4 *>(ah + 32 * s_), fl = *reinterpret_cast<const u32x4 *>(al + 32 * s_);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fh), __builtin_bit_cast(f16x8, Bh[s_]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fh), __builtin_bit_cast(f16x8, Bl[s_]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fl), __builtin_bit_cast(f16x8, Bh[s_]), acc, 0, 0, 0);
                }
                // D^T: column = o (this lane's), rows r = (i & 3) + 8 (i >> 2) + 4 kg.  kg 0 assembles the pieces of rows 0-7 and 16-23, kg 1
                // those of rows 8-15 and 24-31: four values of each piece are the partner's
                // (every select below picks between two NAMED values: written as acc[kg ? i : j] hipcc turns the choice into a dynamic index
                // into the accumulator vector -- sixteen compare + select pairs per element, ~600 idle cycles of hazard padding per tile)
                float av[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { av[i] = acc[i]; asm volatile("" : "+v"(av[i])); }
                const bool hi = kg != 0;
                float snd[8], rcv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { snd[e] = hi ? av[e] : av[4 + e]; snd[4 + e] = hi ? av[8 + e] : av[12 + e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) rcv[e] = __shfl_xor(snd[e], 32);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float own = (hi ? av[4 + 8 * half + e] : av[8 * half + e]) * tsc, got = rcv[4 * half + e] * tsc;
                        v[e] = hi ? got : own;
                        v[4 + e] = hi ? own : got;
                    }
                    const int r0 = rc0 + 32 * rt + 16 * half + 8 * kg;       // first row slot of this piece
                    if (r0 < rc0 + nrows) {
                        unsigned h[4], l[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) split2(v[2 * e], v[2 * e + 1], h[e], l[e]);
                        const int kb = (r0 >> 4) * ks + k;                   // k-block of dense indices 16 kb .. + 15 = (k, rows 16 g .. + 15)
                        u32x4 *dst = planes + (((size_t)z * kb_total + kb) * MT + mt) * 128 + li + 32 * kg;
                        dst[0] = (u32x4){h[0], h[1], h[2], h[3]};
                        dst[64] = (u32x4){l[0], l[1], l[2], l[3]};
                    }
                }
            }
        }
    }
}

// Yt[b][a][o][p] -> Y[b][o][p][a]: block (p chunk of 64, o, b) through a [na][65] LDS tile.  psum / psq (may be null): the block's
// sum and sum of squares of (y - pivot), pivot = Y[0][o][0][0] -- the partial moments the BatchNorm that follows would otherwise
// read the whole tensor for (csrc/bn_act.hip bn_stats_kernel: the same pivot, summed in float64 by the caller), at
// [o][b * chunks + chunk].
// map (may be null) int32 [b][p]: column pp of cloud b is point map[b][pp] of a Y with p_dst points (< 0: the column is padding, not
// written) -- the query points of one rigid part of a posed cloud, computed as a launch of their own
// pivot_pos (may be null; with map): the column of cloud 0 that is point 0 -- the pivot is Y[0][o][0][0] whatever the column order
__global__ __launch_bounds__(256) void dense_untranspose_kernel(int nb, int o_total, int p, int na, int p_dst, const int32_t *__restrict__ map,
                                                                const int32_t *__restrict__ pivot_pos,
                                                                const float *__restrict__ yt, float *__restrict__ y,
                                                                float *__restrict__ psum, float *__restrict__ psq,
                                                                const float *__restrict__ bn_scale = nullptr, const float *__restrict__ bn_shift = nullptr,
                                                                float slope = 1.f) {
    extern __shared__ float tile[];
    __shared__ float red[2][256];
    // block -> (chunk of 64 columns fastest, o, b).  (An XCD-owning map -- all chunks of one (o, b) in one L2 -- did not help the mapped
    // pass: 3.45 -> 3.51 ms with dword stores, 2.81 / 2.81 against 2.86 / 2.75 ms with the 16-byte stores that did: 3.51 -> 2.93, against
    // 2.1-2.2 for the unmapped pass.)
    const int chunks = (p + 63) >> 6, t = threadIdx.x;
    const int p0 = (int)(blockIdx.x % (unsigned)chunks) * 64, grp_ = (int)(blockIdx.x / (unsigned)chunks), o = grp_ % o_total, b = grp_ / o_total;
    const int nchunk_x = chunks, nb_z = nb;
    const int np = min(64, p - p0);
    const float pivot = psum ? yt[(size_t)o * p + (pivot_pos ? *pivot_pos : 0)] : 0.f;        // Yt[0][0][o][column of point 0]
    float s = 0.f, q = 0.f;
    for (int i = t; i < na * 64; i += 256) {
        const int a = i >> 6, pp = i & 63;
        if (pp < np) {
            const float v = yt[(((size_t)b * na + a) * o_total + o) * p + p0 + pp];
            tile[a * 65 + pp] = v;
            if (psum != nullptr && (map == nullptr || map[(size_t)b * p + p0 + pp] >= 0)) {      // (padding columns of a mapped launch: not in the moments)
                const float d = v - pivot;
                s += d; q = fmaf(d, d, q);
            }
        }
    }
    red[0][t] = s; red[1][t] = q;
    __syncthreads();
    // 16-byte stores: a row of Y is na floats = na / 4 pieces (na % 4 == 0); mapped rows are scattered, 240 bytes each
    const int nq = na >> 2;
    float *rows_y = y + ((size_t)b * o_total + o) * (size_t)(map ? p_dst : p) * na;
    for (int i = t; i < np * nq; i += 256) {
        const int pp = i / nq, a4 = i - pp * nq;
        const int q = map ? map[(size_t)b * p + p0 + pp] : p0 + pp;
        if (map != nullptr && (unsigned)q >= (unsigned)p_dst) continue;
        const float *src = tile + (4 * a4) * 65 + pp;
        f32x4 v = {src[0], src[65], src[130], src[195]};
        if (bn_scale != nullptr) {                                         // (block-uniform) BatchNorm + leaky_relu of channel o on the way out
            const float sc = bn_scale[o], sh = bn_shift[o];
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float u = fmaf(v[e], sc, sh); v[e] = u > 0.f ? u : u * slope; }
        }
        *reinterpret_cast<f32x4 *>(rows_y + (size_t)q * na + 4 * a4) = v;
    }
    if (psum == nullptr) return;                                           // (block-uniform)
    for (int h = 128; h > 0; h >>= 1) {                                    // (fixed order: bit-reproducible)
        if (t < h) { red[0][t] += red[0][t + h]; red[1][t] += red[1][t + h]; }
        __syncthreads();
    }
    if (t == 0) {
        const size_t at = ((size_t)o * nb_z + b) * nchunk_x + (p0 >> 6);
        psum[at] = red[0][0]; psq[at] = red[1][0];
    }
}

// Z[b][o][k][a][r] = 0 for the row slots r >= ceil16(n_rows[b]) the trimmed backward product does not write (the GEMMs that follow
// contract over them against zero feature rows: 0 x garbage must not be NaN)
__global__ __launch_bounds__(256) void dense_zero_tail_kernel(int rows_ok, int na, int rp, long long ldz, const int32_t *__restrict__ n_rows,
                                                              float *__restrict__ z) {
    const int b = blockIdx.y, row = blockIdx.x;                           // one block per (o, k) row of a cloud
    const int r0 = (min(n_rows[b], rp) + 15) & ~15, tail = rp - r0;
    if (tail <= 0) return;
    float *dst = z + ((long long)b * rows_ok + row) * ldz;
    for (int e = threadIdx.x; e < na * tail; e += 256) {
        const int a = e / tail;
        dst[(long long)a * rp + r0 + (e - a * tail)] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// the product
// ------------------------------------------------------------------------------------------------------------------------
struct KcArgs {
    int MT, N, KS, na, zcount;            // row tiles of the stored operand, columns, k-steps, anchors per cloud, clouds x anchors
    int tiles_m, blocks_n, mask_tiles;    // row blocks (MT / MI), column blocks (of 4 wave tiles), wave tiles in the mask table
    const u32x4 *A;                       // planes [z][2 KS][MT][2][64]
    const float *scale;                   // [z][32 MT]
    const f32x4 *strT; long long strB, strA;     // k-side table: element kk of (cloud, anchor) at strT[b strB + a strA + kk]
    const f32x4 *colT; long long colB, colA;     // column-side table
    const unsigned *mask;                 // [b][mask_tiles wave tiles][KS][64 lanes] mask bits
    float neg_inv_sigma;                  // FORM 1
    const int32_t *n_rows; int ks, trim;  // referenced rows per cloud; trim 1: the columns are dense indices cut at the cloud's prefix (backward), 3: not cut,
                                          // 2: the k axis is cut at the cloud's prefix (forward), 0: not cut
    float *C; long long cB, cA, ldm; int rp; long long kstride;     // element (row, n) of (b, a) at C[b cB + a cA + row ldm + (n / rp) kstride + n % rp]
    int row_slots;                        // row slots of the dense index range (the launcher's rp; `rp` above is the output's column split)
    const int32_t *steps;                 // (may be null) [b][blocks_n][KS + 1]: count, then the k-steps this column block runs (dense_steps_kernel)
};

// DBG (timing ablations, `make ABLATION=1` + EAP_DENSE_DEBUG, WRONG results): 1 = no mask (all lanes kept), 2 = no k-side table
// reads (constants), 4 = no weight evaluation at all (the B fragments of the prologue for every k-block), 8 = no DMA inside the k-loop,
// 16 = no fragment reads of the stored operand inside the k-loop
template <int MI, int FORM, int DBG = 0>
__global__ __launch_bounds__(256, MI == 4 ? 2 : 1) void kc_gemm_kernel(KcArgs g) {
    static_assert(MI == 8 || MI == 4, "the DMA piece schedule below: 8 or 4 KB of the stored operand per wave and k-step");
    constexpr unsigned WB = MI * 1024u;                    // bytes of the stored operand a wave moves per k-step
    constexpr unsigned SUB_A = MI * 2048u;                 // bytes of the stored operand per k-block (16 k) of a block
    constexpr unsigned STR_OFF = 2 * SUB_A;                // the k-side table's slice: 2 x 256 bytes (+ 2 x 256 of duplicate landing space)
    constexpr unsigned BIT_OFF = STR_OFF + 1024u;          // the mask bits of the step: one dword per lane and wave
    constexpr unsigned STAGE = BIT_OFF + 1024u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- block -> (z, row block, column block): an XCD owns whole (cloud, anchor) pairs, so the stored operand of a pair is
    //      fetched into ONE L2 and shared there by the pair's tiles_m x blocks_n workgroups, which run at the same time ----
    const int per_z = g.tiles_m * g.blocks_n;
    int id = blockIdx.x, z, local;
    {
        const int groups = g.zcount / 8 * 8;
        if (id < groups * per_z) {
            const int xcd = id & 7, slot = id >> 3;
            z = (slot / per_z) * 8 + xcd;
            local = slot % per_z;
        } else {
            const int r = id - groups * per_z;
            z = groups + r / per_z;
            local = r % per_z;
        }
    }
    const int bm = local % g.tiles_m, bn = local / g.tiles_m;
    const int b = z / g.na, a = z - b * g.na;
    // this cloud's prefix of the dense index range (see dense_kr): whole 16-row groups of its own referenced rows
    const int used = (g.trim == 1 || g.trim == 2) ? ((min(g.n_rows[b], g.row_slots) + 15) & ~15) * g.ks : 0;      // (trim 3 / 0: dense indices, every slot)
    const int N = g.trim == 1 ? min(g.N, used) : g.N;
    if (256 * bn >= N) return;                             // (block-uniform, before any barrier)
    // the k-steps this column block runs: the listed ones (dense_steps_kernel: those with a weight that is not masked out), else all of
    // them / the cloud's own prefix.  The list is copied into LDS behind the ring and read one step ahead of its use with a plain
    // ds_read + v_readfirstlane: LDS reads return in order, so the counted waits of the products stay counted (a scalar load in flight
    // turns every lgkmcnt wait into "everything", the first version's 2 ms)
    const int32_t *lstG = g.steps ? g.steps + ((size_t)b * g.blocks_n + bn) * (size_t)(g.KS + 1) : nullptr;
    const int KS = lstG ? __builtin_amdgcn_readfirstlane(lstG[0]) : (g.trim == 2 ? max(min(g.KS, used / KC_BK), 1) : g.KS);
    int *lstL = reinterpret_cast<int *>(smem + 4 * STAGE);
    if (lstG != nullptr) {
        for (int i = threadIdx.x; i < KS; i += 256) lstL[i] = lstG[1 + i];
        __syncthreads();
    }
    auto step_at = [&](int i) __attribute__((always_inline)) { return lstG ? __builtin_amdgcn_readfirstlane(lstL[i]) : i; };

    const int t = threadIdx.x, lane = t & 63, li = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wt = bn * 4 + wave;                          // this wave's 64 columns
    const bool active = 64 * wt < N;                       // (wave-uniform)

    // ---- per-lane column constants ----
    const f32x4 *colz = g.colT + b * g.colB + a * g.colA;
    f32x4 cc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) cc[j] = colz[min(64 * wt + 32 * j + li, N - 1)];
    const float nis = g.neg_inv_sigma;

    // ---- DMA.  Per k-step a wave moves 10 pieces: its 8 KB of the block's 32 KB of the stored operand (contiguous in memory AND in
    //      the stage: the immediate offset of global_load_lds goes into both addresses, tools/microbench/glds_offset.hip, so four
    //      pieces share one M0 and one base; MI = 4: 4 KB), the k-side table's slice (2 x 16 float4; waves 0, 1 land k-block `wave`, waves 2, 3 a
    //      duplicate nobody reads: uniform piece counts, one s_waitcnt immediate for all waves) and its own 256 bytes of mask bits ----
    const unsigned char *Aw = reinterpret_cast<const unsigned char *>(g.A) + ((size_t)z * 2 * g.KS * g.MT + (size_t)bm * MI) * 2048u
                              + (size_t)(wave >> 1) * ((size_t)g.MT * 2048u) + (size_t)(wave & 1) * WB;         // k-block wave / 2, second half of its runs for odd waves
    const size_t step_bytes = (size_t)g.MT * 4096u;                                                              // two k-blocks
    const unsigned char *strw = reinterpret_cast<const unsigned char *>(g.strT + b * g.strB + a * g.strA) + (size_t)(wave & 1) * 256u;
    const unsigned char *bitw = reinterpret_cast<const unsigned char *>(g.mask) + ((size_t)b * g.mask_tiles + min(wt, g.mask_tiles - 1)) * (size_t)g.KS * 256u;
    const unsigned lds0 = lds_addr(smem);
    const unsigned ldsA = lds0 + (unsigned)(wave >> 1) * SUB_A + (unsigned)(wave & 1) * WB;
    const unsigned voff16 = (unsigned)lane * 16u, voff4 = (unsigned)lane * 4u;
    // (M0 is the compiler's scratch register, nothing of ours lives in it across statements: set in the statement that uses it)
#define DMA2(src, dst, o0, o1) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #o0 "\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #o1 \
                                            :: "v"(voff16), "s"(src), "s"(dst) : "memory")
#define DMA1(src, dst, o0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:" #o0 :: "v"(voff16), "s"(src), "s"(dst) : "memory")
#define DMA4B(src, dst) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" :: "v"(voff4), "s"(src), "s"(dst) : "memory")
    // the three sources of a k-step
    struct StepSrc { const unsigned char *a, *str, *bit; };
    auto src_of = [&](int step) __attribute__((always_inline)) {
        return StepSrc{Aw + (size_t)step * step_bytes, strw + (size_t)step * 512u, bitw + (size_t)step * 256u};
    };
    auto dma_slot = [&](const StepSrc &ss, unsigned stage_off, int slot) __attribute__((always_inline)) {
        const unsigned char *src = ss.a;
        const unsigned dst = ldsA + stage_off;
        if constexpr (MI == 8) {
            if (slot == 0) DMA2(src, dst, 0, 1024);
            if (slot == 1) DMA2(src, dst, 2048, 3072);
            if (slot == 2) DMA1(src + 4096, dst + 4096u, 0);
            if (slot == 3) DMA2(src + 4096, dst + 4096u, 1024, 2048);
            if (slot == 4) DMA1(src + 4096, dst + 4096u, 3072);
        } else {
            if (slot == 0) DMA2(src, dst, 0, 1024);
            if (slot == 2) DMA1(src, dst, 2048);
            if (slot == 3) DMA1(src, dst, 3072);
        }
        if (slot == 5) {
            DMA4B(ss.str, lds0 + stage_off + STR_OFF + (unsigned)wave * 256u);
            DMA4B(ss.bit, lds0 + stage_off + BIT_OFF + (unsigned)wave * 256u);
        }
    };
    auto issue = [&](int step, unsigned stage_off) __attribute__((always_inline)) {
        const StepSrc ss = src_of(step);
#pragma unroll
        for (int slot = 0; slot < 6; ++slot) dma_slot(ss, stage_off, slot);
    };

    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LEAN (the 128-row variant): 256 registers, so that TWO workgroups share a CU (72 KB of LDS each) and one wave's matrix
    // instructions run in the shadow of the other's reads and barriers -- the k-side entries without their unused fourth component and
    // for ONE k-block at a time (read at the end of the k-block that consumed the previous ones)
    constexpr bool LEAN = MI == 4;
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    using SV = std::conditional_t<(LEAN && FORM == 1), f32x3, f32x4>;
    // B operand of one k-block: [tile j][plane] 8 halves per lane
    struct BFrag { u32x4 h[2], l[2]; };
    auto frag_a = [&](const unsigned char *st, int sub, int plane, u32x4 (&f)[MI]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MI; ++i) f[i] = *reinterpret_cast<const u32x4 *>(st + (unsigned)sub * SUB_A + (unsigned)(2 * i + plane) * 1024u + (unsigned)lane * 16u);
    };
    auto frag_loop = [&](const unsigned char *st, int sub, int plane, u32x4 (&f)[MI]) __attribute__((always_inline)) {
        if constexpr (DBG & 16) {
#pragma unroll
            for (int i = 0; i < MI; ++i) f[i] = (u32x4){0x3c003c00u + (unsigned)sub, 0x3c003c00u, 0x3c003c00u + (unsigned)plane, 0x3c003c00u};
        } else frag_a(st, sub, plane, f);
    };
    // weights of tile j, elements e0, e0 + 1 of a k-block (k-side entries sv[e]; mask bits 16 tb + 8 j + e of mb) -> word e0 / 2 of h, l.
    // The mask is the fma's addend: 1.0 for a list member, 0.0 otherwise -- the clamp then returns 0 (the product term is <= 0).
    auto gen2 = [&](const SV (&sv)[8], unsigned mb, int tb, int j, int e0, BFrag &f, bool prologue = false) __attribute__((always_inline)) {
        if ((DBG & 4) && !prologue) return;
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const SV s = sv[e0 + e];
            const int keep = (DBG & 1) ? -1 : ((int)(mb << (31 - (16 * tb + 8 * j + e0 + e))) >> 31);
            const float one = __int_as_float(keep & 0x3f800000);
            if constexpr (FORM == 1) {
                const float tx = cc[j].x - s.x, ty = cc[j].y - s.y, tz = cc[j].z - s.z;
                const float d2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
                v[e] = __builtin_amdgcn_fmed3f(fmaf(d2, nis, one), 0.f, 1.f);                    // (v_fma_f32 ... clamp)
            } else {
                const float y = fmaf(cc[j].y, s.y, fmaf(cc[j].x, s.x, (cc[j].w - 1.0f) + s.w));
                v[e] = __builtin_amdgcn_fmed3f(fmaf(cc[j].z, s.z, y) + one, 0.f, 1.f);
            }
        }
        unsigned h, l;
        split2(v[0], v[1], h, l);
        f.h[j][e0 / 2] = h;
        f.l[j][e0 / 2] = l;
    };
    auto load_sv = [&](const f32x4 *str, SV (&sv)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if constexpr (DBG & 2) sv[e] = SV{0.01f * (float)e, 0.02f, 0.03f * (float)kg};
            else sv[e] = *reinterpret_cast<const SV *>(str + 8 * kg + e);
        }
    };
    auto mm = [&](const u32x4 &fa, const u32x4 &fb, f32x16 &c) __attribute__((always_inline)) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa), __builtin_bit_cast(f16x8, fb), c, 0, 0, 0);
    };
#define SB() __builtin_amdgcn_sched_barrier(0)
    // issue order inside a product: one matrix instruction, then NV vector instructions in its shadow (hipcc on its own puts the
    // weight evaluation in front of the sixteen matrix instructions; one wave per SIMD has nobody else to fill the pipe)
#define PIPE_HALF(NV, ND)                                                         \
    _Pragma("unroll") for (int i_ = 0; i_ < MI; ++i_) {                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                        \
        if constexpr ((ND) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (ND) > 0 ? (ND) : 1, 0);  \
        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);                       \
    }
#define PIPE(NV, ND0, ND1) PIPE_HALF(NV, ND0) PIPE_HALF(NV, ND1)
    // One k-block u: acc += A(u) x Bc.  In its shadow: the weights of k-block u + 1 are generated into Bn (k-side entries `sv`, read
    // during the previous k-block; mask bits of `mb`) in eight portions spread over the three products; the second plane of A(u) and
    // the k-side entries of k-block u + 2 (`nsv`) are read at the head of the first product, the first plane of A(u + 1) between
    // the second and the third -- no LDS read is waited for less than a product after its issue.
    auto kblock = [&](const unsigned char *st, int sub, const unsigned char *nst, int nsub, const f32x4 *nnstr, unsigned mb, int tb,
                      const BFrag &Bc, BFrag &Bn, u32x4 (&ah)[MI], const SV (&sv)[8], SV (&nsv)[8], auto dma) __attribute__((always_inline)) {
        u32x4 al[MI];
        // vector instructions per matrix instruction of the three products (MI = 4: half the matrix work per generated weight --
        // the kernel is bound by the weight evaluation there) and LDS reads behind the first / second MI matrix instructions
        constexpr int NV1 = (MI == 8 ? 3 : 6) + (FORM ? 1 : 0), NV3 = (MI == 8 ? 2 : 4) + (FORM ? 1 : 0);
        SB();
        frag_loop(st, sub, 1, al);
        if constexpr (!LEAN) load_sv(nnstr, nsv);
        dma(0);
#pragma unroll
        for (int i = 0; i < MI; ++i) { mm(ah[i], Bc.h[0], acc[i][0]); mm(ah[i], Bc.h[1], acc[i][1]); }
        gen2(sv, mb, tb, 0, 0, Bn); gen2(sv, mb, tb, 0, 2, Bn); gen2(sv, mb, tb, 0, 4, Bn);
        if constexpr (LEAN) { PIPE(NV1, 1, 0) } else { PIPE(NV1, (MI == 8 ? 1 : 2), 1) }     // the MI (+ 8) LDS reads behind the matrix instructions
        SB();
        dma(1);
#pragma unroll
        for (int i = 0; i < MI; ++i) { mm(ah[i], Bc.l[0], acc[i][0]); mm(ah[i], Bc.l[1], acc[i][1]); }
        gen2(sv, mb, tb, 0, 6, Bn); gen2(sv, mb, tb, 1, 0, Bn); gen2(sv, mb, tb, 1, 2, Bn);
        PIPE(NV1, 0, 0)
        SB();
        frag_loop(nst, nsub, 0, ah);
        dma(2);
#pragma unroll
        for (int i = 0; i < MI; ++i) { mm(al[i], Bc.h[0], acc[i][0]); mm(al[i], Bc.h[1], acc[i][1]); }
        gen2(sv, mb, tb, 1, 4, Bn); gen2(sv, mb, tb, 1, 6, Bn);
        PIPE(NV3, 1, 0)                      // the next k-block's MI fragment reads behind the first MI matrix instructions
        SB();
        if constexpr (LEAN) { load_sv(nnstr, nsv); SB(); }      // (nsv IS sv here: after its last use)
    };

    // ---- prologue: stages 0, 1, 2 <- k-steps 0, 1, 2 (past the last step: the last step again, into a stage nobody reads) ----
    const int last = KS - 1;
    issue(step_at(0), 0);
    issue(step_at(min(1, last)), STAGE);
    issue(step_at(min(2, last)), 2 * STAGE);
    int ahead = lstG ? lstL[min(3, last)] : 0;              // (a VGPR, all lanes equal) the k-step whose pieces the first pass of the loop issues
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // (a wave whose 64 columns lie past N runs the same instruction stream on clamped columns and an all-zero mask: a branch
    // around the k-blocks would put the 256 accumulators across a control-flow join)
    BFrag B0, B1;
    u32x4 ah[MI];
    SV sv[8], sw[8];              // (LEAN: sw unused)
    unsigned mbc = *reinterpret_cast<const unsigned *>(smem + BIT_OFF + (unsigned)t * 4u);        // mask bits of step 0
    {
        load_sv(reinterpret_cast<const f32x4 *>(smem + STR_OFF), sv);
#pragma unroll
        for (int e = 0; e < 8; e += 2) { gen2(sv, mbc, 0, 0, e, B0, true); gen2(sv, mbc, 0, 1, e, B0, true); }
        if constexpr (DBG & 4) B1 = B0;
        frag_a(smem, 0, 0, ah);
        load_sv(reinterpret_cast<const f32x4 *>(smem + STR_OFF) + 16, sv);                       // k-block 1
    }

    // ---- k-loop, four stages.  The barrier that opens step s certifies stage (s + 1) % 4 -- its pieces were issued during step
    //      s - 2 and the counted wait in front of the barrier leaves only step s - 1's 10 (MI = 4: 6) pieces (k-step s + 2) in flight -- and frees
    //      stage (s + 3) % 4 (read during step s - 1) for the pieces of k-step s + 3, which are issued one or two at a time at the
    //      head of the six products of the step ----
    unsigned s0 = 0, s1 = STAGE, s2 = 2 * STAGE, s3 = 3 * STAGE;
    for (int s = 0; s < KS; ++s) {
        if (s > 0) {
            if constexpr (MI == 8) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        // (the list entry was read a whole step ago)
        const int nxt = lstG ? __builtin_amdgcn_readfirstlane(ahead) : min(s + 3, last);
        const StepSrc ssn = src_of(nxt);
        if (lstG != nullptr) ahead = lstL[min(s + 4, last)];
        const unsigned dst = s3;
        const unsigned mbn = *reinterpret_cast<const unsigned *>(smem + s1 + BIT_OFF + (unsigned)t * 4u);      // mask bits of step s + 1
        const f32x4 *str1 = reinterpret_cast<const f32x4 *>(smem + s1 + STR_OFF);
        // k-block 2s: generates k-block 2s + 1 (sv = its k-side entries, mask bits 16..31 of this step's dword); reads the entries of
        // k-block 2s + 2 (first slice of the next stage) for the next call
        kblock(smem + s0, 0, smem + s0, 1, str1, mbc, 1, B0, B1, ah, sv, LEAN ? sv : sw, [&](int slot) __attribute__((always_inline)) {
            if constexpr (!(DBG & 8)) dma_slot(ssn, dst, slot);
        });
        // k-block 2s + 1: generates k-block 2s + 2 (mask bits 0..15 of the NEXT step's dword); reads the entries of k-block 2s + 3
        // (past the end: stale data nobody uses)
        kblock(smem + s0, 1, smem + s1, 0, str1 + 16, mbn, 0, B1, B0, ah, LEAN ? sv : sw, sv, [&](int slot) __attribute__((always_inline)) {
            if constexpr (!(DBG & 8)) dma_slot(ssn, dst, 3 + slot);
        });
        mbc = mbn;
        { const unsigned r = s0; s0 = s1; s1 = s2; s2 = s3; s3 = r; }
    }
#undef SB
#undef PIPE
#undef PIPE_HALF
#undef DMA1
#undef DMA2
#undef DMA4B

    // ---- epilogue: the row scales come off, D[i][j] of a 32 x 32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 kg ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the look-ahead pieces of the last steps land in the ring: before it is reused
    __syncthreads();
    float *isc = reinterpret_cast<float *>(smem);
    if (t < 32 * MI) isc[t] = 1.0f / g.scale[(size_t)z * (32 * g.MT) + 32 * MI * bm + t];
    __syncthreads();
    if (!active) return;
    float *Cz = g.C + b * g.cB + a * g.cA + (long long)(32 * MI * bm) * g.ldm;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = 64 * wt + 32 * j + li;
        if (n >= N) continue;
        int k = n / g.rp, r = n - k * g.rp;                  // (untrimmed: plain columns, rp = N)
        if (g.trim & 1) dense_kr(n, g.ks, k, r);
        const long long coff = (long long)k * g.kstride + r;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kg;
                Cz[(long long)row * g.ldm + coff] = acc[i][j][r] * isc[row];
            }
    }
}

template <int MI, int FORM, int DBG = 0>
int kc_launch(const KcArgs &g, hipStream_t s) {
    constexpr size_t shmem = 4 * (2 * (size_t)MI * 2048u + 2048u) + KC_LIST_BYTES;      // the ring + the k-step list
    // per DEVICE (a process may drive several GPUs: the attribute belongs to the function object of the current device) and
    // race-free: a bit per device id, set after the call succeeded -- two threads may both make the (idempotent) call
    static std::atomic<unsigned long long> set_on{0};
    int dev = 0;
    if (int e = eap::hip_fail(hipGetDevice(&dev), "so3_dense: hipGetDevice")) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(set_on.load(std::memory_order_acquire) & bit)) {
        if (int e = eap::hip_fail(hipFuncSetAttribute(reinterpret_cast<const void *>(kc_gemm_kernel<MI, FORM, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                      (int)shmem), "so3_dense: shared memory attribute"))
            return e;
        set_on.fetch_or(bit, std::memory_order_release);
    }
    const long long blocks = (long long)g.zcount * g.tiles_m * g.blocks_n;
    if (blocks > 0x7fffffffLL) return eap::bad_arg("so3_dense: too many workgroups");
    hipLaunchKernelGGL((kc_gemm_kernel<MI, FORM, DBG>), dim3((unsigned)blocks), dim3(256), shmem, s, g);
    eap::set_kernel(MI == 8 ? (FORM ? "kc_gemm_kernel<8, 1>" : "kc_gemm_kernel<8, 0>") : (FORM ? "kc_gemm_kernel<4, 1>" : "kc_gemm_kernel<4, 0>"));
    return eap::check_launch("so3_dense product");
}

int g_dense_form = 1;
int g_dense_rows = 0;       // 0: 256-row blocks where o % 256 == 0, else 128; 128: always 128-row blocks (A/B runs)

inline int ceil_to(int v, int q) { return (v + q - 1) / q * q; }

}  // namespace

// 1 (default): the weights from the squared distance; 0: from the expanded square (see the head of this file).  The tables and the
// product of one layer must be built under the same setting.  -> the previous setting
// rows per workgroup of the product: 0 = 256 where the width allows (default), 128 = always 128 (two workgroups per CU); -> old setting
extern "C" int eap_so3_dense_block_rows(int rows) {
    const int old = g_dense_rows;
    if (rows == 0 || rows == 128) g_dense_rows = rows;
    return old;
}

extern "C" int eap_so3_dense_form(int form) {
    const int old = g_dense_form;
    if (form == 0 || form == 1) g_dense_form = form;
    return old;
}

extern "C" int eap_so3_dense_supported(int p, int na, int ks, int rp, int o) {
    return p > 0 && (p % 32) == 0 && na > 0 && (na % 4) == 0 && na <= 64 && ks > 0 && (ks % 2) == 0 && rp > 0 && (rp % 16) == 0 && rp <= 32 * MEMB_WORDS &&
           (o % 128) == 0;
}

extern "C" int64_t eap_so3_dense_mask_words(int b, int p, int ks, int rp, int dir) {
    const int n = dir ? p : ks * rp, kd = dir ? ceil_to(ks * rp, KC_BK) : p;
    const int wtiles = 4 * ((n + 255) / 256), steps = kd / KC_BK;
    return ((int64_t)b * wtiles * steps * 64 + 1) / 2;               // (in 64-bit words: the table holds 32-bit words)
}

extern "C" int eap_so3_dense_member(int b, int p, int n_sup, int nn, int rp, int rows_ld, const int32_t *idx, const int32_t *rows,
                                    const int32_t *n_rows, int32_t *slot_of, uint32_t *memb, int32_t *flags, eap_stream_t stream) {
    if (b <= 0 || p <= 0) return 0;
    if (rp <= 0 || rp > 32 * MEMB_WORDS || b > 65535) return eap::bad_arg("so3_dense_member: 0 < rp <= 512, b <= 65535");
    hipStream_t s = eap::S(stream);
    if (int e = eap::hip_fail(hipMemsetAsync(slot_of, 0xff, sizeof(int32_t) * (size_t)b * n_sup, s), "so3_dense_member memset")) return e;
    if (int e = eap::hip_fail(hipMemsetAsync(flags, 0, sizeof(int32_t) * (size_t)b, s), "so3_dense_member memset")) return e;
    hipLaunchKernelGGL(dense_slots_kernel, dim3(eap::cdiv(rp, 256), b), dim3(256), 0, s, n_sup, rp, rows_ld, rows, n_rows, slot_of);
    hipLaunchKernelGGL(dense_member_kernel, dim3(eap::cdiv(p, 256), b), dim3(256), 0, s, p, n_sup, nn, rp, idx, slot_of, memb, flags);
    return eap::check_launch("so3_dense_member");
}

extern "C" int eap_so3_dense_masks(int b, int p, int ks, int rp, int dir, const uint32_t *memb, uint64_t *mask, eap_stream_t stream) {
    if (b <= 0) return 0;
    if (!eap_so3_dense_supported(p, 4, ks, rp, 256) || b > 65535) return eap::bad_arg("so3_dense_masks: shape not taken");
    const int n = dir ? p : ks * rp, kd = dir ? ceil_to(ks * rp, KC_BK) : p;
    const int wtiles = 4 * ((n + 255) / 256), steps = kd / KC_BK;
    hipLaunchKernelGGL(dense_mask_kernel, dim3(eap::cdiv((long long)wtiles * steps, 4), b), dim3(256), 0, eap::S(stream), p, ks, rp, dir, wtiles, steps, memb,
                       reinterpret_cast<unsigned *>(mask));
    return eap::check_launch("so3_dense_masks");
}

extern "C" int eap_so3_dense_tables_f32(int b, int p, int n_sup, int na, int ks, int rp, int rows_ld, float sigma, const float *q_xyz,
                                        const float *s_xyz, const int32_t *rows, const float *rk, const float *row_rot, float *centre, float *pt,
                                        float *kr, eap_stream_t stream) {
    if (b <= 0) return 0;
    if (b > 65535 || na > 65535 || !(sigma > 0.f)) return eap::bad_arg("so3_dense_tables: b, na <= 65535, sigma > 0");
    hipStream_t s = eap::S(stream);
    const int p_pad = ceil_to(p, KC_BK), kd_pad = ceil_to(ks * rp, KC_BK);
    hipLaunchKernelGGL(dense_centre_kernel, dim3(b), dim3(256), 0, s, n_sup, s_xyz, centre);
    hipLaunchKernelGGL(dense_points_kernel, dim3(eap::cdiv(p_pad, 256), b), dim3(256), 0, s, p, p_pad, g_dense_form, 1.0f / sigma, q_xyz, centre,
                       reinterpret_cast<f32x4 *>(pt));
    hipLaunchKernelGGL(dense_rows_kernel, dim3(eap::cdiv(kd_pad, 256), na, b), dim3(256), 0, s, n_sup, na, ks, rp, kd_pad, rows_ld, g_dense_form, 1.0f / sigma, s_xyz,
                       centre, rows, rk, row_rot, reinterpret_cast<f32x4 *>(kr));
    return eap::check_launch("so3_dense_tables");
}

// seg / seg_pitch: a row's l elements in l / seg segments of seg elements whose starts are seg_pitch floats apart (seg <= 0: one
// segment: src is plain [b,m,l,na]).  mapped (the forward's G: seg = rp, l = ks rp): element l of the OUTPUT is the pair (k, r)
// whose dense index is l (eap_so3_dense_product_f32), read from segment k, position r; with n_rows [b] the k-blocks past a
// cloud's own prefix are left unwritten.
// rowmax (may be null): max |src| per (cloud, row, anchor) as float bit patterns, [b][m][na] -- saves the pass that finds them
// (an upper bound will do: the scale is the power of two that brings it below 2^15)
// colmap (may be null; not with mapped) int32 [b][l]: element i of every row is element colmap[b][i] of a source row of seg_pitch floats
// (negative: zero) -- the columns of dY [b,o,P,na] that are the query points of one rigid part, seg_pitch = P na
extern "C" int eap_so3_dense_split_f32(int b, int m, int l, int na, int seg, int64_t seg_pitch, int mapped, const int32_t *n_rows, const uint32_t *rowmax,
                                       const int32_t *colmap, const float *src, float *scale, void *planes, eap_stream_t stream) {
    if (b <= 0) return 0;
    if (colmap != nullptr) {
        if (mapped || seg_pitch <= 0 || (l & 7) != 0 || (reinterpret_cast<uintptr_t>(colmap) & 15) != 0)
            return eap::bad_arg("so3_dense_split: a column map (16-byte aligned, a multiple of 8 columns) comes with the source row pitch and without the dense-index mapping");
        seg = l;
    }
    if (seg <= 0) { seg = l; seg_pitch = (int64_t)l * na; }
    if ((m % 32) != 0 || (na % 4) != 0 || na > 64 || b > 65535 || m > 65535 * 32 || (reinterpret_cast<uintptr_t>(src) & 15) || l % seg != 0 ||
        (seg_pitch & 3) != 0 || (colmap == nullptr && seg_pitch < (int64_t)seg * na) || (mapped && ((seg % 16) != 0 || ((l / seg) % 2) != 0)))
        return eap::bad_arg("so3_dense_split: m % 32, na % 4, na <= 64, 16-byte aligned source, whole segments of a 16-byte aligned pitch (mapped: seg % 16, an even number of segments)");
    hipStream_t s = eap::S(stream);
    const int kb_total = ceil_to(l, KC_BK) / 16;
    float *scale2 = scale + (size_t)b * na * m;
    if (rowmax != nullptr)
        hipLaunchKernelGGL(dense_scale_kernel, dim3(eap::cdiv((long long)b * m * na, 256)), dim3(256), 0, s, (long long)b * m * na, m, na, rowmax, scale, scale2);
    else
        hipLaunchKernelGGL(dense_rowmax_kernel, dim3(m, b), dim3(256), 0, s, m, l, na, seg, (long long)(seg_pitch / 4), colmap, reinterpret_cast<const f32x4 *>(src), scale, scale2);
    if ((long long)kb_total * (m / 32) * b > 0x7fffffffLL) return eap::bad_arg("so3_dense_split: too many workgroups");
    hipLaunchKernelGGL(dense_split_kernel<false>, dim3((unsigned)((long long)kb_total * (m / 32) * b)), dim3(256), 0, s, b, m, l, na, kb_total, seg, (long long)(seg_pitch / 4), mapped ? 1 : 0,
                       n_rows, colmap, reinterpret_cast<const f32x4 *>(src), scale2, reinterpret_cast<u32x4 *>(planes), SplitBn{nullptr, nullptr, 1.f, 1.f});
    return eap::check_launch("so3_dense_split");
}

// eap_so3_dense_split_f32 for dY = the gradient behind a training-mode BatchNorm + leaky_relu, formed on the way in (dense_split_kernel<true>):
// grad = dL/dy' [b,m,l_src,na], act = y' (same shape), coef float [5][m] = k1, k2, k3, beta, 1 / gamma, rowbound [b,m,na] = an UPPER BOUND on
// max_l |gx| as float bit patterns (the scale of a row is the power of two that brings it below 2^15: a bound that is too large costs
// nothing but dynamic range, include/eap_hip.h).  colmap as eap_so3_dense_split_f32's (may be null: l = l_src).
extern "C" int eap_so3_dense_split_bn_f32(int b, int m, int l, int l_src, int na, const uint32_t *rowbound, const int32_t *colmap, const float *grad,
                                          const float *act, const float *coef, float slope, float *scale, void *planes, eap_stream_t stream) {
    if (b <= 0) return 0;
    if (colmap == nullptr && l != l_src) return eap::bad_arg("so3_dense_split_bn: without a column map l = l_src");
    if ((m % 32) != 0 || (na % 4) != 0 || na > 64 || b > 65535 || m > 65535 * 32 || ((reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(act)) & 15) ||
        (colmap != nullptr && ((l & 7) != 0 || (reinterpret_cast<uintptr_t>(colmap) & 15) != 0)) || rowbound == nullptr || coef == nullptr || !(slope > 0.f))
        return eap::bad_arg("so3_dense_split_bn: m % 32, na % 4, na <= 64, 16-byte aligned sources, a bound per row, a positive slope (column map: 16-byte aligned, l % 8)");
    hipStream_t s = eap::S(stream);
    const int kb_total = ceil_to(l, KC_BK) / 16;
    float *scale2 = scale + (size_t)b * na * m;
    hipLaunchKernelGGL(dense_scale_kernel, dim3(eap::cdiv((long long)b * m * na, 256)), dim3(256), 0, s, (long long)b * m * na, m, na, rowbound, scale, scale2);
    if ((long long)kb_total * (m / 32) * b > 0x7fffffffLL) return eap::bad_arg("so3_dense_split_bn: too many workgroups");
    const int seg = colmap ? l : l_src;
    hipLaunchKernelGGL(dense_split_kernel<true>, dim3((unsigned)((long long)kb_total * (m / 32) * b)), dim3(256), 0, s, b, m, l, na, kb_total, seg, (long long)l_src * na / 4, 0,
                       nullptr, colmap, reinterpret_cast<const f32x4 *>(grad), scale2, reinterpret_cast<u32x4 *>(planes),
                       SplitBn{reinterpret_cast<const f32x4 *>(act), coef, slope, 1.0f / slope});
    return eap::check_launch("so3_dense_split_bn");
}

// eap_so3_dense_untranspose_map_f32 / eap_so3_dense_untranspose_f32 (map null) with y = leaky_relu(bn_scale[o] yt + bn_shift[o], slope) on the
// way out: the training-mode BatchNorm + activation behind the dense forward without a pass of its own (the moments come from a
// statistics pass over Yt, eap_bn_stats_f32 on its [b na, o, p] view)
extern "C" int eap_so3_dense_untranspose_bnact_f32(int b, int o, int p, int na, int p_dst, const int32_t *map, const float *yt, const float *bn_scale,
                                                   const float *bn_shift, float slope, float *y, eap_stream_t stream) {
    if (b <= 0) return 0;
    if ((long long)eap::cdiv(p, 64) * o * b > 0x7fffffffLL || bn_scale == nullptr || bn_shift == nullptr || (map != nullptr && p_dst <= 0) || (map == nullptr && p_dst != p))
        return eap::bad_arg("so3_dense_untranspose_bnact: scale and shift per channel; p_dst = p without a map");
    hipLaunchKernelGGL(dense_untranspose_kernel, dim3((unsigned)((long long)eap::cdiv(p, 64) * o * b)), dim3(256), sizeof(float) * (size_t)na * 65, eap::S(stream), b, o, p, na,
                       p_dst, map, nullptr, yt, y, nullptr, nullptr, bn_scale, bn_shift, slope);
    return eap::check_launch("so3_dense_untranspose_bnact");
}

// ldz (dir 0): floats between consecutive (o, k) rows of Z, >= na rp (the columns past na rp are not written)
// n_rows [b] (may be null: no trimming): referenced rows per cloud -- a cloud's products end at ceil16(n_rows[b]) row slots (dir 0: the
// slots past it are ZEROED in Z, not computed)
// The forward's stored operand in one kernel (dense_gplanes_kernel): W3 float [o ks][c] (row (o, k) of the conv weights, channels contiguous),
// Ft float [b][na][rp][c] (the referenced feature rows, channels contiguous; empty slots zero), bound float [b][na][o] >= max_{k,r} |G[b,o,(k,r),a]|,
// n_rows [b] (may be null) -> scale [b][na][o] and the planes eap_so3_dense_product_f32 (dir 1) reads.  -> 0 also when the shape is not
// taken (eap_so3_dense_gplanes_supported): the caller then makes G with a GEMM and eap_so3_dense_split_f32.
extern "C" int eap_so3_dense_gplanes_supported(int o, int c, int na, int ks, int rp) {
    return (o % 32) == 0 && (c == 64 || c == 128) && na > 0 && ks > 0 && (ks % 2) == 0 && rp > 0 && (rp % 16) == 0;
}

extern "C" int eap_so3_dense_gplanes_f32(int b, int o, int c, int na, int ks, int rp, const float *W3, const float *Ft, const int32_t *n_rows,
                                         const float *bound, float *scale, void *planes, eap_stream_t stream) {
    if (b <= 0) return 0;
    if (!eap_so3_dense_gplanes_supported(o, c, na, ks, rp) || (long long)b * na > 65535 ||
        ((reinterpret_cast<uintptr_t>(W3) | reinterpret_cast<uintptr_t>(Ft) | reinterpret_cast<uintptr_t>(planes)) & 15) != 0)
        return eap::bad_arg("so3_dense_gplanes: shape not taken (eap_so3_dense_gplanes_supported), 16-byte aligned operands");
    const int kb_total = ceil_to(ks * rp, KC_BK) / 16;
    // rows per LDS chunk (a multiple of 16): all of them where their two planes fit in 84 KB (144 rows at c = 128, 288 at c = 64: the bench
    // layers in one piece, as before the chunking), else chunks of that size
    const int fit = (int)((84u * 1024u) / ((size_t)(2 * c + 16) * 2)) / 16 * 16;
    const int rch = min(rp, max(fit, 32));
    const size_t shmem = (size_t)rch * (2 * c + 16) * 2;
    hipStream_t s = eap::S(stream);
    auto launch = [&](auto kern) -> int {
        if (int e = eap::hip_fail(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem),
                                  "so3_dense_gplanes: shared memory attribute"))
            return e;
        hipLaunchKernelGGL(kern, dim3(eap::cdiv(o / 32, 4), b * na), dim3(256), shmem, s, o, na, ks, rp, rch, kb_total, W3, Ft, n_rows, bound, scale,
                           reinterpret_cast<u32x4 *>(planes));
        return eap::check_launch("so3_dense_gplanes");
    };
    return c == 128 ? launch(dense_gplanes_kernel<8>) : launch(dense_gplanes_kernel<4>);
}

// steps (may be null): the k-step lists of eap_so3_dense_steps for this direction -- column blocks run their listed k-steps only
extern "C" int eap_so3_dense_product_steps_f32(int dir, int b, int o, int p, int na, int ks, int rp, int64_t ldz, float sigma, const int32_t *n_rows,
                                               const void *planes, const float *scale, const float *pt, const float *kr, const uint64_t *mask,
                                               const int32_t *steps, float *out, eap_stream_t stream) {
    if (b <= 0) return 0;
    if (!eap_so3_dense_supported(p, na, ks, rp, o)) return eap::bad_arg("so3_dense_product: shape not taken (eap_so3_dense_supported)");
    const int p_pad = ceil_to(p, KC_BK), kd_pad = ceil_to(ks * rp, KC_BK);
    KcArgs g;
    g.MT = o / 32; g.na = na; g.zcount = b * na; g.row_slots = rp;
    g.N = dir ? p : ks * rp;
    g.KS = (dir ? kd_pad : p) / KC_BK;
    const bool wide = (o % 256) == 0 && g_dense_rows != 128;     // 256-row blocks; else 128-row blocks (half the matrix work per generated weight, two workgroups per CU)
    g.tiles_m = wide ? o / 256 : o / 128;
    g.blocks_n = (g.N + 255) / 256;
    g.mask_tiles = 4 * g.blocks_n;
    g.A = reinterpret_cast<const u32x4 *>(planes);
    g.scale = scale;
    const f32x4 *ptT = reinterpret_cast<const f32x4 *>(pt), *krT = reinterpret_cast<const f32x4 *>(kr);
    if (dir == 0) {
        g.strT = ptT; g.strB = p_pad; g.strA = 0;
        g.colT = krT; g.colB = (long long)na * kd_pad; g.colA = kd_pad;
        if (ldz < (int64_t)na * rp) return eap::bad_arg("so3_dense_product: ldz < na rp");
        g.cB = (long long)o * ks * ldz; g.cA = rp; g.ldm = (long long)ks * ldz; g.rp = rp; g.kstride = ldz;
    } else {
        g.strT = krT; g.strB = (long long)na * kd_pad; g.strA = kd_pad;
        g.colT = ptT; g.colB = p_pad; g.colA = 0;
        g.cB = (long long)na * o * p; g.cA = (long long)o * p; g.ldm = p; g.rp = p; g.kstride = 0;
    }
    g.mask = reinterpret_cast<const unsigned *>(mask);
    g.steps = (steps != nullptr && g.KS <= KC_LIST_BYTES / 4) ? steps : nullptr;      // (longer contraction axes run every k-step)
    g.C = out;
    g.neg_inv_sigma = -1.0f / sigma;
    g.n_rows = n_rows; g.ks = ks; g.trim = dir ? 2 : 1;       // (columns / k axis are dense indices either way; without n_rows every slot counts)
    if (n_rows == nullptr) g.trim = dir ? 0 : 3;
    if (dir == 0 && n_rows != nullptr) {
        hipLaunchKernelGGL(dense_zero_tail_kernel, dim3(o * ks, b), dim3(256), 0, eap::S(stream), o * ks, na, rp, (long long)ldz, n_rows, out);
        if (int e = eap::check_launch("so3_dense zero tail")) return e;
    }
#ifdef EAP_ABLATION
    if (const char *d = wide ? getenv("EAP_DENSE_DEBUG") : nullptr) {
        switch (atoi(d)) {
            case 1: return kc_launch<8, 1, 1>(g, eap::S(stream));
            case 2: return kc_launch<8, 1, 2>(g, eap::S(stream));
            case 3: return kc_launch<8, 1, 3>(g, eap::S(stream));
            case 4: return kc_launch<8, 1, 4>(g, eap::S(stream));
            case 8: return kc_launch<8, 1, 8>(g, eap::S(stream));
            case 16: return kc_launch<8, 1, 16>(g, eap::S(stream));
            case 20: return kc_launch<8, 1, 20>(g, eap::S(stream));
            case 28: return kc_launch<8, 1, 28>(g, eap::S(stream));
            default: break;
        }
    }
#endif
    if (!wide) return g_dense_form ? kc_launch<4, 1>(g, eap::S(stream)) : kc_launch<4, 0>(g, eap::S(stream));
    return g_dense_form ? kc_launch<8, 1>(g, eap::S(stream)) : kc_launch<8, 0>(g, eap::S(stream));
}

extern "C" int eap_so3_dense_product_f32(int dir, int b, int o, int p, int na, int ks, int rp, int64_t ldz, float sigma, const int32_t *n_rows,
                                         const void *planes, const float *scale, const float *pt, const float *kr, const uint64_t *mask, float *out, eap_stream_t stream) {
    return eap_so3_dense_product_steps_f32(dir, b, o, p, na, ks, rp, ldz, sigma, n_rows, planes, scale, pt, kr, mask, nullptr, out, stream);
}

// keys int32 [b,p]: the 16-row groups every point's list touches, one bit per group (see dense_keys_kernel) -- the sort key that makes the
// mask block-sparse
extern "C" int eap_so3_dense_point_keys(int b, int p, const uint32_t *memb, int32_t *keys, eap_stream_t stream) {
    if (b <= 0 || p <= 0) return 0;
    const long long total = (long long)b * p;
    hipLaunchKernelGGL(dense_keys_kernel, dim3((unsigned)eap::cdiv(total, 256)), dim3(256), 0, eap::S(stream), total, memb, keys);
    return eap::check_launch("so3_dense_point_keys");
}

// int32 words of the k-step lists of one direction: [b][column blocks of 256][k-steps + 1]
extern "C" int64_t eap_so3_dense_steps_words(int b, int p, int ks, int rp, int dir) {
    const int n = dir ? p : ks * rp, kd = dir ? ceil_to(ks * rp, KC_BK) : p;
    return (int64_t)b * ((n + 255) / 256) * (kd / KC_BK + 1);
}

// the k-step lists of one direction from its mask table (eap_so3_dense_masks, same b, p, ks, rp, dir); skip = 0: every k-step is listed
extern "C" int eap_so3_dense_steps(int b, int p, int ks, int rp, int dir, int skip, const int32_t *n_rows, const uint64_t *mask, int32_t *steps,
                                   eap_stream_t stream) {
    if (b <= 0) return 0;
    if (!eap_so3_dense_supported(p, 4, ks, rp, 256) || b > 65535) return eap::bad_arg("so3_dense_steps: shape not taken");
    const int n = dir ? p : ks * rp, kd = dir ? ceil_to(ks * rp, KC_BK) : p;
    const int blocks_n = (n + 255) / 256, KS = kd / KC_BK;
    if ((size_t)KS * sizeof(int) > 48 * 1024) return eap::bad_arg("so3_dense_steps: more than 12288 k-steps");
    hipLaunchKernelGGL(dense_steps_kernel, dim3(blocks_n, b), dim3(256), sizeof(int) * (size_t)KS, eap::S(stream), KS, blocks_n, 4 * blocks_n, dir, ks, rp, skip,
                       n_rows, reinterpret_cast<const unsigned *>(mask), steps);
    return eap::check_launch("so3_dense_steps");
}

// psum, psq (may be null): float [o][b * ceil(p / 64)] partial sums of (y - y[0,o,0,0]) and of its square per 64-point chunk
extern "C" int eap_so3_dense_untranspose_f32(int b, int o, int p, int na, const float *yt, float *y, float *psum, float *psq, eap_stream_t stream) {
    if (b <= 0) return 0;
    if ((long long)eap::cdiv(p, 64) * o * b > 0x7fffffffLL) return eap::bad_arg("so3_dense_untranspose: too many workgroups");
    hipLaunchKernelGGL(dense_untranspose_kernel, dim3((unsigned)((long long)eap::cdiv(p, 64) * o * b)), dim3(256), sizeof(float) * (size_t)na * 65, eap::S(stream), b, o, p, na, p, nullptr, nullptr,
                       yt, y, psum, psum ? psq : nullptr);
    return eap::check_launch("so3_dense_untranspose");
}

// the same re-ordering into a Y [b,o,p_dst,na] that other launches fill too: column pp of cloud b goes to point map[b][pp] (int32 [b,p];
// negative: padding, dropped)
extern "C" int eap_so3_dense_untranspose_map_f32(int b, int o, int p, int na, int p_dst, const int32_t *map, const float *yt, float *y,
                                                 eap_stream_t stream) {
    if (b <= 0) return 0;
    if (o > 65535 || b > 65535 || map == nullptr || p_dst <= 0) return eap::bad_arg("so3_dense_untranspose_map: o, b <= 65535, a map, p_dst > 0");
    hipLaunchKernelGGL(dense_untranspose_kernel, dim3((unsigned)((long long)eap::cdiv(p, 64) * o * b)), dim3(256), sizeof(float) * (size_t)na * 65, eap::S(stream), b, o, p, na, p_dst, map, nullptr,
                       yt, y, nullptr, nullptr);
    return eap::check_launch("so3_dense_untranspose_map");
}

// ... and with the channel moments of eap_so3_dense_untranspose_f32 (over the columns whose map entry is not negative); pivot_pos int32 [1]
// (device): the column of cloud 0 that is point 0, so that the pivot is Y[0][o][0][0] as the BatchNorm behind it expects
extern "C" int eap_so3_dense_untranspose_map_stats_f32(int b, int o, int p, int na, int p_dst, const int32_t *map, const int32_t *pivot_pos, const float *yt,
                                                       float *y, float *psum, float *psq, eap_stream_t stream) {
    if (b <= 0) return 0;
    if (o > 65535 || b > 65535 || map == nullptr || pivot_pos == nullptr || p_dst <= 0 || psum == nullptr || psq == nullptr)
        return eap::bad_arg("so3_dense_untranspose_map_stats: o, b <= 65535, a map, the pivot column, both partial arrays");
    hipLaunchKernelGGL(dense_untranspose_kernel, dim3((unsigned)((long long)eap::cdiv(p, 64) * o * b)), dim3(256), sizeof(float) * (size_t)na * 65, eap::S(stream), b, o, p, na, p_dst, map,
                       pivot_pos, yt, y, psum, psq);
    return eap::check_launch("so3_dense_untranspose_map_stats");
}
