/*
 * oracle/native_ops.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of the reference's native (CUDA-only)
 * operators.  The reference has no CPU implementation of these ops (every C++
 * entry asserts is_cuda: vgtk/vgtk/cuda/zpconv_cuda.cpp:L37-39), so each
 * function below restates the algorithm of the cited .cu kernel, one loop nest
 * per CUDA thread index.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path never does.
 *
 * Parity status: these ops are pinned by CODE READING plus hand-built
 * known-answer cases (tests/test_oracle_native.py); no reference test or
 * importable reference implementation exists for them.
 * Where the same math is reachable through the reference's importable Python
 * "naive" path (inter/intra zpconv with an index shared across (a,k)), the
 * golden fixtures cross-check it.
 *
 * Floating-point evaluation order: squared distances are evaluated exactly as
 * written in the reference source, left to right, one rounding per operation
 * (no FMA contraction) -- build with -ffp-contract=off.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* ball_query: vgtk/vgtk/cuda/grouping_cuda_kernel.cu:L68-113               */
/*   new_xyz [b,3,m], xyz [b,3,n] -> idx int32 [b,m,nsample] (zero-init by   */
/*   the host wrapper grouping_cuda.cpp:L80-82).                             */
/*   - first `nsample` support indices, in index order, with d2 < r*r        */
/*   - repeat-padding ONLY when cnt < nsample-1 (L100-105); with exactly     */
/*     nsample-1 hits the last slot keeps its zero initialisation            */
/*   - radius2 = radius*radius is formed in float, then cast to scalar_t     */
/* ------------------------------------------------------------------------ */
#define DEF_BALL_QUERY(NAME, T)                                                 \
void NAME(int b, int n, int m, float radius, int nsample,                      \
          const T *new_xyz, const T *xyz, int32_t *idx) {                      \
    const T radius2 = (T)(radius * radius);                                    \
    memset(idx, 0, sizeof(int32_t) * (size_t)b * m * nsample);                 \
    for (int bi = 0; bi < b; ++bi) {                                           \
        const T *X = xyz + (size_t)bi * 3 * n;                                 \
        const T *Q = new_xyz + (size_t)bi * 3 * m;                             \
        int32_t *I = idx + (size_t)bi * m * nsample;                           \
        for (int j = 0; j < m; ++j) {                                          \
            const T qx = Q[0 * m + j], qy = Q[1 * m + j], qz = Q[2 * m + j];   \
            int cnt = 0;                                                       \
            for (int k = 0; k < n && cnt < nsample; ++k) {                     \
                const T x = X[0 * n + k], y = X[1 * n + k], z = X[2 * n + k];  \
                const T dx = qx - x, dy = qy - y, dz = qz - z;                 \
                const T d2 = (T)((T)((T)(dx * dx) + (T)(dy * dy)) + (T)(dz * dz)); \
                if (d2 < radius2) { I[j * nsample + cnt] = k; ++cnt; }         \
            }                                                                  \
            if (cnt < nsample - 1) {                                           \
                for (int k = 0; k + cnt < nsample; ++k)                        \
                    I[j * nsample + k + cnt] = I[j * nsample + k];             \
            }                                                                  \
        }                                                                      \
    }                                                                          \
}
DEF_BALL_QUERY(oracle_ball_query_f32, float)
DEF_BALL_QUERY(oracle_ball_query_f64, double)

/* ------------------------------------------------------------------------ */
/* gather_points fwd/bwd: vgtk/vgtk/cuda/gathering_cuda_kernel.cu:L43-98     */
/*   fwd: out[b,c,m] = pts[b,c,idx[b,m]]   bwd: gpts[b,c,idx[b,m]] += g[b,c,m]*/
/* ------------------------------------------------------------------------ */
void oracle_gather_points_fwd_f32(int b, int c, int n, int m, const float *pts,
                                  const int32_t *idx, float *out) {
    for (int bi = 0; bi < b; ++bi)
        for (int ci = 0; ci < c; ++ci)
            for (int j = 0; j < m; ++j)
                out[((size_t)bi * c + ci) * m + j] =
                    pts[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + j]];
}

#define DEF_GATHER_BWD(NAME, T)                                                 \
void NAME(int b, int c, int n, int m, const T *grad_out, const int32_t *idx,   \
          T *grad_pts) {                                                       \
    memset(grad_pts, 0, sizeof(T) * (size_t)b * c * n);                        \
    for (int bi = 0; bi < b; ++bi)                                             \
        for (int j = 0; j < m; ++j)                                            \
            for (int ci = 0; ci < c; ++ci)                                     \
                grad_pts[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + j]] += \
                    grad_out[((size_t)bi * c + ci) * m + j];                   \
}
DEF_GATHER_BWD(oracle_gather_points_bwd_f32, float)
DEF_GATHER_BWD(oracle_gather_points_bwd_f64, double)

/* ------------------------------------------------------------------------ */
/* inter zpconv fwd/bwd: vgtk/vgtk/cuda/zpconv_cuda_kernel.cu:L33-116        */
/*   idx,w [b,np,na,ks,ann]; feats [b,c,nq,na]; out [b,c,ks,np,na]           */
/*   out[b,c,k,p,a] += feats[b,c,idx,a] * w      (thread order = flat idx)   */
/* ------------------------------------------------------------------------ */
#define DEF_INTER_ZP(NAMEF, NAMEB, T)                                           \
void NAMEF(int b, int np, int nq, int na, int ks, int ann, int c,              \
           const int32_t *idx, const T *w, const T *feats, T *out) {           \
    memset(out, 0, sizeof(T) * (size_t)b * c * ks * np * na);                  \
    for (int bn = 0; bn < b; ++bn)                                             \
      for (int pn = 0; pn < np; ++pn)                                          \
        for (int an = 0; an < na; ++an)                                        \
          for (int k = 0; k < ks; ++k)                                         \
            for (int ni = 0; ni < ann; ++ni) {                                 \
                const size_t q = ((((size_t)bn * np + pn) * na + an) * ks + k) * ann + ni; \
                const int qn = idx[q];                                         \
                const T ww = w[q];                                             \
                for (int ci = 0; ci < c; ++ci)                                 \
                    out[((((size_t)bn * c + ci) * ks + k) * np + pn) * na + an] += \
                        feats[(((size_t)bn * c + ci) * nq + qn) * na + an] * ww; \
            }                                                                  \
}                                                                              \
void NAMEB(int b, int np, int nq, int na, int ks, int ann, int c,              \
           const int32_t *idx, const T *w, const T *gout, T *gfeats) {         \
    memset(gfeats, 0, sizeof(T) * (size_t)b * c * nq * na);                    \
    for (int bn = 0; bn < b; ++bn)                                             \
      for (int pn = 0; pn < np; ++pn)                                          \
        for (int an = 0; an < na; ++an)                                        \
          for (int k = 0; k < ks; ++k)                                         \
            for (int ni = 0; ni < ann; ++ni) {                                 \
                const size_t q = ((((size_t)bn * np + pn) * na + an) * ks + k) * ann + ni; \
                const int qn = idx[q];                                         \
                const T ww = w[q];                                             \
                for (int ci = 0; ci < c; ++ci)                                 \
                    gfeats[(((size_t)bn * c + ci) * nq + qn) * na + an] +=     \
                        gout[((((size_t)bn * c + ci) * ks + k) * np + pn) * na + an] * ww; \
            }                                                                  \
}
DEF_INTER_ZP(oracle_inter_zpconv_fwd_f32, oracle_inter_zpconv_bwd_f32, float)
DEF_INTER_ZP(oracle_inter_zpconv_fwd_f64, oracle_inter_zpconv_bwd_f64, double)

/* ------------------------------------------------------------------------ */
/* intra zpconv fwd/bwd: vgtk/vgtk/cuda/zpconv_cuda_kernel.cu:L120-195       */
/*   idx [na_out,ann]; w [na_out,ks,ann]; feats [b,c,np,na_in];              */
/*   out [b,c,ks,np,na_out]                                                  */
/* ------------------------------------------------------------------------ */
#define DEF_INTRA_ZP(NAMEF, NAMEB, T)                                           \
void NAMEF(int b, int np, int na_in, int na_out, int ks, int ann, int c,       \
           const int32_t *idx, const T *w, const T *feats, T *out) {           \
    memset(out, 0, sizeof(T) * (size_t)b * c * ks * np * na_out);              \
    for (int bn = 0; bn < b; ++bn)                                             \
      for (int pn = 0; pn < np; ++pn)                                          \
        for (int an = 0; an < na_out; ++an)                                    \
          for (int k = 0; k < ks; ++k)                                         \
            for (int ni = 0; ni < ann; ++ni) {                                 \
                const int qan = idx[an * ann + ni];                            \
                const T ww = w[(an * ks + k) * ann + ni];                      \
                for (int ci = 0; ci < c; ++ci)                                 \
                    out[((((size_t)bn * c + ci) * ks + k) * np + pn) * na_out + an] += \
                        feats[(((size_t)bn * c + ci) * np + pn) * na_in + qan] * ww; \
            }                                                                  \
}                                                                              \
void NAMEB(int b, int np, int na_in, int na_out, int ks, int ann, int c,       \
           const int32_t *idx, const T *w, const T *gout, T *gfeats) {         \
    memset(gfeats, 0, sizeof(T) * (size_t)b * c * np * na_in);                 \
    for (int bn = 0; bn < b; ++bn)                                             \
      for (int pn = 0; pn < np; ++pn)                                          \
        for (int an = 0; an < na_out; ++an)                                    \
          for (int k = 0; k < ks; ++k)                                         \
            for (int ni = 0; ni < ann; ++ni) {                                 \
                const int qan = idx[an * ann + ni];                            \
                const T ww = w[(an * ks + k) * ann + ni];                      \
                for (int ci = 0; ci < c; ++ci)                                 \
                    gfeats[(((size_t)bn * c + ci) * np + pn) * na_in + qan] += \
                        gout[((((size_t)bn * c + ci) * ks + k) * np + pn) * na_out + an] * ww; \
            }                                                                  \
}
DEF_INTRA_ZP(oracle_intra_zpconv_fwd_f32, oracle_intra_zpconv_bwd_f32, float)
DEF_INTRA_ZP(oracle_intra_zpconv_fwd_f64, oracle_intra_zpconv_bwd_f64, double)

/* ------------------------------------------------------------------------ */
/* furthest point sampling: grouping_cuda_kernel.cu:L352-466                 */
/*   xyz [b,3,n] -> idx int32 [b,m]; first index 0; temp init 1e10           */
/*   (grouping_cuda.cpp:L166-168); points with |x|^2 <= 1e-3 are skipped     */
/*   (L385-387).  Tie-break: the kernel's per-thread scan keeps the FIRST    */
/*   strict maximum and the tree reduction keeps the lower slot on ties      */
/*   (`v2 > v1 ? i2 : i1`), so ties resolve to the smallest (k mod block)    */
/*   then smallest k within a thread; `block` below reproduces that.         */
/* ------------------------------------------------------------------------ */
static int fps_block(int n) {  /* opt_n_threads: grouping_cuda_kernel.cu:L29-33 */
    int p = (int)(log((double)n) / log(2.0));
    int t = 1 << p;
    if (t > 1024) t = 1024;
    if (t < 1) t = 1;
    return t;
}
#define DEF_FPS(NAME, T)                                                        \
void NAME(int b, int n, int m, const T *xyz, int32_t *idx) {                   \
    if (m <= 0) return;                                                        \
    const int block = fps_block(n);                                            \
    T *temp = (T *)malloc(sizeof(T) * (size_t)n);                              \
    T *bd = (T *)malloc(sizeof(T) * (size_t)block);                            \
    int *bi_ = (int *)malloc(sizeof(int) * (size_t)block);                     \
    for (int bn = 0; bn < b; ++bn) {                                           \
        const T *X = xyz + (size_t)bn * 3 * n;                                 \
        int32_t *I = idx + (size_t)bn * m;                                     \
        for (int k = 0; k < n; ++k) temp[k] = (T)1e10;                         \
        int old = 0;                                                           \
        I[0] = 0;                                                              \
        for (int j = 1; j < m; ++j) {                                          \
            const T x1 = X[old], y1 = X[n + old], z1 = X[2 * n + old];         \
            for (int t = 0; t < block; ++t) { bd[t] = (T)-1; bi_[t] = 0; }     \
            for (int k = 0; k < n; ++k) {                                      \
                const int t = k % block;                                       \
                const T x2 = X[k], y2 = X[n + k], z2 = X[2 * n + k];           \
                const T mag = (T)((T)((T)(x2 * x2) + (T)(y2 * y2)) + (T)(z2 * z2)); \
                if (mag <= (T)1e-3) continue;                                  \
                const T dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;              \
                const T d = (T)((T)((T)(dx * dx) + (T)(dy * dy)) + (T)(dz * dz)); \
                const T d2 = d < temp[k] ? d : temp[k];                        \
                temp[k] = d2;                                                  \
                if (d2 > bd[t]) { bd[t] = d2; bi_[t] = k; }                    \
            }                                                                  \
            for (int s = block / 2; s >= 1; s /= 2)                            \
                for (int t = 0; t < s; ++t)                                    \
                    if (bd[t + s] > bd[t]) { bd[t] = bd[t + s]; bi_[t] = bi_[t + s]; } \
            old = bi_[0];                                                      \
            I[j] = old;                                                        \
        }                                                                      \
    }                                                                          \
    free(temp); free(bd); free(bi_);                                           \
}
DEF_FPS(oracle_fps_f32, float)
DEF_FPS(oracle_fps_f64, double)

/* ------------------------------------------------------------------------ */
/* chamfer fwd/bwd: extensions/chamfer_dist/chamfer.cu:L15-145, L173-201      */
/*   xyz1 [b,n,3], xyz2 [b,m,3] -> dist1 [b,n] idx1 [b,n] (one direction;     */
/*   the host calls it twice with the clouds swapped, L159-164).             */
/*   Arg-min keeps the first minimum (strict `<` inside a 512-point tile,    */
/*   strict `>` across tiles).  float only, like the reference.              */
/* ------------------------------------------------------------------------ */
void oracle_chamfer_nn_f32(int b, int n, const float *xyz1, int m,
                           const float *xyz2, float *dist, int32_t *indexes) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float x1 = xyz1[((size_t)i * n + j) * 3 + 0];
            const float y1 = xyz1[((size_t)i * n + j) * 3 + 1];
            const float z1 = xyz1[((size_t)i * n + j) * 3 + 2];
            float best = 0.f;
            int besti = 0;
            for (int k = 0; k < m; ++k) {
                const float x2 = xyz2[((size_t)i * m + k) * 3 + 0] - x1;
                const float y2 = xyz2[((size_t)i * m + k) * 3 + 1] - y1;
                const float z2 = xyz2[((size_t)i * m + k) * 3 + 2] - z1;
                const float d = (float)((float)((float)(x2 * x2) + (float)(y2 * y2)) + (float)(z2 * z2));
                if (k == 0 || d < best) { best = d; besti = k; }
            }
            dist[(size_t)i * n + j] = best;
            indexes[(size_t)i * n + j] = besti;
        }
}

/* one direction of the gradient (chamfer.cu:L173-201); accumulates into
 * grad_xyz1 / grad_xyz2, which the caller zero-initialises (L208-209). */
void oracle_chamfer_grad_f32(int b, int n, const float *xyz1, int m,
                             const float *xyz2, const float *grad_dist1,
                             const int32_t *idx1, float *grad_xyz1,
                             float *grad_xyz2) {
    for (int i = 0; i < b; ++i)
        for (int j = 0; j < n; ++j) {
            const float *p1 = xyz1 + ((size_t)i * n + j) * 3;
            const int j2 = idx1[(size_t)i * n + j];
            const float *p2 = xyz2 + ((size_t)i * m + j2) * 3;
            const float g = grad_dist1[(size_t)i * n + j] * 2;
            for (int d = 0; d < 3; ++d) {
                const float v = g * (p1[d] - p2[d]);
                grad_xyz1[((size_t)i * n + j) * 3 + d] += v;
                grad_xyz2[((size_t)i * m + j2) * 3 + d] += -v;
            }
        }
}

/* ------------------------------------------------------------------------ */
/* anchor_query (S^2 variant): grouping_cuda_kernel.cu:L181-247               */
/*   grouped_xyz [b,3,np,nn], anchors [na,3], kernel_points [ks,2]           */
/*   -> w [b,np,na,ks,nn] = (kw-|x|)^2 + ((kh-theta)|x|)^2                   */
/* ------------------------------------------------------------------------ */
#define DEF_ANCHOR_QUERY(NAME, T, SQRT, ACOS)                                   \
void NAME(int b, int np, int nn, int na, int ks, const T *gxyz,                \
          const T *anchors, const T *kpts, T *w) {                             \
    for (int bi = 0; bi < b; ++bi)                                             \
      for (int idx = 0; idx < np * nn; ++idx) {                                \
        const int pi = idx / nn, ni = idx % nn;                                \
        const T *g = gxyz + (size_t)bi * 3 * np * nn;                          \
        const T x = g[idx], y = g[(size_t)np * nn + idx], z = g[(size_t)2 * np * nn + idx]; \
        const T norm = (T)(SQRT((T)((T)((T)(x * x) + (T)(y * y)) + (T)(z * z))) + (T)1e-6); \
        for (int ai = 0; ai < na; ++ai) {                                      \
            const T dot = (T)((T)((T)(x * anchors[ai * 3]) + (T)(y * anchors[ai * 3 + 1])) + (T)(z * anchors[ai * 3 + 2])); \
            const T theta = ACOS(dot / norm);                                  \
            for (int ki = 0; ki < ks; ++ki) {                                  \
                const T kw = kpts[ki * 2], kh = kpts[ki * 2 + 1];              \
                const T a = kw - norm, c = (kh - theta) * norm;                \
                w[((((size_t)bi * np + pi) * na + ai) * ks + ki) * nn + ni] =  \
                    (T)((T)(a * a) + (T)(c * c));                              \
            }                                                                  \
        }                                                                      \
      }                                                                        \
}
DEF_ANCHOR_QUERY(oracle_anchor_query_f32, float, sqrtf, acosf)
DEF_ANCHOR_QUERY(oracle_anchor_query_f64, double, sqrt, acos)

/* ------------------------------------------------------------------------ */
/* initial_anchor_query: grouping_cuda_kernel.cu:L117-167                     */
/*   centers [b,3,nc], xyz [m,3], kernel_points [ks,na,3]                    */
/*   -> w, cnt [b,ks,nc,na]  (summation order here: pm-major, the CUDA       */
/*   atomics have no defined order)                                          */
/* ------------------------------------------------------------------------ */
#define DEF_INIT_ANCHOR_QUERY(NAME, T, SQRT)                                    \
void NAME(int b, int nc, int m, int na, int ks, float radius, float sigma,     \
          const T *centers, const T *xyz, const T *kpts, T *w, T *cnt) {       \
    memset(w, 0, sizeof(T) * (size_t)b * ks * nc * na);                        \
    memset(cnt, 0, sizeof(T) * (size_t)b * ks * nc * na);                      \
    for (int bn = 0; bn < b; ++bn)                                             \
      for (int pm = 0; pm < m; ++pm)                                           \
        for (int kn = 0; kn < ks; ++kn) {                                      \
          const T x = xyz[3 * pm], y = xyz[3 * pm + 1], z = xyz[3 * pm + 2];   \
          const T *C = centers + (size_t)bn * 3 * nc;                          \
          for (int pn = 0; pn < nc; ++pn) {                                    \
            const T cx = C[pn], cy = C[nc + pn], cz = C[2 * nc + pn];          \
            const T dc = SQRT((T)((T)((T)((cx - x) * (cx - x)) + (T)((cy - y) * (cy - y))) + (T)((cz - z) * (cz - z)))); \
            if (dc <= (T)radius)                                               \
              for (int an = 0; an < na; ++an) {                                \
                const T kx = kpts[(kn * na + an) * 3] + cx;                    \
                const T ky = kpts[(kn * na + an) * 3 + 1] + cy;                \
                const T kz = kpts[(kn * na + an) * 3 + 2] + cz;                \
                const T dk = SQRT((T)((T)((T)((kx - x) * (kx - x)) + (T)((ky - y) * (ky - y))) + (T)((kz - z) * (kz - z)))); \
                const T wt = (T)1 - (T)((T)(dk * dk) / (T)sigma);              \
                const size_t o = (((size_t)bn * ks + kn) * nc + pn) * na + an; \
                if (wt > (T)0) w[o] += wt;                                     \
                cnt[o] += (T)1;                                                \
              }                                                                \
          }                                                                    \
        }                                                                      \
}
DEF_INIT_ANCHOR_QUERY(oracle_initial_anchor_query_f32, float, sqrtf)
DEF_INIT_ANCHOR_QUERY(oracle_initial_anchor_query_f64, double, sqrt)
