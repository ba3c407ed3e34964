// csrc/narrow_contract.hip -- the pointwise contraction y[b,o,n] = sum_c W[o,c] x[b,c,n] for at most FOUR output channels:
// the last layer of the pose head's dense translation branch (nn.Conv2d(c, 3 * num_heads, 1),
// SPConvNets/models/model_utils.py:L537-552) and the attention logit of InvPPOutBlockOurs (nn.Conv2d(c, 1, 1),
// SPConvNets/utils/base_so3conv.py:L905-912) over the [b, c, P*A] feature map.
//
// With 1-4 rows there is nothing for a matrix core to do: the op is one streaming pass over x (forward: read x, 4 bytes per
// element; dX: write it; dW: read it), HBM-bound.  A 64-row GEMM tile spends its time on 61 rows of zeros (2.4 ms per call at
// c = 256, 16 clouds of 4096 points); these kernels run the forward in 0.3-0.4 ms.
//   forward / dX : one thread per 4 consecutive columns (16-byte loads and stores), W in LDS
//   dW           : lanes hold the gradient of their 4 columns and accumulate per-channel partial sums over a column slab;
//                  partials [slab][cloud][o][c] are summed in a fixed order by the caller (deterministic, no atomics)
#include "common.h"
#include <type_traits>

namespace {

constexpr int TB = 256;
constexpr int MAXC = 2048;          // W rows staged in LDS: 4 * 2048 floats = 32 KB
constexpr int CS = 16;              // channels per accumulator group of the dW kernel

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int O>
__global__ __launch_bounds__(TB) void narrow_fwd_kernel(int c, long n, const float *__restrict__ W, const float *__restrict__ x,
                                                       float *__restrict__ y) {
    __shared__ float w[O * MAXC];
    for (int i = threadIdx.x; i < O * c; i += TB) w[i] = W[i];
    __syncthreads();
    const long col = ((long)blockIdx.x * TB + threadIdx.x) * 4;
    if (col >= n) return;
    const float *xb = x + (size_t)blockIdx.y * c * n + col;
    f32x4 acc[O];
#pragma unroll
    for (int o = 0; o < O; ++o) acc[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int ci = 0;
    for (; ci + 4 <= c; ci += 4) {              // four independent 16-byte loads in flight
        f32x4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4 *>(xb + (size_t)(ci + j) * n);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int o = 0; o < O; ++o) acc[o] += v[j] * w[o * c + ci + j];
    }
    for (; ci < c; ++ci) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + (size_t)ci * n);
#pragma unroll
        for (int o = 0; o < O; ++o) acc[o] += v * w[o * c + ci];
    }
    float *yb = y + (size_t)blockIdx.y * O * n + col;
#pragma unroll
    for (int o = 0; o < O; ++o) *reinterpret_cast<f32x4 *>(yb + (size_t)o * n) = acc[o];
}

// dx[b,c,n] = sum_o W[o,c] g[b,o,n]
template <int O>
__global__ __launch_bounds__(TB) void narrow_dx_kernel(int c, long n, const float *__restrict__ W, const float *__restrict__ g,
                                                      float *__restrict__ dx) {
    __shared__ float w[O * MAXC];
    for (int i = threadIdx.x; i < O * c; i += TB) w[i] = W[i];
    __syncthreads();
    const long col = ((long)blockIdx.x * TB + threadIdx.x) * 4;
    if (col >= n) return;
    const float *gb = g + (size_t)blockIdx.y * O * n + col;
    f32x4 gv[O];
#pragma unroll
    for (int o = 0; o < O; ++o) gv[o] = *reinterpret_cast<const f32x4 *>(gb + (size_t)o * n);
    float *db = dx + (size_t)blockIdx.y * c * n + col;
    for (int ci = 0; ci < c; ++ci) {
        f32x4 v = gv[0] * w[ci];
#pragma unroll
        for (int o = 1; o < O; ++o) v += gv[o] * w[o * c + ci];
        *reinterpret_cast<f32x4 *>(db + (size_t)ci * n) = v;
    }
}

// partial[(slab * B + b) * O * c + o * c + ci] = sum over the slab's columns of g[b,o,n] x[b,ci,n]
// grid (slabs, channel groups of CS, clouds); a block walks its slab in steps of TB * 4 columns
template <int O>
__global__ __launch_bounds__(TB) void narrow_dw_kernel(int c, long n, long slab_cols, const float *__restrict__ g,
                                                      const float *__restrict__ x, float *__restrict__ partial) {
    const int slab = blockIdx.x, cg = blockIdx.y, b = blockIdx.z, nb = gridDim.z;
    const int c0 = cg * CS, cn = min(CS, c - c0);
    const long beg = (long)slab * slab_cols, end = min(n, beg + slab_cols);
    const float *gb = g + (size_t)b * O * n, *xb = x + ((size_t)b * c + c0) * n;
    float acc[O][CS];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int j = 0; j < CS; ++j) acc[o][j] = 0.f;
    for (long col = beg + (long)threadIdx.x * 4; col < end; col += TB * 4) {
        f32x4 gv[O];
#pragma unroll
        for (int o = 0; o < O; ++o) gv[o] = *reinterpret_cast<const f32x4 *>(gb + (size_t)o * n + col);
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            if (j < cn) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + (size_t)j * n + col);
#pragma unroll
                for (int o = 0; o < O; ++o) acc[o][j] += (gv[o].x * v.x + gv[o].y * v.y) + (gv[o].z * v.z + gv[o].w * v.w);
            }
        }
    }
    // block reduction: wave shuffles, then the four waves through LDS
    __shared__ float red[TB / 64][O * CS];
#pragma unroll
    for (int o = 0; o < O; ++o)
#pragma unroll
        for (int j = 0; j < CS; ++j) {
            float v = acc[o][j];
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][o * CS + j] = v;
        }
    __syncthreads();
    if (threadIdx.x < O * CS) {
        const int o = threadIdx.x / CS, j = threadIdx.x % CS;
        if (j < cn) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < TB / 64; ++wv) v += red[wv][threadIdx.x];
            partial[((size_t)slab * nb + b) * O * c + (size_t)o * c + c0 + j] = v;
        }
    }
}

inline bool dims_ok(int b, int o, int c, long n) { return b > 0 && b <= 65535 && o >= 1 && o <= 4 && c >= 1 && c <= MAXC && n > 0 && (n & 3) == 0; }

inline int dw_slabs(long n) {
    const long per = (long)TB * 4 * 8;                   // at least eight steps of the block per slab
    long s = (n + per - 1) / per;
    return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

template <typename F>
int by_rows(int o, F &&f) {
    switch (o) {
        case 1: return f(std::integral_constant<int, 1>{});
        case 2: return f(std::integral_constant<int, 2>{});
        case 3: return f(std::integral_constant<int, 3>{});
        default: return f(std::integral_constant<int, 4>{});
    }
}

}  // namespace

extern "C" int eap_narrow_contract_supported(int b, int o, int c, int64_t n) { return dims_ok(b, o, c, (long)n) ? 1 : 0; }

extern "C" int eap_narrow_contract_fwd_f32(int b, int o, int c, int64_t n, const float *W, const float *x, float *y, eap_stream_t stream) {
    if (b <= 0 || n <= 0) return 0;
    if (!dims_ok(b, o, c, (long)n)) return eap::bad_arg("narrow_contract_fwd: 1-4 output channels, at most 2048 input channels, row length a multiple of 4");
    by_rows(o, [&](auto oc) {
        hipLaunchKernelGGL(narrow_fwd_kernel<decltype(oc)::value>, dim3(eap::cdiv(n / 4, TB), b), dim3(TB), 0, eap::S(stream), c, (long)n, W, x, y);
        return 0;
    });
    return eap::check_launch("narrow_contract_fwd");
}

extern "C" int eap_narrow_contract_dx_f32(int b, int o, int c, int64_t n, const float *W, const float *g, float *dx, eap_stream_t stream) {
    if (b <= 0 || n <= 0) return 0;
    if (!dims_ok(b, o, c, (long)n)) return eap::bad_arg("narrow_contract_dx: 1-4 output channels, at most 2048 input channels, row length a multiple of 4");
    by_rows(o, [&](auto oc) {
        hipLaunchKernelGGL(narrow_dx_kernel<decltype(oc)::value>, dim3(eap::cdiv(n / 4, TB), b), dim3(TB), 0, eap::S(stream), c, (long)n, W, g, dx);
        return 0;
    });
    return eap::check_launch("narrow_contract_dx");
}

// number of [b][o][c] partials eap_narrow_contract_dw_f32 writes (the caller sums them in order)
extern "C" int eap_narrow_contract_dw_slabs(int64_t n) { return n > 0 ? dw_slabs((long)n) : 0; }

extern "C" int eap_narrow_contract_dw_f32(int b, int o, int c, int64_t n, const float *g, const float *x, float *partial, eap_stream_t stream) {
    if (b <= 0 || n <= 0) return 0;
    if (!dims_ok(b, o, c, (long)n)) return eap::bad_arg("narrow_contract_dw: 1-4 output channels, at most 2048 input channels, row length a multiple of 4");
    const int slabs = dw_slabs((long)n);
    long slab_cols = ((long)n + slabs - 1) / slabs;
    slab_cols = (slab_cols + TB * 4 - 1) / (TB * 4) * (TB * 4);
    by_rows(o, [&](auto oc) {
        hipLaunchKernelGGL(narrow_dw_kernel<decltype(oc)::value>, dim3(slabs, (c + CS - 1) / CS, b), dim3(TB), 0, eap::S(stream), c, (long)n,
                           slab_cols, g, x, partial);
        return 0;
    });
    return eap::check_launch("narrow_contract_dw");
}
