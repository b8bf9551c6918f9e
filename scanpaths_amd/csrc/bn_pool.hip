// HBM-bound NHWC kernels of the encoder: BatchNorm2d (train/eval, +ReLU, +residual), column/row reductions,
// MaxPool(3,2,ceil), layout packing.  All reductions are two-stage with a fixed order (deterministic) and
// accumulate in fp64 so E[x^2]-E[x]^2 does not cancel (the reference uses Welford-style stats in torch).
// Access pattern: channels are the contiguous axis, every thread moves float4 (16 B/lane), a 16-lane group
// covers 256 contiguous bytes of one pixel row.
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ int64_t sp_cdiv_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int CB = 64;        // channels per block column (16 float4 lanes)
constexpr int RL = 16;        // row lanes per block (256 threads = 16 x 16)
constexpr int MAX_G = 256;    // row-slab count upper bound

// slab count of the wide (256 channels per block column) backward reduction: more, shorter slabs keep >= 4 blocks per CU
inline int pick_G_wide(int64_t M, int C) {
    const int64_t colblocks = sp_cdiv(C, 256);
    int64_t g = sp_cdiv(1024, colblocks);
    g = std::min<int64_t>(g, sp_cdiv(M, 4 * 8));
    g = std::min<int64_t>(g, 1024);
    return (int)std::max<int64_t>(g, 1);
}

inline int pick_G(int64_t M, int C) {
    const int64_t colblocks = sp_cdiv(C, CB);
    int64_t g = sp_cdiv(2048, colblocks);              // ~8 blocks per CU overall
    g = std::min<int64_t>(g, sp_cdiv(M, RL * 8));      // at least 8 rows per thread
    g = std::min<int64_t>(g, MAX_G);
    return (int)std::max<int64_t>(g, 1);
}

// Generic column reduction: for every channel c accumulate NV values produced by f(row, c4) over a row slab.
// partial layout: [G][NV][C] doubles.
template <int NV, typename F>
__device__ __forceinline__ void col_reduce(int64_t M, int C, int G, double* partial, F f) {
    __shared__ double sh[NV][RL][CB + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * CB + tx * 4;
    const int g = blockIdx.y;
    const int64_t rows_per = sp_cdiv_dev(M, G);
    const int64_t r0 = (int64_t)g * rows_per, r1 = min(M, r0 + rows_per);
    double acc[NV][4];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[v][k] = 0.0;
    if (c < C) {
        for (int64_t r = r0 + ty; r < r1; r += RL) {
            float4 vals[NV];
            f(r, c, vals);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                acc[v][0] += (double)vals[v].x;
                acc[v][1] += (double)vals[v].y;
                acc[v][2] += (double)vals[v].z;
                acc[v][3] += (double)vals[v].w;
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int k = 0; k < 4; ++k) sh[v][ty][tx * 4 + k] = acc[v][k];
    __syncthreads();
    for (int i = threadIdx.x; i < NV * CB; i += 256) {
        const int v = i / CB, cc = i % CB;
        double s = 0.0;
#pragma unroll
        for (int y = 0; y < RL; ++y) s += sh[v][y][cc];
        const int cg = blockIdx.x * CB + cc;
        if (cg < C) partial[((int64_t)g * NV + v) * C + cg] = s;
    }
}

// ---------------------------------------------------------------- BN statistics
__global__ __launch_bounds__(256) void bn_stats_partial(const float* x, int64_t M, int C, int G, double* partial) {
    col_reduce<2>(M, C, G, partial, [&](int64_t r, int c, float4* v) {
        const float4 a = *reinterpret_cast<const float4*>(x + r * C + c);
        v[0] = a;
        v[1] = make_float4(a.x * a.x, a.y * a.y, a.z * a.z, a.w * a.w);
    });
}

// Second stage of the column reductions: block = 32 channels x 8 slab lanes; each thread adds every 8th slab partial
// (independent loads, G/8 deep instead of a G-deep dependent chain: 66 us -> ~8 us per call, ~110 calls per train step), the
// 8 lane sums are combined in fixed order through LDS -> bitwise reproducible.  NV values per channel, layout [G][NV][C].
// FC channels x FL slab lanes per block (FC * FL = 256): 32 x 8 for the <= 256 slabs of the column-reduction kernels, 8 x 32 for
// the per-M-tile partials a conv epilogue leaves (up to M / 256 of them: 1280 at the 80x128 maps)
constexpr int FIN_C = 32, FIN_L = 8;
template <int NV, int FC = FIN_C, int FL = FIN_L>
__device__ __forceinline__ bool final_reduce(const double* __restrict__ partial, int C, int G, int& c, double (&out)[NV]) {
    __shared__ double shf[NV][FL][FC];
    const int tx = threadIdx.x & (FC - 1), ty = threadIdx.x / FC;
    c = blockIdx.x * FC + tx;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    if (c < C) {
        // four slab rows per round: their loads are independent and go out together (one at a time, every add waited for its own
        // load: a chain of G / FL global round trips, 17-24 us per launch for a few MB); the adds keep their order -> same bits
        int g = ty;
        for (; g + 3 * FL < G; g += 4 * FL) {
            double t[4][NV];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) t[u][v] = partial[((int64_t)(g + u * FL) * NV + v) * C + c];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += t[u][v];
        }
        for (; g < G; g += FL)
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += partial[((int64_t)g * NV + v) * C + c];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) shf[v][ty][tx] = acc[v];
    __syncthreads();
    if (ty != 0 || c >= C) return false;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l < FL; ++l) s += shf[v][l][tx];
        out[v] = s;
    }
    return true;
}

__global__ __launch_bounds__(FIN_C * FIN_L) void bn_stats_final(const double* partial, int64_t M, int C, int G, float eps,
                                                                 float momentum, float* mean, float* invstd, float* rmean,
                                                                 float* rvar) {
    int c;
    double r[2];
    if (!final_reduce<2>(partial, C, G, c, r)) return;
    const double s = r[0], q = r[1];
    const double mu = s / (double)M;
    double var = q / (double)M - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

__global__ void bn_eval_stats(const float* rmean, const float* rvar, int C, float eps, float* mean, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rmean[c];
    invstd[c] = 1.f / sqrtf(rvar[c] + eps);
}

// ---------------------------------------------------------------- BN apply (+res, +relu)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* x, const float* mean, const float* invstd,
                                                       const float* gamma, const float* beta, const float* res,
                                                       int relu, int64_t n4, int C, float* y, unsigned* amax) {
    __shared__ float sh4[4];
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % C);
        const float4 a = reinterpret_cast<const float4*>(x)[i];
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(beta + c);
        float4 o;
        o.x = (a.x - mu.x) * is.x * ga.x + be.x;
        o.y = (a.y - mu.y) * is.y * ga.y + be.y;
        o.z = (a.z - mu.z) * is.z * ga.z + be.z;
        o.w = (a.w - mu.w) * is.w * ga.w + be.w;
        if (res) {
            const float4 rr = reinterpret_cast<const float4*>(res)[i];
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        if (relu) {
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        reinterpret_cast<float4*>(y)[i] = o;
        mx = amax4(mx, o.x, o.y, o.z, o.w);
    }
    if (amax) block_amax_commit(mx, amax, sh4);
}

// ---------------------------------------------------------------- BN backward
__global__ __launch_bounds__(256) void bn_bwd_partial(const float* dy, const float* x, const float* y, const float* mean,
                                                      const float* invstd, int relu, int64_t M, int C, int G,
                                                      double* partial) {
    col_reduce<2>(M, C, G, partial, [&](int64_t r, int c, float4* v) {
        float4 d = *reinterpret_cast<const float4*>(dy + r * C + c);
        const float4 a = *reinterpret_cast<const float4*>(x + r * C + c);
        if (relu) {
            const float4 yy = *reinterpret_cast<const float4*>(y + r * C + c);
            d.x = yy.x > 0.f ? d.x : 0.f; d.y = yy.y > 0.f ? d.y : 0.f;
            d.z = yy.z > 0.f ? d.z : 0.f; d.w = yy.w > 0.f ? d.w : 0.f;
        }
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        v[0] = d;
        v[1] = make_float4(d.x * (a.x - mu.x) * is.x, d.y * (a.y - mu.y) * is.y, d.z * (a.z - mu.z) * is.z,
                           d.w * (a.w - mu.w) * is.w);
    });
}

// coef[0][c] = sum dy_eff / M ; coef[1][c] = sum dy_eff*xhat / M
__global__ __launch_bounds__(FIN_C * FIN_L) void bn_bwd_final(const double* partial, int64_t M, int C, int G, float* dgamma,
                                                               float* dbeta, float* coef) {
    int c;
    double r[2];
    if (!final_reduce<2>(partial, C, G, c, r)) return;
    const double s1 = r[0], s2 = r[1];
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
    coef[c] = (float)(s1 / (double)M);
    coef[C + c] = (float)(s2 / (double)M);
}

__global__ __launch_bounds__(256) void bn_bwd_apply(const float* dy, const float* x, const float* y, const float* mean,
                                                    const float* invstd, const float* gamma, const float* coef,
                                                    int relu, int training, int64_t n4, int C, float* dx, float* dres,
                                                    unsigned* amax) {
    __shared__ float sh4[4];
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % C);
        float4 d = reinterpret_cast<const float4*>(dy)[i];
        if (relu) {
            const float4 yy = reinterpret_cast<const float4*>(y)[i];
            d.x = yy.x > 0.f ? d.x : 0.f; d.y = yy.y > 0.f ? d.y : 0.f;
            d.z = yy.z > 0.f ? d.z : 0.f; d.w = yy.w > 0.f ? d.w : 0.f;
        }
        if (dres) reinterpret_cast<float4*>(dres)[i] = d;
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
        float4 o;
        if (training) {
            const float4 a = reinterpret_cast<const float4*>(x)[i];
            const float4 mu = *reinterpret_cast<const float4*>(mean + c);
            const float4 k1 = *reinterpret_cast<const float4*>(coef + c);
            const float4 k2 = *reinterpret_cast<const float4*>(coef + C + c);
            o.x = ga.x * is.x * (d.x - k1.x - (a.x - mu.x) * is.x * k2.x);
            o.y = ga.y * is.y * (d.y - k1.y - (a.y - mu.y) * is.y * k2.y);
            o.z = ga.z * is.z * (d.z - k1.z - (a.z - mu.z) * is.z * k2.z);
            o.w = ga.w * is.w * (d.w - k1.w - (a.w - mu.w) * is.w * k2.w);
        } else {
            o.x = ga.x * is.x * d.x; o.y = ga.y * is.y * d.y; o.z = ga.z * is.z * d.z; o.w = ga.w * is.w * d.w;
        }
        reinterpret_cast<float4*>(dx)[i] = o;
        mx = amax4(mx, o.x, o.y, o.z, o.w);
    }
    if (amax) block_amax_commit(mx, amax, sh4);
}


// ================================================================================================================
// BatchNorm that EMITS the 2xfp16 split operand of its consumer conv (train mode; conv_f16x2.hip layout [row][C/16][2][16]).
// The operand scale needs an upper bound of max|output| BEFORE the apply pass.  It does not have to be tight: a power-of-two scale
// s with bound * s in [8192, 16384) keeps 22 significant bits for every element within 2^-11 of the bound and an absolute error of
// 2^-38 * bound below that, so a bound that is a few times too large costs nothing measurable -- it only has to be >= the maximum
// (fp16 overflow).  Forward: the statistics pass also tracks min / max of x per channel; the output of the per-channel affine map
// (+ReLU) is then bounded exactly, a residual adds its own max|.|.  Backward: |dx| <= |gamma*invstd| * (max|d| + |k1| + |k2| *
// max|xhat|) per channel.  Saved traffic: the consumer's split pass no longer re-reads the tensor (4 B/element forward and
// backward), and the ReLU mask is kept as one bit per element (the backward passes read 1/32 of a word instead of the fp32
// output, twice).  Mask layout: 64 consecutive float4 (= one wave iteration) -> 4 uint64 ballots (x, y, z, w components).
__device__ __forceinline__ float bn_affine(float a, float mu, float is, float ga, float be) {
    return __fmaf_rn(__fmul_rn(__fsub_rn(a, mu), is), ga, be);
}

__global__ __launch_bounds__(256) void bn_stats_mm_partial(const float* __restrict__ x, int64_t M, int C, int G,
                                                           double* __restrict__ partial, float* __restrict__ mm) {
    __shared__ double sh[2][RL][CB + 1];
    __shared__ float shm[2][RL][CB + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int c = blockIdx.x * CB + tx * 4;
    const int g = blockIdx.y;
    const int64_t rows_per = sp_cdiv_dev(M, G);
    const int64_t r0 = (int64_t)g * rows_per, r1 = min(M, r0 + rows_per);
    double s[4] = {0.0, 0.0, 0.0, 0.0}, q[4] = {0.0, 0.0, 0.0, 0.0};
    float mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (c < C)
        for (int64_t r = r0 + ty; r < r1; r += RL) {
            const float4 a = *reinterpret_cast<const float4*>(x + r * C + c);
            const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k] += (double)v[k];
                q[k] += (double)(v[k] * v[k]);
                mn[k] = fminf(mn[k], v[k]);
                mx[k] = fmaxf(mx[k], v[k]);
            }
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sh[0][ty][tx * 4 + k] = s[k];
        sh[1][ty][tx * 4 + k] = q[k];
        shm[0][ty][tx * 4 + k] = mn[k];
        shm[1][ty][tx * 4 + k] = mx[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * CB; i += 256) {
        const int v = i / CB, cc = i % CB;
        double a = 0.0;
        float m = v == 0 ? INFINITY : -INFINITY;
#pragma unroll
        for (int y = 0; y < RL; ++y) {
            a += sh[v][y][cc];
            m = v == 0 ? fminf(m, shm[0][y][cc]) : fmaxf(m, shm[1][y][cc]);
        }
        const int cg = blockIdx.x * CB + cc;
        if (cg < C) {
            partial[((int64_t)g * 2 + v) * C + cg] = a;
            mm[((int64_t)g * 2 + v) * C + cg] = m;
        }
    }
}

// extrema of [G][2][C] float partials (min in slot 0, max in slot 1); valid where final_reduce returned true
template <int FC = FIN_C, int FL = FIN_L>
__device__ __forceinline__ void final_minmax(const float* __restrict__ mm, int C, int G, float& mn, float& mx) {
    __shared__ float shx[2][FL][FC];
    const int tx = threadIdx.x & (FC - 1), ty = threadIdx.x / FC;
    const int c = blockIdx.x * FC + tx;
    float a = INFINITY, b = -INFINITY;
    if (c < C) {
        int g = ty;
        for (; g + 3 * FL < G; g += 4 * FL) {          // (four rows per round, as final_reduce)
            float ta[4], tb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ta[u] = mm[((int64_t)(g + u * FL) * 2 + 0) * C + c];
                tb[u] = mm[((int64_t)(g + u * FL) * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a = fminf(a, ta[u]);
                b = fmaxf(b, tb[u]);
            }
        }
        for (; g < G; g += FL) {
            a = fminf(a, mm[((int64_t)g * 2 + 0) * C + c]);
            b = fmaxf(b, mm[((int64_t)g * 2 + 1) * C + c]);
        }
    }
    shx[0][ty][tx] = a;
    shx[1][ty][tx] = b;
    __syncthreads();
    mn = INFINITY;
    mx = -INFINITY;
#pragma unroll
    for (int l = 0; l < FL; ++l) {
        mn = fminf(mn, shx[0][l][tx]);
        mx = fmaxf(mx, shx[1][l][tx]);
    }
}

// block maximum of a per-thread bound -> one atomicMax (order-free: the result does not depend on the schedule)
__device__ __forceinline__ void commit_bound(float b, unsigned* slot) {
    __shared__ float shb[FIN_C * FIN_L / 64];
    b = wave_max(b);
    if ((threadIdx.x & 63) == 0) shb[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int w = 0; w < FIN_C * FIN_L / 64; ++w) m = fmaxf(m, shb[w]);
        if (m > 0.f) atomicMax(slot, __float_as_uint(m));
    }
}

template <int FC, int FL>
__global__ __launch_bounds__(256) void bn_stats_mm_final(const double* partial, const float* mm, int64_t M, int C, int G,
                                                         float eps, float momentum, const float* gamma, const float* beta, int relu,
                                                         float* mean, float* invstd, float* rmean, float* rvar, float* ext,
                                                         unsigned* bound) {
    int c;
    double r[2];
    float mn, mx;
    final_minmax<FC, FL>(mm, C, G, mn, mx);
    const bool own = final_reduce<2, FC, FL>(partial, C, G, c, r);
    float bnd = 0.f;
    if (own) {
        const double mu = r[0] / (double)M;
        double var = r[1] / (double)M - mu * mu;
        if (var < 0.0) var = 0.0;
        const float muf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
        mean[c] = muf;
        invstd[c] = isf;
        if (rmean) {
            const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * muf;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
        ext[c] = mn;
        ext[C + c] = mx;
        // the affine map is monotone per channel: its extrema sit at the extrema of x (evaluated with the apply pass's own formula)
        const float v1 = bn_affine(mn, muf, isf, gamma[c], beta[c]), v2 = bn_affine(mx, muf, isf, gamma[c], beta[c]);
        bnd = relu ? fmaxf(fmaxf(v1, v2), 0.f) : fmaxf(fabsf(v1), fabsf(v2));
    }
    commit_bound(bnd, bound);
}

// y = relu?(bn(x) + res): optional fp32 output, split operand, optional ReLU bit mask.  scale from the bound (+ max|res|).
__global__ __launch_bounds__(256) void bn_apply_split_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ res,
                                                             const unsigned* __restrict__ bound, const unsigned* __restrict__ res_amax,
                                                             int relu, int64_t n4, int C, float* __restrict__ y,
                                                             uint16_t* __restrict__ planes, float* __restrict__ y_scale,
                                                             unsigned long long* __restrict__ mask) {
    const float total = __uint_as_float(*bound) + (res_amax ? __uint_as_float(*res_amax) : 0.f);
    const float s = scale_of(__float_as_uint(total));
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // when the grid stride is a multiple of the row length (in float4) a thread meets the same four channels in every iteration:
    // their parameters are loaded once (four 16-byte parameter loads per 16 bytes of data kept the L1 path, not HBM, busy)
    const bool hoist = stride % (C / 4) == 0;
    float4 mu, is, ga, be;
    if (hoist) {
        const int c = (int)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4) % C);
        mu = *reinterpret_cast<const float4*>(mean + c);
        is = *reinterpret_cast<const float4*>(invstd + c);
        ga = *reinterpret_cast<const float4*>(gamma + c);
        be = *reinterpret_cast<const float4*>(beta + c);
    }
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n4; base += stride) {
        const int64_t i = base + lane;
        const bool live = i < n4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            const float4 a = reinterpret_cast<const float4*>(x)[i];
            if (!hoist) {
                const int c = (int)((i * 4) % C);
                mu = *reinterpret_cast<const float4*>(mean + c);
                is = *reinterpret_cast<const float4*>(invstd + c);
                ga = *reinterpret_cast<const float4*>(gamma + c);
                be = *reinterpret_cast<const float4*>(beta + c);
            }
            o.x = bn_affine(a.x, mu.x, is.x, ga.x, be.x);
            o.y = bn_affine(a.y, mu.y, is.y, ga.y, be.y);
            o.z = bn_affine(a.z, mu.z, is.z, ga.z, be.z);
            o.w = bn_affine(a.w, mu.w, is.w, ga.w, be.w);
            if (res) {
                const float4 rr = reinterpret_cast<const float4*>(res)[i];
                o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
            }
        }
        if (relu) {
            if (mask) {
                const unsigned long long b0 = __ballot(live && o.x > 0.f), b1 = __ballot(live && o.y > 0.f);
                const unsigned long long b2 = __ballot(live && o.z > 0.f), b3 = __ballot(live && o.w > 0.f);
                if (lane < 4) mask[(base >> 6) * 4 + lane] = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
            }
            o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        }
        if (live && y) reinterpret_cast<float4*>(y)[i] = o;
        if (planes) {                  // (wave-uniform)
            ushort4 pa, pb;
            split2(o.x, s, pa.x, pb.x);
            split2(o.y, s, pa.y, pb.y);
            split2(o.z, s, pa.z, pb.z);
            split2(o.w, s, pa.w, pb.w);
            store_planes_quad(planes, i, live, pa, pb);
        }
    }
    if (planes && blockIdx.x == 0 && threadIdx.x < 8) reinterpret_cast<uint2*>(planes + 8 * n4)[threadIdx.x] = make_uint2(0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        y_scale[0] = s;
        y_scale[1] = total;          // an upper bound of max|y| (what a later residual add / re-split may rely on)
    }
}

__device__ __forceinline__ float4 mask_select(const unsigned long long* __restrict__ mask, int64_t i, float4 d) {
    const unsigned long long* w = mask + (i >> 6) * 4;
    const int b = (int)(i & 63);
    d.x = ((w[0] >> b) & 1ull) ? d.x : 0.f;
    d.y = ((w[1] >> b) & 1ull) ? d.y : 0.f;
    d.z = ((w[2] >> b) & 1ull) ? d.z : 0.f;
    d.w = ((w[3] >> b) & 1ull) ? d.w : 0.f;
    return d;
}

// TX lanes along the channels (x 4 channels each), 256 / TX row lanes: 16 x 16 for narrow maps, 64 x 4 for C >= 256 (a wave then
// reads 1 KB contiguous per row instead of four 256-byte pieces)
template <int TX>
__global__ __launch_bounds__(256) void bn_bwd_mm_partial(const float* __restrict__ dy, const float* __restrict__ x,
                                                         const unsigned long long* __restrict__ mask,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         int64_t M, int C, int G, double* __restrict__ partial,
                                                         float* __restrict__ dmax) {
    constexpr int CBW = TX * 4, RLW = 256 / TX;
    __shared__ double sh[2][RLW][CBW + 1];
    __shared__ float shm[RLW][CBW + 1];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = blockIdx.x * CBW + tx * 4;
    const int g = blockIdx.y;
    const int64_t rows_per = sp_cdiv_dev(M, G);
    const int64_t r0 = (int64_t)g * rows_per, r1 = min(M, r0 + rows_per);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    float mx[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        const float m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
        for (int64_t r = r0 + ty; r < r1; r += RLW) {
            float4 d = *reinterpret_cast<const float4*>(dy + r * C + c);
            const float4 a = *reinterpret_cast<const float4*>(x + r * C + c);
            if (mask) d = mask_select(mask, (r * C + c) >> 2, d);
            const float dv[4] = {d.x, d.y, d.z, d.w}, av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s1[k] += (double)dv[k];
                s2[k] += (double)(dv[k] * (av[k] - m4[k]) * i4[k]);
                mx[k] = fmaxf(mx[k], fabsf(dv[k]));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sh[0][ty][tx * 4 + k] = s1[k];
        sh[1][ty][tx * 4 + k] = s2[k];
        shm[ty][tx * 4 + k] = mx[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * CBW; i += 256) {
        const int v = i / CBW, cc = i % CBW;
        double a = 0.0;
        float m = 0.f;
#pragma unroll
        for (int y = 0; y < RLW; ++y) {
            a += sh[v][y][cc];
            m = fmaxf(m, shm[y][cc]);
        }
        const int cg = blockIdx.x * CBW + cc;
        if (cg < C) {
            partial[((int64_t)g * 2 + v) * C + cg] = a;
            if (v == 0) dmax[(int64_t)g * C + cg] = m;
        }
    }
}

template <int FC, int FL>
__global__ __launch_bounds__(256) void bn_bwd_mm_final(const double* partial, const float* dmax, int64_t M, int C, int G,
                                                       const float* gamma, const float* mean, const float* invstd,
                                                       const float* ext, float* dgamma, float* dbeta, float* coef, unsigned* bound) {
    __shared__ float shd[FL][FC];
    const int tx = threadIdx.x & (FC - 1), ty = threadIdx.x / FC;
    const int cc = blockIdx.x * FC + tx;
    float dm = 0.f;
    if (cc < C)
        for (int g = ty; g < G; g += FL) dm = fmaxf(dm, dmax[(int64_t)g * C + cc]);
    shd[ty][tx] = dm;
    int c;
    double r[2];
    const bool own = final_reduce<2, FC, FL>(partial, C, G, c, r);      // (contains the __syncthreads that publishes shd)
    float bnd = 0.f;
    if (own) {
        dm = 0.f;
#pragma unroll
        for (int l = 0; l < FL; ++l) dm = fmaxf(dm, shd[l][tx]);
        dbeta[c] = (float)r[0];
        dgamma[c] = (float)r[1];
        const float k1 = (float)(r[0] / (double)M), k2 = (float)(r[1] / (double)M);
        coef[c] = k1;
        coef[C + c] = k2;
        const float xh = fmaxf(fabsf(ext[c] - mean[c]), fabsf(ext[C + c] - mean[c])) * invstd[c];
        bnd = fabsf(gamma[c] * invstd[c]) * (dm + fabsf(k1) + fabsf(k2) * xh) * 1.0001f;
    }
    commit_bound(bnd, bound);
}

__global__ __launch_bounds__(256) void bn_bwd_apply_split_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                 const unsigned long long* __restrict__ mask,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ coef,
                                                                 const unsigned* __restrict__ bound, int64_t n4, int C,
                                                                 float* __restrict__ dx, float* __restrict__ dres,
                                                                 uint16_t* __restrict__ planes, float* __restrict__ dx_scale) {
    const float s = scale_of(*bound);
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool hoist = stride % (C / 4) == 0;            // see bn_apply_split_kernel
    float4 is, ga, mu, k1, k2;
    if (hoist) {
        const int c = (int)((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4) % C);
        is = *reinterpret_cast<const float4*>(invstd + c);
        ga = *reinterpret_cast<const float4*>(gamma + c);
        mu = *reinterpret_cast<const float4*>(mean + c);
        k1 = *reinterpret_cast<const float4*>(coef + c);
        k2 = *reinterpret_cast<const float4*>(coef + C + c);
    }
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n4; base += stride) {      // wave-uniform
        const int64_t i = base + lane;
        const bool live = i < n4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            float4 d = reinterpret_cast<const float4*>(dy)[i];
            if (mask) d = mask_select(mask, i, d);
            if (dres) reinterpret_cast<float4*>(dres)[i] = d;
            const float4 a = reinterpret_cast<const float4*>(x)[i];
            if (!hoist) {
                const int c = (int)((i * 4) % C);
                is = *reinterpret_cast<const float4*>(invstd + c);
                ga = *reinterpret_cast<const float4*>(gamma + c);
                mu = *reinterpret_cast<const float4*>(mean + c);
                k1 = *reinterpret_cast<const float4*>(coef + c);
                k2 = *reinterpret_cast<const float4*>(coef + C + c);
            }
            o.x = ga.x * is.x * (d.x - k1.x - (a.x - mu.x) * is.x * k2.x);
            o.y = ga.y * is.y * (d.y - k1.y - (a.y - mu.y) * is.y * k2.y);
            o.z = ga.z * is.z * (d.z - k1.z - (a.z - mu.z) * is.z * k2.z);
            o.w = ga.w * is.w * (d.w - k1.w - (a.w - mu.w) * is.w * k2.w);
            if (dx) reinterpret_cast<float4*>(dx)[i] = o;
        }
        if (planes) {                  // (wave-uniform)
            ushort4 pa, pb;
            split2(o.x, s, pa.x, pb.x);
            split2(o.y, s, pa.y, pb.y);
            split2(o.z, s, pa.z, pb.z);
            split2(o.w, s, pa.w, pb.w);
            store_planes_quad(planes, i, live, pa, pb);
        }
    }
    if (planes && blockIdx.x == 0 && threadIdx.x < 8) reinterpret_cast<uint2*>(planes + 8 * n4)[threadIdx.x] = make_uint2(0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        dx_scale[0] = s;
        dx_scale[1] = __uint_as_float(*bound);
    }
}

// ---------------------------------------------------------------- column sums / row sums
__global__ __launch_bounds__(256) void colsum_partial(const float* x, int64_t M, int C, int ld, int G, double* partial) {
    col_reduce<1>(M, C, G, partial, [&](int64_t r, int c, float4* v) {
        v[0] = *reinterpret_cast<const float4*>(x + r * ld + c);
    });
}
__global__ __launch_bounds__(FIN_C * FIN_L) void colsum_final(const double* partial, int C, int G, float* out, int beta) {
    int c;
    double r[1];
    if (!final_reduce<1>(partial, C, G, c, r)) return;
    const double s = r[0];
    out[c] = (beta ? out[c] : 0.f) + (float)s;
}

// Few rows (the decode loop's bias / vector gradients: M = batch or streams x batch rows, 228 partial + final launch pairs per training
// step): ONE launch -- a thread owns four columns and adds the M rows in row order in fp64 (as the two-stage form accumulates: the
// fp32 result is the correctly rounded sum either way), four row loads in flight.
constexpr int COLSUM_SMALL_M = 128;
__global__ __launch_bounds__(64) void colsum_small_kernel(const float* __restrict__ x, int M, int C, int ld, float* __restrict__ out, int beta) {
    const int c = ((int)blockIdx.x * 64 + (int)threadIdx.x) * 4;
    if (c >= C) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int r = 0;
    for (; r + 3 < M; r += 4) {
        const float4 v0 = *reinterpret_cast<const float4*>(x + (int64_t)r * ld + c), v1 = *reinterpret_cast<const float4*>(x + (int64_t)(r + 1) * ld + c);
        const float4 v2 = *reinterpret_cast<const float4*>(x + (int64_t)(r + 2) * ld + c), v3 = *reinterpret_cast<const float4*>(x + (int64_t)(r + 3) * ld + c);
        a0 += (double)v0.x; a1 += (double)v0.y; a2 += (double)v0.z; a3 += (double)v0.w;
        a0 += (double)v1.x; a1 += (double)v1.y; a2 += (double)v1.z; a3 += (double)v1.w;
        a0 += (double)v2.x; a1 += (double)v2.y; a2 += (double)v2.z; a3 += (double)v2.w;
        a0 += (double)v3.x; a1 += (double)v3.y; a2 += (double)v3.z; a3 += (double)v3.w;
    }
    for (; r < M; ++r) {
        const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)r * ld + c);
        a0 += (double)v.x; a1 += (double)v.y; a2 += (double)v.z; a3 += (double)v.w;
    }
    float4 o = beta ? *reinterpret_cast<const float4*>(out + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    o.x += (float)a0; o.y += (float)a1; o.z += (float)a2; o.w += (float)a3;
    *reinterpret_cast<float4*>(out + c) = o;
}

// one wave per row
__global__ __launch_bounds__(256) void rowsum_kernel(const float* x, int64_t M, int C, float scale, float* out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave0; r < M; r += nwaves) {
        float s = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 a = *reinterpret_cast<const float4*>(x + r * C + c);
            s += (a.x + a.y) + (a.z + a.w);
        }
        s = wave_sum(s);
        if (lane == 0) out[r] = s * scale;
    }
}
__global__ __launch_bounds__(256) void rowsum_bwd_kernel(const float* dout, int64_t n4, int C, float scale, float* dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = (i * 4) / C;
        const float g = dout[r] * scale;
        float4 a = reinterpret_cast<float4*>(dx)[i];
        a.x += g; a.y += g; a.z += g; a.w += g;
        reinterpret_cast<float4*>(dx)[i] = a;
    }
}

// ---------------------------------------------------------------- maxpool 3x3 s2 ceil, NHWC
// argmax rule = first maximum in row-major window order (strict >), as ATen's CPU max_pool2d.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* x, int N, int H, int W, int C, float* y, int Ho,
                                                          int Wo, unsigned char* amx) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int xo = (int)(r % Wo); r /= Wo;
        const int yo = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 am = make_uchar4(0, 0, 0, 0);                  // window position (ky*3 + kx) of the FIRST maximum per channel
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = yo * 2 + ky;
            if (iy >= H) break;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = xo * 2 + kx;
                if (ix >= W) break;
                const float4 a = *reinterpret_cast<const float4*>(x + (((int64_t)n * H + iy) * W + ix) * C + c4 * 4);
                const unsigned char q = (unsigned char)(ky * 3 + kx);
                if (a.x > m.x) { m.x = a.x; am.x = q; }
                if (a.y > m.y) { m.y = a.y; am.y = q; }
                if (a.z > m.z) { m.z = a.z; am.z = q; }
                if (a.w > m.w) { m.w = a.w; am.w = q; }
            }
        }
        reinterpret_cast<float4*>(y)[i] = m;
        if (amx) reinterpret_cast<uchar4*>(amx)[i] = am;
    }
}

// gather form (no atomics): each input element sums dy of the windows whose FIRST maximum it is.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* dy, const float* x, const float* y, int N, int H,
                                                          int W, int C, float* dx, int Ho, int Wo) {
    const int64_t total = (int64_t)N * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int n = (int)(r / H);
        const float v = x[i];
        float g = 0.f;
        const int oy_lo = max(0, (iy - 1) / 2), oy_hi = min(Ho - 1, iy / 2);   // windows with 2*oy <= iy <= 2*oy+2
        const int ox_lo = max(0, (ix - 1) / 2), ox_hi = min(Wo - 1, ix / 2);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (iy - 2 * oy > 2) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (ix - 2 * ox > 2) continue;
                const int64_t o = (((int64_t)n * Ho + oy) * Wo + ox) * C + c;
                if (y[o] != v) continue;
                // am I the first element equal to the max in this window (row-major scan, strict >)?
                const int mine = (iy - 2 * oy) * 3 + (ix - 2 * ox);
                bool first = true;
                for (int q = 0; q < mine; ++q) {
                    const int yy = oy * 2 + q / 3, xx = ox * 2 + q % 3;
                    if (yy < H && xx < W && x[(((int64_t)n * H + yy) * W + xx) * C + c] == v) { first = false; break; }
                }
                if (first) g += dy[o];
            }
        }
        dx[i] = g;
    }
}

// the same with the forward's saved window positions: no value comparisons, no reads of x / y; float4 over the channels
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const float* dy, const unsigned char* amx, int N, int H, int W,
                                                              int C, float* dx, int Ho, int Wo) {
    const int C4 = C / 4;
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int ix = (int)(r % W); r /= W;
        const int iy = (int)(r % H);
        const int n = (int)(r / H);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        const int oy_lo = max(0, (iy - 1) / 2), oy_hi = min(Ho - 1, iy / 2);
        const int ox_lo = max(0, (ix - 1) / 2), ox_hi = min(Wo - 1, ix / 2);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (iy - 2 * oy > 2) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (ix - 2 * ox > 2) continue;
                const int64_t o = (((int64_t)n * Ho + oy) * Wo + ox) * C4 + c4;
                const uchar4 a = reinterpret_cast<const uchar4*>(amx)[o];
                const unsigned char mine = (unsigned char)((iy - 2 * oy) * 3 + (ix - 2 * ox));
                if (a.x == mine || a.y == mine || a.z == mine || a.w == mine) {
                    const float4 d = reinterpret_cast<const float4*>(dy)[o];
                    g.x += a.x == mine ? d.x : 0.f;
                    g.y += a.y == mine ? d.y : 0.f;
                    g.z += a.z == mine ? d.z : 0.f;
                    g.w += a.w == mine ? d.w : 0.f;
                }
            }
        }
        reinterpret_cast<float4*>(dx)[i] = g;
    }
}

// ---------------------------------------------------------------- layout helpers
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* x, int N, int C, int H, int W, int Cp, float* y) {
    const int64_t total = (int64_t)N * H * W * Cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        int64_t r = i / Cp;
        const int w = (int)(r % W); r /= W;
        const int hh = (int)(r % H);
        const int n = (int)(r / H);
        y[i] = c < C ? x[(((int64_t)n * C + c) * H + hh) * W + w] : 0.f;
    }
}
__global__ __launch_bounds__(256) void pad_lastdim_kernel(const float* x, int64_t rows, int Cin, int Cout, float* y) {
    const int64_t total = rows * Cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cout);
        const int64_t r = i / Cout;
        y[i] = c < Cin ? x[r * Cin + c] : 0.f;
    }
}
__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, float* o, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
        reinterpret_cast<float4*>(o)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}
// out = sum of up to 32 equally sized tensors in list order (gradient fan-in of a tensor used by every decode step: one
// pass over T inputs instead of T-1 read-read-write adds)
struct SumList {
    const float* p[32];
    int n;
    // row sparsity (masked-step sparsity of the backward pass, see SumListMixed): term k belongs to decode step step[k] and is exactly
    // zero -- and not read -- for a sample b with row_last[b] < step[k]; quads_per_sample float4 per sample; row_last NULL: dense
    const int* row_last;
    int step[32];
    int64_t quads_per_sample;
};
__global__ __launch_bounds__(256) void sum_n_rows_kernel(SumList l, float* o, int64_t n4, unsigned* amax) {
    __shared__ float sh4[4];
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int lastb = l.row_last[i / l.quads_per_sample];
        double ax = 0.0, ay = 0.0, az = 0.0, aw = 0.0;      // (fp64 accumulation in list order: the live terms of sum_n_kernel's sum)
        bool first = true;
        for (int k = 0; k < l.n; ++k) {
            if (l.step[k] > lastb) continue;
            const float4 v = reinterpret_cast<const float4*>(l.p[k])[i];
            if (first) { ax = v.x; ay = v.y; az = v.z; aw = v.w; first = false; }
            else { ax += (double)v.x; ay += (double)v.y; az += (double)v.z; aw += (double)v.w; }
        }
        const float4 acc = make_float4((float)ax, (float)ay, (float)az, (float)aw);
        reinterpret_cast<float4*>(o)[i] = acc;
        mx = amax4(mx, acc.x, acc.y, acc.z, acc.w);
    }
    if (amax) block_amax_commit(mx, amax, sh4);
}
__global__ __launch_bounds__(256) void sum_n_kernel(SumList l, float* o, int64_t n4, unsigned* amax) {
    __shared__ float sh4[4];
    float mx = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        // fp64 accumulation like every reduction of the path (free on an HBM-bound pass): a fan-in of T decode steps may cancel --
        // the scalar gradient of object_head.drt_layer_1.bias came out 2e-4 of itself off on EVERY back-end with fp32 sums here
        // (tests/diagnostics/drt_bias_probe.py), 50x the fp32 reference's own error
        const float4 v0 = reinterpret_cast<const float4*>(l.p[0])[i];
        double ax = v0.x, ay = v0.y, az = v0.z, aw = v0.w;
        for (int k = 1; k < l.n; ++k) {
            const float4 v = reinterpret_cast<const float4*>(l.p[k])[i];
            ax += (double)v.x; ay += (double)v.y; az += (double)v.z; aw += (double)v.w;
        }
        const float4 acc = make_float4((float)ax, (float)ay, (float)az, (float)aw);
        reinterpret_cast<float4*>(o)[i] = acc;
        mx = amax4(mx, acc.x, acc.y, acc.z, acc.w);
    }
    if (amax) block_amax_commit(mx, amax, sh4);
}
// The same fan-in when some inputs exist only as 2xfp16 split operands ([16-element group][plane][16] fp16 + device scale, as written
// by lstm_bwd_kernel): input k contributes (plane0 + plane1) / scale_k -- exactly the value the GEMMs read (fp16 + fp16 of one
// element is exact in fp32, the scale is a power of two).  One thread per 16-element group; list order.
struct SumListMixed {
    const float* f[32];          // fp32 form, or NULL
    const uint16_t* pl[32];      // split form (used when f[k] is NULL)
    const float* sc[32];
    int n;
    // row sparsity (masked-step sparsity of the backward pass): term k is the gate gradient of decode step step[k]; for a sample b with
    // row_last[b] < step[k] it is exactly zero and is not read.  groups_per_sample 16-element groups per sample; row_last NULL: dense.
    const int* row_last;
    int step[32];
    int64_t groups_per_sample;
};
__global__ __launch_bounds__(256) void sum_n_mixed_kernel(SumListMixed l, float* o, int64_t n16, unsigned* amax) {
    __shared__ float sh4[4];
    float mx = 0.f;
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        double acc[16];                      // fp64 accumulation over the (up to 32) contributions, see sum_n_kernel
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.0;
        const int lastb = l.row_last ? l.row_last[i / l.groups_per_sample] : 0x7fffffff;
        for (int k = 0; k < l.n; ++k) {
            if (l.step[k] > lastb) continue;         // an exactly-zero term
            if (l.f[k]) {
                const float4* q = reinterpret_cast<const float4*>(l.f[k]) + i * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = q[j];
                    acc[4 * j] += (double)v.x; acc[4 * j + 1] += (double)v.y; acc[4 * j + 2] += (double)v.z; acc[4 * j + 3] += (double)v.w;
                }
            } else {
                const h8* q = reinterpret_cast<const h8*>(l.pl[k]) + i * 4;        // 64 bytes: plane 0 (2 x 8 halves), plane 1
                const float inv = 1.f / l.sc[k][0];
                const h8 a0 = q[0], a1 = q[1], b0 = q[2], b1 = q[3];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    acc[e] += (double)(((float)a0[e] + (float)b0[e]) * inv);
                    acc[8 + e] += (double)(((float)a1[e] + (float)b1[e]) * inv);
                }
            }
        }
        float4* w = reinterpret_cast<float4*>(o) + i * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 r4 = make_float4((float)acc[4 * j], (float)acc[4 * j + 1], (float)acc[4 * j + 2], (float)acc[4 * j + 3]);
            w[j] = r4;
            mx = amax4(mx, r4.x, r4.y, r4.z, r4.w);
        }
    }
    if (amax) block_amax_commit(mx, amax, sh4);
}
__global__ __launch_bounds__(256) void add_tail_kernel(const float* a, const float* b, float* o, int64_t start, int64_t n) {
    const int64_t i = start + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* dy, const float* y, int64_t n, float* dx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

inline int ew_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(n, 256), 2048)); }

}  // namespace

extern "C" int64_t sp_bn_workspace(int64_t M, int C) {
    return (int64_t)pick_G(M, C) * 2 * C * (int64_t)sizeof(double) + 2 * (int64_t)C * (int64_t)sizeof(float);
}

extern "C" int sp_bn_stats(const float* x, int64_t M, int C, float eps, float momentum, float* mean, float* invstd,
                           float* running_mean, float* running_var, void* workspace, void* stream) {
    if (!x || !mean || !invstd || !workspace) return SP_ENULL;
    if (C % 4 || M <= 0) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int G = pick_G(M, C);
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(bn_stats_partial, dim3((unsigned)sp_cdiv(C, CB), G), dim3(256), 0, s, x, M, C, G, partial);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_stats_final, dim3((unsigned)sp_cdiv(C, FIN_C)), dim3(FIN_C * FIN_L), 0, s, partial, M, C, G, eps, momentum,
                       mean, invstd, running_mean, running_var);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                                float* invstd, void* stream) {
    if (!running_mean || !running_var || !mean || !invstd) return SP_ENULL;
    hipLaunchKernelGGL(bn_eval_stats, dim3((unsigned)sp_cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream, running_mean,
                       running_var, C, eps, mean, invstd);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const float* residual, int relu, int64_t M, int C, float* y, unsigned* y_amax, void* stream) {
    if (!x || !mean || !invstd || !gamma || !beta || !y) return SP_ENULL;
    if (C % 4) return SP_EINVAL;
    const int64_t n4 = M * C / 4;
    SP_RESET_AMAX(y_amax, stream);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma,
                       beta, residual, relu, n4, C, y, y_amax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_backward(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                              const float* gamma, int relu, int training, int64_t M, int C, float* dx, float* dres,
                              float* dgamma, float* dbeta, void* workspace, unsigned* dx_amax, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !dx || !dgamma || !dbeta || !workspace) return SP_ENULL;
    if (relu && !y) return SP_ENULL;
    if (C % 4 || M <= 0) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int G = pick_G(M, C);
    double* partial = (double*)workspace;
    float* coef = (float*)(partial + (int64_t)G * 2 * C);
    hipLaunchKernelGGL(bn_bwd_partial, dim3((unsigned)sp_cdiv(C, CB), G), dim3(256), 0, s, dy, x, y, mean, invstd, relu, M,
                       C, G, partial);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_bwd_final, dim3((unsigned)sp_cdiv(C, FIN_C)), dim3(FIN_C * FIN_L), 0, s, partial, M, C, G, dgamma, dbeta,
                       coef);
    SP_LAUNCH_CHECK();
    const int64_t n4 = M * C / 4;
    SP_RESET_AMAX(dx_amax, s);
    hipLaunchKernelGGL(bn_bwd_apply, dim3(ew_blocks(n4)), dim3(256), 0, s, dy, x, y, mean, invstd, gamma, coef, relu,
                       training, n4, C, dx, dres, dx_amax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int64_t sp_bn_split_workspace(int64_t M, int C) {
    const int64_t G = std::max(pick_G(M, C), pick_G_wide(M, C));
    return G * 2 * C * (int64_t)sizeof(double) + G * 2 * C * (int64_t)sizeof(float) + 2 * (int64_t)C * (int64_t)sizeof(float);
}

extern "C" int64_t sp_bn_mask_words(int64_t M, int C) { return sp_cdiv(M * C / 4, 64) * 4; }

extern "C" int sp_bn_fwd_split(const float* x, int64_t M, int C, float eps, float momentum, const float* gamma, const float* beta,
                               const float* residual, const unsigned* res_amax, int relu, float* mean, float* invstd,
                               float* running_mean, float* running_var, float* ext, float* y, void* planes, float* y_scale,
                               unsigned* bound, unsigned long long* mask, void* workspace, const double* pre_partial,
                               const float* pre_mm, int pre_G, void* stream) {
    if (!x || !gamma || !beta || !mean || !invstd || !ext || !y_scale || !bound || !workspace) return SP_ENULL;
    if ((residual && !res_amax) || (relu && !mask) || (!y && !planes)) return SP_ENULL;
    if (C % 4 || (planes && C % 16) || M <= 0 || ((uintptr_t)planes & 15)) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int G = pick_G(M, C);
    const double* partial = (double*)workspace;
    const float* mm = (float*)((double*)workspace + (int64_t)G * 2 * C);
    SP_RESET_AMAX(bound, s);
    if (pre_partial) {                 // first stage done by the producing conv's epilogue (sp_conv_igemm_f16x2_stats)
        if (!pre_mm || pre_G < 1) return SP_EINVAL;
        partial = pre_partial;
        mm = pre_mm;
        G = pre_G;
    } else {
        hipLaunchKernelGGL(bn_stats_mm_partial, dim3((unsigned)sp_cdiv(C, CB), G), dim3(256), 0, s, x, M, C, G, (double*)workspace,
                           (float*)((double*)workspace + (int64_t)G * 2 * C));
        SP_LAUNCH_CHECK();
    }
    if (G > 256)
        hipLaunchKernelGGL((bn_stats_mm_final<8, 32>), dim3((unsigned)sp_cdiv(C, 8)), dim3(256), 0, s, partial, mm, M, C, G, eps,
                           momentum, gamma, beta, relu, mean, invstd, running_mean, running_var, ext, bound);
    else
        hipLaunchKernelGGL((bn_stats_mm_final<FIN_C, FIN_L>), dim3((unsigned)sp_cdiv(C, FIN_C)), dim3(256), 0, s, partial, mm, M, C,
                           G, eps, momentum, gamma, beta, relu, mean, invstd, running_mean, running_var, ext, bound);
    SP_LAUNCH_CHECK();
    const int64_t n4 = M * C / 4;
    hipLaunchKernelGGL(bn_apply_split_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, x, mean, invstd, gamma, beta, residual, bound,
                       residual ? res_amax : nullptr, relu, n4, C, y, (uint16_t*)planes, y_scale, relu ? mask : nullptr);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_bn_bwd_split(const float* dy, const float* x, const unsigned long long* mask, const float* mean,
                               const float* invstd, const float* gamma, const float* ext, int64_t M, int C, float* dx,
                               float* dres, void* planes, float* dx_scale, unsigned* bound, float* dgamma, float* dbeta,
                               void* workspace, void* stream) {
    if (!dy || !x || !mean || !invstd || !gamma || !ext || !dx_scale || !bound || !dgamma || !dbeta || !workspace) return SP_ENULL;
    if (!dx && !planes) return SP_ENULL;
    if (C % 4 || (planes && C % 16) || M <= 0 || ((uintptr_t)planes & 15)) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = C >= 256 && C % 256 == 0;
    const int G = wide ? pick_G_wide(M, C) : pick_G(M, C);
    double* partial = (double*)workspace;
    float* dmax = (float*)(partial + (int64_t)G * 2 * C);
    float* coef = dmax + (int64_t)G * 2 * C;
    SP_RESET_AMAX(bound, s);
    if (wide)
        hipLaunchKernelGGL(bn_bwd_mm_partial<64>, dim3((unsigned)(C / 256), G), dim3(256), 0, s, dy, x, mask, mean, invstd, M, C, G,
                           partial, dmax);
    else
        hipLaunchKernelGGL(bn_bwd_mm_partial<16>, dim3((unsigned)sp_cdiv(C, CB), G), dim3(256), 0, s, dy, x, mask, mean, invstd, M, C,
                           G, partial, dmax);
    SP_LAUNCH_CHECK();
    if (G > 256)
        hipLaunchKernelGGL((bn_bwd_mm_final<8, 32>), dim3((unsigned)sp_cdiv(C, 8)), dim3(256), 0, s, partial, dmax, M, C, G, gamma,
                           mean, invstd, ext, dgamma, dbeta, coef, bound);
    else
        hipLaunchKernelGGL((bn_bwd_mm_final<FIN_C, FIN_L>), dim3((unsigned)sp_cdiv(C, FIN_C)), dim3(256), 0, s, partial, dmax, M, C,
                           G, gamma, mean, invstd, ext, dgamma, dbeta, coef, bound);
    SP_LAUNCH_CHECK();
    const int64_t n4 = M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_split_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, dy, x, mask, mean, invstd, gamma, coef,
                       bound, n4, C, dx, dres, (uint16_t*)planes, dx_scale);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int64_t sp_colsum_workspace(int64_t M, int C) { return (int64_t)pick_G(M, C) * C * (int64_t)sizeof(double); }

extern "C" int sp_colsum(const float* x, int64_t M, int C, int ld, float* out, int beta, void* workspace, void* stream) {
    if (!x || !out || !workspace) return SP_ENULL;
    if (C % 4 || ld % 4 || M <= 0) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (M <= COLSUM_SMALL_M && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0) {
        hipLaunchKernelGGL(colsum_small_kernel, dim3((unsigned)sp_cdiv(C, 256)), dim3(64), 0, s, x, (int)M, C, ld, out, beta);
        SP_LAUNCH_CHECK();
        return SP_OK;
    }
    const int G = pick_G(M, C);
    hipLaunchKernelGGL(colsum_partial, dim3((unsigned)sp_cdiv(C, CB), G), dim3(256), 0, s, x, M, C, ld, G, (double*)workspace);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_final, dim3((unsigned)sp_cdiv(C, FIN_C)), dim3(FIN_C * FIN_L), 0, s, (const double*)workspace, C, G, out,
                       beta);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_rowsum(const float* x, int64_t M, int C, float scale, float* out, void* stream) {
    if (!x || !out) return SP_ENULL;
    if (C % 4) return SP_EINVAL;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(M, 4), 4096));
    hipLaunchKernelGGL(rowsum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, M, C, scale, out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_rowsum_bwd(const float* dout, int64_t M, int C, float scale, float* dx, void* stream) {
    if (!dout || !dx) return SP_ENULL;
    if (C % 4) return SP_EINVAL;
    const int64_t n4 = M * C / 4;
    hipLaunchKernelGGL(rowsum_bwd_kernel, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, dout, n4, C, scale, dx);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_maxpool3s2_fwd(const float* x, int N, int H, int W, int C, float* y, int Ho, int Wo, void* stream) {
    return sp_maxpool3s2_fwd_idx(x, N, H, W, C, y, nullptr, Ho, Wo, stream);
}

extern "C" int sp_maxpool3s2_fwd_idx(const float* x, int N, int H, int W, int C, float* y, unsigned char* argmax, int Ho, int Wo,
                                     void* stream) {
    if (!x || !y) return SP_ENULL;
    if (C % 4) return SP_EINVAL;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, N, H, W, C, y, Ho, Wo,
                       argmax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_maxpool3s2_bwd_idx(const float* dy, const unsigned char* argmax, int N, int H, int W, int C, float* dx, int Ho,
                                     int Wo, void* stream) {
    if (!dy || !argmax || !dx) return SP_ENULL;
    if (C % 4) return SP_EINVAL;
    const int64_t total = (int64_t)N * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(ew_blocks(total) * 4), dim3(256), 0, (hipStream_t)stream, dy, argmax, N, H, W,
                       C, dx, Ho, Wo);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_maxpool3s2_bwd(const float* dy, const float* x, const float* y, int N, int H, int W, int C, float* dx,
                                 int Ho, int Wo, void* stream) {
    if (!dy || !x || !y || !dx) return SP_ENULL;
    const int64_t total = (int64_t)N * H * W * C;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_blocks(total) * 4), dim3(256), 0, (hipStream_t)stream, dy, x, y, N, H, W,
                       C, dx, Ho, Wo);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_nchw_to_nhwc_pad(const float* x, int N, int C, int H, int W, int Cp, float* y, void* stream) {
    if (!x || !y) return SP_ENULL;
    if (Cp < C) return SP_EINVAL;
    const int64_t total = (int64_t)N * H * W * Cp;
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, N, C, H, W, Cp, y);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_pad_lastdim(const float* x, int64_t rows, int Cin, int Cout, float* y, void* stream) {
    if (!x || !y) return SP_ENULL;
    const int64_t total = rows * Cout;
    hipLaunchKernelGGL(pad_lastdim_kernel, dim3(ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, rows, Cin, Cout, y);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_add(const float* a, const float* b, float* out, int64_t n, void* stream) {
    if (!a || !b || !out) return SP_ENULL;
    const int64_t n4 = n / 4;
    if (n4 > 0) {
        hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n4)), dim3(256), 0, (hipStream_t)stream, a, b, out, n4);
        SP_LAUNCH_CHECK();
    }
    if (n4 * 4 < n) {
        hipLaunchKernelGGL(add_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, b, out, n4 * 4, n);
        SP_LAUNCH_CHECK();
    }
    return SP_OK;
}

extern "C" int sp_sum_n_rows(const float* const* inputs, int count, int64_t n, float* out, unsigned* out_amax, const int* row_last,
                             const int* steps, int nsamples, void* stream) {
    if (!inputs || !out) return SP_ENULL;
    if (count < 1 || count > 32 || n % 4) return SP_EINVAL;
    if ((row_last != nullptr) != (steps != nullptr)) return SP_ENULL;
    if (row_last && (nsamples < 1 || n % nsamples || (n / nsamples) % 4)) return SP_EINVAL;      // whole float4 per sample
    SumList l;
    l.n = count;
    l.row_last = row_last;
    l.quads_per_sample = row_last ? n / nsamples / 4 : 1;
    for (int k = 0; k < count; ++k) {
        if (!inputs[k]) return SP_ENULL;
        l.p[k] = inputs[k];
        l.step[k] = steps ? steps[k] : -1;
    }
    SP_RESET_AMAX(out_amax, stream);
    if (row_last) hipLaunchKernelGGL(sum_n_rows_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, l, out, n / 4, out_amax);
    else hipLaunchKernelGGL(sum_n_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, l, out, n / 4, out_amax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
extern "C" int sp_sum_n(const float* const* inputs, int count, int64_t n, float* out, unsigned* out_amax, void* stream) {
    return sp_sum_n_rows(inputs, count, n, out, out_amax, nullptr, nullptr, 0, stream);
}

extern "C" int sp_sum_n_mixed_rows(const float* const* inputs, const void* const* planes, const float* const* scales, int count, int64_t n,
                                   float* out, unsigned* out_amax, const int* row_last, const int* steps, int nsamples, void* stream) {
    if (!inputs || !planes || !scales || !out) return SP_ENULL;
    if (count < 1 || count > 32 || n % 16) return SP_EINVAL;
    if ((row_last != nullptr) != (steps != nullptr)) return SP_ENULL;
    if (row_last && (nsamples < 1 || n % nsamples || (n / nsamples) % 16)) return SP_EINVAL;      // whole 16-element groups per sample
    SumListMixed l;
    l.n = count;
    l.row_last = row_last;
    l.groups_per_sample = row_last ? n / nsamples / 16 : 1;
    for (int k = 0; k < count; ++k) {
        l.step[k] = steps ? steps[k] : -1;
        if (!inputs[k] && (!planes[k] || !scales[k])) return SP_ENULL;
        if (!inputs[k] && ((uintptr_t)planes[k] & 15)) return SP_EINVAL;
        l.f[k] = inputs[k];
        l.pl[k] = (const uint16_t*)planes[k];
        l.sc[k] = scales[k];
    }
    SP_RESET_AMAX(out_amax, stream);
    hipLaunchKernelGGL(sum_n_mixed_kernel, dim3(ew_blocks(n / 16)), dim3(256), 0, (hipStream_t)stream, l, out, n / 16, out_amax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
extern "C" int sp_sum_n_mixed(const float* const* inputs, const void* const* planes, const float* const* scales, int count, int64_t n,
                              float* out, unsigned* out_amax, void* stream) {
    return sp_sum_n_mixed_rows(inputs, planes, scales, count, n, out, out_amax, nullptr, nullptr, 0, stream);
}

extern "C" int sp_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, void* stream) {
    if (!dy || !y || !dx) return SP_ENULL;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, n, dx);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
