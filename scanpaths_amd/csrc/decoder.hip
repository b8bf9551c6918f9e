// Small HBM-/latency-bound kernels of the attentive ConvLSTM decoder (AiR/models/baseline_attention.py).
// Big contractions live in conv_igemm.hip / conv_wgrad.hip; this file holds the per-step pointwise cell,
// the memory-list attention, the composed predict_head epilogue (with the wave-shuffle softmax) and glue.
#include "common.h"
#include <algorithm>

namespace {

inline int ew_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(n, 256), 2048)); }

__device__ __forceinline__ float sigmoidf_(float x) { return sp_sigmoid(x); }      // common.h: the fused cell epilogue uses the same pair

// ------------------------------------------------------------------------------------------------
// ConvLSTM cell pointwise (:44-54): gates gate-major [i|f|o|g] x C.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_fwd_kernel(const float* xg, const float* hg, const float* c_prev,
                                                       int64_t rows, int C, float* gates, float* c_out, float* h_out) {
    const int C4 = C / 4;
    const int64_t total = rows * C4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / C4;
        const int c = (int)(idx - r * C4) * 4;
        const float* px = xg + r * 4 * C + c;
        float4 pi = *reinterpret_cast<const float4*>(px);
        float4 pf = *reinterpret_cast<const float4*>(px + C);
        float4 po = *reinterpret_cast<const float4*>(px + 2 * C);
        float4 pg = *reinterpret_cast<const float4*>(px + 3 * C);
        if (hg) {
            const float* ph = hg + r * 4 * C + c;
            const float4 a = *reinterpret_cast<const float4*>(ph), b = *reinterpret_cast<const float4*>(ph + C);
            const float4 d = *reinterpret_cast<const float4*>(ph + 2 * C), e = *reinterpret_cast<const float4*>(ph + 3 * C);
            pi.x += a.x; pi.y += a.y; pi.z += a.z; pi.w += a.w;
            pf.x += b.x; pf.y += b.y; pf.z += b.z; pf.w += b.w;
            po.x += d.x; po.y += d.y; po.z += d.z; po.w += d.w;
            pg.x += e.x; pg.y += e.y; pg.z += e.z; pg.w += e.w;
        }
        float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_prev) cp = *reinterpret_cast<const float4*>(c_prev + r * C + c);
        float4 gi, gf, go, gg, cn, hn;
#define CELL(k)                                   \
        gi.k = sigmoidf_(pi.k);                   \
        gf.k = sigmoidf_(pf.k);                   \
        go.k = sigmoidf_(po.k);                   \
        gg.k = sp_tanh(pg.k);                       \
        cn.k = gf.k * cp.k + gi.k * gg.k;         \
        hn.k = go.k * cn.k;
        CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
        float* pgt = gates + r * 4 * C + c;
        *reinterpret_cast<float4*>(pgt) = gi;
        *reinterpret_cast<float4*>(pgt + C) = gf;
        *reinterpret_cast<float4*>(pgt + 2 * C) = go;
        *reinterpret_cast<float4*>(pgt + 3 * C) = gg;
        *reinterpret_cast<float4*>(c_out + r * C + c) = cn;
        *reinterpret_cast<float4*>(h_out + r * C + c) = hn;
    }
}

// ------------------------------------------------------------------------------------------------
// get_channel_semantic + ReLU (baseline_attention.py:246-250,284,324): pooled[s][b][c] = relu(mean_p(a[s][b][p] * vf[b][p][c])).
// HBM-bound on vf (one read); as a batched GEMM with M = S = 2 rows it ran a 128-row MFMA tile for 2 rows (0.14 ms fwd,
// 0.22 + 0.06 ms bwd per decode step).  fwd: block = (pixel chunk, sample), a thread owns 4 channels of every second pixel
// (256 threads = 128 channel quads x 2 pixel lanes; C <= 512), chunk partials reduced in fixed order by the finish kernel.
// bwd: one wave per pixel: d_a[s][b][p] = <dz[b][s][:], vf[b][p][:]> / P (wave reduction) and the rank-S update
// d_vf[b][p][:] = sum_s a[s][b][p] * dz[b][s][:] / P of the same row in one pass over vf.  S <= 2.
// ------------------------------------------------------------------------------------------------
constexpr int SP_PCH = 160;      // pixels per forward block
__global__ __launch_bounds__(256) void sempool_fwd_kernel(const float* __restrict__ a, const float* __restrict__ vf, int S, int B,
                                                          int P, int C, float* __restrict__ partial) {
    __shared__ f32x4 red[2][128];
    const int ch = blockIdx.x, b = blockIdx.y, nch = gridDim.x;
    const int c4 = threadIdx.x & 127, pl = threadIdx.x >> 7, C4 = C / 4;
    const int p0 = ch * SP_PCH, p1 = min(P, p0 + SP_PCH);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4)
        for (int p = p0 + pl; p < p1; p += 2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(vf + ((int64_t)b * P + p) * C + c4 * 4);
            acc0 += a[((int64_t)0 * B + b) * P + p] * v;
            if (S > 1) acc1 += a[((int64_t)1 * B + b) * P + p] * v;
        }
    if (pl == 1) { red[0][c4] = acc0; red[1][c4] = acc1; }
    __syncthreads();
    if (pl == 0 && c4 < C4) {
        acc0 += red[0][c4];
        acc1 += red[1][c4];
        f32x4* dst = reinterpret_cast<f32x4*>(partial + (((int64_t)ch * B + b) * S) * C) + c4;
        dst[0] = acc0;
        if (S > 1) dst[C4] = acc1;
    }
    (void)nch;
}
// out (b, s, :) at out + b * osb + s * oss  = relu(alpha * sum_chunks partial [chunk][b][s][c])
__global__ __launch_bounds__(256) void sempool_finish_kernel(const float* __restrict__ partial, int nch, int64_t n, float alpha,
                                                             float* __restrict__ out, int S, int C, int64_t osb, int64_t oss) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < nch; ++k) s += partial[(int64_t)k * n + i];
        const int c = (int)(i % C);
        const int64_t bs = i / C;
        out[(bs / S) * osb + (bs % S) * oss + c] = fmaxf(alpha * s, 0.f);
    }
}
// dz, out: row (b, s) at b * zsb + s * zss; a [S][B][P], vf [B][P][C] -> da [S][B][P], dvf [B][P][C]
__global__ __launch_bounds__(256) void sempool_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ out,
                                                          const float* __restrict__ a, const float* __restrict__ vf, int S,
                                                          int B, int P, int C, float alpha, float* __restrict__ da,
                                                          float* __restrict__ dvf, const int* __restrict__ row_last, int row_step,
                                                          int64_t zsb, int64_t zss) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * 256) >> 6;
    const int nq = C / 4;                         // channel quads; lane handles quads lane, lane+64
    for (int64_t bp = wave; bp < (int64_t)B * P; bp += nwaves) {
        const int b = (int)(bp / P), p = (int)(bp % P);
        if (row_last && row_last[b] < row_step) {      // row sparsity: dz of this sample is exactly zero -> zero gradients, vf not read (wave-uniform)
            for (int q = lane; q < nq; q += 64) *reinterpret_cast<f32x4*>(dvf + bp * C + q * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            if (lane == 0) {
                da[((int64_t)0 * B + b) * P + p] = 0.f;
                if (S > 1) da[((int64_t)1 * B + b) * P + p] = 0.f;
            }
            continue;
        }
        const float a0 = a[((int64_t)0 * B + b) * P + p], a1 = S > 1 ? a[((int64_t)1 * B + b) * P + p] : 0.f;
        float d0 = 0.f, d1 = 0.f;
        for (int q = lane; q < nq; q += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(vf + bp * C + q * 4);
            f32x4 g0 = *reinterpret_cast<const f32x4*>(dz + (int64_t)b * zsb + q * 4);
            const f32x4 o0 = *reinterpret_cast<const f32x4*>(out + (int64_t)b * zsb + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g0[e] = o0[e] > 0.f ? g0[e] : 0.f;
            d0 += g0[0] * v[0] + g0[1] * v[1] + g0[2] * v[2] + g0[3] * v[3];
            f32x4 w = a0 * g0;
            if (S > 1) {
                f32x4 g1 = *reinterpret_cast<const f32x4*>(dz + (int64_t)b * zsb + zss + q * 4);
                const f32x4 o1 = *reinterpret_cast<const f32x4*>(out + (int64_t)b * zsb + zss + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) g1[e] = o1[e] > 0.f ? g1[e] : 0.f;
                d1 += g1[0] * v[0] + g1[1] * v[1] + g1[2] * v[2] + g1[3] * v[3];
                w += a1 * g1;
            }
            *reinterpret_cast<f32x4*>(dvf + bp * C + q * 4) = alpha * w;
        }
        d0 = wave_sum(d0);
        d1 = wave_sum(d1);
        if (lane == 0) {
            da[((int64_t)0 * B + b) * P + p] = alpha * d0;
            if (S > 1) da[((int64_t)1 * B + b) * P + p] = alpha * d1;
        }
    }
}

// ConvLSTM cell with the rank-1 gate terms fused in (baseline_attention.py:40-50): the i/f/o pre-activations receive
//   sum_k spcol[b,p,k] * wc[b, g*C + c, k]      (conv3x3(W, spatial (x) semantic) as a 9-tap 1-channel conv per stream with the
// per-sample contracted filter).  As a separate batched GEMM with beta = 1 this term cost a full read-modify-write of the gate
// tensor (0.77 ms per decode step at bs 32); here it is KP FMAs per gate value on data the cell already holds in registers.
// Block = 64 channels x 64 pixels per iteration of ONE sample: the sample's filter slice (3 gates x 64 channels x KP) is staged
// once per block in LDS as [k][gate][channel] (conflict-free b128 reads, broadcast across the pixel lanes), the pixel taps as
// [pixel][k]; a thread owns one channel quad of 4 pixels (48 accumulators).  grid (C/64, ceil(P/RB), B).
constexpr int R1_RB = 256;     // pixels per block
__global__ __launch_bounds__(256) void lstm_rank1_fwd_kernel(const float* __restrict__ xg, const float* __restrict__ hg,
                                                             const float* __restrict__ c_prev, const float* __restrict__ spcol,
                                                             const float* __restrict__ wc, int P, int C, int KP,
                                                             float* __restrict__ gates, float* __restrict__ c_out,
                                                             float* __restrict__ h_out, unsigned* __restrict__ h_amax) {
    extern __shared__ __attribute__((aligned(16))) float r1s[];
    __shared__ float sh4[4];
    float hmx = 0.f;
    float* wcT = r1s;                       // [KP][3][64]
    float* sp = r1s + KP * 192;             // [64][KP]
    const int t = threadIdx.x, q = t & 15, rg = t >> 4;
    const int c0 = blockIdx.x * 64, b = blockIdx.z;
    const int p_begin = blockIdx.y * R1_RB, p_end = min(P, p_begin + R1_RB);
    const int N3 = 3 * C;
    for (int i = t; i < 192 * KP; i += 256) {            // global [n][k] (k fastest) -> LDS [k][g][j]
        const int k = i % KP, nj = i / KP;                // nj = g*64 + j
        const int g = nj >> 6, j = nj & 63;
        wcT[(k * 3 + g) * 64 + j] = wc[((int64_t)b * N3 + g * C + c0 + j) * KP + k];
    }
    for (int p0 = p_begin; p0 < p_end; p0 += 64) {
        __syncthreads();
        for (int i = t; i < 64 * KP; i += 256) {
            const int pr = p0 + i / KP;
            sp[i] = pr < p_end ? spcol[((int64_t)b * P + p0) * KP + i] : 0.f;
        }
        __syncthreads();
        f32x4 acc[4][3];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[rr][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < KP; ++k) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wcT + (k * 3 + 0) * 64 + q * 4);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(wcT + (k * 3 + 1) * 64 + q * 4);
            const f32x4 w2 = *reinterpret_cast<const f32x4*>(wcT + (k * 3 + 2) * 64 + q * 4);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float sv = sp[(rg * 4 + rr) * KP + k];
                acc[rr][0] += sv * w0;
                acc[rr][1] += sv * w1;
                acc[rr][2] += sv * w2;
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int pr = p0 + rg * 4 + rr;
            if (pr >= p_end) continue;
            const int64_t r = (int64_t)b * P + pr;
            const int c = c0 + q * 4;
            const float* px = xg + r * 4 * C + c;
            f32x4 pi = *reinterpret_cast<const f32x4*>(px) + acc[rr][0];
            f32x4 pf = *reinterpret_cast<const f32x4*>(px + C) + acc[rr][1];
            f32x4 po = *reinterpret_cast<const f32x4*>(px + 2 * C) + acc[rr][2];
            f32x4 pg = *reinterpret_cast<const f32x4*>(px + 3 * C);
            if (hg) {
                const float* ph = hg + r * 4 * C + c;
                pi += *reinterpret_cast<const f32x4*>(ph);
                pf += *reinterpret_cast<const f32x4*>(ph + C);
                po += *reinterpret_cast<const f32x4*>(ph + 2 * C);
                pg += *reinterpret_cast<const f32x4*>(ph + 3 * C);
            }
            f32x4 cp = {0.f, 0.f, 0.f, 0.f};
            if (c_prev) cp = *reinterpret_cast<const f32x4*>(c_prev + r * C + c);
            f32x4 gi, gf, go, gg, cn, hn;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gi[e] = sigmoidf_(pi[e]);
                gf[e] = sigmoidf_(pf[e]);
                go[e] = sigmoidf_(po[e]);
                gg[e] = sp_tanh(pg[e]);
                cn[e] = gf[e] * cp[e] + gi[e] * gg[e];
                hn[e] = go[e] * cn[e];
            }
            float* pgt = gates + r * 4 * C + c;
            *reinterpret_cast<f32x4*>(pgt) = gi;
            *reinterpret_cast<f32x4*>(pgt + C) = gf;
            *reinterpret_cast<f32x4*>(pgt + 2 * C) = go;
            *reinterpret_cast<f32x4*>(pgt + 3 * C) = gg;
            *reinterpret_cast<f32x4*>(c_out + r * C + c) = cn;
            *reinterpret_cast<f32x4*>(h_out + r * C + c) = hn;
            hmx = amax4(hmx, hn[0], hn[1], hn[2], hn[3]);
        }
    }
    if (h_amax) block_amax_commit(hmx, h_amax, sh4);
}

// Gradient of the per-sample contracted rank-1 filters:  dwc[b,n,k] = sum_p dpre[b,p,n] * spcol[b,p,k]  (n < 3C).
// HBM-bound (one read of the i/f/o part of dpre).  Block = a 256-wide strip of n x a chunk of pixels of one sample; a thread
// owns 4 consecutive n (16-byte loads: a wave reads 1 KB of a dpre row) and keeps 4 x KP sums in registers; the 4 waves take
// every 4th pixel (its KP taps are a wave-wide LDS broadcast) and are combined through LDS in fixed order; chunk partials are
// reduced by rank1_dwc_reduce_kernel.  As a batched TN GEMM with a 128-wide N tile for 20 columns this took 0.45 ms.
constexpr int R1_MAXKP = 24, R1_PCH = 256;
__global__ __launch_bounds__(256) void rank1_dwc_kernel(const float* __restrict__ dpre, const float* __restrict__ spcol, int P,
                                                        int ld, int N3, int KP, int nchunk, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float sp[R1_PCH * R1_MAXKP];      // 24 KB: the chunk's taps
    __shared__ f32x4 red[3][64][R1_MAXKP / 4 + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + lane * 4, ch = blockIdx.y, b = blockIdx.z;
    const int p_begin = ch * R1_PCH, np = min(P, p_begin + R1_PCH) - p_begin;
    for (int i = threadIdx.x; i < np * KP; i += 256) sp[i] = spcol[((int64_t)b * P + p_begin) * KP + i];
    __syncthreads();
    f32x4 acc[R1_MAXKP];
#pragma unroll
    for (int k = 0; k < R1_MAXKP; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n < N3) {
        const float* src = dpre + ((int64_t)b * P + p_begin) * ld + n;
        for (int pq = wv; pq < np; pq += 16) {              // 4 independent 16-byte row loads in flight per thread
            f32x4 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                g[u] = pq + 4 * u < np ? *reinterpret_cast<const f32x4*>(src + (int64_t)(pq + 4 * u) * ld) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* spr = sp + min(pq + 4 * u, np - 1) * KP;
#pragma unroll
                for (int k = 0; k < R1_MAXKP; ++k)
                    if (k < KP) acc[k] += spr[k] * g[u];
            }
        }
    }
    // combine the 4 pixel lanes (waves) in fixed order, R1_MAXKP/4 taps per pass
    for (int k0 = 0; k0 < KP; k0 += R1_MAXKP / 4) {
        __syncthreads();
        if (wv > 0) {
#pragma unroll
            for (int k = 0; k < R1_MAXKP; ++k)
                if (k >= k0 && k < k0 + R1_MAXKP / 4) red[wv - 1][lane][k - k0] = acc[k];
        }
        __syncthreads();
        if (wv == 0 && n < N3) {
#pragma unroll
            for (int k = 0; k < R1_MAXKP; ++k)
                if (k >= k0 && k < k0 + R1_MAXKP / 4 && k < KP) {
                    const f32x4 v = ((acc[k] + red[0][lane][k - k0]) + red[1][lane][k - k0]) + red[2][lane][k - k0];
                    float* dst = partial + (((int64_t)ch * gridDim.z + b) * N3 + n) * KP + k;
                    dst[0] = v[0];
                    dst[KP] = v[1];
                    dst[2 * KP] = v[2];
                    dst[3 * KP] = v[3];
                }
        }
    }
}
__global__ __launch_bounds__(256) void rank1_dwc_reduce_kernel(const float* __restrict__ partial, int64_t n, int nchunk,
                                                               float* __restrict__ dwc) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int c = 0; c < nchunk; ++c) s += partial[(int64_t)c * n + i];
        dwc[i] = s;
    }
}

// planes != NULL: also writes dpre as the 2xfp16 split operand of the two h-gate GEMMs that consume it (data and weight gradient),
// with the operand scale from an upper bound of max|dpre| (bn_pool.hip explains why a bound is enough): gates lie in (0,1), |g| < 1,
// |c_t| <= t + 1, so with D = max|dc| + max|dh| (>= |dc + dh*o|):  |d_g|, |d_i| <= D,  |d_o| <= max|dh| * c_bound / 4,
// |d_f| <= D * cprev_bound / 4.  The separate split pass re-read the 671 MB tensor every step.
__global__ __launch_bounds__(256) void lstm_bwd_kernel(const float* dh, const float* dc, const float* gates,
                                                       const float* c_prev, const float* c_out, int64_t rows, int C,
                                                       float* dpre, float* dc_prev, unsigned* dpre_amax, unsigned* dcp_amax,
                                                       const unsigned* dh_amax, const unsigned* dc_amax, float c_bound,
                                                       float cprev_bound, uint16_t* planes, float* dpre_scale,
                                                       const int* row_last, int row_step, int row_P) {
    // row_last (nullable): samples with row_last[r / row_P] < row_step receive no loss gradient at this decode step or any later one
    // (loss masks): dh and dc are exactly zero there and so is every output -- written as zeros without reading the five inputs
    __shared__ float sh4[4];
    float dmx = 0.f, cmx = 0.f;
    const int C4 = C / 4;
    const int64_t total = rows * C4;
    float s = 1.f, bound = 0.f;
    if (planes) {
        const float a_dh = dh_amax ? __uint_as_float(*dh_amax) : 0.f, a_dc = dc_amax ? __uint_as_float(*dc_amax) : 0.f;
        const float D = a_dh + a_dc;
        bound = fmaxf(D, 0.25f * fmaxf(a_dh * c_bound, D * cprev_bound)) * 1.0001f;
        s = scale_of(__float_as_uint(bound));
    }
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < total; base += stride) {     // wave-uniform
        const int64_t idx = base + lane;
        const bool live = idx < total;
        const int64_t r = live ? idx / C4 : 0;
        const int c = live ? (int)(idx - r * C4) * 4 : 0;
        float4 di = make_float4(0.f, 0.f, 0.f, 0.f), df = di, dov = di, dg = di;
        const bool active = !row_last || row_last[r / row_P] >= row_step;
        if (live && !active) {
            if (dpre) {
                float* pd = dpre + r * 4 * C + c;
                *reinterpret_cast<float4*>(pd) = di;
                *reinterpret_cast<float4*>(pd + C) = di;
                *reinterpret_cast<float4*>(pd + 2 * C) = di;
                *reinterpret_cast<float4*>(pd + 3 * C) = di;
            }
            *reinterpret_cast<float4*>(dc_prev + r * C + c) = di;
        }
        if (live && active) {
            const float* pgt = gates + r * 4 * C + c;
            const float4 gi = *reinterpret_cast<const float4*>(pgt), gf = *reinterpret_cast<const float4*>(pgt + C);
            const float4 go = *reinterpret_cast<const float4*>(pgt + 2 * C), gg = *reinterpret_cast<const float4*>(pgt + 3 * C);
            const float4 cn = *reinterpret_cast<const float4*>(c_out + r * C + c);
            float4 cp = make_float4(0.f, 0.f, 0.f, 0.f), vdh = cp, vdc = cp;
            if (c_prev) cp = *reinterpret_cast<const float4*>(c_prev + r * C + c);
            if (dh) vdh = *reinterpret_cast<const float4*>(dh + r * C + c);
            if (dc) vdc = *reinterpret_cast<const float4*>(dc + r * C + c);
            float4 dcp;
#define CELLB(k)                                              \
        {                                                     \
            const float dct = vdc.k + vdh.k * go.k;           \
            dov.k = vdh.k * cn.k * go.k * (1.f - go.k);       \
            di.k = dct * gg.k * gi.k * (1.f - gi.k);          \
            df.k = dct * cp.k * gf.k * (1.f - gf.k);          \
            dg.k = dct * gi.k * (1.f - gg.k * gg.k);          \
            dcp.k = dct * gf.k;                               \
        }
            CELLB(x) CELLB(y) CELLB(z) CELLB(w)
#undef CELLB
            if (dpre) {                // NULL: every consumer reads the split form (planes) -- 671 MB per step not written
                float* pd = dpre + r * 4 * C + c;
                *reinterpret_cast<float4*>(pd) = di;
                *reinterpret_cast<float4*>(pd + C) = df;
                *reinterpret_cast<float4*>(pd + 2 * C) = dov;
                *reinterpret_cast<float4*>(pd + 3 * C) = dg;
            }
            *reinterpret_cast<float4*>(dc_prev + r * C + c) = dcp;
            dmx = amax4(amax4(dmx, di.x, di.y, di.z, di.w), df.x, df.y, df.z, df.w);
            dmx = amax4(amax4(dmx, dov.x, dov.y, dov.z, dov.w), dg.x, dg.y, dg.z, dg.w);
            cmx = amax4(cmx, dcp.x, dcp.y, dcp.z, dcp.w);
        }
        if (planes) {                  // (wave-uniform; the 64 lanes of a wave own 64 consecutive float4 of ONE row: C % 256 == 0)
            const float4 v4[4] = {di, df, dov, dg};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                ushort4 pa, pb;
                split2(v4[g].x, s, pa.x, pb.x);
                split2(v4[g].y, s, pa.y, pb.y);
                split2(v4[g].z, s, pa.z, pb.z);
                split2(v4[g].w, s, pa.w, pb.w);
                store_planes_quad(planes, r * C + g * C4 + (c >> 2), live, pa, pb);
            }
        }
    }
    if (dpre_amax) block_amax_commit(dmx, dpre_amax, sh4);
    if (dcp_amax) block_amax_commit(cmx, dcp_amax, sh4);
    if (planes && blockIdx.x == 0 && threadIdx.x < 8)
        reinterpret_cast<uint2*>(planes + 8 * (rows * C))[threadIdx.x] = make_uint2(0u, 0u);       // 64-byte zero block after 2*4C*rows halves
    if (planes && blockIdx.x == 0 && threadIdx.x == 0) {
        dpre_scale[0] = s;
        dpre_scale[1] = bound;
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 zero-padded im2col of 1-channel maps into columns [koff, koff+9) of col[r][p][ldk], and adjoint.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col1_kernel(const float* maps, int R, int H, int W, int koff, int ldk, float* col) {
    const int64_t total = (int64_t)R * H * W * 9;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 9);
        int64_t q = i / 9;
        const int x = (int)(q % W); q /= W;
        const int y = (int)(q % H);
        const int r = (int)(q / H);
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        float v = 0.f;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = maps[((int64_t)r * H + yy) * W + xx];
        col[(((int64_t)r * H + y) * W + x) * ldk + koff + tap] = v;
    }
}
// all S streams in one launch: maps [S][R][H][W] -> col [R][H*W][ldk], stream s in columns [9 s, 9 s + 9), columns >= 9 S zero (the decode
// loop ran a zero-fill and one launch per stream per step); and its adjoint dmaps [S][R][H][W]
__global__ __launch_bounds__(256) void im2col1_multi_kernel(const float* __restrict__ maps, int S, int R, int H, int W, int ldk, float* __restrict__ col) {
    const int64_t total = (int64_t)R * H * W * ldk;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % ldk);
        int64_t q = i / ldk;
        const int x = (int)(q % W); q /= W;
        const int y = (int)(q % H);
        const int r = (int)(q / H);
        float v = 0.f;
        if (k < 9 * S) {
            const int s = k / 9, tap = k - 9 * s;
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) v = maps[(((int64_t)s * R + r) * H + yy) * W + xx];
        }
        col[i] = v;
    }
}
__global__ __launch_bounds__(256) void col2im1_multi_kernel(const float* __restrict__ dcol, int S, int R, int H, int W, int ldk, float* __restrict__ dmaps) {
    const int64_t total = (int64_t)S * R * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t q = i;
        const int x = (int)(q % W); q /= W;
        const int y = (int)(q % H); q /= H;
        const int r = (int)(q % R);
        const int s = (int)(q / R);
        float acc = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {          // same order as col2im1_kernel: identical sums
            const int y0 = y - (tap / 3 - 1), x0 = x - (tap % 3 - 1);
            if ((unsigned)y0 < (unsigned)H && (unsigned)x0 < (unsigned)W)
                acc += dcol[(((int64_t)r * H + y0) * W + x0) * ldk + 9 * s + tap];
        }
        dmaps[i] = acc;
    }
}
__global__ __launch_bounds__(256) void col2im1_kernel(const float* dcol, int R, int H, int W, int koff, int ldk, float* dmaps) {
    const int64_t total = (int64_t)R * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t q = i;
        const int x = (int)(q % W); q /= W;
        const int y = (int)(q % H);
        const int r = (int)(q / H);
        float s = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // col[(y0,x0)][tap] = map[y0+ky-1][x0+kx-1]  ->  (y0,x0) = (y-ky+1, x-kx+1)
            const int y0 = y - (tap / 3 - 1), x0 = x - (tap % 3 - 1);
            if ((unsigned)y0 < (unsigned)H && (unsigned)x0 < (unsigned)W)
                s += dcol[(((int64_t)r * H + y0) * W + x0) * ldk + koff + tap];
        }
        dmaps[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// attention over a memory list: one block per row r.
// ------------------------------------------------------------------------------------------------
constexpr int MAXT = 40;

__global__ __launch_bounds__(256) void listatt_fwd_kernel(const float* L, const float* u, int T, int R, int D, float* mem,
                                                          float* alpha) {
    __shared__ float sh16[16];
    __shared__ float sc[MAXT];
    const int r = blockIdx.x;
    for (int t0 = 0; t0 < T; t0 += 4) {          // four list entries per round (block_sum4_256: same bits as one at a time)
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d = threadIdx.x; d < D; d += 256) {
            const float ud = u[d];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t0 + k < T) s[k] += L[((int64_t)(t0 + k) * R + r) * D + d] * ud;
        }
        block_sum4_256(s, sh16);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t0 + k < T) sc[t0 + k] = s[k];
        }
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, sc[t]);
    float den = 0.f;
    for (int t = 0; t < T; ++t) den += expf(sc[t] - mx);
    const float inv = 1.f / den;
    if (threadIdx.x < T) alpha[(int64_t)threadIdx.x * R + r] = expf(sc[threadIdx.x] - mx) * inv;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) acc += expf(sc[t] - mx) * inv * L[((int64_t)t * R + r) * D + d];
        mem[(int64_t)r * D + d] = acc;
    }
}

__global__ __launch_bounds__(256) void listatt_bwd_kernel(const float* dmem, const float* L, const float* u,
                                                          const float* alpha, int T, int R, int D, float* dL,
                                                          float* du_partial) {
    __shared__ float sh16[16];
    __shared__ float gs[MAXT];
    const int r = blockIdx.x;
    const float* dm = dmem + (int64_t)r * D;
    for (int t0 = 0; t0 < T; t0 += 4) {          // four list entries per round (block_sum4_256: same bits as one at a time)
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d = threadIdx.x; d < D; d += 256) {
            const float dmd = dm[d];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t0 + k < T) s[k] += L[((int64_t)(t0 + k) * R + r) * D + d] * dmd;
        }
        block_sum4_256(s, sh16);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t0 + k < T) gs[t0 + k] = s[k];
        }
    }
    __syncthreads();
    float mean = 0.f;
    for (int t = 0; t < T; ++t) mean += alpha[(int64_t)t * R + r] * gs[t];
    for (int d = threadIdx.x; d < D; d += 256) {
        const float dmd = dm[d], ud = u[d];
        float du = 0.f;
        for (int t = 0; t < T; ++t) {
            const float a = alpha[(int64_t)t * R + r];
            const float ds = a * (gs[t] - mean);
            const int64_t o = ((int64_t)t * R + r) * D + d;
            du += ds * L[o];
            dL[o] = a * dmd + ds * ud;
        }
        du_partial[(int64_t)r * D + d] = du;
    }
}

// ------------------------------------------------------------------------------------------------
// out = relu(a * b[i % nb])  (get_spatial_semantic with mean_c(vf) hoisted)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mulrelu_fwd_kernel(const float* a, const float* b, int64_t n, int64_t nb, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = fmaxf(a[i] * b[i % nb], 0.f);
}
__global__ __launch_bounds__(256) void mulrelu_bwd_kernel(const float* dout, const float* a, const float* b, const float* out,
                                                          int64_t n, int64_t nb, float* da, float* db_partial) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float g = out[i] > 0.f ? dout[i] : 0.f;
        da[i] = g * b[i % nb];
        db_partial[i] = g * a[i];
    }
}

// out[r][:] = sel[r] ? a[r][:] : b[r][:]   and its adjoint
__global__ __launch_bounds__(256) void select_rows_kernel(const float* a, const float* b, const unsigned char* sel, int64_t rows,
                                                          int64_t len, float* out) {
    const int64_t n = rows * len;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = sel[i / len] ? a[i] : b[i];
}
__global__ __launch_bounds__(256) void select_rows_bwd_kernel(const float* dout, const unsigned char* sel, int64_t rows,
                                                              int64_t len, float* da, float* db) {
    const int64_t n = rows * len;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const bool s = sel[i / len] != 0;
        const float g = dout[i];
        da[i] = s ? g : 0.f;
        db[i] = s ? 0.f : g;
    }
}

// ------------------------------------------------------------------------------------------------
// composed predict_head epilogue.  One block per (head, sample).
//   Z row p holds at column base = hd*HC : [0] terminate map, [1] action map, [2+tap] the 49 taps of the
//   7x7 stride-5 pad-2 duration conv composed with the 5x5 head conv; cb[hd][j] = composed biases,
//   cb[hd][51] = drt_layer_1.bias.  Padding semantics of the ORIGINAL 7x7 conv are kept: taps that fall
//   outside the (zero-padded) intermediate map contribute neither value nor composed bias.
// ------------------------------------------------------------------------------------------------
constexpr int NTAP = 49;
constexpr int IDX_BD = 51;
constexpr int MAXS = 256;   // dh*dw upper bound

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* Z, int B, int Hm, int Wm, int ldz, int HC,
                                                       const float* cb, int cb_per_sample, const float* w2,
                                                       const float* b2, int softmax, float* logits, float* amap, float* mu, float* sigma2, float* drt,
                                                       int dh, int dw, const float* dpre, int zc, int parts) {
    // parts: bit 0 = the saliency part (terminate logit, action map, softmax), bit 1 = the duration part (mu, sigma2).  The decode loop
    // evaluates bit 0 per step (the action map feeds the next memory update) and bit 1 ONCE for all T steps behind the loop (B = T x
    // batch rows: nothing of the recurrence depends on it), see models/scanpath_model.py decode()
    __shared__ float sh4[4];
    __shared__ float sdrt[MAXS];
    const int hd = blockIdx.y, b = blockIdx.x;
    const int P = Hm * Wm, S = dh * dw;
    const float* z = Z + (int64_t)b * P * ldz + hd * zc;
    const float* c = cb + (cb_per_sample ? ((int64_t)b * gridDim.y + hd) : (int64_t)hd) * HC;
    float* lg = logits + ((int64_t)hd * B + b) * (P + 1);
    float* am = amap + ((int64_t)hd * B + b) * P;
    if (parts & 1) {
    // terminate logit + action map
    float s0 = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) {
        s0 += z[(int64_t)p * ldz];
        const float v = fmaxf(z[(int64_t)p * ldz + 1] + c[1], 0.f);
        lg[1 + p] = v;
        am[p] = v;
    }
    s0 = block_sum_256(s0, sh4);
    const float yterm = s0 / (float)P + c[0];
    if (threadIdx.x == 0) lg[0] = yterm;
    }
    if (parts & 2) {
    // duration branch
    for (int s = threadIdx.x; s < S; s += 256) {
        const int sy = s / dw, sx = s % dw;
        float acc = c[IDX_BD];
        if (dpre) acc += dpre[((int64_t)hd * B + b) * S + s];   // window sums + composed tap biases (head_direct.hip)
        else
        for (int ky = 0; ky < 7; ++ky) {
            const int py = 5 * sy - 2 + ky;
            if ((unsigned)py >= (unsigned)Hm) continue;
            for (int kx = 0; kx < 7; ++kx) {
                const int px = 5 * sx - 2 + kx;
                if ((unsigned)px >= (unsigned)Wm) continue;
                const int tap = ky * 7 + kx;
                acc += z[(int64_t)(py * Wm + px) * ldz + 2 + tap] + c[2 + tap];
            }
        }
        acc = fmaxf(acc, 0.f);
        sdrt[s] = acc;
        drt[((int64_t)hd * B + b) * S + s] = acc;
    }
    __syncthreads();
    float t0 = 0.f, t1 = 0.f;
    for (int s = threadIdx.x; s < S; s += 256) {
        t0 += w2[s] * sdrt[s];
        t1 += w2[S + s] * sdrt[s];
    }
    t0 = block_sum_256(t0, sh4);
    t1 = block_sum_256(t1, sh4);
    if (threadIdx.x == 0) {
        mu[hd * B + b] = t0 + b2[0];
        sigma2[hd * B + b] = expf(t1 + b2[1]);
    }
    }
    if (softmax && (parts & 1)) {
        __syncthreads();   // lg[] written by this block is visible to it after the barrier
        float mx = -INFINITY;
        for (int a = threadIdx.x; a <= P; a += 256) mx = fmaxf(mx, lg[a]);
        mx = block_max_256(mx, sh4);
        float den = 0.f;
        for (int a = threadIdx.x; a <= P; a += 256) den += expf(lg[a] - mx);
        den = block_sum_256(den, sh4);
        const float inv = 1.f / den;
        for (int a = threadIdx.x; a <= P; a += 256) lg[a] = expf(lg[a] - mx) * inv;
    }
}

// amap (pre-softmax action map) is needed in backward when softmax != 0; pass it as `amap` ([nh][B][P]).
// block-wide fp64 sum for blockDim.x == 256 (the bias gradients below are sums with heavy cancellation over sites / pixels)
__device__ __forceinline__ double block_sum_256_d(double v, double* sh4d) {
    v = wave_sum_d(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4d[w] = v;
    __syncthreads();
    return sh4d[0] + sh4d[1] + sh4d[2] + sh4d[3];
}

__global__ __launch_bounds__(256) void head_bwd_kernel(const float* dlogits, const float* damap, const float* dmu,
                                                       const float* dsigma2, const float* logits, const float* amap,
                                                       const float* sigma2,
                                                       const float* drt, int B, int Hm, int Wm, int ldz, int HC,
                                                       const float* w2, int softmax, float* dZ, float* dcb_partial,
                                                       float* dw2_partial, float* db2_partial, int dh, int dw, float* ddpre,
                                                       int zc, int parts, int* live, int64_t dl_ld) {
    // dl_ld: elements between the rows (head, sample) of dlogits (P + 1 when dense; a slice of the stacked outputs' gradient is read in place)
    // parts: as head_fwd_kernel.  parts == 1: dZ and dcb[0], dcb[1] (the other dcb entries zero), no duration gradients; parts == 2:
    // ddpre, the dw2 / db2 partials and dcb[IDX_BD] (the other entries zero), no dZ.  live (nullable, parts & 2): live[hd * B + b] = does this
    // (head slot, row) receive ANY duration gradient -- an exact test on dmu / dsigma2 -- so that the duration sites' backward kernels can
    // skip the slots whose ddpre is exactly zero (rows behind their last loss step, AiR's unselected head, the step right at a scanpath's end)
    __shared__ float sh4[4];
    __shared__ double sh4d[4];
    __shared__ float sdd[MAXS];
    const int hd = blockIdx.y, b = blockIdx.x, nh = gridDim.y;
    const int P = Hm * Wm, S = dh * dw;
    float* dcb = dcb_partial + ((int64_t)b * nh + hd) * HC;
    if (!(parts & 1)) {          // the duration part alone (one launch for all T decode steps: B = T x batch rows)
        const float dt0 = dmu[hd * B + b];
        const float dt1 = dsigma2[hd * B + b] * sigma2[hd * B + b];
        if (live && threadIdx.x == 0) live[hd * B + b] = (dt0 != 0.f || dt1 != 0.f) ? 1 : 0;
        double sumdd = 0.0;
        float* pw = dw2_partial + (((int64_t)b * nh + hd) * 2) * S;
        for (int s = threadIdx.x; s < S; s += 256) {
            const float dv = drt[((int64_t)hd * B + b) * S + s];
            const float dd = dv > 0.f ? dt0 * w2[s] + dt1 * w2[S + s] : 0.f;
            ddpre[((int64_t)hd * B + b) * S + s] = dd;
            sumdd += (double)dd;
            pw[s] = dt0 * dv;
            pw[S + s] = dt1 * dv;
        }
        sumdd = block_sum_256_d(sumdd, sh4d);
        if (threadIdx.x == 0) {
            db2_partial[((int64_t)b * nh + hd) * 2 + 0] = dt0;
            db2_partial[((int64_t)b * nh + hd) * 2 + 1] = dt1;
        }
        for (int j = threadIdx.x; j < HC; j += 256) dcb[j] = j == IDX_BD ? (float)sumdd : 0.f;
        return;
    }
    const float* dl = dlogits + ((int64_t)hd * B + b) * dl_ld;
    const float* lg = logits + ((int64_t)hd * B + b) * (P + 1);
    const float* am = amap + ((int64_t)hd * B + b) * P;
    float* dz = dZ + (int64_t)b * P * ldz + hd * zc;
    float dot = 0.f;
    if (softmax) {
        for (int a = threadIdx.x; a <= P; a += 256) dot += lg[a] * dl[a];
        dot = block_sum_256(dot, sh4);
    }
    const float d0 = softmax ? lg[0] * (dl[0] - dot) : dl[0];
    // duration branch gradients (parts & 2)
    const bool dur = (parts & 2) != 0;
    const float dt0 = dur ? dmu[hd * B + b] : 0.f;
    const float dt1 = dur ? dsigma2[hd * B + b] * sigma2[hd * B + b] : 0.f;
    if (dur && live && threadIdx.x == 0) live[hd * B + b] = (dt0 != 0.f || dt1 != 0.f) ? 1 : 0;
    double sumdd = 0.0;                  // fp64: per-site terms of either sign (drt_layer_1.bias, round 4 per-parameter bars)
    for (int s = threadIdx.x; s < S; s += 256) {
        const float dv = dur ? drt[((int64_t)hd * B + b) * S + s] : 0.f;
        const float dd = dv > 0.f ? dt0 * w2[s] + dt1 * w2[S + s] : 0.f;
        sdd[s] = dd;
        if (!dur) continue;
        if (ddpre) ddpre[((int64_t)hd * B + b) * S + s] = dd;
        sumdd += (double)dd;
        float* pw = dw2_partial + (((int64_t)b * nh + hd) * 2) * S;
        pw[s] = dt0 * dv;
        pw[S + s] = dt1 * dv;
    }
    sumdd = block_sum_256_d(sumdd, sh4d);   // also a barrier: sdd visible
    if (threadIdx.x == 0) {
        if (dur) {
            db2_partial[((int64_t)b * nh + hd) * 2 + 0] = dt0;
            db2_partial[((int64_t)b * nh + hd) * 2 + 1] = dt1;
        }
        dcb[0] = d0;
        dcb[IDX_BD] = (float)sumdd;
    }
    // composed-bias gradient of the taps: sum over sites where the tap is in range
    if (threadIdx.x < NTAP) {
        const int ky = threadIdx.x / 7, kx = threadIdx.x % 7;
        double s = 0.0;
        if (!ddpre && dur)
        for (int sy = 0; sy < dh; ++sy) {
            if ((unsigned)(5 * sy - 2 + ky) >= (unsigned)Hm) continue;
            for (int sx = 0; sx < dw; ++sx)
                if ((unsigned)(5 * sx - 2 + kx) < (unsigned)Wm) s += (double)sdd[sy * dw + sx];
        }
        dcb[2 + threadIdx.x] = (float)s;
    } else if (threadIdx.x > IDX_BD && threadIdx.x < HC) {
        dcb[threadIdx.x] = 0.f;
    }
    // dZ, coalesced over the HC columns of this head
    double s1 = 0.0;                     // fp64: sal_layer_3.bias = a sum over all pixels of the masked map gradient
    const float invP = 1.f / (float)P;
    for (int64_t i = threadIdx.x; i < (int64_t)P * zc; i += 256) {
        const int p = (int)(i / zc), j = (int)(i % zc);
        float v = 0.f;
        if (j == 0) {
            v = d0 * invP;
        } else if (j == 1) {
            float g = softmax ? lg[1 + p] * (dl[1 + p] - dot) : dl[1 + p];
            if (damap) g += damap[((int64_t)hd * B + b) * P + p];
            v = am[p] > 0.f ? g : 0.f;
            s1 += (double)v;
        } else if (j < 2 + NTAP && !ddpre && dur) {
            const int tap = j - 2, ky = tap / 7, kx = tap % 7;
            const int py = p / Wm, px = p % Wm;
            const int ty = py + 2 - ky, tx = px + 2 - kx;
            if (ty >= 0 && tx >= 0 && ty % 5 == 0 && tx % 5 == 0) {
                const int sy = ty / 5, sx = tx / 5;
                if (sy < dh && sx < dw) v = sdd[sy * dw + sx];
            }
        }
        dz[(int64_t)p * ldz + j] = v;
    }
    s1 = block_sum_256_d(s1, sh4d);
    if (threadIdx.x == 0) dcb[1] = (float)s1;
}

}  // namespace

extern "C" int sp_lstm_pointwise_fwd(const float* xg, const float* hg, const float* c_prev, int64_t rows, int C,
                                     float* gates, float* c_out, float* h_out, void* stream) {
    if (!xg || !gates || !c_out || !h_out) return SP_ENULL;
    if (C % 4) return SP_EINVAL;
    hipLaunchKernelGGL(lstm_fwd_kernel, dim3(ew_blocks(rows * C / 4)), dim3(256), 0, (hipStream_t)stream, xg, hg, c_prev,
                       rows, C, gates, c_out, h_out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_lstm_rank1_fwd(const float* xg, const float* hg, const float* c_prev, const float* spcol, const float* wc,
                                 int B, int P, int C, int KP, float* gates, float* c_out, float* h_out, unsigned* h_amax,
                                 void* stream) {
    if (!xg || !spcol || !wc || !gates || !c_out || !h_out) return SP_ENULL;
    if (C % 64 || B < 1 || P < 1 || KP < 1 || KP > 64) return SP_EINVAL;
    const size_t lds = (size_t)KP * (192 + 64) * sizeof(float);
    SP_RESET_AMAX(h_amax, stream);
    hipLaunchKernelGGL(lstm_rank1_fwd_kernel, dim3(C / 64, (P + R1_RB - 1) / R1_RB, B), dim3(256), lds, (hipStream_t)stream, xg, hg,
                       c_prev, spcol, wc, P, C, KP, gates, c_out, h_out, h_amax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int64_t sp_rank1_dwc_workspace(int B, int P, int N3, int KP) {
    return (int64_t)((P + R1_PCH - 1) / R1_PCH) * B * N3 * KP * (int64_t)sizeof(float);
}

extern "C" int sp_rank1_dwc(const float* dpre, const float* spcol, int B, int P, int ld, int N3, int KP, void* workspace,
                            float* dwc, void* stream) {
    if (!dpre || !spcol || !workspace || !dwc) return SP_ENULL;
    if (B < 1 || P < 1 || N3 < 4 || N3 % 4 || ld % 4 || N3 > ld || KP < 1 || KP > R1_MAXKP) return SP_EINVAL;
    const int nchunk = (P + R1_PCH - 1) / R1_PCH;
    hipLaunchKernelGGL(rank1_dwc_kernel, dim3((N3 + 255) / 256, nchunk, B), dim3(256), 0, (hipStream_t)stream, dpre, spcol, P, ld,
                       N3, KP, nchunk, (float*)workspace);
    SP_LAUNCH_CHECK();
    const int64_t n = (int64_t)B * N3 * KP;
    hipLaunchKernelGGL(rank1_dwc_reduce_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, n,
                       nchunk, dwc);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sempool_fwd_sbc(const float* a, const float* vf, int S, int B, int P, int C, float alpha, void* workspace, float* out, int sbc,
                                  void* stream);
extern "C" int sp_sempool_bwd_rows_sbc(const float* dout, const float* out, const float* a, const float* vf, int S, int B, int P, int C, float alpha,
                                       float* da, float* dvf, const int* row_last, int row_step, int sbc, void* stream);
extern "C" int64_t sp_sempool_workspace(int S, int B, int P, int C) {
    return (int64_t)((P + SP_PCH - 1) / SP_PCH) * B * S * C * (int64_t)sizeof(float);
}

extern "C" int sp_sempool_fwd(const float* a, const float* vf, int S, int B, int P, int C, float alpha, void* workspace,
                              float* out, void* stream) {
    return sp_sempool_fwd_sbc(a, vf, S, B, P, C, alpha, workspace, out, 0, stream);
}
extern "C" int sp_sempool_fwd_sbc(const float* a, const float* vf, int S, int B, int P, int C, float alpha, void* workspace,
                                  float* out, int sbc, void* stream) {
    if (!a || !vf || !workspace || !out) return SP_ENULL;
    if (S < 1 || S > 2 || B < 1 || P < 1 || C % 4 || C > 512) return SP_EINVAL;
    const int64_t osb = sbc ? C : (int64_t)S * C, oss = sbc ? (int64_t)B * C : C;
    const int nch = (P + SP_PCH - 1) / SP_PCH;
    hipLaunchKernelGGL(sempool_fwd_kernel, dim3(nch, B), dim3(256), 0, (hipStream_t)stream, a, vf, S, B, P, C, (float*)workspace);
    SP_LAUNCH_CHECK();
    const int64_t n = (int64_t)B * S * C;
    hipLaunchKernelGGL(sempool_finish_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nch,
                       n, alpha, out, S, C, osb, oss);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sempool_bwd_rows(const float* dout, const float* out, const float* a, const float* vf, int S, int B, int P, int C,
                                   float alpha, float* da, float* dvf, const int* row_last, int row_step, void* stream) {
    return sp_sempool_bwd_rows_sbc(dout, out, a, vf, S, B, P, C, alpha, da, dvf, row_last, row_step, 0, stream);
}
extern "C" int sp_sempool_bwd_rows_sbc(const float* dout, const float* out, const float* a, const float* vf, int S, int B, int P, int C,
                                       float alpha, float* da, float* dvf, const int* row_last, int row_step, int sbc, void* stream) {
    if (!dout || !out || !a || !vf || !da || !dvf) return SP_ENULL;
    if (S < 1 || S > 2 || B < 1 || P < 1 || C % 4) return SP_EINVAL;
    const int64_t zsb = sbc ? C : (int64_t)S * C, zss = sbc ? (int64_t)B * C : C;
    hipLaunchKernelGGL(sempool_bwd_kernel, dim3(ew_blocks((int64_t)B * P * 64)), dim3(256), 0, (hipStream_t)stream, dout, out, a,
                       vf, S, B, P, C, alpha, da, dvf, row_last, row_step, zsb, zss);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
extern "C" int sp_sempool_bwd(const float* dout, const float* out, const float* a, const float* vf, int S, int B, int P, int C,
                              float alpha, float* da, float* dvf, void* stream) {
    return sp_sempool_bwd_rows(dout, out, a, vf, S, B, P, C, alpha, da, dvf, nullptr, 0, stream);
}

extern "C" int sp_lstm_pointwise_bwd(const float* dh, const float* dc, const float* gates, const float* c_prev,
                                     const float* c_out, int64_t rows, int C, float* dpre, float* dc_prev, unsigned* dpre_amax,
                                     void* stream) {
    return sp_lstm_pointwise_bwd_split(dh, dc, gates, c_prev, c_out, rows, C, dpre, dc_prev, dpre_amax, nullptr, nullptr, nullptr,
                                       0.f, 0.f, nullptr, nullptr, stream);
}

extern "C" int sp_lstm_pointwise_bwd_rows(const float* dh, const float* dc, const float* gates, const float* c_prev,
                                          const float* c_out, int64_t rows, int C, float* dpre, float* dc_prev,
                                          unsigned* dpre_amax, unsigned* dcp_amax, const unsigned* dh_amax,
                                          const unsigned* dc_amax, float c_bound, float cprev_bound, void* planes,
                                          float* dpre_scale, const int* row_last, int row_step, int rows_per_sample, void* stream) {
    if (!gates || !c_out || (!dpre && !planes) || !dc_prev) return SP_ENULL;      // dpre may be NULL when the split form is written
    if (C % 4) return SP_EINVAL;
    if (planes && (!dpre_scale || C % 256 || ((uintptr_t)planes & 15) || (dh && !dh_amax) || (dc && !dc_amax))) return SP_EINVAL;
    if (row_last && (rows_per_sample < 1 || rows % rows_per_sample)) return SP_EINVAL;
    SP_RESET_AMAX(dpre_amax, stream);
    SP_RESET_AMAX(dcp_amax, stream);
    hipLaunchKernelGGL(lstm_bwd_kernel, dim3(ew_blocks(rows * C / 4)), dim3(256), 0, (hipStream_t)stream, dh, dc, gates,
                       c_prev, c_out, rows, C, dpre, dc_prev, dpre_amax, dcp_amax, dh_amax, dc_amax, c_bound, cprev_bound,
                       (uint16_t*)planes, dpre_scale, row_last, row_step, row_last ? rows_per_sample : 1);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_lstm_pointwise_bwd_split(const float* dh, const float* dc, const float* gates, const float* c_prev,
                                           const float* c_out, int64_t rows, int C, float* dpre, float* dc_prev,
                                           unsigned* dpre_amax, unsigned* dcp_amax, const unsigned* dh_amax,
                                           const unsigned* dc_amax, float c_bound, float cprev_bound, void* planes,
                                           float* dpre_scale, void* stream) {
    return sp_lstm_pointwise_bwd_rows(dh, dc, gates, c_prev, c_out, rows, C, dpre, dc_prev, dpre_amax, dcp_amax, dh_amax, dc_amax, c_bound,
                                      cprev_bound, planes, dpre_scale, nullptr, 0, 1, stream);
}

extern "C" int sp_im2col3x3_1ch(const float* maps, int R, int H, int W, int koff, int ldk, float* col, void* stream) {
    if (!maps || !col) return SP_ENULL;
    if (koff + 9 > ldk) return SP_EINVAL;
    hipLaunchKernelGGL(im2col1_kernel, dim3(ew_blocks((int64_t)R * H * W * 9)), dim3(256), 0, (hipStream_t)stream, maps, R, H,
                       W, koff, ldk, col);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_im2col3x3_multi(const float* maps, int S, int R, int H, int W, int ldk, float* col, void* stream) {
    if (!maps || !col) return SP_ENULL;
    if (S < 1 || 9 * S > ldk) return SP_EINVAL;
    hipLaunchKernelGGL(im2col1_multi_kernel, dim3(ew_blocks((int64_t)R * H * W * ldk)), dim3(256), 0, (hipStream_t)stream, maps, S, R, H, W, ldk, col);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
extern "C" int sp_col2im3x3_multi(const float* dcol, int S, int R, int H, int W, int ldk, float* dmaps, void* stream) {
    if (!dcol || !dmaps) return SP_ENULL;
    if (S < 1 || 9 * S > ldk) return SP_EINVAL;
    hipLaunchKernelGGL(col2im1_multi_kernel, dim3(ew_blocks((int64_t)S * R * H * W)), dim3(256), 0, (hipStream_t)stream, dcol, S, R, H, W, ldk, dmaps);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_col2im3x3_1ch(const float* dcol, int R, int H, int W, int koff, int ldk, float* dmaps, void* stream) {
    if (!dcol || !dmaps) return SP_ENULL;
    if (koff + 9 > ldk) return SP_EINVAL;
    hipLaunchKernelGGL(col2im1_kernel, dim3(ew_blocks((int64_t)R * H * W)), dim3(256), 0, (hipStream_t)stream, dcol, R, H, W,
                       koff, ldk, dmaps);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_listatt_fwd(const float* L, const float* u, int T, int R, int D, float* mem, float* alpha, void* stream) {
    if (!L || !u || !mem || !alpha) return SP_ENULL;
    if (T < 1 || T > MAXT) return SP_EINVAL;
    hipLaunchKernelGGL(listatt_fwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, L, u, T, R, D, mem, alpha);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_listatt_bwd(const float* dmem, const float* L, const float* u, const float* alpha, int T, int R, int D,
                              float* dL, float* du_partial, void* stream) {
    if (!dmem || !L || !u || !alpha || !dL || !du_partial) return SP_ENULL;
    if (T < 1 || T > MAXT) return SP_EINVAL;
    hipLaunchKernelGGL(listatt_bwd_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, dmem, L, u, alpha, T, R, D, dL,
                       du_partial);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_mulrelu_fwd(const float* a, const float* b, int64_t n, int64_t nb, float* out, void* stream) {
    if (!a || !b || !out) return SP_ENULL;
    hipLaunchKernelGGL(mulrelu_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n, nb, out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_mulrelu_bwd(const float* dout, const float* a, const float* b, const float* out, int64_t n, int64_t nb,
                              float* da, float* db_partial, void* stream) {
    if (!dout || !a || !b || !out || !da || !db_partial) return SP_ENULL;
    hipLaunchKernelGGL(mulrelu_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dout, a, b, out, n, nb, da,
                       db_partial);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_select_rows(const float* a, const float* b, const unsigned char* sel, int64_t rows, int64_t len, float* out,
                              void* stream) {
    if (!a || !b || !sel || !out) return SP_ENULL;
    hipLaunchKernelGGL(select_rows_kernel, dim3(ew_blocks(rows * len)), dim3(256), 0, (hipStream_t)stream, a, b, sel, rows, len,
                       out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_select_rows_bwd(const float* dout, const unsigned char* sel, int64_t rows, int64_t len, float* da, float* db,
                                  void* stream) {
    if (!dout || !sel || !da || !db) return SP_ENULL;
    hipLaunchKernelGGL(select_rows_bwd_kernel, dim3(ew_blocks(rows * len)), dim3(256), 0, (hipStream_t)stream, dout, sel, rows,
                       len, da, db);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_head_finish_parts_fwd(const float* Z, int B, int Hm, int Wm, int ldz, int nheads, int HC, const float* cb,
                                        int cb_per_sample, const float* w2, const float* b2, int softmax, float* logits, float* amap, float* mu,
                                        float* sigma2, float* drt, const float* dpre, int zc, int parts, void* stream) {
    if (parts < 1 || parts > 3 || B < 1 || nheads < 1) return SP_EINVAL;
    if (!cb) return SP_ENULL;
    if ((parts & 1) && (!Z || !logits || !amap)) return SP_ENULL;
    if ((parts & 2) && (!w2 || !b2 || !mu || !sigma2 || !drt)) return SP_ENULL;
    if (parts == 2 && !dpre) return SP_ENULL;          // without Z the duration sites come from sp_drt_direct_fwd
    const int dh = (Hm + 4 - 7) / 5 + 1, dw = (Wm + 4 - 7) / 5 + 1;
    if (zc <= 0) zc = HC;
    if (HC < 52 || dh * dw > MAXS || dh < 1 || dw < 1 || zc < 2 || ((parts & 2) && !dpre && zc < 2 + NTAP) || ((parts & 1) && nheads * zc > ldz))
        return SP_EINVAL;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(B, nheads), dim3(256), 0, (hipStream_t)stream, Z, B, Hm, Wm, ldz, HC, cb,
                       cb_per_sample, w2, b2,
                       softmax, logits, amap, mu, sigma2, drt, dh, dw, dpre, zc, parts);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
extern "C" int sp_head_finish_fwd(const float* Z, int B, int Hm, int Wm, int ldz, int nheads, int HC, const float* cb,
                                  int cb_per_sample, const float* w2, const float* b2, int softmax, float* logits, float* amap, float* mu,
                                  float* sigma2, float* drt, const float* dpre, int zc, void* stream) {
    return sp_head_finish_parts_fwd(Z, B, Hm, Wm, ldz, nheads, HC, cb, cb_per_sample, w2, b2, softmax, logits, amap, mu, sigma2, drt, dpre, zc, 3,
                                    stream);
}

extern "C" int sp_head_finish_parts_bwd_ld(const float* dlogits, int64_t dlogits_ld, const float* damap, const float* dmu, const float* dsigma2,
                                           const float* logits, const float* amap, const float* sigma2, const float* drt, int B, int Hm, int Wm,
                                           int ldz, int nheads, int HC, const float* w2, int softmax, float* dZ, float* dcb_partial,
                                           float* dw2_partial, float* db2_partial, float* ddpre, int zc, int parts, int* live, void* stream);
extern "C" int sp_head_finish_parts_bwd(const float* dlogits, const float* damap, const float* dmu, const float* dsigma2,
                                        const float* logits,
                                        const float* amap, const float* sigma2, const float* drt, int B, int Hm, int Wm, int ldz,
                                        int nheads, int HC, const float* w2, int softmax, float* dZ, float* dcb_partial,
                                        float* dw2_partial, float* db2_partial, float* ddpre, int zc, int parts, int* live, void* stream) {
    return sp_head_finish_parts_bwd_ld(dlogits, 0, damap, dmu, dsigma2, logits, amap, sigma2, drt, B, Hm, Wm, ldz, nheads, HC, w2, softmax, dZ,
                                       dcb_partial, dw2_partial, db2_partial, ddpre, zc, parts, live, stream);
}
extern "C" int sp_head_finish_parts_bwd_ld(const float* dlogits, int64_t dlogits_ld, const float* damap, const float* dmu, const float* dsigma2,
                                           const float* logits, const float* amap, const float* sigma2, const float* drt, int B, int Hm, int Wm,
                                           int ldz, int nheads, int HC, const float* w2, int softmax, float* dZ, float* dcb_partial,
                                           float* dw2_partial, float* db2_partial, float* ddpre, int zc, int parts, int* live, void* stream) {
    if (parts < 1 || parts > 3 || B < 1 || nheads < 1) return SP_EINVAL;
    if (dlogits_ld == 0) dlogits_ld = (int64_t)Hm * Wm + 1;
    if (dlogits_ld < (int64_t)Hm * Wm + 1) return SP_EINVAL;
    if (!dcb_partial) return SP_ENULL;
    if ((parts & 1) && (!dlogits || !logits || !amap || !dZ)) return SP_ENULL;
    if ((parts & 2) && (!dmu || !dsigma2 || !sigma2 || !drt || !w2 || !dw2_partial || !db2_partial)) return SP_ENULL;
    if (parts == 2 && !ddpre) return SP_ENULL;
    const int dh = (Hm + 4 - 7) / 5 + 1, dw = (Wm + 4 - 7) / 5 + 1;
    if (zc <= 0) zc = HC;
    if (HC < 52 || dh * dw > MAXS || dh < 1 || dw < 1 || zc < 2 || ((parts & 2) && !ddpre && zc < 2 + NTAP) || ((parts & 1) && nheads * zc > ldz))
        return SP_EINVAL;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(B, nheads), dim3(256), 0, (hipStream_t)stream, dlogits, damap, dmu, dsigma2, logits,
                       amap,
                       sigma2, drt, B, Hm, Wm, ldz, HC, w2, softmax, dZ, dcb_partial, dw2_partial, db2_partial, dh, dw, ddpre, zc, parts, live,
                       dlogits_ld);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
extern "C" int sp_head_finish_bwd(const float* dlogits, const float* damap, const float* dmu, const float* dsigma2,
                                  const float* logits,
                                  const float* amap, const float* sigma2, const float* drt, int B, int Hm, int Wm, int ldz,
                                  int nheads, int HC, const float* w2, int softmax, float* dZ, float* dcb_partial,
                                  float* dw2_partial, float* db2_partial, float* ddpre, int zc, void* stream) {
    return sp_head_finish_parts_bwd(dlogits, damap, dmu, dsigma2, logits, amap, sigma2, drt, B, Hm, Wm, ldz, nheads, HC, w2, softmax, dZ,
                                    dcb_partial, dw2_partial, db2_partial, ddpre, zc, 3, nullptr, stream);
}
