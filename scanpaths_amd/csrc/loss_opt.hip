// Supervised loss (value + gradient in one pass) and the fused clip_grad_norm_ + Adam step.
//   CrossEntropyLoss          AiR/models/loss.py:10-14   (soft targets, eps inside the log, masked mean)
//   MLPLogNormalDistribution  AiR/models/loss.py:27-32
//   loss = L_a + lambda_1 L_d AiR/train.py:192-197
//   clip_grad_norm_(12.5) + Adam(lr, betas, eps, L2 weight decay)   AiR/train.py:116-117,200-202
#include "common.h"
#include <algorithm>

namespace {

constexpr float EPS = 1e-7f;

// one block per (b,t) row of the [B*T][A] logits
__global__ __launch_bounds__(256) void loss_rows_kernel(const float* z, const float* gt, const float* amask, const float* mu,
                                                        const float* sigma2, const float* dur, const float* dmask, int A,
                                                        float lambda1, const float* mask_sums, float* row_la,
                                                        float* row_ld, float* dz, float* dmu, float* dsigma2) {
    __shared__ float sh4[4];
    const int row = blockIdx.x;
    const float* zr = z + (int64_t)row * A;
    const float* gr = gt + (int64_t)row * A;
    float* dzr = dz + (int64_t)row * A;
    const float inv_a = 1.f / mask_sums[0], inv_d = 1.f / mask_sums[1];
    const float am = amask[row];

    float mx = -INFINITY;
    for (int a = threadIdx.x; a < A; a += 256) mx = fmaxf(mx, zr[a]);
    mx = block_max_256(mx, sh4);
    float den = 0.f;
    for (int a = threadIdx.x; a < A; a += 256) den += expf(zr[a] - mx);
    den = block_sum_256(den, sh4);
    const float inv = 1.f / den;
    float la = 0.f, qs = 0.f;
    for (int a = threadIdx.x; a < A; a += 256) {
        const float p = expf(zr[a] - mx) * inv;
        const float g = gr[a];
        if (g != 0.f) {
            la -= g * logf(p + EPS);
            qs += g * p / (p + EPS);
        }
    }
    la = block_sum_256(la, sh4);
    qs = block_sum_256(qs, sh4);
    const float w = am * inv_a;
    for (int a = threadIdx.x; a < A; a += 256) {
        const float p = expf(zr[a] - mx) * inv;
        const float q = gr[a] * p / (p + EPS);
        dzr[a] = w * (p * qs - q);
    }
    if (threadIdx.x == 0) {
        row_la[row] = la * am;
        float ld = 0.f, gm = 0.f, gs = 0.f;
        if (dmask[row] == 1.f) {
            const float d = dur[row], s2 = sigma2[row];
            const float Lg = logf(d + EPS);
            const float diff = Lg - mu[row];
            const float logpdf = logf(1.f / (d + EPS) * 1.f / sqrtf(2.f * 3.14159265358979323846f * s2)) -
                                 diff * diff / (2.f * s2);
            ld = -logpdf;
            gm = -lambda1 * inv_d * (diff / s2);
            gs = -lambda1 * inv_d * (-0.5f / s2 + diff * diff / (2.f * s2 * s2));
        }
        row_ld[row] = ld;
        dmu[row] = gm;
        dsigma2[row] = gs;
    }
}

__global__ void loss_final_kernel(const float* row_la, const float* row_ld, int rows, float lambda1, const float* mask_sums,
                                  float* out3) {
    // single thread, fixed order: deterministic
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double a = 0.0, d = 0.0;
        for (int i = 0; i < rows; ++i) {
            a += (double)row_la[i];
            d += (double)row_ld[i];
        }
        const float la = (float)(a / (double)mask_sums[0]);
        const float ld = (float)(d / (double)mask_sums[1]);
        out3[0] = la + lambda1 * ld;
        out3[1] = la;
        out3[2] = ld;
    }
}

// generic deterministic sum: partials per block, then one block finishes
template <typename TIn, bool SQUARE>
__global__ __launch_bounds__(256) void reduce_partial(const TIn* x, int64_t n, double* partial) {
    __shared__ double shd[4];
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = (double)x[i];
        s += SQUARE ? v * v : v;
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) shd[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = shd[0] + shd[1] + shd[2] + shd[3];
}
__global__ void reduce_final_d(const double* partial, int nblk, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nblk; ++i) s += partial[i];
        *out = s;
    }
}
__global__ void reduce_final_f(const double* partial, int nblk, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < nblk; ++i) s += partial[i];
        *out = (float)s;
    }
}

inline int red_blocks(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(n, 1024), 1024)); }

__global__ __launch_bounds__(256) void clip_adam_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                        const double* sumsq, float gscale, float clip, float lr, float beta1,
                                                        float beta2, float eps, float wd, float bc1, float bc2) {
    const float norm = (float)(sqrt(*sumsq) * (double)gscale);
    float coef = 1.f;
    if (clip > 0.f) coef = fminf(1.f, clip / (norm + 1e-6f));
    const float gs = coef * gscale;
    const float step = lr / bc1;
    const float rs = 1.f / sqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] * gs + wd * pi;
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi) * rs + eps);
    }
}

__global__ __launch_bounds__(256) void scale_by_kernel(const float* x, const float* sc, int64_t n, float* out) {
    const float k = *sc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = x[i] * k;
}

// ---- RL (self-critical) log-probabilities, models/loss.py:34-45 -- value and per-element derivative in one pass ----------
// LogAction:   out[b] = sum_t log(p[b,t] + eps) * mask[b,t] / sum(mask)                       dcoef = d out[b] / d p[b,t]
// LogDuration: item = log(1/(d+eps) * 1/sqrt(2 pi s2)) - (log(d+eps) - mu)^2 / (2 s2);  out[b] = sum_t item * mask / sum(mask)
// msum: device scalar sum(mask) over the WHOLE tensor (the reference divides every row by the global mask sum).
__global__ __launch_bounds__(64) void log_action_kernel(const float* p, const float* mask, int T, const float* msum, float* out,
                                                        float* dcoef) {
    const int b = blockIdx.x;
    const float inv = 1.f / msum[0];
    float acc = 0.f;
    for (int t = threadIdx.x; t < T; t += 64) {
        const float v = p[(int64_t)b * T + t], m = mask[(int64_t)b * T + t];
        acc += logf(v + EPS) * m;
        dcoef[(int64_t)b * T + t] = m / (v + EPS) * inv;
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[b] = acc * inv;
}

__global__ __launch_bounds__(64) void log_duration_kernel(const float* d, const float* mu, const float* s2, const float* mask,
                                                          int T, const float* msum, float* out, float* dmu, float* ds2) {
    const int b = blockIdx.x;
    const float inv = 1.f / msum[0];
    float acc = 0.f;
    for (int t = threadIdx.x; t < T; t += 64) {
        const int64_t i = (int64_t)b * T + t;
        const float m = mask[i], sg = s2[i];
        const float Lg = logf(d[i] + EPS), r = Lg - mu[i];
        const float item = logf(1.f / (d[i] + EPS) * 1.f / sqrtf(2.f * 3.14159265358979323846f * sg)) - r * r / (2.f * sg);
        acc += item * m;
        dmu[i] = m * inv * (r / sg);
        ds2[i] = m * inv * (-0.5f / sg + r * r / (2.f * sg * sg));
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) out[b] = acc * inv;
}

// out[b,t] = coef[b,t] * g[b]
__global__ __launch_bounds__(256) void rowscale_kernel(const float* coef, const float* g, int64_t n, int T, float* out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = coef[i] * g[i / T];
}

// last[b] = the last decode step at which ANY output gradient of sample b is non-zero (NaN counts as non-zero); -1: none.
// One block per sample; steps are visited from the last one down and the block stops at the first live step, so a sample whose
// scanpath fills all T steps costs one row per tensor and a sample without any loss term reads its rows once.
struct RowsLastList {
    const float* g[8];
    long long row_len[8];
    int count;
};

__global__ __launch_bounds__(256) void rows_last_kernel(RowsLastList l, int nstack, int B, int T, int* __restrict__ last) {
    const int b = blockIdx.x;
    int t = T - 1;
    for (; t >= 0; --t) {
        int live = 0;
        for (int k = 0; k < l.count; ++k) {
            const long long rl = l.row_len[k];
            for (int h = 0; h < nstack; ++h) {
                const float* row = l.g[k] + (((long long)h * B + b) * T + t) * rl;
                for (long long i = threadIdx.x; i < rl; i += 256) live |= (row[i] != 0.f);
            }
        }
        if (__syncthreads_or(live)) break;
    }
    if (threadIdx.x == 0) last[b] = t;
}

}  // namespace

extern "C" int sp_rows_last(const float* const* grads, const int64_t* row_len, int count, int nstack, int B, int T, int* last,
                            void* stream) {
    if (!grads || !row_len || !last) return SP_ENULL;
    if (count < 1 || count > 8 || nstack < 1 || B < 1 || T < 1) return SP_EINVAL;
    RowsLastList l;
    l.count = 0;
    for (int k = 0; k < count; ++k) {
        if (!grads[k]) continue;                  // a gradient that never arrived is a zero gradient
        if (row_len[k] < 1) return SP_EINVAL;
        l.g[l.count] = grads[k];
        l.row_len[l.count++] = row_len[k];
    }
    hipLaunchKernelGGL(rows_last_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, l, nstack, B, T, last);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_scale_by(const float* x, const float* scale, int64_t n, float* out, void* stream) {
    if (!x || !scale || !out) return SP_ENULL;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(n, 256), 2048));
    hipLaunchKernelGGL(scale_by_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, scale, n, out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_abi_version(void) { return SP_ABI_VERSION; }

int sp_tuning_values[SP_TUNE_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1};

extern "C" int sp_set_tuning(const char* name, int value) {
    if (!name) return SP_ENULL;
    const char* names[SP_TUNE_COUNT] = {"amax_reset", "hw_splits", "h2_halo", "h2_dbg", "hw_dbg", "b3_dbg", "row_order", "hw_cap"};
    for (int i = 0; i < SP_TUNE_COUNT; ++i) {
        const char *a = names[i], *b = name;
        while (*a && *a == *b) { ++a; ++b; }
        if (!*a && !*b) {
#ifndef SP_TIMING_VARIANTS
            if (i != SP_TUNE_AMAX_RESET) return SP_EINVAL;      // schedule / timing selectors exist in the timing build only
#endif
            sp_tuning_values[i] = value;
            return SP_OK;
        }
    }
    return SP_EINVAL;
}

// 1 in libscanpaths_amd_timing.so (wrong-result timing modes and A/B switches compiled in), 0 in the product library
extern "C" int sp_timing_build(void) {
#ifdef SP_TIMING_VARIANTS
    return 1;
#else
    return 0;
#endif
}

extern "C" int64_t sp_scanpath_loss_workspace(int B, int T) { return 2 * (int64_t)B * T * (int64_t)sizeof(float); }

extern "C" int sp_scanpath_loss(const float* z, const float* gt, const float* amask, const float* mu, const float* sigma2,
                                const float* dur, const float* dmask, int B, int T, int A, float lambda1,
                                const float* mask_sums, float* out3, float* dz, float* dmu, float* dsigma2, void* workspace,
                                void* stream) {
    if (!z || !gt || !amask || !mu || !sigma2 || !dur || !dmask || !mask_sums || !out3 || !dz || !dmu || !dsigma2 ||
        !workspace)
        return SP_ENULL;
    hipStream_t s = (hipStream_t)stream;
    float* row_la = (float*)workspace;
    float* row_ld = row_la + (int64_t)B * T;
    hipLaunchKernelGGL(loss_rows_kernel, dim3(B * T), dim3(256), 0, s, z, gt, amask, mu, sigma2, dur, dmask, A, lambda1,
                       mask_sums, row_la, row_ld, dz, dmu, dsigma2);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, s, row_la, row_ld, B * T, lambda1, mask_sums, out3);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int64_t sp_sumsq_workspace(int64_t n) { return (int64_t)red_blocks(n) * (int64_t)sizeof(double); }

extern "C" int sp_sumsq(const float* g, int64_t n, double* out, void* workspace, void* stream) {
    if (!g || !out || !workspace) return SP_ENULL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = red_blocks(n);
    hipLaunchKernelGGL((reduce_partial<float, true>), dim3(nb), dim3(256), 0, s, g, n, (double*)workspace);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_final_d, dim3(1), dim3(64), 0, s, (const double*)workspace, nb, out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sum(const float* x, int64_t n, float* out, void* workspace, void* stream) {
    if (!x || !out || !workspace) return SP_ENULL;
    hipStream_t s = (hipStream_t)stream;
    const int nb = red_blocks(n);
    hipLaunchKernelGGL((reduce_partial<float, false>), dim3(nb), dim3(256), 0, s, x, n, (double*)workspace);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_final_f, dim3(1), dim3(64), 0, s, (const double*)workspace, nb, out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float gscale,
                            float clip, float lr, float beta1, float beta2, float eps, float weight_decay, float bc1,
                            float bc2, void* stream) {
    if (!p || !g || !m || !v || !sumsq) return SP_ENULL;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(n, 256), 4096));
    hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, gscale, clip,
                       lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_log_action(const float* p, const float* mask, int B, int T, const float* mask_sum, float* out, float* dcoef,
                             void* stream) {
    if (!p || !mask || !mask_sum || !out || !dcoef) return SP_ENULL;
    if (B < 1 || T < 1) return SP_EINVAL;
    hipLaunchKernelGGL(log_action_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, p, mask, T, mask_sum, out, dcoef);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_log_duration(const float* d, const float* mu, const float* sigma2, const float* mask, int B, int T,
                               const float* mask_sum, float* out, float* dmu, float* dsigma2, void* stream) {
    if (!d || !mu || !sigma2 || !mask || !mask_sum || !out || !dmu || !dsigma2) return SP_ENULL;
    if (B < 1 || T < 1) return SP_EINVAL;
    hipLaunchKernelGGL(log_duration_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, d, mu, sigma2, mask, T, mask_sum, out, dmu,
                       dsigma2);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_rowscale(const float* coef, const float* g, int B, int T, float* out, void* stream) {
    if (!coef || !g || !out) return SP_ENULL;
    if (B < 1 || T < 1) return SP_EINVAL;
    const int64_t n = (int64_t)B * T;
    hipLaunchKernelGGL(rowscale_kernel, dim3((unsigned)std::min<int64_t>(sp_cdiv(n, 256), 2048)), dim3(256), 0, (hipStream_t)stream,
                       coef, g, n, T, out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
