// Post-hoc scanpath sampling on the device (reference models/sampling.py:16-77).
//   random_sample   : per (b,t) one categorical draw from the step's action distribution with the terminate action masked
//                     for t < min_length (:18-21), the probability of the chosen action gathered from the UNMASKED
//                     distribution (:22-23), a log-normal duration exp(eps*sigma2 + mu) (sigma2 used as the scale -- the
//                     reference's quirk, :26-27), and the first-terminate scan for the scanpath length (:29-34, incl. the
//                     "terminate at t=0 -> T" quirk).
//   generate_scanpath: index -> pixel mapping ((a-1)%Wm+0.5)*W/Wm, ((a-1)/Wm+0.5)*H/Hm, masks (:48-77).
// RNG: Philox4x32-10 keyed by (seed), counter = (row, draw) -- reproducible for a given seed on any device count; the
// stream necessarily differs from torch's (the reference's CPU and CUDA streams differ from each other too).
#include "common.h"

namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0, h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    c[0] = h1 ^ c[1] ^ k0;
    c[1] = l1;
    c[2] = h0 ^ c[3] ^ k1;
    c[3] = l0;
}
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0,1)

// one block per (b,t) row
__global__ __launch_bounds__(256) void sample_rows_kernel(const float* probs, const float* mu, const float* sigma2, int T, int A,
                                                          int min_length, uint64_t seed, int64_t* actions, float* aprob,
                                                          float* durations) {
    __shared__ float part[256];
    __shared__ float sh4[4];
    const int row = blockIdx.x;
    const int t = row % T;
    const float* p = probs + (int64_t)row * A;
    const int a_lo = (t < min_length) ? 1 : 0;                 // terminate action masked for the first min_length steps
    const int per = (A + 255) / 256;
    const int s0 = threadIdx.x * per, s1 = min(A, s0 + per);
    float s = 0.f;
    for (int a = max(s0, a_lo); a < s1; ++a) s += p[a];
    part[threadIdx.x] = s;
    const float total = block_sum_256(s, sh4);
    uint32_t c[4] = {(uint32_t)row, (uint32_t)((uint64_t)row >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    if (threadIdx.x == 0) {
        const float target = u01(c[0]) * total;
        float run = 0.f;
        int seg = 255;
        for (int i = 0; i < 256; ++i) {                          // coarse search over the 256 segment sums
            if (run + part[i] >= target) { seg = i; break; }
            run += part[i];
        }
        int chosen = -1;
        const int b0 = seg * per, b1 = min(A, b0 + per);
        for (int a = max(b0, a_lo); a < b1; ++a) {
            run += p[a];
            if (run >= target && p[a] > 0.f) { chosen = a; break; }
        }
        if (chosen < 0) {                                        // rounding fell off the end: last positive entry
            for (int a = A - 1; a >= a_lo; --a)
                if (p[a] > 0.f) { chosen = a; break; }
            if (chosen < 0) chosen = a_lo;
        }
        actions[row] = chosen;
        aprob[row] = p[chosen];
        // Box-Muller on two more words of the same Philox block
        const float r = sqrtf(-2.f * logf(u01(c[1]))), th = 6.283185307179586f * u01(c[2]);
        const float eps = r * cosf(th);
        durations[row] = expf(eps * sigma2[row] + mu[row]);
    }
}

// one thread per sample: first-terminate scan, masks, pixel coordinates
__global__ void scanpath_kernel(const int64_t* actions, const float* durations, int B, int T, int map_w, float xg, float yg,
                                float* length, float* amask, float* dmask, float* fix /* [B][T][3] */, int* nfix) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int len = 0;
    for (int t = 0; t < T; ++t)                                   // reference scan: first index t > 0 ... else T
        if (len == 0 && actions[(int64_t)b * T + t] == 0) len = t;
    length[b] = (float)(len == 0 ? T : len);
    int n = 0;
    bool done = false;
    for (int t = 0; t < T; ++t) {
        const int64_t i = (int64_t)b * T + t;
        float am = 0.f, dm = 0.f, x = 0.f, y = 0.f, d = 0.f;
        if (!done) {
            am = 1.f;
            const int64_t a = actions[i];
            if (a == 0) {
                done = true;
            } else {
                const int64_t idx = a - 1;
                x = (float)(idx % map_w) * xg + 0.5f * xg;
                y = (float)(idx / map_w) * yg + 0.5f * yg;
                d = durations[i];
                dm = 1.f;
                ++n;
            }
        }
        amask[i] = am;
        dmask[i] = dm;
        fix[i * 3 + 0] = x;
        fix[i * 3 + 1] = y;
        fix[i * 3 + 2] = d;
    }
    nfix[b] = n;
}

// ---- beam search over the per-step action distributions (BASELINE.json config 5: "beam-4 scanpath sampling") -------------
// The reference only samples (models/sampling.py:16-46); beam search is a build-side decoder over the SAME eval-mode outputs.
// The per-step distributions do not depend on earlier choices (the decoder is not conditioned on sampled actions), so the K
// best sequences under  score = sum_t log p_t(a_t)  with "terminate (action 0) ends the sequence, allowed from t >= min_length"
// are found exactly by a width-K beam: at step t every live beam is extended by the step's K most probable allowed actions,
// finished beams compete with their final score, the best K survive.  Ties: higher score, then the earlier candidate in the
// order (beam 0..K-1; finished beam itself, else actions by descending probability then ascending index).
// One 256-thread block per sample.  Scores are accumulated in float64.
constexpr int BEAM_MAX = 8;

__global__ __launch_bounds__(256) void beam_kernel(const float* __restrict__ probs, int T, int A, int min_length, int K,
                                                   int64_t* __restrict__ actions /* [B][K][T] */, double* __restrict__ scores) {
    __shared__ float tv[BEAM_MAX];
    __shared__ int ti[BEAM_MAX];
    __shared__ float rv[256];
    __shared__ int ri[256];
    __shared__ double bscore[BEAM_MAX];
    __shared__ int bdone[BEAM_MAX];
    __shared__ int nlive;
    __shared__ int16_t seq[2][BEAM_MAX][64];                 // T <= 64
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* P = probs + (int64_t)b * T * A;
    if (tid == 0) {
        nlive = 1;                                           // one empty beam
        bscore[0] = 0.0;
        bdone[0] = 0;
    }
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < T; ++t) {
        const float* p = P + (int64_t)t * A;
        const int a_lo = t < min_length ? 1 : 0;
        // K rounds of block arg-max (largest probability, lowest index on ties), skipping the already chosen entries
        for (int k = 0; k < K; ++k) {
            float bv = -1.f;
            int bi = 0x7fffffff;
            for (int a = a_lo + tid; a < A; a += 256) {
                bool used = false;
                for (int q = 0; q < k; ++q) used |= (ti[q] == a);
                const float v = p[a];
                if (!used && (v > bv || (v == bv && a < bi))) { bv = v; bi = a; }
            }
            rv[tid] = bv;
            ri[tid] = bi;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if (tid < o) {
                    const float v2 = rv[tid + o];
                    const int i2 = ri[tid + o];
                    if (v2 > rv[tid] || (v2 == rv[tid] && i2 < ri[tid])) { rv[tid] = v2; ri[tid] = i2; }
                }
                __syncthreads();
            }
            if (tid == 0) { tv[k] = rv[0]; ti[k] = ri[0]; }
            __syncthreads();
        }
        if (tid == 0) {
            // candidates in tie-break order
            double cs[BEAM_MAX * (BEAM_MAX + 1)];
            int cb[BEAM_MAX * (BEAM_MAX + 1)], ca[BEAM_MAX * (BEAM_MAX + 1)];
            int nc = 0;
            for (int q = 0; q < nlive; ++q) {
                if (bdone[q]) { cs[nc] = bscore[q]; cb[nc] = q; ca[nc] = -1; ++nc; continue; }
                for (int k = 0; k < K; ++k) {
                    if (ti[k] == 0x7fffffff || !(tv[k] > 0.f)) continue;          // fewer than K positive entries
                    cs[nc] = bscore[q] + log((double)tv[k]);
                    cb[nc] = q;
                    ca[nc] = ti[k];
                    ++nc;
                }
            }
            const int nxt = cur ^ 1;
            int taken[BEAM_MAX];
            int nn = 0;
            for (; nn < K && nn < nc; ++nn) {
                int best = -1;
                for (int c = 0; c < nc; ++c) {
                    bool used = false;
                    for (int q = 0; q < nn; ++q) used |= (taken[q] == c);
                    if (!used && (best < 0 || cs[c] > cs[best])) best = c;
                }
                taken[nn] = best;
            }
            double ns[BEAM_MAX];
            int nd[BEAM_MAX];
            for (int q = 0; q < nn; ++q) {
                const int c = taken[q];
                for (int u = 0; u < t; ++u) seq[nxt][q][u] = seq[cur][cb[c]][u];
                if (ca[c] < 0) {                                  // finished earlier: stays finished, padded with terminate
                    seq[nxt][q][t] = 0;
                    nd[q] = 1;
                } else {
                    seq[nxt][q][t] = (int16_t)ca[c];
                    nd[q] = ca[c] == 0;
                }
                ns[q] = cs[c];
            }
            for (int q = 0; q < nn; ++q) { bscore[q] = ns[q]; bdone[q] = nd[q]; }
            nlive = nn;
        }
        __syncthreads();
        cur ^= 1;
    }
    for (int i = tid; i < K * T; i += 256) {
        const int q = i / T, t = i % T;
        actions[((int64_t)b * K + q) * T + t] = q < nlive ? (int64_t)seq[cur][q][t] : 0;
    }
    if (tid < K) scores[(int64_t)b * K + tid] = tid < nlive ? bscore[tid] : -INFINITY;
}

// ---- dataset targets (AiR/dataset/dataset.py:111-147, blur_sigma = None): ragged fixations -> soft one-hot targets, masks ----
// one thread per (sample, step).  Arithmetic as numpy >= 2 evaluates the reference's expressions (float32 scalars with weak
// python floats): cell = int32(float32(x) / float32(origin / map)), duration = (float32(T_end) - float32(T_start)) / 1000f.
// f64_div = 1 evaluates the cell divisions in float64 instead (what numpy 1.x -- the reference's pinned environment -- does).
__global__ void collate_kernel(const float* X, const float* Y, const float* Ts, const float* Te, const int64_t* start,
                               const int* count, const double* origin_w, const double* origin_h, int B, int T, int map_h, int map_w,
                               int f64_div, float* target /* [B][T][1+map_h*map_w], pre-zeroed */, float* duration, float* amask,
                               float* dmask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i % T;
    const int n = min(count[b], T);
    const int A = 1 + map_h * map_w;
    float d = 0.f, am = 0.f, dm = 0.f;
    int px = -1, py = -1;
    if (t < n) {
        const int64_t s = start[b] + t;
        const double dsx = origin_w[b] / (double)map_w, dsy = origin_h[b] / (double)map_h;
        if (f64_div) {
            px = (int)((double)X[s] / dsx);
            py = (int)((double)Y[s] / dsy);
        } else {
            px = (int)(X[s] / (float)dsx);
            py = (int)(Y[s] / (float)dsy);
        }
        // numpy 1.x (the reference's pinned 1.19.2): float32 scalar / python float is evaluated in float64, then stored as float32
        d = f64_div ? (float)((double)(Te[s] - Ts[s]) / 1000.0) : (Te[s] - Ts[s]) / 1000.0f;
        am = 1.f;
        dm = 1.f;
    } else if (t == n) {
        am = 1.f;                                           // the step after the last fixation (:136-137)
    }
    duration[i] = d;
    amask[i] = am;
    dmask[i] = dm;
    float* row = target + (int64_t)i * A;
    if (px == -1 || py == -1) row[0] = 1.f;                  // terminate target (:141-142)
    else if ((unsigned)px < (unsigned)map_w && (unsigned)py < (unsigned)map_h)
        row[1 + py * map_w + px] = 1.f;                      // (:144-147)
    // a fixation outside the image (the reference raises IndexError there) leaves the step without a target
}

}  // namespace

extern "C" int sp_beam_search(const float* probs, int B, int T, int A, int min_length, int K, int64_t* actions, double* scores,
                              void* stream) {
    if (!probs || !actions || !scores) return SP_ENULL;
    if (B < 1 || T < 1 || T > 64 || A < 2 || A > 32767 || K < 1 || K > BEAM_MAX) return SP_EINVAL;
    hipLaunchKernelGGL(beam_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, probs, T, A, min_length, K, actions, scores);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// blur_sigma targets (AiR/dataset/dataset.py:144-147): every non-terminate step's one-hot map is replaced by
//   scipy.ndimage.gaussian_filter(map, sigma) / sum   (mode 'reflect', truncate 4.0: radius = int(4 sigma + 0.5)).
// The filter of a delta is separable: pass 1 along y (double sum, rounded to float32 as scipy stores the intermediate), pass 2
// along x (double sum of w[j] * float32 intermediate, rounded to float32), then the division by the float32 total.  One block
// per (sample, step); the hot cell is found by scanning the row.  scipy sums the normaliser pairwise in float32, here it is a
// double sum rounded once: agreement to ~1 ulp of float32, not bitwise (tests hold 2e-7 relative).
constexpr int BLUR_MAXR = 64;
__global__ __launch_bounds__(256) void blur_targets_kernel(float* target, int A, int map_h, int map_w, double sigma) {
    __shared__ double w[2 * BLUR_MAXR + 1];
    __shared__ float col[1024], rowv[1024];
    __shared__ int hot;
    __shared__ double red[256];
    float* row = target + (int64_t)blockIdx.x * A;
    const int t = threadIdx.x;
    if (t == 0) hot = -1;
    __syncthreads();
    for (int i = 1 + t; i < A; i += 256)
        if (row[i] == 1.f) hot = i - 1;
    __syncthreads();
    if (hot < 0) return;                                  // terminate target or a fixation outside the map: nothing to blur
    const int y0 = hot / map_w, x0 = hot % map_w;
    const int r = (int)(4.0 * sigma + 0.5);
    if (t == 0) {
        double sum = 0.0;
        for (int j = 0; j <= 2 * r; ++j) {
            const double x = (double)(j - r);
            w[j] = exp(-0.5 / (sigma * sigma) * x * x);
            sum += w[j];
        }
        for (int j = 0; j <= 2 * r; ++j) w[j] /= sum;
    }
    __syncthreads();
    auto reflect = [](int i, int n) {                     // scipy 'reflect': (d c b a | a b c d | d c b a)
        const int p2 = 2 * n;
        i = ((i % p2) + p2) % p2;
        return i < n ? i : p2 - 1 - i;
    };
    for (int y = t; y < map_h; y += 256) {
        double s = 0.0;
        for (int j = 0; j <= 2 * r; ++j)
            if (reflect(y + j - r, map_h) == y0) s += w[j];
        col[y] = (float)s;
    }
    for (int x = t; x < map_w; x += 256) {
        double s = 0.0;
        for (int j = 0; j <= 2 * r; ++j)
            if (reflect(x + j - r, map_w) == x0) s += w[j];
        rowv[x] = (float)s;                               // only used as the set of matching weights below
    }
    __syncthreads();
    // pass 2: out[y][x] = float( sum_{j matching} w[j] * (double)col[y] ); with a single matching j this is float(w[j] * col[y])
    double part = 0.0;
    for (int i = t; i < map_h * map_w; i += 256) {
        const int y = i / map_w, x = i % map_w;
        double s = 0.0;
        for (int j = 0; j <= 2 * r; ++j)
            if (reflect(x + j - r, map_w) == x0) s += w[j] * (double)col[y];
        const float v = (float)s;
        row[1 + i] = v;
        part += (double)v;
    }
    red[t] = part;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (t < o) red[t] += red[t + o];
        __syncthreads();
    }
    const float total = (float)red[0];
    for (int i = t; i < map_h * map_w; i += 256) row[1 + i] = row[1 + i] / total;
}

// in-place on target [rows][1 + map_h*map_w] (rows = B*T), after sp_collate_targets
extern "C" int sp_blur_targets(float* target, int rows, int map_h, int map_w, double sigma, void* stream) {
    if (!target) return SP_ENULL;
    if (rows <= 0 || map_h < 1 || map_w < 1 || map_h > 1024 || map_w > 1024 || !(sigma > 0.0) || (int)(4.0 * sigma + 0.5) > BLUR_MAXR)
        return SP_EINVAL;
    hipLaunchKernelGGL(blur_targets_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, target, 1 + map_h * map_w, map_h, map_w, sigma);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_collate_targets(const float* X, const float* Y, const float* T_start, const float* T_end, const int64_t* start,
                                  const int* count, const double* origin_w, const double* origin_h, int B, int T, int map_h,
                                  int map_w, int f64_div, float* target, float* duration, float* action_mask,
                                  float* duration_mask, void* stream) {
    if (!X || !Y || !T_start || !T_end || !start || !count || !origin_w || !origin_h || !target || !duration || !action_mask ||
        !duration_mask)
        return SP_ENULL;
    if (B < 1 || T < 1 || map_h < 1 || map_w < 1) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(target, 0, sizeof(float) * (size_t)B * T * (1 + (size_t)map_h * map_w), s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(collate_kernel, dim3((B * T + 127) / 128), dim3(128), 0, s, X, Y, T_start, T_end, start, count, origin_w,
                       origin_h, B, T, map_h, map_w, f64_div, target, duration, action_mask, duration_mask);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sample_actions(const float* probs, const float* mu, const float* sigma2, int B, int T, int A, int min_length,
                                 uint64_t seed, int64_t* actions, float* action_probs, float* durations, void* stream) {
    if (!probs || !mu || !sigma2 || !actions || !action_probs || !durations) return SP_ENULL;
    if (B < 1 || T < 1 || A < 2) return SP_EINVAL;
    hipLaunchKernelGGL(sample_rows_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, probs, mu, sigma2, T, A, min_length,
                       seed, actions, action_probs, durations);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_generate_scanpath(const int64_t* actions, const float* durations, int B, int T, int map_w, int map_h, int width,
                                    int height, float* length, float* action_masks, float* duration_masks, float* fix, int* nfix,
                                    void* stream) {
    if (!actions || !durations || !length || !action_masks || !duration_masks || !fix || !nfix) return SP_ENULL;
    if (map_w < 1 || map_h < 1) return SP_EINVAL;
    hipLaunchKernelGGL(scanpath_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, actions, durations, B, T, map_w,
                       (float)width / (float)map_w, (float)height / (float)map_h, length, action_masks, duration_masks, fix, nfix);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
