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

}  // namespace

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
