// predict_head without the dense 5x5 GEMM  (reference: models/baseline_attention.py:149-158 -- 5x5 head conv 512->512, then
// sal_layer_2/3 (1x1, 512->1) and drt_layer_1 (7x7 stride 5 pad 2, 512->1) on its output).
//
// Everything downstream of the 5x5 conv is linear up to the ReLUs that follow the 1-channel maps, so the head conv is composed
// with the three projections (G = wcomp x W5, models/scanpath_model.py:_compose_heads) and evaluated in two exact forms:
//
//   * saliency maps (2 per head): conv5x5(h; g) = sum_tap shift_tap( h . g[tap] ) -- one 1x1 GEMM with 25 columns per map
//     ("tap partials" T [B,P,50*heads]) followed by a 25-tap spatial gather (sal_gather_*).  2*P*512*50 flop per head instead
//     of 2*P*12800*50.
//   * duration sites: drt[s] = sum_{7x7 taps k inside the map} conv5x5(h; g_k)[5s-2+k] -- the composite is an 11x11 stride-5
//     window on h whose weights depend only on WHICH taps fall inside the zero-padded intermediate map, i.e. on a border class
//     of the site (first / interior / last per axis).  compose11_* builds W11[head][class][11x11][C] (and the matching sum of
//     composed tap biases); drt_direct_* evaluate the S = dh*dw window dot products and their two gradients directly.
//     2*S*121*512 flop per head instead of 2*P*12800*49.
//
// HBM-bound VALU kernels; every reduction has a fixed order (no atomics) so results are run-to-run identical.
#include "common.h"

namespace {

constexpr int NTAP = 49, NV = 121, MAXSITE = 32, MAXCLS = 8;

struct AxisCls {
    int n;                            // sites along this axis
    int ncls;                         // distinct in-range tap masks
    int len;                          // map extent
    unsigned char mask[MAXCLS];       // bit k: tap k of the 7-tap axis lands inside [0, len)
    unsigned char cls[MAXSITE];       // site -> class
};

static bool make_axis(int len, AxisCls& a) {
    a.n = (len + 4 - 7) / 5 + 1;
    a.ncls = 0;
    a.len = len;
    if (a.n < 1 || a.n > MAXSITE) return false;
    for (int s = 0; s < a.n; ++s) {
        unsigned m = 0;
        for (int k = 0; k < 7; ++k) {
            const int pos = 5 * s - 2 + k;
            if (pos >= 0 && pos < len) m |= 1u << k;
        }
        int id = -1;
        for (int j = 0; j < a.ncls; ++j)
            if (a.mask[j] == m) id = j;
        if (id < 0) {
            if (a.ncls == MAXCLS) return false;
            id = a.ncls;
            a.mask[a.ncls++] = (unsigned char)m;
        }
        a.cls[s] = (unsigned char)id;
    }
    return true;
}

// ---- W11 / cbsum from the composed head filters -------------------------------------------------------------
// G physical [nheads*HC][5][5][C]; rows hd*HC + 2 + tap are the 49 duration taps.  grid (121, ncls, nheads).
__global__ __launch_bounds__(128) void compose11_fwd_kernel(const float* __restrict__ G, const float* __restrict__ cb, int HC,
                                                            int C4, AxisCls ay, AxisCls ax, float* __restrict__ W11,
                                                            float* __restrict__ cbsum) {
    const int v = blockIdx.x, cls = blockIdx.y, hd = blockIdx.z, ncls = gridDim.y;
    const int vy = v / 11, vx = v % 11;
    const unsigned my = ay.mask[cls / ax.ncls], mx = ax.mask[cls % ax.ncls];
    const f32x4* G4 = reinterpret_cast<const f32x4*>(G);
    f32x4* O4 = reinterpret_cast<f32x4*>(W11);
    for (int c4 = threadIdx.x; c4 < C4; c4 += blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < 7; ++ky) {
            const int uy = vy - ky;
            if (!((my >> ky) & 1u) || uy < 0 || uy > 4) continue;
            for (int kx = 0; kx < 7; ++kx) {
                const int ux = vx - kx;
                if (!((mx >> kx) & 1u) || ux < 0 || ux > 4) continue;
                const int64_t row = (int64_t)hd * HC + 2 + ky * 7 + kx;
                acc += G4[(row * 25 + uy * 5 + ux) * C4 + c4];
            }
        }
        O4[(((int64_t)hd * ncls + cls) * NV + v) * C4 + c4] = acc;
    }
    if (v == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int ky = 0; ky < 7; ++ky)
            for (int kx = 0; kx < 7; ++kx)
                if (((my >> ky) & 1u) && ((mx >> kx) & 1u)) s += cb[(int64_t)hd * HC + 2 + ky * 7 + kx];
        cbsum[hd * ncls + cls] = s;
    }
}

// grid (25, 49, nheads): dG row (hd*HC+2+tap), 5x5 position u
__global__ __launch_bounds__(128) void compose11_bwd_kernel(const float* __restrict__ dW11, const float* __restrict__ dcbsum,
                                                            int HC, int C4, int ncls, AxisCls ay, AxisCls ax,
                                                            float* __restrict__ dG, float* __restrict__ dcb) {
    const int u = blockIdx.x, tap = blockIdx.y, hd = blockIdx.z;
    const int ky = tap / 7, kx = tap % 7, uy = u / 5, ux = u % 5;
    const int v = (ky + uy) * 11 + kx + ux;
    const f32x4* S4 = reinterpret_cast<const f32x4*>(dW11);
    f32x4* O4 = reinterpret_cast<f32x4*>(dG);
    const int64_t row = (int64_t)hd * HC + 2 + tap;
    for (int c4 = threadIdx.x; c4 < C4; c4 += blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int cls = 0; cls < ncls; ++cls) {
            if (!((ay.mask[cls / ax.ncls] >> ky) & 1u) || !((ax.mask[cls % ax.ncls] >> kx) & 1u)) continue;
            acc += S4[(((int64_t)hd * ncls + cls) * NV + v) * C4 + c4];
        }
        O4[(row * 25 + u) * C4 + c4] = acc;
    }
    if (u == 0 && threadIdx.x == 0) {
        float s = 0.f;
        for (int cls = 0; cls < ncls; ++cls)
            if (((ay.mask[cls / ax.ncls] >> ky) & 1u) && ((ax.mask[cls % ax.ncls] >> kx) & 1u)) s += dcbsum[hd * ncls + cls];
        dcb[row] = s;
    }
}

// ---- saliency maps from the tap partials --------------------------------------------------------------------
// T [B][P][ldt], column (src*2 + o)*25 + tap; Z2 [B][P][nsel*2]; hmap [B][nsel] = source head of output slot i.
__global__ __launch_bounds__(256) void sal_gather_fwd_kernel(const float* __restrict__ T, int B, int Hm, int Wm, int ldt,
                                                             int nsel, const int* __restrict__ hmap, float* __restrict__ Z2) {
    const int P = Hm * Wm, J = nsel * 2;
    const int64_t n = (int64_t)B * P * J;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i % J);
        const int64_t bp = i / J;
        const int p = (int)(bp % P), b = (int)(bp / P);
        const int py = p / Wm, px = p % Wm;
        const int col0 = (hmap[b * nsel + (j >> 1)] * 2 + (j & 1)) * 25;
        const float* t = T + (int64_t)b * P * ldt + col0;
        float acc = 0.f;
        for (int uy = 0; uy < 5; ++uy) {
            const int qy = py + uy - 2;
            if ((unsigned)qy >= (unsigned)Hm) continue;
            for (int ux = 0; ux < 5; ++ux) {
                const int qx = px + ux - 2;
                if ((unsigned)qx < (unsigned)Wm) acc += t[(int64_t)(qy * Wm + qx) * ldt + uy * 5 + ux];
            }
        }
        Z2[i] = acc;
    }
}

__global__ __launch_bounds__(256) void sal_gather_bwd_kernel(const float* __restrict__ dZ2, int B, int Hm, int Wm, int ldt,
                                                             int nsel, int nsrc, const int* __restrict__ hmap,
                                                             float* __restrict__ dT) {
    const int P = Hm * Wm, J = nsel * 2;
    const int64_t n = (int64_t)B * P * ldt;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % ldt);
        const int64_t bq = i / ldt;
        const int q = (int)(bq % P), b = (int)(bq / P);
        float v = 0.f;
        if (col < nsrc * 50) {
            const int src = col / 50, o = (col % 50) / 25, u = col % 25;
            const int py = q / Wm - (u / 5 - 2), px = q % Wm - (u % 5 - 2);
            if ((unsigned)py < (unsigned)Hm && (unsigned)px < (unsigned)Wm)
                for (int s = 0; s < nsel; ++s)
                    if (hmap[b * nsel + s] == src) v += dZ2[((int64_t)b * P + py * Wm + px) * J + s * 2 + o];
        }
        dT[i] = v;
    }
}

// ---- duration sites ------------------------------------------------------------------------------------------
// Dpre[i][b][s] = cbsum[src][cls(s)] + <W11[src][cls(s)], h[b, 5s-4 .. 5s+6, :]>.  grid (S, B, ceil(nsel/2)), 256 threads;
// a block evaluates two head slots on one read of the window (AiR: good + poor head).
__global__ __launch_bounds__(256) void drt_fwd_kernel(const float* __restrict__ h, const float* __restrict__ W11,
                                                      const float* __restrict__ cbsum, const int* __restrict__ hmap, int B,
                                                      int C4, int nsel, int ncls, AxisCls ay, AxisCls ax,
                                                      float* __restrict__ Dpre) {
    __shared__ float sh4[4];
    const int s = blockIdx.x, b = blockIdx.y, i0 = blockIdx.z * 2;
    const bool two = i0 + 1 < nsel;
    const int Hm = ay.len, Wm = ax.len, S = ay.n * ax.n;
    const int sy = s / ax.n, sx = s % ax.n;
    const int cls = ay.cls[sy] * ax.ncls + ax.cls[sx];
    const int src0 = hmap[b * nsel + i0], src1 = two ? hmap[b * nsel + i0 + 1] : src0;
    const int oy = 5 * sy - 4, ox = 5 * sx - 4;
    const f32x4* H4 = reinterpret_cast<const f32x4*>(h) + (int64_t)b * Hm * Wm * C4;
    const f32x4* W40 = reinterpret_cast<const f32x4*>(W11) + ((int64_t)src0 * ncls + cls) * NV * C4;
    const f32x4* W41 = reinterpret_cast<const f32x4*>(W11) + ((int64_t)src1 * ncls + cls) * NV * C4;
    float acc0 = 0.f, acc1 = 0.f;
    for (int idx = threadIdx.x; idx < NV * C4; idx += 256) {
        const int v = idx / C4, c4 = idx - v * C4;
        const int qy = oy + v / 11, qx = ox + v % 11;
        if ((unsigned)qy >= (unsigned)Hm || (unsigned)qx >= (unsigned)Wm) continue;
        const f32x4 a = H4[(int64_t)(qy * Wm + qx) * C4 + c4], w = W40[idx];
        acc0 += a[0] * w[0] + a[1] * w[1] + a[2] * w[2] + a[3] * w[3];
        if (two) {
            const f32x4 u = W41[idx];
            acc1 += a[0] * u[0] + a[1] * u[1] + a[2] * u[2] + a[3] * u[3];
        }
    }
    acc0 = block_sum_256(acc0, sh4);
    if (two) acc1 = block_sum_256(acc1, sh4);
    if (threadIdx.x == 0) {
        Dpre[((int64_t)i0 * B + b) * S + s] = acc0 + cbsum[src0 * ncls + cls];
        if (two) Dpre[((int64_t)(i0 + 1) * B + b) * S + s] = acc1 + cbsum[src1 * ncls + cls];
    }
}

// dh[b][q][c] (+)= sum over the sites whose window covers q, over the head slots.
__global__ __launch_bounds__(256) void drt_bwd_data_kernel(const float* __restrict__ dD, const float* __restrict__ W11,
                                                           const int* __restrict__ hmap, int B, int C4, int nsel, int ncls,
                                                           AxisCls ay, AxisCls ax, int accumulate, float* __restrict__ dh) {
    const int Hm = ay.len, Wm = ax.len, P = Hm * Wm, S = ay.n * ax.n;
    const f32x4* W4 = reinterpret_cast<const f32x4*>(W11);
    f32x4* O4 = reinterpret_cast<f32x4*>(dh);
    const int64_t n = (int64_t)B * P * C4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(t % C4);
        const int64_t bq = t / C4;
        const int q = (int)(bq % P), b = (int)(bq / P);
        const int qy = q / Wm, qx = q % Wm;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int sy_lo = qy >= 6 ? (qy - 6 + 4) / 5 : 0, sy_hi = min(ay.n - 1, (qy + 4) / 5);
        const int sx_lo = qx >= 6 ? (qx - 6 + 4) / 5 : 0, sx_hi = min(ax.n - 1, (qx + 4) / 5);
        for (int sy = sy_lo; sy <= sy_hi; ++sy)
            for (int sx = sx_lo; sx <= sx_hi; ++sx) {
                const int v = (qy - 5 * sy + 4) * 11 + (qx - 5 * sx + 4);
                const int cls = ay.cls[sy] * ax.ncls + ax.cls[sx];
                for (int i = 0; i < nsel; ++i) {
                    const float g = dD[((int64_t)i * B + b) * S + sy * ax.n + sx];
                    const int src = hmap[b * nsel + i];
                    acc += g * W4[(((int64_t)src * ncls + cls) * NV + v) * C4 + c4];
                }
            }
        if (accumulate) acc += O4[t];
        O4[t] = acc;
    }
}

// per-(sample, slot) partial of dW11: slab[b][i][cls][v][c] = sum_{s in cls} dD[i][b][s] * h[b][win(s)+v][c].
// grid (121*ncls, B, ceil(nsel/2)), 128 threads over c4; two head slots per read of h.
__global__ __launch_bounds__(128) void drt_bwd_weight_kernel(const float* __restrict__ dD, const float* __restrict__ h, int B,
                                                             int C4, int nsel, int ncls, AxisCls ay, AxisCls ax,
                                                             float* __restrict__ slab) {
    const int v = blockIdx.x % NV, cls = blockIdx.x / NV, b = blockIdx.y, i0 = blockIdx.z * 2;
    const bool two = i0 + 1 < nsel;
    const int Hm = ay.len, Wm = ax.len, S = ay.n * ax.n;
    const int vy = v / 11, vx = v % 11;
    const f32x4* H4 = reinterpret_cast<const f32x4*>(h) + (int64_t)b * Hm * Wm * C4;
    const float* g0 = dD + ((int64_t)i0 * B + b) * S;
    const float* g1 = dD + ((int64_t)(two ? i0 + 1 : i0) * B + b) * S;
    f32x4* O40 = reinterpret_cast<f32x4*>(slab) + ((((int64_t)b * nsel + i0) * ncls + cls) * NV + v) * C4;
    f32x4* O41 = O40 + (int64_t)ncls * NV * C4;
    for (int c4 = threadIdx.x; c4 < C4; c4 += blockDim.x) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int sy = 0; sy < ay.n; ++sy) {
            const int qy = 5 * sy - 4 + vy;
            if ((unsigned)qy >= (unsigned)Hm) continue;
            for (int sx = 0; sx < ax.n; ++sx) {
                const int qx = 5 * sx - 4 + vx;
                if ((unsigned)qx >= (unsigned)Wm || ay.cls[sy] * ax.ncls + ax.cls[sx] != cls) continue;
                const f32x4 a = H4[(int64_t)(qy * Wm + qx) * C4 + c4];
                acc0 += g0[sy * ax.n + sx] * a;
                acc1 += g1[sy * ax.n + sx] * a;
            }
        }
        O40[c4] = acc0;
        if (two) O41[c4] = acc1;
    }
}

// dW11[k] = sum of the slabs whose source head is k, in (b, i) order.  one thread per float4 of dW11.
__global__ __launch_bounds__(256) void drt_slab_reduce_kernel(const float* __restrict__ slab, const int* __restrict__ hmap,
                                                              int B, int nsel, int nheads, int64_t per_head4,
                                                              float* __restrict__ dW11) {
    const f32x4* S4 = reinterpret_cast<const f32x4*>(slab);
    f32x4* O4 = reinterpret_cast<f32x4*>(dW11);
    const int64_t n = (int64_t)nheads * per_head4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const int k = (int)(t / per_head4);
        const int64_t r = t % per_head4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int bi = 0; bi < B * nsel; ++bi)
            if (hmap[bi] == k) acc += S4[(int64_t)bi * per_head4 + r];
        O4[t] = acc;
    }
}

// dcbsum[k][cls] = sum_{(b,i): hmap == k} sum_{s in cls} dD[i][b][s].  grid (ncls, nheads), one wave.
__global__ __launch_bounds__(64) void drt_dcbsum_kernel(const float* __restrict__ dD, const int* __restrict__ hmap, int B,
                                                        int nsel, AxisCls ay, AxisCls ax, float* __restrict__ dcbsum) {
    const int cls = blockIdx.x, k = blockIdx.y, ncls = gridDim.x, S = ay.n * ax.n;
    float acc = 0.f;
    for (int bi = threadIdx.x; bi < B * nsel; bi += 64) {
        if (hmap[bi] != k) continue;
        const int b = bi / nsel, i = bi % nsel;
        const float* g = dD + ((int64_t)i * B + b) * S;
        for (int s = 0; s < S; ++s)
            if (ay.cls[s / ax.n] * ax.ncls + ax.cls[s % ax.n] == cls) acc += g[s];
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) dcbsum[k * ncls + cls] = acc;
}

static inline int ew_grid(int64_t n) {
    const int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

}  // namespace

extern "C" int sp_head_num_classes(int Hm, int Wm) {
    AxisCls ay, ax;
    if (!make_axis(Hm, ay) || !make_axis(Wm, ax)) return -1;
    return ay.ncls * ax.ncls;
}

extern "C" int sp_head_compose11_fwd(const float* G, const float* cb, int nheads, int HC, int C, int Hm, int Wm, float* W11,
                                     float* cbsum, void* stream) {
    if (!G || !cb || !W11 || !cbsum) return SP_ENULL;
    AxisCls ay, ax;
    if (HC < 2 + NTAP || C % 4 || nheads < 1 || !make_axis(Hm, ay) || !make_axis(Wm, ax)) return SP_EINVAL;
    hipLaunchKernelGGL(compose11_fwd_kernel, dim3(NV, ay.ncls * ax.ncls, nheads), dim3(128), 0, (hipStream_t)stream, G, cb, HC,
                       C / 4, ay, ax, W11, cbsum);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_head_compose11_bwd(const float* dW11, const float* dcbsum, int nheads, int HC, int C, int Hm, int Wm,
                                     float* dG, float* dcb, void* stream) {
    if (!dW11 || !dcbsum || !dG || !dcb) return SP_ENULL;
    AxisCls ay, ax;
    if (HC < 2 + NTAP || C % 4 || nheads < 1 || !make_axis(Hm, ay) || !make_axis(Wm, ax)) return SP_EINVAL;
    hipError_t e = hipMemsetAsync(dG, 0, (size_t)nheads * HC * 25 * C * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(dcb, 0, (size_t)nheads * HC * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(compose11_bwd_kernel, dim3(25, NTAP, nheads), dim3(128), 0, (hipStream_t)stream, dW11, dcbsum, HC, C / 4,
                       ay.ncls * ax.ncls, ay, ax, dG, dcb);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sal_gather_fwd(const float* T, int B, int Hm, int Wm, int ldt, int nsel, const int* hmap, float* Z2,
                                 void* stream) {
    if (!T || !hmap || !Z2) return SP_ENULL;
    if (B < 1 || nsel < 1 || ldt < 50) return SP_EINVAL;
    hipLaunchKernelGGL(sal_gather_fwd_kernel, dim3(ew_grid((int64_t)B * Hm * Wm * nsel * 2)), dim3(256), 0, (hipStream_t)stream,
                       T, B, Hm, Wm, ldt, nsel, hmap, Z2);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_sal_gather_bwd(const float* dZ2, int B, int Hm, int Wm, int ldt, int nsel, int nsrc, const int* hmap,
                                 float* dT, void* stream) {
    if (!dZ2 || !hmap || !dT) return SP_ENULL;
    if (B < 1 || nsel < 1 || nsrc < 1 || ldt < nsrc * 50) return SP_EINVAL;
    hipLaunchKernelGGL(sal_gather_bwd_kernel, dim3(ew_grid((int64_t)B * Hm * Wm * ldt)), dim3(256), 0, (hipStream_t)stream, dZ2,
                       B, Hm, Wm, ldt, nsel, nsrc, hmap, dT);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_drt_direct_fwd(const float* h, const float* W11, const float* cbsum, const int* hmap, int B, int Hm, int Wm,
                                 int C, int nsel, float* Dpre, void* stream) {
    if (!h || !W11 || !cbsum || !hmap || !Dpre) return SP_ENULL;
    AxisCls ay, ax;
    if (C % 4 || B < 1 || nsel < 1 || !make_axis(Hm, ay) || !make_axis(Wm, ax)) return SP_EINVAL;
    hipLaunchKernelGGL(drt_fwd_kernel, dim3(ay.n * ax.n, B, (nsel + 1) / 2), dim3(256), 0, (hipStream_t)stream, h, W11, cbsum, hmap, B,
                       C / 4, nsel, ay.ncls * ax.ncls, ay, ax, Dpre);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_drt_direct_bwd_data(const float* dDpre, const float* W11, const int* hmap, int B, int Hm, int Wm, int C,
                                      int nsel, int accumulate, float* dh, void* stream) {
    if (!dDpre || !W11 || !hmap || !dh) return SP_ENULL;
    AxisCls ay, ax;
    if (C % 4 || B < 1 || nsel < 1 || !make_axis(Hm, ay) || !make_axis(Wm, ax)) return SP_EINVAL;
    hipLaunchKernelGGL(drt_bwd_data_kernel, dim3(ew_grid((int64_t)B * Hm * Wm * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       dDpre, W11, hmap, B, C / 4, nsel, ay.ncls * ax.ncls, ay, ax, accumulate, dh);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int64_t sp_drt_direct_bwd_weight_workspace(int B, int Hm, int Wm, int C, int nsel) {
    AxisCls ay, ax;
    if (!make_axis(Hm, ay) || !make_axis(Wm, ax)) return -1;
    return (int64_t)B * nsel * ay.ncls * ax.ncls * NV * C * (int64_t)sizeof(float);
}

extern "C" int sp_drt_direct_bwd_weight(const float* dDpre, const float* h, const int* hmap, int B, int Hm, int Wm, int C,
                                        int nsel, int nheads, void* workspace, float* dW11, float* dcbsum, void* stream) {
    if (!dDpre || !h || !hmap || !workspace || !dW11 || !dcbsum) return SP_ENULL;
    AxisCls ay, ax;
    if (C % 4 || B < 1 || nsel < 1 || nheads < 1 || !make_axis(Hm, ay) || !make_axis(Wm, ax)) return SP_EINVAL;
    const int ncls = ay.ncls * ax.ncls;
    hipLaunchKernelGGL(drt_bwd_weight_kernel, dim3(NV * ncls, B, (nsel + 1) / 2), dim3(128), 0, (hipStream_t)stream, dDpre, h, B, C / 4,
                       nsel, ncls, ay, ax, (float*)workspace);
    SP_LAUNCH_CHECK();
    const int64_t per_head4 = (int64_t)ncls * NV * (C / 4);
    hipLaunchKernelGGL(drt_slab_reduce_kernel, dim3(ew_grid(nheads * per_head4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, hmap, B, nsel, nheads, per_head4, dW11);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(drt_dcbsum_kernel, dim3(ncls, nheads), dim3(64), 0, (hipStream_t)stream, dDpre, hmap, B, nsel, ay, ax,
                       dcbsum);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
