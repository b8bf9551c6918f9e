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
        double s = 0.0;
        for (int cls = 0; cls < ncls; ++cls)
            if (((ay.mask[cls / ax.ncls] >> ky) & 1u) && ((ax.mask[cls % ax.ncls] >> kx) & 1u)) s += (double)dcbsum[hd * ncls + cls];
        dcb[row] = (float)s;
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
                                                             float* __restrict__ dT, const int* __restrict__ row_last, int row_step) {
    const int P = Hm * Wm, J = nsel * 2;
    const int64_t n = (int64_t)B * P * ldt;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int col = (int)(i % ldt);
        const int64_t bq = i / ldt;
        const int q = (int)(bq % P), b = (int)(bq / P);
        float v = 0.f;
        // row sparsity: dZ2 of a sample behind its last loss step is exactly zero (not read)
        if (col < nsrc * 50 && !(row_last && row_last[b] < row_step)) {
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
// Dpre[i][b][s] = cbsum[src][cls(s)] + <W11[src][cls(s)], h[b, 5s-4 .. 5s+6, :]>.  grid (S, ceil(B/4), ceil(nsel/2)), 256 threads;
// a block evaluates two head slots on one read of the window (AiR: good + poor head), and its site for a GROUP of up to four
// samples: when they use the same source heads (always, except COCO's per-sample heads) the 2 x 248 KB of W11 are read once per
// group instead of once per sample -- the kernel is bound by those L2 reads.  Per (sample, slot) the sum order is that of the round-2 kernel for INTERIOR sites only: the
// loop runs over the in-map rectangle of taps, so at border sites a thread's partial sums cover other elements (same value to rounding).
constexpr int DRT_GB = 4;
// acc + <a, w> as ONE fixed chain of fused multiply-adds: with the compiler free to fuse (or not) every a * b + c of the unrolled group loop
// on its own, a row's sum depended -- in the last bit -- on its POSITION in the group of four (found when round 6 evaluated the sites of all
// decode steps in one launch: rows that changed position differed by one ulp from the per-step launches)
__device__ __forceinline__ float drt_dot4(const f32x4& a, const f32x4& w, float acc) {
    return __builtin_fmaf(a[3], w[3], __builtin_fmaf(a[2], w[2], __builtin_fmaf(a[1], w[1], __builtin_fmaf(a[0], w[0], acc))));
}
__global__ __launch_bounds__(256) void drt_fwd_kernel(const float* __restrict__ h, const float* __restrict__ W11,
                                                      const float* __restrict__ cbsum, const int* __restrict__ hmap, int B,
                                                      int C4, int nsel, int ncls, AxisCls ay, AxisCls ax,
                                                      float* __restrict__ Dpre) {
    __shared__ float sh4[4];
    // Eight site rows (the 40 x 64 map of the benchmark) on eight XCDs: under the round-robin dispatch the workgroups of one XCD then walk
    // ONE site row -- sites left to right, then the next group of samples -- whose 11-row windows overlap by 6 of 11 columns, so the x-overlap
    // (2.2 of the 4.9 reads per element of h) is served from that XCD's L2 instead of the fabric (the launch over all T x B hidden states of
    // a training step read 12.8 GB fabric-side for 2.7 GB of h: profiles/r06_pmc_hbm_kernels.json).  Speed only; any other map keeps the
    // plain order.
    int s = blockIdx.x, gy = blockIdx.y;
    if (ay.n == 8 && ((gridDim.x * gridDim.y) & 7u) == 0) {
        const unsigned bid = blockIdx.y * gridDim.x + blockIdx.x, q = bid >> 3;
        s = (int)(bid & 7u) * ax.n + (int)(q % (unsigned)ax.n);
        gy = (int)(q / (unsigned)ax.n);
    }
    const int b0 = gy * DRT_GB, i0 = blockIdx.z * 2;
    const int nb = min(DRT_GB, B - b0);
    const bool two = i0 + 1 < nsel;
    const int Hm = ay.len, Wm = ax.len, S = ay.n * ax.n;
    const int sy = s / ax.n, sx = s % ax.n;
    const int cls = ay.cls[sy] * ax.ncls + ax.cls[sx];
    const int oy = 5 * sy - 4, ox = 5 * sx - 4;
    int src0[DRT_GB], src1[DRT_GB];
    bool same = true;
#pragma unroll
    for (int j = 0; j < DRT_GB; ++j) {
        const int b = min(b0 + j, B - 1);
        src0[j] = hmap[b * nsel + i0];
        src1[j] = two ? hmap[b * nsel + i0 + 1] : src0[j];
        same = same && src0[j] == src0[0] && src1[j] == src1[0];
    }
    const int64_t wstride = (int64_t)ncls * NV * C4;
    const f32x4* W4c = reinterpret_cast<const f32x4*>(W11) + (int64_t)cls * NV * C4;
    const f32x4* H4 = reinterpret_cast<const f32x4*>(h);
    const int64_t hb = (int64_t)Hm * Wm * C4;
    float acc0[DRT_GB], acc1[DRT_GB];
#pragma unroll
    for (int j = 0; j < DRT_GB; ++j) acc0[j] = acc1[j] = 0.f;
    // the taps inside the map form a rectangle [vy0, vy1] x [vx0, vx1] of the 11 x 11 window: a branch-free loop over it (with the
    // bounds test inside the loop the compiler kept one iteration's loads in flight at a time)
    const int vy0 = max(0, -oy), vy1 = min(10, Hm - 1 - oy), vx0 = max(0, -ox), vx1 = min(10, Wm - 1 - ox);
    const int nvx = vx1 - vx0 + 1, cnt = max(0, vy1 - vy0 + 1) * max(0, nvx);
#pragma unroll 4
    for (int it = threadIdx.x; it < cnt * C4; it += 256) {
        const int k = it / C4, c4 = it - k * C4;
        const int ky = k / nvx, vy = vy0 + ky, vx = vx0 + (k - ky * nvx);
        const int idx = (vy * 11 + vx) * C4 + c4;
        const int64_t ho = (int64_t)((oy + vy) * Wm + ox + vx) * C4 + c4;
        f32x4 a[DRT_GB];
#pragma unroll
        for (int j = 0; j < DRT_GB; ++j) a[j] = H4[(int64_t)min(b0 + j, B - 1) * hb + ho];
        if (same) {
            const f32x4 w = W4c[src0[0] * wstride + idx];
#pragma unroll
            for (int j = 0; j < DRT_GB; ++j) acc0[j] = drt_dot4(a[j], w, acc0[j]);
            if (two) {
                const f32x4 u = W4c[src1[0] * wstride + idx];
#pragma unroll
                for (int j = 0; j < DRT_GB; ++j) acc1[j] = drt_dot4(a[j], u, acc1[j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < DRT_GB; ++j) {
                const f32x4 w = W4c[src0[j] * wstride + idx];
                acc0[j] = drt_dot4(a[j], w, acc0[j]);
                if (two) {
                    const f32x4 u = W4c[src1[j] * wstride + idx];
                    acc1[j] = drt_dot4(a[j], u, acc1[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DRT_GB; ++j) {
        if (j >= nb) break;                       // (uniform)
        const float r0 = block_sum_256(acc0[j], sh4);
        const float r1 = two ? block_sum_256(acc1[j], sh4) : 0.f;
        if (threadIdx.x == 0) {
            Dpre[((int64_t)i0 * B + b0 + j) * S + s] = r0 + cbsum[src0[j] * ncls + cls];
            if (two) Dpre[((int64_t)(i0 + 1) * B + b0 + j) * S + s] = r1 + cbsum[src1[j] * ncls + cls];
        }
    }
}

// Which (row, head slot) pairs carry an exactly-zero duration gradient (round 6: the duration branch of ALL T decode steps is evaluated in
// one launch behind the decode loop, B = T x batch "virtual" rows, row b = decode step b / rowB of sample b % rowB):
//   row_last (nullable): the sample's last decode step with any loss gradient (functional._OutputGate) -- row b is DEAD when
//                        row_last[b % rowB] < row_step + b / rowB (per-step launches: rowB = B, i.e. the old row_last[b] < row_step);
//   live (nullable):     live[i * B + b] = 0 when slot i of row b received dmu == 0 and dsigma2 == 0 exactly (sp_head_finish_parts_bwd):
//                        AiR's unselected head, the step right at a scanpath's end (action mask on, duration mask off).
// Skipping them leaves every sum bit-identical (their terms are exact zeros; the reduce order of the others does not change).
__device__ __forceinline__ bool drt_row_dead(const int* __restrict__ row_last, int row_step, int rowB, int b) {
    return row_last != nullptr && row_last[b % rowB] < row_step + b / rowB;
}

// dh[b][q][c] (+)= sum over the sites whose window covers q, over the head slots.  A thread owns (pixel q, channel quad c4) for a
// GROUP of up to four samples: the up to 9 sites x nsel filter rows it needs are the same for every sample that uses the same
// source heads, so they are read once per group (the kernel is bound by these L2 reads: 18 float4 of W11 per float4 of dh).
// (launch bounds: 6 waves per SIMD = at most 80 VGPRs, and drt_bwd_weight_kernel's 10 KB of LDS: both then fit beside a resident
// workgroup of the h-gate conv's data gradient -- 2 x 216 registers, 148 KB -- that runs on the side stream during these launches)
__global__ __launch_bounds__(256, 6) void drt_bwd_data_kernel(const float* __restrict__ dD, const float* __restrict__ W11,
                                                           const int* __restrict__ hmap, int B, int C4, int nsel, int ncls,
                                                           AxisCls ay, AxisCls ax, int accumulate, float* __restrict__ dh,
                                                           const int* __restrict__ live, const int* __restrict__ row_last, int row_step,
                                                           int rowB) {
    const int Hm = ay.len, Wm = ax.len, P = Hm * Wm, S = ay.n * ax.n;
    const f32x4* W4 = reinterpret_cast<const f32x4*>(W11);
    f32x4* O4 = reinterpret_cast<f32x4*>(dh);
    const int NG = (B + DRT_GB - 1) / DRT_GB;
    const int64_t n = (int64_t)NG * P * C4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(t % C4);
        const int64_t gq = t / C4;
        const int q = (int)(gq % P), b0 = (int)(gq / P) * DRT_GB;
        const int nb = min(DRT_GB, B - b0);
        const int qy = q / Wm, qx = q % Wm;
        f32x4 acc[DRT_GB];
#pragma unroll
        for (int j = 0; j < DRT_GB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (row, slot) pairs of this group with an exactly-zero gradient (drt_row_dead / live above): bit (i * DRT_GB + j) of `on`
        unsigned on = 0;
        for (int i = 0; i < nsel && i < 8; ++i)
#pragma unroll
            for (int j = 0; j < DRT_GB; ++j)
                if (j < nb && !drt_row_dead(row_last, row_step, rowB, b0 + j) && (!live || live[(int64_t)i * B + b0 + j])) on |= 1u << (i * DRT_GB + j);
        const bool flagged = (live != nullptr || row_last != nullptr) && nsel <= 8;
        const int sy_lo = qy >= 6 ? (qy - 6 + 4) / 5 : 0, sy_hi = min(ay.n - 1, (qy + 4) / 5);
        const int sx_lo = qx >= 6 ? (qx - 6 + 4) / 5 : 0, sx_hi = min(ax.n - 1, (qx + 4) / 5);
        if (!flagged || on)
        for (int sy = sy_lo; sy <= sy_hi; ++sy)
            for (int sx = sx_lo; sx <= sx_hi; ++sx) {
                const int v = (qy - 5 * sy + 4) * 11 + (qx - 5 * sx + 4);
                const int cls = ay.cls[sy] * ax.ncls + ax.cls[sx];
                for (int i = 0; i < nsel; ++i) {
                    const unsigned oni = flagged ? (on >> (i * DRT_GB)) & ((1u << DRT_GB) - 1u) : ~0u;
                    if (!oni) continue;                     // no row of the group has a gradient in this slot: the filter row is not read
                    const int src0 = hmap[b0 * nsel + i];
                    f32x4 w0 = W4[(((int64_t)src0 * ncls + cls) * NV + v) * C4 + c4];
#pragma unroll
                    for (int j = 0; j < DRT_GB; ++j) {
                        if (j >= nb) break;
                        if (!((oni >> j) & 1u)) continue;
                        const int b = b0 + j;
                        const float g = dD[((int64_t)i * B + b) * S + sy * ax.n + sx];
                        const int src = hmap[b * nsel + i];
                        const f32x4 w = src == src0 ? w0 : W4[(((int64_t)src * ncls + cls) * NV + v) * C4 + c4];
                        acc[j] += g * w;
                    }
                }
            }
#pragma unroll
        for (int j = 0; j < DRT_GB; ++j) {
            if (j >= nb) break;
            const int64_t o = ((int64_t)(b0 + j) * P + q) * C4 + c4;
            f32x4 r = acc[j];
            if (accumulate) r += O4[o];
            O4[o] = r;
        }
    }
}

// per-(sample, slot) partial of dW11: slab[b][i][cls][v][c] = sum_{s in cls} dD[i][b][s] * h[b][win(s)+v][c].
// grid (121*ncls, B, ceil(nsel/2)), 128 threads over c4; two head slots per read of h.
// The block first compacts the sites of its class whose tap lands inside the map into LDS (site order kept: the sum order, hence the
// result, is that of the plain nested loop), then streams their h rows eight loads at a time.  With the class test inside the load
// loop every thread had ONE load in flight (the interior class of the 40x64 map has 66 sites: 66 dependent L2 round trips per block,
// 292 us per launch for 0.4 GFLOP); the compacted form is bound by the L2 reads instead.
constexpr int DRT_MAXS = MAXSITE * MAXSITE;
__global__ __launch_bounds__(128) void drt_bwd_weight_kernel(const float* __restrict__ dD, const float* __restrict__ h, int B,
                                                             int C4, int nsel, int ncls, AxisCls ay, AxisCls ax,
                                                             float* __restrict__ slab, const int* __restrict__ row_last, int row_step,
                                                             const int* __restrict__ live, int rowB) {
    __shared__ unsigned short s_pix[DRT_MAXS];          // (a map has < 65536 pixels: make_axis caps a side at 5 * MAXSITE)
    __shared__ float s_g0[DRT_MAXS], s_g1[DRT_MAXS];
    __shared__ int s_cnt[2];
    const int v = blockIdx.x % NV, cls = blockIdx.x / NV, b = blockIdx.y, i0 = blockIdx.z * 2;
    const bool two = i0 + 1 < nsel;
    // slots with an exactly-zero gradient (drt_row_dead / live, see drt_bwd_data_kernel): their slabs are neither computed nor written --
    // drt_slab_reduce_kernel applies the same test and does not read them (block-uniform)
    const bool rdead = drt_row_dead(row_last, row_step, rowB, b);
    const bool on0 = !rdead && (!live || live[(int64_t)i0 * B + b]);
    const bool on1 = two && !rdead && (!live || live[(int64_t)(i0 + 1) * B + b]);
    if (!on0 && !on1) return;
    const int Hm = ay.len, Wm = ax.len, S = ay.n * ax.n;
    const int vy = v / 11, vx = v % 11;
    const f32x4* H4 = reinterpret_cast<const f32x4*>(h) + (int64_t)b * Hm * Wm * C4;
    const float* g0 = dD + ((int64_t)i0 * B + b) * S;
    const float* g1 = dD + ((int64_t)(two ? i0 + 1 : i0) * B + b) * S;
    f32x4* O40 = reinterpret_cast<f32x4*>(slab) + ((((int64_t)b * nsel + i0) * ncls + cls) * NV + v) * C4;
    f32x4* O41 = O40 + (int64_t)ncls * NV * C4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int n = 0;
    for (int s0 = 0; s0 < S; s0 += 128) {
        const int s = s0 + tid;
        const int sy = s / ax.n, sx = s - sy * ax.n;
        const int qy = 5 * sy - 4 + vy, qx = 5 * sx - 4 + vx;
        const bool valid = s < S && (unsigned)qy < (unsigned)Hm && (unsigned)qx < (unsigned)Wm && ay.cls[sy] * ax.ncls + ax.cls[sx] == cls;
        const unsigned long long m = __ballot(valid);
        if (lane == 0) s_cnt[wave] = __popcll(m);
        __syncthreads();
        const int pos = n + (wave ? s_cnt[0] : 0) + __popcll(m & ((1ull << lane) - 1ull));
        if (valid) {
            s_pix[pos] = (unsigned short)(qy * Wm + qx);
            s_g0[pos] = g0[s];
            s_g1[pos] = g1[s];
        }
        n += s_cnt[0] + s_cnt[1];
        __syncthreads();
    }
    for (int c4 = tid; c4 < C4; c4 += blockDim.x) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < n; k += 8) {
            f32x4 a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = H4[(int64_t)s_pix[min(k + u, n - 1)] * C4 + c4];      // (clamped: always a legal row)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k + u < n) {
                    acc0 += s_g0[k + u] * a[u];
                    acc1 += s_g1[k + u] * a[u];
                }
        }
        if (on0) O40[c4] = acc0;
        if (on1) O41[c4] = acc1;
    }
}

// dW11[k] = sum of the slabs whose source head is k, in (b, i) order.  one thread per float4 of dW11.
__global__ __launch_bounds__(256) void drt_slab_reduce_kernel(const float* __restrict__ slab, const int* __restrict__ hmap,
                                                              int B, int nsel, int nheads, int64_t per_head4,
                                                              float* __restrict__ dW11, const int* __restrict__ row_last, int row_step,
                                                              const int* __restrict__ live, int rowB) {
    const f32x4* S4 = reinterpret_cast<const f32x4*>(slab);
    f32x4* O4 = reinterpret_cast<f32x4*>(dW11);
    const int64_t n = (int64_t)nheads * per_head4;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (int64_t)gridDim.x * 256) {
        const int k = (int)(t / per_head4);
        const int64_t r = t % per_head4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int bi = 0; bi < B * nsel; ++bi) {
            if (hmap[bi] != k) continue;
            const int b = bi / nsel, i = bi - b * nsel;
            if (drt_row_dead(row_last, row_step, rowB, b) || (live && !live[(int64_t)i * B + b])) continue;      // slab not written: an exact zero
            acc += S4[(int64_t)bi * per_head4 + r];
        }
        O4[t] = acc;
    }
}

// dcbsum[k][cls] = sum_{(b,i): hmap == k} sum_{s in cls} dD[i][b][s].  grid (ncls, nheads), one wave.
__global__ __launch_bounds__(64) void drt_dcbsum_kernel(const float* __restrict__ dD, const int* __restrict__ hmap, int B,
                                                        int nsel, AxisCls ay, AxisCls ax, float* __restrict__ dcbsum) {
    const int cls = blockIdx.x, k = blockIdx.y, ncls = gridDim.x, S = ay.n * ax.n;
    double acc = 0.0;                    // fp64 like every reduction of the path: the terms cancel
    for (int bi = threadIdx.x; bi < B * nsel; bi += 64) {
        if (hmap[bi] != k) continue;
        const int b = bi / nsel, i = bi % nsel;
        const float* g = dD + ((int64_t)i * B + b) * S;
        for (int s = 0; s < S; ++s)
            if (ay.cls[s / ax.n] * ax.ncls + ax.cls[s % ax.n] == cls) acc += (double)g[s];
    }
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) dcbsum[k * ncls + cls] = (float)acc;
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

extern "C" int sp_sal_gather_bwd_rows(const float* dZ2, int B, int Hm, int Wm, int ldt, int nsel, int nsrc, const int* hmap,
                                      float* dT, const int* row_last, int row_step, void* stream) {
    if (!dZ2 || !hmap || !dT) return SP_ENULL;
    if (B < 1 || nsel < 1 || nsrc < 1 || ldt < nsrc * 50) return SP_EINVAL;
    hipLaunchKernelGGL(sal_gather_bwd_kernel, dim3(ew_grid((int64_t)B * Hm * Wm * ldt)), dim3(256), 0, (hipStream_t)stream, dZ2,
                       B, Hm, Wm, ldt, nsel, nsrc, hmap, dT, row_last, row_step);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
extern "C" int sp_sal_gather_bwd(const float* dZ2, int B, int Hm, int Wm, int ldt, int nsel, int nsrc, const int* hmap,
                                 float* dT, void* stream) {
    return sp_sal_gather_bwd_rows(dZ2, B, Hm, Wm, ldt, nsel, nsrc, hmap, dT, nullptr, 0, stream);
}

extern "C" int sp_drt_direct_fwd(const float* h, const float* W11, const float* cbsum, const int* hmap, int B, int Hm, int Wm,
                                 int C, int nsel, float* Dpre, void* stream) {
    if (!h || !W11 || !cbsum || !hmap || !Dpre) return SP_ENULL;
    AxisCls ay, ax;
    if (C % 4 || B < 1 || nsel < 1 || !make_axis(Hm, ay) || !make_axis(Wm, ax)) return SP_EINVAL;
    hipLaunchKernelGGL(drt_fwd_kernel, dim3(ay.n * ax.n, (B + DRT_GB - 1) / DRT_GB, (nsel + 1) / 2), dim3(256), 0, (hipStream_t)stream, h, W11, cbsum, hmap, B,
                       C / 4, nsel, ay.ncls * ax.ncls, ay, ax, Dpre);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_drt_direct_bwd_data_live(const float* dDpre, const float* W11, const int* hmap, int B, int Hm, int Wm, int C,
                                           int nsel, int accumulate, float* dh, const int* live, const int* row_last, int row_step, int rowB,
                                           void* stream) {
    if (!dDpre || !W11 || !hmap || !dh) return SP_ENULL;
    AxisCls ay, ax;
    if (C % 4 || B < 1 || nsel < 1 || !make_axis(Hm, ay) || !make_axis(Wm, ax)) return SP_EINVAL;
    if (row_last && (rowB < 1 || B % rowB)) return SP_EINVAL;          // B = (decode steps) x rowB virtual rows
    hipLaunchKernelGGL(drt_bwd_data_kernel, dim3(ew_grid((int64_t)((B + DRT_GB - 1) / DRT_GB) * Hm * Wm * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       dDpre, W11, hmap, B, C / 4, nsel, ay.ncls * ax.ncls, ay, ax, accumulate, dh, live, row_last, row_step, row_last ? rowB : B);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
extern "C" int sp_drt_direct_bwd_data(const float* dDpre, const float* W11, const int* hmap, int B, int Hm, int Wm, int C,
                                      int nsel, int accumulate, float* dh, void* stream) {
    return sp_drt_direct_bwd_data_live(dDpre, W11, hmap, B, Hm, Wm, C, nsel, accumulate, dh, nullptr, nullptr, 0, B, stream);
}

extern "C" int64_t sp_drt_direct_bwd_weight_workspace(int B, int Hm, int Wm, int C, int nsel) {
    AxisCls ay, ax;
    if (!make_axis(Hm, ay) || !make_axis(Wm, ax)) return -1;
    return (int64_t)B * nsel * ay.ncls * ax.ncls * NV * C * (int64_t)sizeof(float);
}

extern "C" int sp_drt_direct_bwd_weight_live(const float* dDpre, const float* h, const int* hmap, int B, int Hm, int Wm, int C,
                                             int nsel, int nheads, void* workspace, float* dW11, float* dcbsum, const int* live,
                                             const int* row_last, int row_step, int rowB, void* stream) {
    if (!dDpre || !h || !hmap || !workspace || !dW11 || !dcbsum) return SP_ENULL;
    AxisCls ay, ax;
    if (C % 4 || B < 1 || nsel < 1 || nheads < 1 || !make_axis(Hm, ay) || !make_axis(Wm, ax)) return SP_EINVAL;
    if (row_last && (rowB < 1 || B % rowB)) return SP_EINVAL;
    if (!row_last) rowB = B;
    const int ncls = ay.ncls * ax.ncls;
    hipLaunchKernelGGL(drt_bwd_weight_kernel, dim3(NV * ncls, B, (nsel + 1) / 2), dim3(128), 0, (hipStream_t)stream, dDpre, h, B, C / 4,
                       nsel, ncls, ay, ax, (float*)workspace, row_last, row_step, live, rowB);
    SP_LAUNCH_CHECK();
    const int64_t per_head4 = (int64_t)ncls * NV * (C / 4);
    hipLaunchKernelGGL(drt_slab_reduce_kernel, dim3(ew_grid(nheads * per_head4)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, hmap, B, nsel, nheads, per_head4, dW11, row_last, row_step, live, rowB);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(drt_dcbsum_kernel, dim3(ncls, nheads), dim3(64), 0, (hipStream_t)stream, dDpre, hmap, B, nsel, ay, ax,
                       dcbsum);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
extern "C" int sp_drt_direct_bwd_weight_rows(const float* dDpre, const float* h, const int* hmap, int B, int Hm, int Wm, int C,
                                             int nsel, int nheads, void* workspace, float* dW11, float* dcbsum, const int* row_last,
                                             int row_step, void* stream) {
    return sp_drt_direct_bwd_weight_live(dDpre, h, hmap, B, Hm, Wm, C, nsel, nheads, workspace, dW11, dcbsum, nullptr, row_last, row_step, B, stream);
}
extern "C" int sp_drt_direct_bwd_weight(const float* dDpre, const float* h, const int* hmap, int B, int Hm, int Wm, int C,
                                        int nsel, int nheads, void* workspace, float* dW11, float* dcbsum, void* stream) {
    return sp_drt_direct_bwd_weight_rows(dDpre, h, hmap, B, Hm, Wm, C, nsel, nheads, workspace, dW11, dcbsum, nullptr, 0, stream);
}
