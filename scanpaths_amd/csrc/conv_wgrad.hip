// Weight gradient as a TN GEMM on the fp32 matrix pipe:
//
//   dW[co][tap][ci] = alpha * sum_m dY[m][co] * X[pix(m,tap)][ci]   (+ beta * dW)
//
// Both operands are reduction-major in memory (NHWC: the pixel index m is the slow axis), so the K-tile
// (32 pixels) is staged as [pixel][channel] rows exactly as it lies in HBM (512-byte coalesced runs) and
// the MFMA fragments are read with conflict-free ds_read_b32 (consecutive lanes = consecutive channels).
// The reduction over pixels is split over `splits` workgroup columns, each writing a partial slab; a
// second kernel sums the slabs in a fixed order -> bitwise run-to-run reproducible (the reference asks for
// cudnn.deterministic, AiR/train.py:40-41), no float atomics.
#include "common.h"
#include <algorithm>

namespace {

struct WgradArgs {
    const float* X;
    const float* dY;
    float* out;        // final dW (splits == 1) or slab base
    int64_t M;         // pixels
    int Hi, Wi, Ci, ldx;
    int Ho, Wo, Co, ldy;
    int KH, KW, stride, pad, dil;
    int Ntot;          // taps * Ci
    int ldo;
    int tiles_n;
    int splits;
    int64_t rows_per_split;   // multiple of 32
    int64_t slab_stride;      // elements between split slabs (splits > 1)
    float alpha;
    int beta;
    int64_t strideX, strideY, strideO;
};

constexpr int BM = 128, BN = 128, BKP = 32, LDS_LD = 132;

__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 stages x (A,B) x [32][132]
    constexpr int TILE = BKP * LDS_LD;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = lid % p.tiles_n, tmi = lid / p.tiles_n;
    const int co0 = tmi * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int bz = blockIdx.z;

    const float* X = p.X + (int64_t)bz * p.strideX;
    const float* dY = p.dY + (int64_t)bz * p.strideY;

    const int64_t m_begin = (int64_t)split * p.rows_per_split;
    const int64_t m_end = min(p.M, m_begin + p.rows_per_split);
    const int nkt = (int)((m_end - m_begin + BKP - 1) / BKP);

    // thread's fixed column group
    const int cq = t & 31;          // float4 column group
    const int prow0 = t >> 5;       // rows prow0 + 8*j
    const int co_col = co0 + cq * 4;
    const bool a_colok = co_col < p.Co;
    const int ncol = n0 + cq * 4;
    const bool b_colok = ncol < p.Ntot;
    const int tap = b_colok ? ncol / p.Ci : 0;
    const int ci = b_colok ? ncol - tap * p.Ci : 0;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int HoWo = p.Ho * p.Wo;

    float4 ra[4], rb[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m_begin + (int64_t)kt * BKP + prow0 + 8 * j;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
            if (m < m_end) {
                if (a_colok) va = *reinterpret_cast<const float4*>(dY + m * p.ldy + co_col);
                if (b_colok) {
                    const int b = (int)(m / HoWo);
                    const int rem = (int)(m - (int64_t)b * HoWo);
                    const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
                    const int iy = yo * p.stride - p.pad + ky * p.dil;
                    const int ix = xo * p.stride - p.pad + kx * p.dil;
                    if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi)
                        vb = *reinterpret_cast<const float4*>(X + (((int64_t)b * p.Hi + iy) * p.Wi + ix) * p.ldx + ci);
                }
            }
            ra[j] = va;
            rb[j] = vb;
        }
    };
    auto store_tile = [&](float* sA, float* sB) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4*>(sA + (prow0 + 8 * j) * LDS_LD + cq * 4) = ra[j];
            *reinterpret_cast<float4*>(sB + (prow0 + 8 * j) * LDS_LD + cq * 4) = rb[j];
        }
    };

    // two-level accumulation (see conv_igemm.hip): fold the running chunk into `tot` every 8 K-tiles (256 pixels)
    constexpr int CHUNK = 8;
    f32x16 acc[2][2], tot[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                tot[i][j][r] = 0.f;
            }

    if (nkt > 0) {
        load_tile(0);
        store_tile(smem, smem + TILE);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) load_tile(kt + 1);
        const float* sA = smem + (kt & 1) * 2 * TILE;
        const float* sB = sA + TILE;
        const float* pa = sA + wm * 64 + l32;
        const float* pb = sB + wn * 64 + l32;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = kg * 8 + 4 * h + e;
                const float a0 = pa[k * LDS_LD], a1 = pa[k * LDS_LD + 32];
                const float b0 = pb[k * LDS_LD], b1 = pb[k * LDS_LD + 32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
        if ((kt & (CHUNK - 1)) == CHUNK - 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    tot[i][j] += acc[i][j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
        }
        if (more) {
            float* nA = smem + ((kt + 1) & 1) * 2 * TILE;
            store_tile(nA, nA + TILE);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) tot[i][j] += acc[i][j];

    float* out = p.out + (int64_t)bz * p.strideO + (p.splits > 1 ? (int64_t)split * p.slab_stride : 0);
    const bool direct = p.splits == 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + l32;
        if (n >= p.Ntot) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (co < p.Co) {
                    float* dst = out + (int64_t)co * p.ldo + n;
                    if (direct) {
                        float v = p.alpha * tot[i][j][r];
                        if (p.beta) v += *dst;
                        *dst = v;
                    } else {
                        *dst = tot[i][j][r];
                    }
                }
            }
        }
    }
}

// out[co][n] = alpha * sum_s slab[s][co][n] (+ beta*out)
__global__ void wgrad_reduce_kernel(const float* slab, float* out, int Co, int Ntot, int ldo, int splits,
                                    int64_t slab_stride, float alpha, int beta) {
    const int64_t total = (int64_t)Co * Ntot;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i / Ntot), n = (int)(i - (int64_t)co * Ntot);
        const int64_t off = (int64_t)co * ldo + n;
        float s = 0.f;
        for (int k = 0; k < splits; ++k) s += slab[(int64_t)k * slab_stride + off];
        s *= alpha;
        if (beta) s += out[off];
        out[off] = s;
    }
}

int choose_splits(const sp_wgrad_desc* d) {
    if (d->nbatch > 1) return 1;
    const int64_t M = (int64_t)d->N_img * d->Ho * d->Wo;
    const int64_t tiles = sp_cdiv(d->Co, BM) * sp_cdiv((int64_t)d->KH * d->KW * d->Ci, BN);
    // 2 workgroups fit per CU (LDS) -> 512 run together; the launch lasts  rounds x (K-tiles of a split + ~3 K-tiles of fixed cost),
    // rounds = ceil(tiles x splits / 512)  (this kernel's K-tile is long: register-staged loads, one barrier pair per tile).  The old
    // rule capped the few-tile shapes at 128 splits: the stem's weight gradient (2 tiles) ran 256 workgroups of 320 K-tiles on half
    // the slots (1.15 ms), the 64 x 64 pointwise one 128 workgroups of 80 (0.32 ms).  >= 8 K-tiles per split.
    const int64_t smax = std::min<int64_t>(512, std::max<int64_t>(1, M / 256));
    int64_t best = 1;
    double best_cost = 1e30;
    for (int64_t sp = 1; sp <= smax; ++sp) {
        const double rounds = (double)sp_cdiv(tiles * sp, 512);
        const double cost = rounds * ((double)sp_cdiv(sp_cdiv(M, sp), BKP) + 3.0) + 0.02 * (double)sp;
        if (cost < best_cost * 0.999) {
            best_cost = cost;
            best = sp;
        }
    }
    return (int)best;
}

}  // namespace

extern "C" int64_t sp_conv_wgrad_workspace(const sp_wgrad_desc* d) {
    if (!d) return 0;
    const int s = choose_splits(d);
    if (s <= 1) return 0;
    return (int64_t)s * d->Co * d->ldo * (int64_t)sizeof(float);
}

extern "C" int sp_conv_wgrad(const sp_wgrad_desc* d, const float* X, const float* dY, float* dW, void* workspace,
                             void* stream) {
    if (!d || !X || !dY || !dW) return SP_ENULL;
    if (d->Ci % 4 || d->ldx % 4 || d->ldy % 4 || d->Co % 4) return SP_EINVAL;
    if (((uintptr_t)X | (uintptr_t)dY) & 15) return SP_EINVAL;
    if (d->nbatch < 1) return SP_EINVAL;
    WgradArgs a;
    a.X = X; a.dY = dY;
    a.M = (int64_t)d->N_img * d->Ho * d->Wo;
    a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.ldy = d->ldy;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.Ntot = d->KH * d->KW * d->Ci;
    a.ldo = d->ldo;
    a.tiles_n = (int)sp_cdiv(a.Ntot, BN);
    a.splits = choose_splits(d);
    if (a.splits > 1 && !workspace) return SP_ENULL;
    a.rows_per_split = sp_cdiv(sp_cdiv(a.M, a.splits), BKP) * BKP;
    a.slab_stride = (int64_t)d->Co * d->ldo;
    a.out = a.splits > 1 ? (float*)workspace : dW;
    a.alpha = d->alpha; a.beta = d->beta;
    a.strideX = d->strideX; a.strideY = d->strideY; a.strideO = d->strideO;
    if (a.M <= 0) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t grid = sp_cdiv(d->Co, BM) * a.tiles_n;
    const size_t lds = 2 * 2 * BKP * LDS_LD * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(wgrad_kernel, dim3((unsigned)grid, (unsigned)a.splits, (unsigned)d->nbatch), dim3(256), lds, s, a);
    SP_LAUNCH_CHECK();
    if (a.splits > 1) {
        const int64_t total = (int64_t)d->Co * a.Ntot;
        const int blocks = (int)std::min<int64_t>(sp_cdiv(total, 256), 4096);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, dW, d->Co,
                           a.Ntot, d->ldo, a.splits, a.slab_stride, d->alpha, d->beta);
        SP_LAUNCH_CHECK();
    }
    return SP_OK;
}
