// Implicit-GEMM convolution / dgrad / batched GEMM on the gfx950 fp32 matrix pipe.
//
//   out[m][n] = relu?( alpha * sum_k A[m][k] * B[k][n] + bias[n] + beta*out[m][n] )
//
// m = output pixel (b, yo, xo) of an NHWC tensor, k = (tap, channel).  A is never materialised: each
// K-tile of 32 channels of one filter tap is gathered straight from the NHWC activation (128-byte
// contiguous runs per pixel -> coalesced HBM/L2 reads), staged through padded LDS rows and fed to
// v_mfma_f32_32x32x2_f32 (exact fp32, bitwise a k-ordered fmaf chain: guide §3 "FP32-input MFMA").
//
// Block tile BM x BN x 32, 256 threads = 4 waves, each wave owns a (BM/WAVES_M) x (BN/WAVES_N)
// sub-tile as TM x TN accumulators of 32x32.  Register-staged double buffering: global loads of tile
// t+1 are issued before the MFMAs of tile t and written to the other LDS buffer afterwards, one
// barrier per K-tile.  LDS rows are padded to 36 floats so ds_read_b128 fragment reads are
// conflict-free (row stride 36 dwords: the 16 lanes of a b128 group land on 16 distinct 4-bank slots).
//
// K ordering trick: a lane's ds_read_b128 returns 4 consecutive k (k0+4h .. k0+4h+3, h = lane>>5);
// MFMA #e consumes element e of both operands, i.e. it sums k = {k0+e, k0+4+e}.  A and B use the same
// permutation, and the sum over k is order-free, so no shuffles are needed.
#include "common.h"
#include <algorithm>

namespace {

struct IgemmArgs {
    const float* X;
    const float* W;
    const float* bias;
    float* C;
    int64_t M;            // rows = N_img*Ho*Wo
    int Hi, Wi, Kc, ldx;
    int Ho, Wo, Nout, ldc;
    int KH, KW, stride, pad, dil;
    int ldw;
    int ncblk;            // ceil(Kc/32)
    int nkt;              // number of K tiles
    int tiles_n;
    float alpha;
    int beta, relu;
    int64_t strideX, strideW, strideC;
    int ksplit, kt_per_split;     // split-K: blockIdx.z handles K-tiles [z*kt_per_split, ...), raw partials to slabs
    int64_t slab_stride;
};

constexpr int BK = 32;
constexpr int LDA = BK + 4;

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, bool SMALLC>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmArgs p) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int LDB = (MODE == 0) ? (BK + 4) : (BN + 4);
    constexpr int A_ELEMS = BM * LDA;
    constexpr int B_ELEMS = (MODE == 0) ? BN * LDB : BK * LDB;
    constexpr int STAGE = A_ELEMS + B_ELEMS;
    constexpr int AR = BM / 32;   // A float4 per thread
    constexpr int BR = BN / 32;   // B float4 per thread
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");

    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int l32 = lane & 31, h = lane >> 5;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = lid % p.tiles_n, tmi = lid / p.tiles_n;
    const int64_t m0 = (int64_t)tmi * BM;
    const int n0 = tn * BN;

    const float* X = p.X + (int64_t)blockIdx.y * p.strideX;
    const float* W = p.W + (int64_t)blockIdx.y * p.strideW;
    float* C = p.C + (int64_t)blockIdx.y * p.strideC;

    // ---- per-thread A rows -------------------------------------------------------------------
    const int kq = t & 7;
    const int HoWo = p.Ho * p.Wo;
    int a_py[AR], a_px[AR];
    int64_t a_boff[AR];
    bool a_ok[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int64_t m = m0 + (t >> 3) + 32 * j;
        a_ok[j] = m < p.M;
        const int64_t mm = a_ok[j] ? m : 0;
        const int b = (int)(mm / HoWo);
        const int rem = (int)(mm - (int64_t)b * HoWo);
        const int yo = rem / p.Wo, xo = rem - yo * p.Wo;
        a_boff[j] = (int64_t)b * p.Hi * p.Wi;
        if (MODE == 0) {
            a_py[j] = yo * p.stride - p.pad;
            a_px[j] = xo * p.stride - p.pad;
        } else {
            a_py[j] = yo + p.pad;
            a_px[j] = xo + p.pad;
        }
    }
    const int taps = p.KH * p.KW;
    const int Ktot = taps * p.Kc;

    // load cursor (uniform): which tap / channel block the NEXT tile to be loaded belongs to
    int ld_ky = 0, ld_kx = 0, ld_cblk = 0, ld_kt = 0;
    int kt_begin = 0, kt_end = p.nkt;
    if (p.ksplit > 1) {           // pure GEMM (one tap): the K-tile index is the channel block
        kt_begin = blockIdx.z * p.kt_per_split;
        kt_end = min(p.nkt, kt_begin + p.kt_per_split);
        ld_kt = kt_begin;
        ld_cblk = kt_begin;
    }

    float4 ra[AR], rb[BR];

    auto load_tile = [&]() {
        // ---- A ----
        int ky, kx, c;
        bool kvalid;
        if (SMALLC) {
            const int tap = ld_kt * 8 + kq;
            kvalid = tap < taps;
            ky = tap / p.KW;
            kx = tap - ky * p.KW;
            c = 0;
        } else {
            ky = ld_ky;
            kx = ld_kx;
            c = ld_cblk * 32 + kq * 4;
            kvalid = c < p.Kc;
        }
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_ok[j] && kvalid) {
                int iy, ix;
                bool ok;
                if (MODE == 0) {
                    iy = a_py[j] + ky * p.dil;
                    ix = a_px[j] + kx * p.dil;
                    ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                } else {
                    const int ty = a_py[j] - ky * p.dil, tx = a_px[j] - kx * p.dil;
                    ok = ty >= 0 && tx >= 0;
                    if (p.stride == 1) {
                        iy = ty;
                        ix = tx;
                    } else {
                        iy = ty / p.stride;
                        ix = tx / p.stride;
                        ok = ok && (iy * p.stride == ty) && (ix * p.stride == tx);
                    }
                    ok = ok && iy < p.Hi && ix < p.Wi;
                }
                if (ok) v = *reinterpret_cast<const float4*>(X + (a_boff[j] + (int64_t)iy * p.Wi + ix) * p.ldx + c);
            }
            ra[j] = v;
        }
        // ---- B ----
        if (MODE == 0) {
            const int k = ld_kt * 32 + kq * 4;
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                const int n = n0 + (t >> 3) + 32 * j;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n < p.Nout && k < Ktot) v = *reinterpret_cast<const float4*>(W + (int64_t)n * p.ldw + k);
                rb[j] = v;
            }
        } else {
            const int tap = ld_ky * p.KW + ld_kx;
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                const int s = t + 256 * j;
                const int krow = s / (BN / 4), nq = s % (BN / 4);
                const int cc = ld_cblk * 32 + krow;
                const int n = n0 + nq * 4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cc < p.Kc && n < p.Nout)
                    v = *reinterpret_cast<const float4*>(W + ((int64_t)cc * taps + tap) * p.ldw + n);
                rb[j] = v;
            }
        }
        // advance the cursor
        ++ld_kt;
        if (!SMALLC) {
            if (++ld_cblk == p.ncblk) {
                ld_cblk = 0;
                if (++ld_kx == p.KW) {
                    ld_kx = 0;
                    ++ld_ky;
                }
            }
        }
    };

    auto store_tile = [&](float* sA, float* sB) {
#pragma unroll
        for (int j = 0; j < AR; ++j)
            *reinterpret_cast<float4*>(sA + ((t >> 3) + 32 * j) * LDA + kq * 4) = ra[j];
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < BR; ++j)
                *reinterpret_cast<float4*>(sB + ((t >> 3) + 32 * j) * LDB + kq * 4) = rb[j];
        } else {
#pragma unroll
            for (int j = 0; j < BR; ++j) {
                const int s = t + 256 * j;
                const int krow = s / (BN / 4), nq = s % (BN / 4);
                *reinterpret_cast<float4*>(sB + krow * LDB + nq * 4) = rb[j];
            }
        }
    };

    // Two-level accumulation: the MFMA chain is an in-order fp32 fmaf chain, whose rounding error grows with
    // the chain length.  K reaches 18432 (sal_conv), so every CHUNK K-tiles (256 k) the running chunk `acc` is
    // folded into `tot` and restarted -- error ~ sqrt(256)+sqrt(K/256) instead of sqrt(K) ulps, which keeps the
    // result within ~2x of a blocked CPU summation (measured against the fp64 oracle).
    constexpr int CHUNK = 8;
    f32x16 acc[TM][TN], tot[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                tot[i][j][r] = 0.f;
            }

    load_tile();
    store_tile(smem, smem + A_ELEMS);
    __syncthreads();

    for (int kt = 0; kt < kt_end - kt_begin; ++kt) {
        const bool more = kt + 1 < kt_end - kt_begin;
        if (more) load_tile();
        const float* sA = smem + (kt & 1) * STAGE;
        const float* sB = sA + A_ELEMS;
        const float* pa = sA + (wm * WM + l32) * LDA + 4 * h;
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(pa + i * 32 * LDA + kg * 8);
            if (MODE == 0) {
                const float* pb = sB + (wn * WN + l32) * LDB + 4 * h;
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(pb + j * 32 * LDB + kg * 8);
            } else {
                const float* pb = sB + (kg * 8 + 4 * h) * LDB + wn * WN + l32;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bf[j].x = pb[0 * LDB + j * 32];
                    bf[j].y = pb[1 * LDB + j * 32];
                    bf[j].z = pb[2 * LDB + j * 32];
                    bf[j].w = pb[3 * LDB + j * 32];
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if ((kt & (CHUNK - 1)) == CHUNK - 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    tot[i][j] += acc[i][j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
        }
        if (more) {
            float* nA = smem + ((kt + 1) & 1) * STAGE;
            store_tile(nA, nA + A_ELEMS);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) tot[i][j] += acc[i][j];

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    if (p.ksplit > 1) {           // raw partial tile -> slab z (dense [M][Nout]); alpha/bias/beta/relu applied by the reducer
        float* slab = p.C + (int64_t)blockIdx.z * p.slab_stride;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + l32;
            if (n >= p.Nout) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (m < p.M) slab[m * p.Nout + n] = tot[i][j][r];
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + l32;
        if (n >= p.Nout) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.M) {
                    float* dst = C + m * p.ldc + n;
                    float v = p.alpha * tot[i][j][r] + bv;
                    if (p.beta) v += *dst;
                    if (p.relu) v = fmaxf(v, 0.f);
                    *dst = v;
                }
            }
        }
    }
}

// out[m][n] = relu?( alpha * sum_z slab[z][m][n] + bias[n] + beta*out[m][n] )
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* slab, int ksplit, int64_t slab_stride, int64_t M, int N,
                                                            int ldc, const float* bias, float alpha, int beta, int relu,
                                                            float* out) {
    const int64_t total = M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = i / N;
        const int n = (int)(i - m * N);
        float s = 0.f;
        for (int z = 0; z < ksplit; ++z) s += slab[(int64_t)z * slab_stride + i];
        float v = alpha * s + (bias ? bias[n] : 0.f);
        float* dst = out + m * ldc + n;
        if (beta) v += *dst;
        if (relu) v = fmaxf(v, 0.f);
        *dst = v;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int MODE, bool SMALLC>
int launch(const IgemmArgs& a, int nbatch, hipStream_t s, void* b_workspace = nullptr) {
    constexpr int LDB = (MODE == 0) ? (BK + 4) : (BN + 4);
    constexpr int STAGE = BM * LDA + ((MODE == 0) ? BN * LDB : BK * LDB);
    const size_t lds = 2 * STAGE * sizeof(float);
    auto kern = igemm_kernel<BM, BN, WAVES_M, WAVES_N, MODE, SMALLC>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int64_t tiles_m = sp_cdiv(a.M, BM);
    IgemmArgs b = a;
    b.tiles_n = (int)sp_cdiv(a.Nout, BN);
    const int64_t grid = tiles_m * b.tiles_n;
    if (grid <= 0 || grid > 0x7fffffff) return SP_EINVAL;
    if (b.ksplit > 1) {
        float* final_out = b.C;
        const float* bias = b.bias;
        b.C = (float*)b_workspace;      // slabs
        b.slab_stride = b.M * b.Nout;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid, 1, (unsigned)b.ksplit), dim3(256), lds, s, b);
        SP_LAUNCH_CHECK();
        const int64_t total = b.M * b.Nout;
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(total, 256), 2048));
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)b_workspace, b.ksplit, b.slab_stride,
                           b.M, b.Nout, b.ldc, bias, b.alpha, b.beta, b.relu, final_out);
        SP_LAUNCH_CHECK();
        return SP_OK;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)nbatch), dim3(256), lds, s, b);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

}  // namespace

extern "C" int sp_conv_igemm(const sp_conv_desc* d, const float* X, const float* W, const float* bias, float* out,
                             void* stream) {
    if (!d || !X || !W || !out) return SP_ENULL;
    if (d->mode != 0 && d->mode != 1) return SP_EINVAL;
    const int taps = d->KH * d->KW;
    const bool smallc = (d->mode == 0 && d->Kc == 4 && taps > 1);
    if (d->Kc % 4 || d->ldx % 4 || d->ldw % 4) return SP_EINVAL;
    if (taps > 1 && !smallc && d->Kc % 32) return SP_EINVAL;
    if (d->mode == 1 && d->Nout % 4) return SP_EINVAL;
    if (((uintptr_t)X | (uintptr_t)W) & 15) return SP_EINVAL;
    if (d->stride < 1 || d->dil < 1 || d->nbatch < 1) return SP_EINVAL;
    IgemmArgs a;
    a.X = X; a.W = W; a.bias = bias; a.C = out;
    a.M = (int64_t)d->N_img * d->Ho * d->Wo;
    a.Hi = d->Hi; a.Wi = d->Wi; a.Kc = d->Kc; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.Nout = d->Nout; a.ldc = d->ldc;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.ldw = d->ldw;
    a.ncblk = (d->Kc + 31) / 32;
    a.nkt = smallc ? (taps + 7) / 8 : taps * a.ncblk;
    a.tiles_n = 0;
    a.alpha = d->alpha; a.beta = d->beta; a.relu = d->relu;
    a.strideX = d->strideX; a.strideW = d->strideW; a.strideC = d->strideC;
    a.ksplit = 1; a.kt_per_split = a.nkt; a.slab_stride = 0;
    if (d->ksplit > 1 && d->workspace && taps == 1 && d->nbatch == 1 && !smallc) {
        a.kt_per_split = (a.nkt + d->ksplit - 1) / d->ksplit;
        a.ksplit = (a.nkt + a.kt_per_split - 1) / a.kt_per_split;
    }
    void* ws = d->workspace;
    if (a.M <= 0 || a.Nout <= 0) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool narrow = d->Nout <= 64;
    if (d->mode == 0) {
        if (smallc) return narrow ? launch<128, 64, 4, 1, 0, true>(a, d->nbatch, s) : launch<128, 128, 2, 2, 0, true>(a, d->nbatch, s);
        return narrow ? launch<128, 64, 4, 1, 0, false>(a, d->nbatch, s, ws) : launch<128, 128, 2, 2, 0, false>(a, d->nbatch, s, ws);
    }
    return narrow ? launch<128, 64, 4, 1, 1, false>(a, d->nbatch, s, ws) : launch<128, 128, 2, 2, 1, false>(a, d->nbatch, s, ws);
}
