// Skinny fp32 GEMM: C[M <= 64][N] = relu?(alpha * A[M][K] * op(B) + bias) on v_mfma_f32_16x16x4_f32 (exact fp32 products and sums).
//
// The decode loop applies three dense layers per step to a handful of rows -- spatial_embed (64 x 2560 x 2560), semantic_embed
// (64 x 512 x 512) and the contraction of the semantic memory with the rank-1 gate filters (32 x 13824 x 512, once per stream):
// AiR/models/baseline_attention.py:207-208,279-286,40-50 -- and their data gradients in backward.  Their cost is the weight matrix
// streamed ONCE (26-28 MB: ~6 us at HBM speed); the 128 x 128-tile kernel of conv_igemm.hip put them on 20-108 workgroups with a serial
// K loop each and took 135-185 us per launch (profiles/r05_async_hgate_window.log), 3 of the 0.565 ms between two h-gate convs.
// Here a workgroup owns a thin column block of the output for ALL rows and a slice of K:
//   layout 0 ("nk", B [N][K]): 16 columns; lane (l16, g4) loads float4 B[n0 + l16][k0 + 4 g4 ..] and float4 A[16 i + l16][k0 + 4 g4 ..]
//            per 16-k block and issues 4 MFMAs per row tile (element e of both: the same k permutation on both sides, the sum is
//            order-free) -- D[row 4 g4 + r][col l16];
//   layout 1 ("kn", B [K][N]): 32 columns; per 16-k block four float2 B[k0 + 4 g4 + e][n0 + 2 l16 ..] (128 contiguous bytes per k row)
//            against the same A fragments; column tile c of the MFMA holds output column n0 + 2 l16 + c, so a lane ends up with two
//            consecutive columns (float2 stores).
// The 4 waves of a workgroup take the 16-k blocks of the slice round-robin and are summed through LDS in wave order; K slices
// (gridDim.y) leave partial [z][M][N] tiles that a second kernel sums in slice order: bitwise reproducible, no atomics.
// FOOTPRINT: both layouts stay under 80 VGPRs and 4 KB of LDS on purpose.  In backward these GEMMs sit in the chain of small launches
// that runs BESIDE the h-gate conv's data gradient (functional._GateConvLstm.backward): that kernel holds 2 x 216 of a SIMD's 512
// registers and 148 of the CU's 160 KB of LDS for 0.6 ms per tile, and a workgroup that fits into the remainder starts at once while one
// that does not waits for a tile to end (the 64-column build of layout 1 -- 128 VGPRs, 16 KB -- took 0.4-0.6 ms per launch there,
// profiles/r05_async_dgrad_window.log, against 15 us alone).
#include "common.h"
#include <algorithm>

namespace {

struct SkArgs {
    const float* A;
    const float* B;
    const float* bias;
    float* C;        // final output (nsplit == 1) ...
    float* part;     // ... or partial tiles [nsplit][M][N]
    int M, N, K, lda, ldb, ldc;
    int kchunk, nsplit;
    float alpha;
    int relu;
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int LAYOUT>
__global__ __launch_bounds__(256, 6) void skinny_kernel(SkArgs p) {      // 6 waves per SIMD: VGPRs + AGPRs <= 80 (see FOOTPRINT above)
    constexpr int NCT = LAYOUT == 0 ? 1 : 2;                  // column tiles per wave
    __shared__ __attribute__((aligned(16))) float red[4][64][4];      // one column tile at a time
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l16 = lane & 15, g4 = lane >> 4;
    const int n0 = blockIdx.x * (LAYOUT == 0 ? 16 : 32);
    const int kbeg = blockIdx.y * p.kchunk, kend = min(p.K, kbeg + p.kchunk);
    const int mt = (p.M + 15) >> 4;                           // row tiles in use (scalar)
    f32x4 acc[4][NCT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = kbeg + 16 * wave; k0 < kend; k0 += 64) {
        float4 a4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * i + l16;
            a4[i] = (i < mt && row < p.M) ? ld4(p.A + (int64_t)row * p.lda + k0 + 4 * g4) : z4;
        }
        if constexpr (LAYOUT == 0) {
            const float4 b4 = (n0 + l16 < p.N) ? ld4(p.B + (int64_t)(n0 + l16) * p.ldb + k0 + 4 * g4) : z4;
            const float be[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < mt) {
                    const float ae[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], be[e], acc[i][0], 0, 0, 0);
                }
        } else {
            float2 b2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                b2[e] = (n0 + 2 * l16 < p.N) ? *reinterpret_cast<const float2*>(p.B + (int64_t)(k0 + 4 * g4 + e) * p.ldb + n0 + 2 * l16)
                                             : make_float2(0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < mt) {
                    const float ae[4] = {a4[i].x, a4[i].y, a4[i].z, a4[i].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], b2[e].x, acc[i][0], 0, 0, 0);
                        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ae[e], b2[e].y, acc[i][1], 0, 0, 0);
                    }
                }
        }
    }
    // the four waves' partial sums, row tile by row tile, in wave order; wave w finishes the lanes' values of column tile / row group
    float* dst = p.nsplit == 1 ? p.C : p.part + (int64_t)blockIdx.y * p.M * p.N;
    const int ldd = p.nsplit == 1 ? p.ldc : p.N;
    const bool fin = p.nsplit == 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                             // (mt is uniform: the barriers below are taken by all or by none)
        if (i >= mt) break;
        float v[NCT][4];
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            __syncthreads();
            *reinterpret_cast<float4*>(red[wave][lane]) = make_float4(acc[i][c][0], acc[i][c][1], acc[i][c][2], acc[i][c][3]);
            __syncthreads();
            if (wave == 0) {
                float4 s = *reinterpret_cast<const float4*>(red[0][lane]);
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const float4 x = *reinterpret_cast<const float4*>(red[w][lane]);
                    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
                }
                v[c][0] = s.x; v[c][1] = s.y; v[c][2] = s.z; v[c][3] = s.w;
            }
        }
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * i + 4 * g4 + r;
                if (row >= p.M) continue;
                if constexpr (LAYOUT == 0) {
                    const int n = n0 + l16;
                    if (n < p.N) {
                        float x = v[0][r];
                        if (fin) {
                            x = p.alpha * x + (p.bias ? p.bias[n] : 0.f);
                            if (p.relu) x = fmaxf(x, 0.f);
                        }
                        dst[(int64_t)row * ldd + n] = x;
                    }
                } else {
                    const int n = n0 + 2 * l16;
                    if (n < p.N) {
                        float2 x = make_float2(v[0][r], v[1][r]);
                        if (fin) {
                            const float2 bv = p.bias ? *reinterpret_cast<const float2*>(p.bias + n) : make_float2(0.f, 0.f);
                            x.x = p.alpha * x.x + bv.x; x.y = p.alpha * x.y + bv.y;
                            if (p.relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); }
                        }
                        *reinterpret_cast<float2*>(dst + (int64_t)row * ldd + n) = x;
                    }
                }
            }
        }
    }
}

// C[m][n] = relu?(alpha * sum_z part[z][m][n] + bias[n]), slices in order
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float* part, int nsplit, int M, int N, int ldc, const float* bias,
                                                            float alpha, int relu, float* C) {
    const int total4 = M * (N / 4);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total4; i += gridDim.x * 256) {
        const int m = i / (N / 4), n = (i - m * (N / 4)) * 4;
        float4 s = ld4(part + (int64_t)m * N + n);
        for (int z = 1; z < nsplit; ++z) {
            const float4 x = ld4(part + ((int64_t)z * M + m) * N + n);
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        const float4 bv = bias ? ld4(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        s.x = alpha * s.x + bv.x; s.y = alpha * s.y + bv.y; s.z = alpha * s.z + bv.z; s.w = alpha * s.w + bv.w;
        if (relu) { s.x = fmaxf(s.x, 0.f); s.y = fmaxf(s.y, 0.f); s.z = fmaxf(s.z, 0.f); s.w = fmaxf(s.w, 0.f); }
        *reinterpret_cast<float4*>(C + (int64_t)m * ldc + n) = s;
    }
}

// K slices so that the launch has ~512 workgroups; a slice is a multiple of 64 k (16 per wave)
void sk_plan(int N, int K, int layout, int& col_wgs, int& nsplit, int& kchunk) {
    col_wgs = (int)sp_cdiv(N, layout == 0 ? 16 : 32);
    const int want = std::max(1, std::min(512 / std::max(col_wgs, 1), K / 64));
    kchunk = (int)(sp_cdiv(sp_cdiv(K, want), 64) * 64);
    nsplit = (int)sp_cdiv(K, kchunk);
}

bool sk_applies(int M, int N, int K, int lda, int ldb, int ldc, int layout) {
    if (M < 1 || M > 64 || N < 16 || K < 16 || K % 16 || lda % 4 || ldb % 4) return false;
    return layout == 0 ? (N % 16 == 0 && ldb >= K && N % 4 == 0) : (N % 64 == 0 && ldb >= N && ldc % 4 == 0);
}

}  // namespace

extern "C" int sp_gemm_skinny_applies(int M, int N, int K, int lda, int ldb, int ldc, int layout) {
    return (layout == 0 || layout == 1) && sk_applies(M, N, K, lda, ldb, ldc, layout) ? 1 : 0;
}

extern "C" int64_t sp_gemm_skinny_workspace(int M, int N, int K, int layout) {
    if (layout != 0 && layout != 1) return 0;
    int cw, ns, kc;
    sk_plan(N, K, layout, cw, ns, kc);
    return ns > 1 ? (int64_t)ns * M * N * (int64_t)sizeof(float) : 0;
}

extern "C" int sp_gemm_skinny(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                              int layout, float alpha, int relu, void* workspace, void* stream) {
    if (!A || !B || !C) return SP_ENULL;
    if ((layout != 0 && layout != 1) || !sk_applies(M, N, K, lda, ldb, ldc, layout)) return SP_EINVAL;
    if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)workspace) & 15) return SP_EINVAL;
    if (layout == 1 && (((uintptr_t)C | (uintptr_t)bias) & 15)) return SP_EINVAL;      // float4 stores / bias loads ("nk" touches them by element)
    SkArgs a;
    a.A = A; a.B = B; a.bias = bias; a.C = C; a.part = (float*)workspace;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.alpha = alpha; a.relu = relu;
    int cw;
    sk_plan(N, K, layout, cw, a.nsplit, a.kchunk);
    if (a.nsplit > 1 && !workspace) return SP_ENULL;
    if (a.nsplit > 1 && ((((uintptr_t)C | (uintptr_t)bias) & 15) || ldc % 4)) return SP_EINVAL;      // the slice reduce moves float4
    hipStream_t s = (hipStream_t)stream;
    if (layout == 0) hipLaunchKernelGGL(skinny_kernel<0>, dim3((unsigned)cw, (unsigned)a.nsplit), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(skinny_kernel<1>, dim3((unsigned)cw, (unsigned)a.nsplit), dim3(256), 0, s, a);
    SP_LAUNCH_CHECK();
    if (a.nsplit > 1) {
        const int blocks = std::max(1, std::min((M * (N / 4) + 255) / 256, 1024));
        hipLaunchKernelGGL(skinny_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)workspace, a.nsplit, M, N, ldc, bias, alpha,
                           relu, C);
        SP_LAUNCH_CHECK();
    }
    return SP_OK;
}
