// Shared helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/scanpaths_amd.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SP_LAUNCH_CHECK()                          \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

// Process-wide switches (include/scanpaths_amd.h sp_set_tuning); -1 = built-in default.  The PRODUCT library
// (libscanpaths_amd.so) knows exactly one: "amax_reset".  Every other selector -- the halo-block A/B switch, a forced split count of the weight
// gradient, wrong-result timing modes (no loads / no MFMAs / ...) -- exists only in the TIMING build (-DSP_TIMING_VARIANTS ->
// libscanpaths_amd_timing.so, `make timing`, loaded by tools/ through SP_LIBRARY=timing); the product build reads no environment variable.
// (Schedule variants that lost their A/B were removed in round 3; their measurements are in DESIGN.md.)
enum { SP_TUNE_AMAX_RESET = 0, SP_TUNE_HW_SPLITS = 1, SP_TUNE_H2_HALO = 2, SP_TUNE_H2_DBG = 3, SP_TUNE_HW_DBG = 4, SP_TUNE_B3_DBG = 5,
       SP_TUNE_ROW_ORDER = 6, SP_TUNE_HW_CAP = 7, SP_TUNE_COUNT = 8 };
extern int sp_tuning_values[SP_TUNE_COUNT];
#ifdef SP_TIMING_VARIANTS
static inline int sp_tuning_get(int key, int dflt) { return sp_tuning_values[key] < 0 ? dflt : sp_tuning_values[key]; }
#else
static inline int sp_tuning_get(int key, int dflt) {      // product build: only amax_reset is settable
    return (key == SP_TUNE_AMAX_RESET && sp_tuning_values[key] >= 0) ? sp_tuning_values[key] : dflt;
}
#endif

// A producer's fused-amax slot is zeroed IN STREAM ORDER by the producer's own launcher, so every launch -- eager or a HIP-graph
// replay on new inputs -- starts from 0 instead of max(old, new).  A one-thread KERNEL, not hipMemsetAsync: 4-byte memset nodes
// captured into a HIP graph did not reliably precede the kernels that follow them on replay (replays at a different input
// amplitude produced wrong operand scales, tools/graph_debug.py), kernel nodes do.
namespace {
__global__ void sp_zero_words_kernel(unsigned* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0u;
}
}  // namespace
// sp_set_tuning("amax_reset", 1): the caller hands in ZEROED slots (the Python host draws every slot from a zero-filled pool and never
// reuses one), so the reset node is only added while the stream is being captured into a graph (a replay re-uses the slot);
// ~350 one-thread launches per training step otherwise.  Default 0: always reset.
static inline bool sp_reset_needed(hipStream_t s) {
    if (sp_tuning_get(SP_TUNE_AMAX_RESET, 0) != 1) return true;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) return true;
    return st != hipStreamCaptureStatusNone;
}
#define SP_RESET_AMAX(ptr, stream)                                                                                          \
    do {                                                                                                                    \
        if ((ptr) && sp_reset_needed((hipStream_t)(stream))) {                                                              \
            hipLaunchKernelGGL(sp_zero_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)(stream), (unsigned*)(ptr), 1);     \
            SP_LAUNCH_CHECK();                                                                                              \
        }                                                                                                                   \
    } while (0)


static inline int64_t sp_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sh4) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[w] = v;
    __syncthreads();
    return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}
// four independent block-wide sums at once (same partition and order as block_sum_256, hence the same bits): one pair of barriers
// for four values -- the attention kernels' per-entry dot products were a chain of T dependent load -> reduce -> barrier rounds
__device__ __forceinline__ void block_sum4_256(float (&v)[4], float* sh16) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sh16[w * 4 + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = sh16[k] + sh16[4 + k] + sh16[8 + k] + sh16[12 + k];
}
__device__ __forceinline__ float block_max_256(float v, float* sh4) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh4[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh4[0], sh4[1]), fmaxf(sh4[2], sh4[3]));
}

// Fused amax of a producer's output (operand scale of the 2xfp16 GEMMs, conv_f16x2.hip): block maximum, then at most one
// atomicMax per block -- skipped when the block cannot raise the current value (monotone, so a stale read is safe).  256 threads.
__device__ __forceinline__ void block_amax_commit(float m, unsigned* amax, float* sh4) {
    m = block_max_256(m, sh4);
    if (threadIdx.x == 0 && m > 0.f) {
        const unsigned b = __float_as_uint(m);
        if (b > *reinterpret_cast<volatile unsigned*>(amax)) atomicMax(amax, b);
    }
}
// ---- gate non-linearities of the ConvLSTM cell (decoder.hip, the fused cell epilogue of conv_f16x2.hip) ----
// sigmoid(x) = rcp(1 + exp2(-x * log2 e)) on the hardware's 1-ulp v_exp_f32 / v_rcp_f32 (6 VALU instructions; `1.f / (1.f + expf(-x))`
// compiles to 27, tanhf to 40, and the fused epilogue evaluates 3 + 1 of them for 64 elements per lane: ~15 us of its ~27 us per tile
// were these).  Error: the argument product adds |x| * 2^-24 relative to exp(-x), which sigmoid damps by s(1 - s): <= 1.3e-8 absolute
// over all x, below half an ulp of the result's range; with the rcp ~2 ulp in total.  tanh(x) = 1 - 2 * rcp(1 + exp2(2x log2 e)):
// absolute error ~1e-7 (one ulp of 1.0; relative accuracy is lost for |x| << 1, where the cell only needs g = tanh(pre) to an
// absolute 1e-7 next to c of order 1).  Saturates correctly: exp2 -> inf gives rcp 0.
__device__ __forceinline__ float sp_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float sp_tanh(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// ---- 2xfp16 operand split (conv_f16x2.hip; shared with the producers that emit split operands directly) ----
// power-of-two scale with amax * s in [8192, 16384)
__device__ __forceinline__ float scale_of(unsigned amax_bits) {
    const float a = __uint_as_float(amax_bits);
    if (!(a > 0.f) || !(a < INFINITY)) return 1.f;
    int e;
    (void)frexpf(a, &e);                 // a = f * 2^e, f in [0.5, 1)
    e = 14 - e;                          // a * 2^(14-e) in [8192, 16384)
    e = max(-126, min(126, e));
    return ldexpf(1.f, e);
}

__device__ __forceinline__ void split2(float v, float s, uint16_t& a, uint16_t& b) {
    const float xs = v * s;
    const _Float16 x1 = (_Float16)xs;
    const float r1 = xs - (float)x1;
    const _Float16 x2 = (_Float16)r1;
    a = __builtin_bit_cast(uint16_t, x1);
    b = __builtin_bit_cast(uint16_t, x2);
}

// Store the two fp16 planes of 4 consecutive elements per lane (pa, pb: float4 index i = wave base + lane) in the split layout
// [16-element group][plane][16] as ONE 16-byte store per lane, 1 KB contiguous per wave: the four lanes of a quad own one 64-byte
// group; lane r of the quad collects (quad_perm DPP moves, no LDS) the 8 halves of plane r>>1 that belong at bytes 16 r .. 16 r + 15.
// Two 8-byte stores per lane straight from pa / pb leave 32-byte holes per instruction and ran at ~2 TB/s.
// Call from wave-uniform control flow (all 64 lanes execute; `live` must be uniform per quad: element counts are multiples of 16).
__device__ __forceinline__ void store_planes_quad(uint16_t* __restrict__ planes, int64_t i, bool live, ushort4 pa, ushort4 pb) {
    const int a0 = (int)((unsigned)pa.x | ((unsigned)pa.y << 16)), a1 = (int)((unsigned)pa.z | ((unsigned)pa.w << 16));
    const int b0 = (int)((unsigned)pb.x | ((unsigned)pb.y << 16)), b1 = (int)((unsigned)pb.z | ((unsigned)pb.w << 16));
    const bool lo = (threadIdx.x & 2) == 0;                 // quad lanes 0, 1 write plane 0
    // all eight moves execute on every lane (a DPP read of a lane that is switched off returns the `old` operand)
    const int a0e = __builtin_amdgcn_update_dpp(0, a0, 0x88, 0xf, 0xf, false), a1e = __builtin_amdgcn_update_dpp(0, a1, 0x88, 0xf, 0xf, false);
    const int a0o = __builtin_amdgcn_update_dpp(0, a0, 0xdd, 0xf, 0xf, false), a1o = __builtin_amdgcn_update_dpp(0, a1, 0xdd, 0xf, 0xf, false);
    const int b0e = __builtin_amdgcn_update_dpp(0, b0, 0x88, 0xf, 0xf, false), b1e = __builtin_amdgcn_update_dpp(0, b1, 0x88, 0xf, 0xf, false);
    const int b0o = __builtin_amdgcn_update_dpp(0, b0, 0xdd, 0xf, 0xf, false), b1o = __builtin_amdgcn_update_dpp(0, b1, 0xdd, 0xf, 0xf, false);
    uint4 w;
    w.x = (unsigned)(lo ? a0e : b0e);
    w.y = (unsigned)(lo ? a1e : b1e);
    w.z = (unsigned)(lo ? a0o : b0o);
    w.w = (unsigned)(lo ? a1o : b1o);
    if (live) reinterpret_cast<uint4*>(planes)[i] = w;
}

__device__ __forceinline__ float amax4(float m, float a, float b, float c, float d) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(a), fabsf(b))), fmaxf(fabsf(c), fabsf(d)));
}

// Bijective XCD-aware remap of a 1-D block id: blocks that share an XCD (same id % 8 under the
// observed round-robin dispatch) receive a contiguous range of logical tile ids, so neighbouring
// tiles (which share operand panels) hit the same per-XCD L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Logical tile id -> (tile_m, tile_n) with 4x8 super-tiles: the 32 workgroups that run together on one XCD (32 CUs, one
// workgroup each for the LDS-heavy GEMM kernels) cover 4 M-tiles x 8 N-tiles, so an XCD's L2 serves 4 activation panels and
// 8 weight slices to 32 tiles (1.75 MB of operand per tile at K=4608) instead of 2 x 16 (2.2 MB).  Falls back to
// n-fastest order when the grid is not a multiple of the super-tile.
// tall = true: 8 M-tiles x 4 N-tiles per XCD group instead of 4 x 8 -- for the channel-block-major K order, where consecutive
// K-tiles re-read the same activation pixels (an activation panel costs ~1.5/9 of its bytes from beyond L2), so the weight slices
// dominate the fabric traffic and fewer distinct N-tiles per group pay.
__device__ __forceinline__ void supertile_map(int lid, int tiles_m, int tiles_n, int& tm, int& tn, bool tall = false) {
    if (tall && (tiles_n & 3) == 0 && (tiles_m & 7) == 0) {
        const int g = lid >> 5, w = lid & 31;
        const int gn = tiles_n >> 2;
        tm = (g / gn) * 8 + (w >> 2);
        tn = (g % gn) * 4 + (w & 3);
        return;
    }
    if ((tiles_n & 7) == 0 && (tiles_m & 3) == 0) {
        const int g = lid >> 5, w = lid & 31;
        const int gn = tiles_n >> 3;
        tm = (g / gn) * 4 + (w >> 3);
        tn = (g % gn) * 8 + (w & 7);
    } else {
        tm = lid / tiles_n;
        tn = lid % tiles_n;
    }
}
