// Probe: cycles per K-tile of the fragment-read + MFMA part of the 2xfp16 / 3-product GEMM main loop for two workgroup shapes
//   A  8 waves (2 per SIMD), wave tile 64x64, ping-pong halves, one barrier per K-tile      (= h2_kernel's structure, no global loads)
//   B  4 waves (1 per SIMD), wave tile 128x64, fragments of K-tile t+1 read under the MFMAs of K-tile t (register double buffer)
// Both: block tile 256x128x32, two-level accumulation (fold every 8 K-tiles), LDS stage = [256 + 128 rows][128 B], the ring's chunk
// swizzle.  LDS content is static (no loads): the question is what the LDS-read / MFMA side alone sustains.  Prints ms per launch and
// MFMA-pipe utilisation at the measured time for a grid of 5120 workgroups x 144 K-tiles (the h-gate conv's forward launch).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/wave_tile_probe.hip -o tools/probes/wave_tile_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int HA = 256 * 128, HB = 128 * 128, STAGE = HA + HB;

__device__ __forceinline__ void fill_lds(unsigned char* smem, int nthreads, bool constant) {
    if (constant) {
        for (int i = threadIdx.x; i < 3 * STAGE / 4; i += nthreads) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + (i & 0xff);   // fp16 ~1.0
        __syncthreads();
        return;
    }
    // random finite fp16 pairs (sign random, exponent 0x30..0x3f, random mantissa): the chip's clock under matrix load depends on the data
    for (int i = threadIdx.x; i < 3 * STAGE / 4; i += nthreads) {
        uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        reinterpret_cast<uint32_t*>(smem)[i] = (h & 0x8fff8fffu) | 0x30003000u;
    }
    __syncthreads();
}

// ---- A: 8 waves, 64x64, ping-pong -------------------------------------------------------------------------------------------------
#define GLDS16(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
// MODE 0: no global loads.  MODE 1: 6 LDS-DMA pieces per lane per K-tile (the ring of h2_kernel: tile kt+2 issued in K-tile kt, counted
// vmcnt).  MODE 2: 2 pieces per K-tile (the halo build's steady state).  MODE 3: 6 pieces by global_load_dwordx4 into registers, written
// to LDS by ds_write_b128 one K-tile later.  Sources: a 64 MiB buffer, every workgroup its own sliding window (L2 / MALL resident).
template <int MODE, int EPI = 0>   // EPI 1: each workgroup writes a contiguous 128 KB block; 2: the real conv layout (512-float rows, 4 column tiles)
__global__ __launch_bounds__(512, 2) void probe8(float* out, int nkt, const unsigned char* src, uint32_t srcmask, uint32_t stride4, uint32_t wsrc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool constant = nkt < 0;
    nkt = constant ? -nkt : nkt;
    fill_lds(smem, 512, constant);
    constexpr int NP = MODE == 2 ? 2 : 6;
    const uint32_t lane_off = (uint32_t)(threadIdx.x * 16);
    uint32_t gpos = (uint32_t)blockIdx.x * 1572864u;                 // 1.5 MiB apart
    uint4 stg[6];
    // MODE 4: the conv loader's access pattern: activation piece j of a lane = 16 B of pixel row (wave + 8 j) * 8 + lane / 8 (256 pixel rows
    // `stride4` bytes apart), channel block kt / 9, filter tap kt % 9 (rows shifted by (ky - 1) * 64 + kx - 1 pixels); weights linear.
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int kt_issue = 0;
    const uint32_t tile_row0 = (uint32_t)(blockIdx.x % 320) * 256u + 128u;      // 320 M-tiles of 256 pixels (+ margin for the tap shifts)
    auto issue = [&](int stage_) {
        if constexpr (MODE == 4) {
            unsigned char* st = smem + stage_ * STAGE;
            const int tap = kt_issue % 9, cb = kt_issue / 9;
            const int shift = (tap / 3 - 1) * 64 + (tap % 3 - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t row = tile_row0 + (uint32_t)((wv + 8 * j) * 8 + (threadIdx.x & 63) / 8) + (uint32_t)shift;
                GLDS16(src + (((uint64_t)row * stride4 + (uint32_t)cb * 128u + (threadIdx.x & 7) * 16u) & srcmask), st + (wv + 8 * j) * 1024);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) GLDS16(src + ((wsrc + (uint32_t)kt_issue * 16384u + j * 8192u + lane_off) & srcmask), st + HA + (wv + 8 * j) * 1024);
            ++kt_issue;
        } else if constexpr (MODE == 5) {
            // the pointwise conv's loader: rows of `stride4` bytes (= 4 K bytes: both planes of every channel), K-tile kt = 128 B of each row;
            // the activation tile is shared by the 4 neighbouring workgroups (output-channel tiles), the weights come from a small region
            unsigned char* st = smem + stage_ * STAGE;
            // srcmask bit 0 clear: workgroups of one activation tile on one XCD (the real kernel's map); bit 1 clear: K-tile-major activations
            // ([K-tile][row][128 B]: every K-tile of a tile is one contiguous 32 KB block)
            const bool xmap = (srcmask & 1u) == 0, ktmajor = (srcmask & 2u) == 0;
            const uint32_t lid = xmap ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
            const uint32_t arow0 = (lid >> 2) * 256u, brow0 = (lid & 3) * 128u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t row = arow0 + (uint32_t)((wv + 8 * j) * 8 + (threadIdx.x & 63) / 8);
                const uint64_t off = ktmajor ? ((uint64_t)kt_issue * (gridDim.x / 4u) * 256u + row) * 128u + (threadIdx.x & 7) * 16u
                                             : (uint64_t)row * stride4 + (uint32_t)kt_issue * 128u + (threadIdx.x & 7) * 16u;
                GLDS16(src + (off & (srcmask | 15u)), st + (wv + 8 * j) * 1024);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t row = brow0 + (uint32_t)((wv + 8 * j) * 8 + (threadIdx.x & 63) / 8);
                GLDS16(src + ((wsrc + row * stride4 + (uint32_t)kt_issue * 128u + (threadIdx.x & 7) * 16u) & srcmask), st + HA + (wv + 8 * j) * 1024);
            }
            ++kt_issue;
        } else if constexpr (MODE == 1 || MODE == 2) {
            unsigned char* st = smem + stage_ * STAGE;
#pragma unroll
            for (int j = 0; j < NP; ++j) GLDS16(src + ((gpos + j * 8192u + lane_off) & srcmask), st + (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) + 8 * j) * 1024);
            gpos += NP * 8192u;
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 6; ++j) stg[j] = *reinterpret_cast<const uint4*>(src + ((gpos + j * 8192u + lane_off) & srcmask));
            gpos += 6 * 8192u;
        }
    };
    auto commit = [&](int stage_) {      // MODE 3: registers -> LDS
        if constexpr (MODE == 3) {
            unsigned char* st = smem + stage_ * STAGE;
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<uint4*>(st + j * 8192 + lane_off) = stg[j];
        }
    };
    auto wait_loads = [&](bool more) {
        if constexpr (MODE == 1 || MODE == 4 || MODE == 5) { if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if constexpr (MODE == 2) { if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    };
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, g4 = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int rot = (l16 >> 1) & 7;
    int offA[2], offB[2];
    for (int pl = 0; pl < 2; ++pl) {
        const int pos = ((g4 >> 1) * 4 + pl * 2 + (g4 & 1)) ^ rot;
        offA[pl] = (wm * 64 + l16) * 128 + pos * 16;
        offB[pl] = HA + (wn * 64 + l16) * 128 + pos * 16;
    }
    f32x4 acc[4][4], tot[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    f16x8 af[4][2], bf[4][2];
    auto rd = [&](int stage) {
        const unsigned char* st = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                af[i][pl] = *reinterpret_cast<const f16x8*>(st + offA[pl] + i * 16 * 128);
                bf[i][pl] = *reinterpret_cast<const f16x8*>(st + offB[pl] + i * 16 * 128);
            }
    };
    auto mm = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
            }
    };
    auto fold = [&](int kt) {
        if ((kt & 7) == 7)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { tot[i][j] += acc[i][j]; for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f; }
    };
    const bool late = wave >= 4;
    if (late) __builtin_amdgcn_s_setprio(1);
    int stage = 0;
    auto prev = [](int st_) { return st_ == 0 ? 2 : st_ - 1; };
    if constexpr (MODE == 1 || MODE == 2 || MODE == 4 || MODE == 5) { issue(0); issue(1); wait_loads(true); __builtin_amdgcn_s_barrier(); }
    if (!late) {
        for (int kt = 0; kt < nkt; ++kt) {
            rd(stage);
            if constexpr (MODE == 3) { if (kt > 0) commit(prev(stage)); }
            if (kt + 2 < nkt) issue(prev(stage));
            mm(); fold(kt);
            wait_loads(kt + 2 < nkt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage = stage == 2 ? 0 : stage + 1;
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt > 0) { mm(); fold(kt - 1); }
            rd(stage);
            if constexpr (MODE == 3) { if (kt > 0) commit(prev(stage)); }
            if (kt + 2 < nkt) issue(prev(stage));
            wait_loads(kt + 2 < nkt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage = stage == 2 ? 0 : stage + 1;
        }
        mm(); fold(nkt - 1);
    }
    if constexpr (EPI) {
        // h2_kernel's epilogue: the wave's 64x64 tile through its private 17 KB slice of the idle ring, float4 stores, 256 B per row
        float* stg = reinterpret_cast<float*>(smem) + wave * (64 * 68);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) stg[(i * 16 + 4 * g4 + r) * 68 + j * 16 + l16] = tot[i][j][r] + acc[i][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int cq4 = lane & 15, rsub = lane >> 4;
        const int64_t tile = blockIdx.x;                       // [tiles][256 rows][128 cols] fp32: 128 KB per workgroup
        constexpr int LDC = EPI == 2 ? 512 : 128;
        float* dst0 = (EPI == 2 ? out + (tile >> 2) * (256 * 512) + (tile & 3) * 128 : out + tile * (256 * 128)) + (wm * 64) * LDC + wn * 64 + 4 * cq4;
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const int row = ps * 4 + rsub;
            *reinterpret_cast<float4*>(dst0 + row * LDC) = *reinterpret_cast<const float4*>(stg + row * 68 + 4 * cq4);
        }
        return;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += tot[i][j][r] + acc[i][j][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// ---- B: 4 waves, 128x64, software-pipelined fragment reads ------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void probe4(float* out, int nkt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fill_lds(smem, 256, nkt < 0);
    nkt = nkt < 0 ? -nkt : nkt;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, g4 = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int rot = (l16 >> 1) & 7;
    int offA[2], offB[2];
    for (int pl = 0; pl < 2; ++pl) {
        const int pos = ((g4 >> 1) * 4 + pl * 2 + (g4 & 1)) ^ rot;
        offA[pl] = (wm * 128 + l16) * 128 + pos * 16;
        offB[pl] = HA + (wn * 64 + l16) * 128 + pos * 16;
    }
    f32x4 acc[8][4], tot[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    f16x8 af[2][8][2], bf[2][4][2];
    auto rd = [&](int stage, int buf) {
        const unsigned char* st = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[buf][i][pl] = *reinterpret_cast<const f16x8*>(st + offA[pl] + i * 16 * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bf[buf][j][pl] = *reinterpret_cast<const f16x8*>(st + offB[pl] + j * 16 * 128);
    };
    auto mm = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[buf][i][0], bf[buf][j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[buf][i][1], bf[buf][j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[buf][i][0], bf[buf][j][0], acc[i][j], 0, 0, 0);
            }
    };
    auto fold = [&](int kt) {
        if ((kt & 7) == 7)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { tot[i][j] += acc[i][j]; for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f; }
    };
    int stage = 0;
    rd(0, 0);
    for (int kt = 0; kt < nkt; kt += 2) {          // unrolled by two so that the fragment buffers are compile-time indices
        int nst = stage == 2 ? 0 : stage + 1;
        rd(nst, 1);                                 // K-tile kt+1 under the MFMAs of K-tile kt
        mm(0); fold(kt);
        __builtin_amdgcn_s_barrier();
        stage = nst; nst = stage == 2 ? 0 : stage + 1;
        rd(nst, 0);
        mm(1); fold(kt + 1);
        __builtin_amdgcn_s_barrier();
        stage = nst;
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += tot[i][j][r] + acc[i][j][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// ---- C: hw_kernel's structure: K = pixels, tiles staged pixel-major ([32 pixels][256 co] / [32 pixels][128 ci] rows of 1024 / 512 B),
// fragments by ds_read_b64_tr_b16 pairs (32 LDS instructions per wave and K-tile instead of 16), same MFMAs, ping-pong, 6 LDS-DMA pieces
typedef short short4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 tr_pair(const unsigned char* base, int off0, int off1) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + off0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + off1));
    typedef short short8v __attribute__((ext_vector_type(8)));
    short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ int rot4p(int q) { return 2 * (q & 1) + 8 * (q >> 1); }
__device__ __forceinline__ int swz16p(int r) { return rot4p(r & 3) + 4 * ((r >> 3) & 1); }
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe_hw(float* out, int nkt, const unsigned char* src, uint32_t srcmask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool constant = nkt < 0;
    nkt = constant ? -nkt : nkt;
    fill_lds(smem, 512, constant);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int q = (lane >> 2) & 3, pp = lane & 3, kg = lane >> 4;
    int offA[4][2][2], offB[4][2][2];
    for (int i = 0; i < 4; ++i) for (int pl = 0; pl < 2; ++pl) for (int s2 = 0; s2 < 2; ++s2) {
        const int row = 8 * kg + q;
        const int pa = (((wm << 2) | (pl << 1) | (pp >> 1)) ^ swz16p(row));
        offA[i][pl][s2] = row * 1024 + pa * 16 + (pp & 1) * 8 + i * 256 + s2 * 4 * 1024;
        const int pb = ((((i & 1) << 3) | (wn << 2) | (pl << 1) | (pp >> 1)) ^ swz16p(row));
        offB[i][pl][s2] = HA + row * 512 + pb * 16 + (pp & 1) * 8 + (i >> 1) * 256 + s2 * 4 * 512;
    }
    f32x4 acc[4][4], tot[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    f16x8 af[4][2], bf[4][2];
    auto rd = [&](int stage) {
        const unsigned char* st = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                af[i][pl] = tr_pair(st, offA[i][pl][0], offA[i][pl][1]);
                bf[i][pl] = tr_pair(st, offB[i][pl][0], offB[i][pl][1]);
            }
    };
    auto mm = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
            }
    };
    auto fold = [&](int kt) {
        if ((kt & 7) == 7)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { tot[i][j] += acc[i][j]; for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f; }
    };
    const uint32_t lane_off = (uint32_t)(threadIdx.x * 16);
    uint32_t gpos = (uint32_t)blockIdx.x * 1572864u;
    auto issue = [&](int stage_) {
        if constexpr (MODE == 1) {
            unsigned char* st = smem + stage_ * STAGE;
#pragma unroll
            for (int j = 0; j < 6; ++j) GLDS16(src + ((gpos + j * 8192u + lane_off) & srcmask), st + (wave + 8 * j) * 1024);
            gpos += 6 * 8192u;
        }
    };
    auto wait_loads = [&](bool more) {
        if constexpr (MODE == 1) { if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    };
    const bool late = wave >= 4;
    int stage = 0;
    auto prev = [](int st_) { return st_ == 0 ? 2 : st_ - 1; };
    if constexpr (MODE == 1) { issue(0); issue(1); wait_loads(true); __builtin_amdgcn_s_barrier(); }
    if (!late) {
        for (int kt = 0; kt < nkt; ++kt) {
            rd(stage);
            if (kt + 2 < nkt) issue(prev(stage));
            mm(); fold(kt);
            wait_loads(kt + 2 < nkt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage = stage == 2 ? 0 : stage + 1;
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            if (kt > 0) { mm(); fold(kt - 1); }
            rd(stage);
            if (kt + 2 < nkt) issue(prev(stage));
            wait_loads(kt + 2 < nkt);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage = stage == 2 ? 0 : stage + 1;
        }
        mm(); fold(nkt - 1);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += tot[i][j][r] + acc[i][j][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

// ---- D: persistent short-K kernel: one workgroup per CU walks its tiles; the 3-stage ring runs ACROSS tile boundaries (the first two
// K-tiles of the next tile are in flight while the current tile finishes), every wave stores its 64x64 result through a private 2 KB
// staging slice (8 rows per pass) so the epilogue of one half overlaps the other half's next K-tile.  NKT K-tiles per tile.
template <int NKT>
__global__ __launch_bounds__(512, 2) void probe_persist(float* out, int ntiles, const unsigned char* src, uint32_t srcmask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    fill_lds(smem, 512, false);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l16 = lane & 15, g4 = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int rot = (l16 >> 1) & 7;
    int offA[2], offB[2];
    for (int pl = 0; pl < 2; ++pl) {
        const int pos = ((g4 >> 1) * 4 + pl * 2 + (g4 & 1)) ^ rot;
        offA[pl] = (wm * 64 + l16) * 128 + pos * 16;
        offB[pl] = HA + (wn * 64 + l16) * 128 + pos * 16;
    }
    f32x4 acc[4][4];
    f16x8 af[4][2], bf[4][2];
    auto zero = [&]() { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f; };
    zero();
    auto rd = [&](int stage) {
        const unsigned char* st = smem + stage * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                af[i][pl] = *reinterpret_cast<const f16x8*>(st + offA[pl] + i * 16 * 128);
                bf[i][pl] = *reinterpret_cast<const f16x8*>(st + offB[pl] + i * 16 * 128);
            }
    };
    auto mm = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
            }
    };
    float* stg = reinterpret_cast<float*>(smem + 3 * STAGE) + wave * 512;          // 2 KB per wave behind the ring
    auto epilogue = [&](int tile) {
        const int cq4 = lane & 15, rsub = lane >> 4;
        float* dst0 = out + (int64_t)tile * (256 * 128) + (wm * 64) * 128 + wn * 64 + 4 * cq4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int hpass = 0; hpass < 2; ++hpass) {      // rows i*16 + 4*g4 + r: pass = the 8 rows with g4 in {2*hpass, 2*hpass+1}
                if ((g4 >> 1) == hpass) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) stg[((g4 & 1) * 4 + r) * 64 + j * 16 + l16] = acc[i][j][r];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // 8 rows x 64 floats = 128 float4: two per lane
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int e = lane + 64 * u, row8 = e >> 4, c4 = e & 15;
                    const int row = i * 16 + 8 * hpass + row8;
                    *reinterpret_cast<float4*>(dst0 - 4 * cq4 + row * 128 + 4 * c4) = *reinterpret_cast<const float4*>(stg + row8 * 64 + 4 * c4);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        zero();
    };
    const uint32_t lane_off = (uint32_t)(threadIdx.x * 16);
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = my_tiles * NKT;
    int gi = 0;                                            // next global K-tile index to issue
    auto issue = [&](int stage_) {
        const int tl = (int)blockIdx.x + (gi / NKT) * (int)gridDim.x;
        const uint32_t gpos = (uint32_t)tl * (uint32_t)(NKT * 49152) + (uint32_t)(gi % NKT) * 49152u;
        unsigned char* st = smem + stage_ * STAGE;
#pragma unroll
        for (int j = 0; j < 6; ++j) GLDS16(src + ((gpos + j * 8192u + lane_off) & srcmask), st + (wave + 8 * j) * 1024);
        ++gi;
    };
    auto prev = [](int st_) { return st_ == 0 ? 2 : st_ - 1; };
    const bool late = wave >= 4;
    if (late) __builtin_amdgcn_s_setprio(1);
    int stage = 0;
    issue(0); if (total > 1) issue(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!late) {
        for (int g = 0; g < total; ++g) {
            rd(stage);
            if (g + 2 < total) issue(prev(stage));
            mm();
            if (g % NKT == NKT - 1) epilogue((int)blockIdx.x + (g / NKT) * (int)gridDim.x);
            if (g + 2 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage = stage == 2 ? 0 : stage + 1;
        }
    } else {
        for (int g = 0; g < total; ++g) {
            if (g > 0) {
                mm();
                if ((g - 1) % NKT == NKT - 1) epilogue((int)blockIdx.x + ((g - 1) / NKT) * (int)gridDim.x);
            }
            rd(stage);
            if (g + 2 < total) issue(prev(stage));
            if (g + 2 < total) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage = stage == 2 ? 0 : stage + 1;
        }
        mm();
        epilogue((int)blockIdx.x + ((total - 1) / NKT) * (int)gridDim.x);
    }
}

int main() {
    float* d; hipMalloc(&d, 4096);
    unsigned char* src; const uint32_t maxbytes = 1024u << 20; hipMalloc(&src, (size_t)maxbytes + (1 << 20)); { std::vector<uint32_t> hbuf((size_t)(maxbytes >> 2) + (1 << 18)); uint32_t x = 12345u; for (auto& v : hbuf) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = (x & 0x8fff8fffu) | 0x30003000u; } hipMemcpy(src, hbuf.data(), hbuf.size() * 4, hipMemcpyHostToDevice); }
    unsigned char* srcc; hipMalloc(&srcc, (size_t)maxbytes + (1 << 20)); hipMemset(srcc, 0x3c, (size_t)maxbytes + (1 << 20));
    unsigned char* srcr = src;
    int nkt = 144; const int grid = 5120, lds = 3 * STAGE;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe4), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double mfma_cycles_per_simd = 20.0 * 144 * 96 * 16;      // 20 workgroups per CU, 96 MFMAs per K-tile per SIMD, 16 cycles each
    const char* names[8] = {"8 waves 64x64 ping-pong, no loads          ", "8 waves, 6 LDS-DMA pieces / K-tile, linear  ", "8 waves, 2 LDS-DMA pieces / K-tile, linear  ",
                            "8 waves, conv gather, pixel rows 2 KiB apart", "8 waves, conv gather, pixel rows 8 KiB apart", "4 waves 128x64 double-buffer, no loads      ",
                            "hw structure (transposed reads), no loads   ", "hw structure, 6 LDS-DMA pieces / K-tile      "};
    uint32_t srcmask = (1024u << 20) - 1;
    auto launch = [&](int which) {
        const uint32_t wsrc = 900u << 20;           // weights: a 2.4 MiB slice re-read by every workgroup
        switch (which) {
            case 0: hipLaunchKernelGGL(probe8<0>, dim3(grid), dim3(512), lds, 0, d, nkt, src, srcmask, 0u, wsrc); break;
            case 1: hipLaunchKernelGGL(probe8<1>, dim3(grid), dim3(512), lds, 0, d, nkt, src, (64u << 20) - 1, 0u, wsrc); break;
            case 2: hipLaunchKernelGGL(probe8<2>, dim3(grid), dim3(512), lds, 0, d, nkt, src, (64u << 20) - 1, 0u, wsrc); break;
            case 3: hipLaunchKernelGGL(probe8<4>, dim3(grid), dim3(512), lds, 0, d, nkt, src, srcmask, 2048u, wsrc); break;
            case 4: hipLaunchKernelGGL(probe8<4>, dim3(grid), dim3(512), lds, 0, d, nkt, src, srcmask, 8192u, wsrc); break;
            case 5: hipLaunchKernelGGL(probe4, dim3(grid), dim3(256), lds, 0, d, nkt); break;
            case 6: hipLaunchKernelGGL(probe_hw<0>, dim3(2304), dim3(512), lds, 0, d, nkt < 0 ? -320 : 320, src, (64u << 20) - 1); break;
            default: hipLaunchKernelGGL(probe_hw<1>, dim3(2304), dim3(512), lds, 0, d, nkt < 0 ? -320 : 320, src, (64u << 20) - 1); break;
        }
    };
    for (int rep = 0; rep < 4; ++rep)
        for (int which = 0; which < 8; ++which) {
            const bool constant = (rep & 1) == 0;
            src = constant ? srcc : srcr;
            nkt = constant ? -144 : 144;
            for (int w = 0; w < 2; ++w) launch(which);
            hipEventRecord(e0);
            const int n = 10;
            for (int w = 0; w < n; ++w) launch(which);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
            printf("[%s operands] %s: %.3f ms per launch (matrix pipe alone: %.3f ms at 2.4 GHz, %.3f at 2.0 GHz); %s\n", constant ? "constant" : "random  ", names[which], ms,
                   mfma_cycles_per_simd / 2.4e6, mfma_cycles_per_simd / 2.0e6, hipGetErrorString(hipGetLastError()));
        }
    // ---- short-K lifecycle: 5120 workgroups x 4 K-tiles (the encoder's M = 327680, K = 128, N = 512 pointwise conv), operands streamed
    //      from HBM (every workgroup its own 192 KB), with / without the 128 KB-per-workgroup epilogue (671 MB of output)
    float* big; hipMalloc(&big, (size_t)5120 * 256 * 128 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const char* pn[4] = {"short K (4 K-tiles): loads + epilogue   ", "short K: loads, no epilogue             ", "short K: no loads, epilogue             ", "short K: neither                        "};
    for (int rep = 0; rep < 2; ++rep)
        for (int which = 0; which < 4; ++which) {
            auto go = [&]() {
                switch (which) {
                    case 0: hipLaunchKernelGGL((probe8<1, true>), dim3(grid), dim3(512), lds, 0, big, 4, srcr, (1024u << 20) - 1, 0u, 0u); break;
                    case 1: hipLaunchKernelGGL((probe8<1, false>), dim3(grid), dim3(512), lds, 0, d, 4, srcr, (1024u << 20) - 1, 0u, 0u); break;
                    case 2: hipLaunchKernelGGL((probe8<0, true>), dim3(grid), dim3(512), lds, 0, big, 4, srcr, (1024u << 20) - 1, 0u, 0u); break;
                    default: hipLaunchKernelGGL((probe8<0, false>), dim3(grid), dim3(512), lds, 0, d, 4, srcr, (1024u << 20) - 1, 0u, 0u); break;
                }
            };
            go(); go();
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) go();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("%s: %.1f us per launch = %.2f us per workgroup round (20 rounds); %s\n", pn[which], ms * 1e3, ms * 1e3 / 20, hipGetErrorString(hipGetLastError()));
        }
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_persist<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds + 16384);
    // the operand footprint decides how much of the read stream the caches absorb (the real kernel re-reads each activation tile for
    // 4 output-channel tiles: 220 MB from HBM out of 983 MB requested, PMC) -- sweep it for both structures
    for (int fp = 0; fp < 3; ++fp) {
        const uint32_t mask = ((fp == 0 ? 1024u : fp == 1 ? 256u : 64u) << 20) - 1;
        for (int g = -2; g < 3; ++g) {
            auto go = [&]() {
                if (g == -2) hipLaunchKernelGGL((probe8<5, 2>), dim3(grid), dim3(512), lds, 0, big, 4, srcr, ((1024u << 20) - 1) & ~(uint32_t)fp, 512u, 512u << 20);
                else if (g == -1) hipLaunchKernelGGL((probe8<1, 2>), dim3(grid), dim3(512), lds, 0, big, 4, srcr, mask, 0u, 0u);
                else if (g == 0) hipLaunchKernelGGL((probe8<1, true>), dim3(grid), dim3(512), lds, 0, big, 4, srcr, mask, 0u, 0u);
                else hipLaunchKernelGGL(probe_persist<4>, dim3(g == 1 ? 256 : 512), dim3(512), lds + 16384, 0, big, 5120, srcr, mask);
            };
            go(); go();
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) go();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("short K, operand footprint %u MB, %s: %.1f us per launch; %s\n", (mask + 1) >> 20, g == -2 ? (fp == 0 ? "(168 MB) pointwise conv's loads: 512 B rows, 128 B per K-tile, tile shared by 4 workgroups on different XCDs" : fp == 1 ? "(168 MB) the same, the 4 workgroups on one XCD" : "(168 MB) the same, one XCD, K-tile-major activations") : g == -1 ? "one tile per workgroup, output rows of 512 floats shared by 4 tiles" : g == 0 ? "one tile per workgroup" : g == 1 ? "persistent 256" : "persistent 512", ms * 1e3, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
