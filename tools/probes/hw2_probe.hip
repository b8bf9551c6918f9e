// Probe (round 4): main loop of a weight-gradient kernel with a LARGER wave tile, before building it.
//   hw   today: block 256 co x 128 n x 32 pixels, 8 waves (64x64 wave tiles), two-level accumulators, 3-stage ring of 48 KB,
//        32 ds_read_b64_tr_b16 + 48 MFMA per wave and K-tile, 6 LDS-DMA pieces per lane and K-tile
//   hw2  block 256 co x 256 n x 32 pixels, 8 waves (64 co x 128 n wave tiles), SINGLE-level accumulators (128 VGPRs), 2-stage ring of
//        64 KB, 48 ds_read_b64_tr_b16 + 96 MFMA per wave and K-tile (-25 % LDS read bytes per MFMA), 8 LDS-DMA pieces per lane and
//        K-tile (-33 % L2->LDS bytes per MFMA).  Ping-pong halves, one barrier per K-tile; the late half issues its loads FIRST (a
//        2-stage ring has one K-tile in flight: the loads need the whole matrix segment to land).
// Grids: hw 2304 x 320 K-tiles = ONE launch of the h-gate shape (M = 81920 pixels, 2048 x 4608 outputs, 8 splits);
//        hw2 (144 tiles x 64 slabs) x 640 K-tiles = SIXTEEN steps in one deferred launch (4 splits per step): time / 16 compares.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/hw2_probe.hip -o tools/probes/hw2_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
#define GLDS16(src, dst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src), (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

__device__ __forceinline__ void fill_lds(unsigned char* smem, int bytes, int nthreads, bool constant) {
    for (int i = threadIdx.x; i < bytes / 4; i += nthreads) {
        uint32_t h = (uint32_t)i * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        reinterpret_cast<uint32_t*>(smem)[i] = constant ? 0x3c003c00u + (i & 0xff) : ((h & 0x8fff8fffu) | 0x30003000u);
    }
    __syncthreads();
}
__device__ __forceinline__ f16x8 tr_pair(const unsigned char* base, int off0, int off1) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + off0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + off1));
    typedef short short8v __attribute__((ext_vector_type(8)));
    short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ int rot4p(int q) { return 2 * (q & 1) + 8 * (q >> 1); }
__device__ __forceinline__ int swz16p(int r) { return rot4p(r & 3) + 4 * ((r >> 3) & 1); }

// ---- today's structure (reference arm, = wave_tile_probe.hip probe_hw) ----
constexpr int HA = 256 * 128, HB = 128 * 128, STAGE = HA + HB;
template <int MODE>
__global__ __launch_bounds__(512, 2) void probe_hw(float* out, int nkt, const unsigned char* src, uint32_t srcmask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool constant = nkt < 0;
    nkt = constant ? -nkt : nkt;
    fill_lds(smem, 3 * STAGE, 512, constant);
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

// ---- hw2: 256 x 256 block, 64 x 128 wave tiles, single-level accumulation, 2-stage ring ----
constexpr int S2 = 2 * 32 * 1024;     // one stage: A [32 pixels][1024 B] + B [32 pixels][1024 B]
// MODE 0: no loads; 1: 8 LDS-DMA pieces per lane and K-tile, linear source.  EPI 1: 256 KB per workgroup written as float4 through LDS.
// SPLITB 1: the B fragments are read in two halves of 64 columns, the second half under the first half's MFMAs (fewer live registers)
template <int MODE, int EPI, int SPLITB>
__global__ __launch_bounds__(512, 2) void probe_hw2(float* out, int nkt, const unsigned char* src, uint32_t srcmask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool constant = nkt < 0;
    nkt = constant ? -nkt : nkt;
    fill_lds(smem, 2 * S2, 512, constant);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int q = (lane >> 2) & 3, pp = lane & 3, kg = lane >> 4;
    // base offsets: tiles are reached through immediate offsets (A: i * 256; B: (i >> 1) * 256 with (i & 1) in the swizzled position)
    int offA[2][2], offB[2][2][2];      // [pl][s2], [i&1][pl][s2]
    for (int pl = 0; pl < 2; ++pl) for (int s2 = 0; s2 < 2; ++s2) {
        const int row = 8 * kg + q;
        const int pa = (((wm << 2) | (pl << 1) | (pp >> 1)) ^ swz16p(row));
        offA[pl][s2] = row * 1024 + pa * 16 + (pp & 1) * 8 + s2 * 4 * 1024;
        for (int i1 = 0; i1 < 2; ++i1) {
            const int pb = (((i1 << 3) | (wn << 2) | (pl << 1) | (pp >> 1)) ^ swz16p(row));
            offB[i1][pl][s2] = 32 * 1024 + row * 1024 + pb * 16 + (pp & 1) * 8 + s2 * 4 * 1024;
        }
    }
    f32x4 acc[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    f16x8 af[4][2], bf[8][2];
    auto rdA = [&](int stage) {
        const unsigned char* st = smem + stage * S2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = tr_pair(st + i * 256, offA[pl][0], offA[pl][1]);
    };
    auto rdB = [&](int stage, int j0, int j1) {
        const unsigned char* st = smem + stage * S2;
#pragma unroll
        for (int j = j0; j < j1; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bf[j][pl] = tr_pair(st + (j >> 1) * 256, offB[j & 1][pl][0], offB[j & 1][pl][1]);
    };
    auto mm = [&](int j0, int j1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = j0; j < j1; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
            }
    };
    const uint32_t lane_off = (uint32_t)(threadIdx.x * 16);
    uint32_t gpos = (uint32_t)blockIdx.x * 1572864u;
    auto issue = [&](int stage_) {
        if constexpr (MODE == 1) {
            unsigned char* st = smem + stage_ * S2;
#pragma unroll
            for (int j = 0; j < 8; ++j) GLDS16(src + ((gpos + j * 8192u + lane_off) & srcmask), st + (wave + 8 * j) * 1024);
            gpos += 8 * 8192u;
        }
    };
    auto wait_all = [&]() {
        if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    const bool late = wave >= 4;
    if constexpr (MODE == 1) { issue(0); wait_all(); }
    if (!late) {
        for (int kt = 0; kt < nkt; ++kt) {
            const int st = kt & 1;
            rdA(st);
            if constexpr (SPLITB) {
                rdB(st, 0, 4);
                if (kt + 1 < nkt) issue(st ^ 1);
                rdB(st, 4, 8);
                mm(0, 4);
                mm(4, 8);
            } else {
                rdB(st, 0, 8);
                if (kt + 1 < nkt) issue(st ^ 1);
                mm(0, 8);
            }
            wait_all();
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            const int st = kt & 1;
            if (kt + 1 < nkt) issue(st ^ 1);
            if (kt > 0) mm(0, 8);
            rdA(st);
            rdB(st, 0, 8);
            wait_all();
        }
        mm(0, 8);
    }
    if constexpr (EPI) {
        // the wave's 64 x 128 tile through a private 34 KB slice of the idle ring (row pitch 132 floats), float4 stores, 512 B per row
        float* stg = reinterpret_cast<float*>(smem) + wave * (64 * 132);       // 8 x 33792 B = 270 KB > ring: two passes of 32 rows
        const int l16 = lane & 15;
        float* dst0 = out + (int64_t)blockIdx.x * (256 * 256) + (wm * 64) * 256 + wn * 128;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float* sg = reinterpret_cast<float*>(smem) + wave * (32 * 132);
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sg[(i2 * 16 + 4 * kg + r) * 132 + j * 16 + l16] = acc[half * 2 + i2][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int c4 = lane & 31, rsub = lane >> 5;
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                const int row = ps * 2 + rsub;
                *reinterpret_cast<float4*>(dst0 + (half * 32 + row) * 256 + 4 * c4) = *reinterpret_cast<const float4*>(sg + row * 132 + 4 * c4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        (void)stg;
        return;
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
    float* d; hipMalloc(&d, 4096);
    const uint32_t maxbytes = 64u << 20;
    unsigned char *srcr, *srcc;
    hipMalloc(&srcr, (size_t)maxbytes + (1 << 20)); hipMalloc(&srcc, (size_t)maxbytes + (1 << 20));
    { std::vector<uint32_t> hbuf((size_t)(maxbytes >> 2) + (1 << 18)); uint32_t x = 12345u; for (auto& v : hbuf) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = (x & 0x8fff8fffu) | 0x30003000u; } hipMemcpy(srcr, hbuf.data(), hbuf.size() * 4, hipMemcpyHostToDevice); }
    hipMemset(srcc, 0x3c, (size_t)maxbytes + (1 << 20));
    float* big; hipMalloc(&big, (size_t)9216 * 256 * 256 * 4);
    const int lds1 = 3 * STAGE, lds2 = 2 * S2;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds1);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw2<0, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw2<1, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw2<1, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw2<0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw2<1, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_hw2<1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[9] = {"hw  (today) 2304 x 320 K-tiles, no loads, one step              ",
                            "hw  (today) 6 pieces / K-tile, one step                          ",
                            "hw2 9216 x 640 K-tiles no loads, per step (/16)                  ",
                            "hw2 8 pieces / K-tile, per step (/16)                            ",
                            "hw2 8 pieces + 256 KB epilogue per workgroup, per step (/16)     ",
                            "hw2 B read in halves, no loads, per step (/16)                   ",
                            "hw2 B read in halves, 8 pieces, per step (/16)                   ",
                            "hw2 B read in halves, 8 pieces + epilogue, per step (/16)        ",
                            "hw2 8 pieces + epilogue, 576 x 640 K-tiles: ONE step alone       "};
    for (int rep = 0; rep < 4; ++rep)
        for (int which = 0; which < 9; ++which) {
            const bool constant = (rep & 1) == 0;
            const unsigned char* src = constant ? srcc : srcr;
            const int sg = constant ? -1 : 1;
            auto launch = [&]() {
                switch (which) {
                    case 0: hipLaunchKernelGGL(probe_hw<0>, dim3(2304), dim3(512), lds1, 0, d, sg * 320, src, maxbytes - 1); break;
                    case 1: hipLaunchKernelGGL(probe_hw<1>, dim3(2304), dim3(512), lds1, 0, d, sg * 320, src, maxbytes - 1); break;
                    case 2: hipLaunchKernelGGL((probe_hw2<0, 0, 0>), dim3(9216), dim3(512), lds2, 0, d, sg * 640, src, maxbytes - 1); break;
                    case 3: hipLaunchKernelGGL((probe_hw2<1, 0, 0>), dim3(9216), dim3(512), lds2, 0, d, sg * 640, src, maxbytes - 1); break;
                    case 4: hipLaunchKernelGGL((probe_hw2<1, 1, 0>), dim3(9216), dim3(512), lds2, 0, big, sg * 640, src, maxbytes - 1); break;
                    case 5: hipLaunchKernelGGL((probe_hw2<0, 0, 1>), dim3(9216), dim3(512), lds2, 0, d, sg * 640, src, maxbytes - 1); break;
                    case 6: hipLaunchKernelGGL((probe_hw2<1, 0, 1>), dim3(9216), dim3(512), lds2, 0, d, sg * 640, src, maxbytes - 1); break;
                    case 7: hipLaunchKernelGGL((probe_hw2<1, 1, 1>), dim3(9216), dim3(512), lds2, 0, big, sg * 640, src, maxbytes - 1); break;
                    default: hipLaunchKernelGGL((probe_hw2<1, 1, 0>), dim3(576), dim3(512), lds2, 0, big, sg * 640, src, maxbytes - 1); break;
                }
            };
            launch();
            hipEventRecord(e0);
            const int n = which >= 2 && which <= 7 ? 2 : 8;
            for (int w = 0; w < n; ++w) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
            const float per_step = which >= 2 && which <= 7 ? ms / 16.f : ms;
            printf("[%s operands] %s: %.3f ms (launch %.3f ms); %s\n", constant ? "constant" : "random  ", names[which], per_step, ms,
                   hipGetErrorString(hipGetLastError()));
            fflush(stdout);
        }
    return 0;
}
