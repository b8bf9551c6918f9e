// Probe (round 5): which small kernels start BESIDE a resident workgroup of the h-gate conv's data gradient, and which wait for one of its
// 0.6-ms tiles to end?  The two-stream backward (functional._GateConvLstm.backward) runs ~45 small launches on the main stream while that
// GEMM holds every CU with one 512-thread workgroup: 2 waves per SIMD x 216 VGPRs, 151 552 B of LDS.  By the arithmetic a 4-wave
// workgroup with <= 80 VGPRs and <= 12 KB of LDS fits into what is left; the trace (profiles/r05_async_dgrad_window.log) says some do and
// some do not.  Here: a stand-in "resident" kernel (same threads / registers / LDS, spins for 3 ms, one workgroup per CU x 2 rounds) on
// stream A, then -- 200 us later -- a small kernel of V registers and L bytes of LDS on stream B; printed: the small kernel's start-to-end
// time.  Tens of microseconds = co-resident, ~ms = it waited for the residents.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/coresidency_probe.hip -o tools/probes/coresidency_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int TOPREG>
__device__ __forceinline__ void touch() {
    if constexpr (TOPREG == 215) asm volatile("v_mov_b32 v215, 0" ::: "v215");
    if constexpr (TOPREG == 255) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    if constexpr (TOPREG == 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if constexpr (TOPREG == 95) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if constexpr (TOPREG == 87) asm volatile("v_mov_b32 v87, 0" ::: "v87");
    if constexpr (TOPREG == 79) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if constexpr (TOPREG == 71) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if constexpr (TOPREG == 63) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if constexpr (TOPREG == 55) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    if constexpr (TOPREG == 47) asm volatile("v_mov_b32 v47, 0" ::: "v47");
    if constexpr (TOPREG == 39) asm volatile("v_mov_b32 v39, 0" ::: "v39");
    if constexpr (TOPREG == 31) asm volatile("v_mov_b32 v31, 0" ::: "v31");
}

template <int TOPREG>
__global__ __launch_bounds__(512) void resident(long long ticks, int* sink) {
    extern __shared__ int lds[];
    touch<TOPREG>();
    lds[threadIdx.x] = threadIdx.x;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (lds[(threadIdx.x + 1) & 511] == -1) sink[0] = 1;
}

template <int TOPREG>
__global__ __launch_bounds__(256) void small(int* sink, int iters) {
    extern __shared__ int lds[];
    touch<TOPREG>();
    int v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1664525 + 1013904223;
    if (v == 12345) sink[1] = v;
    if (v == 54321) sink[2] = lds[0];
}

template <int RTOP, int STOP>
static int run(const char* name, int rlds, int slds, int swgs, int* sink, hipStream_t a, hipStream_t b, int ncu) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(resident<RTOP>), hipFuncAttributeMaxDynamicSharedMemorySize, rlds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, worst = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        const long long ticks = 300000;       // 3 ms at 100 MHz
        hipLaunchKernelGGL(resident<RTOP>, dim3(2 * ncu), dim3(512), rlds, a, ticks, sink);
        // wait ~200 us on the host so that the residents hold every CU
        const long long h0 = clock();
        while ((clock() - h0) * 1000000 / CLOCKS_PER_SEC < 300) {}
        CK(hipEventRecord(e0, b));
        hipLaunchKernelGGL(small<STOP>, dim3(swgs), dim3(256), slds, b, sink, 2000);
        CK(hipEventRecord(e1, b));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; worst = ms > worst ? ms : worst;
        CK(hipDeviceSynchronize());
    }
    printf("%-64s small kernel: %8.3f .. %8.3f ms\n", name, best, worst);
    return 0;
}

int main() {
    int* sink; CK(hipMalloc(&sink, 64));
    hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("%d CUs; resident = 512 threads, 216 VGPRs, 151552 B LDS, 3 ms, 2 rounds; small kernel = 256 threads x 432 workgroups\n", ncu);
    const int R = 151552;
    if (run<215, 31>("resident 216 regs | small 32 regs, 0 LDS", R, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 39>("resident 216 regs | small 40 regs, 0 LDS", R, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 47>("resident 216 regs | small 48 regs, 0 LDS", R, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 55>("resident 216 regs | small 56 regs, 0 LDS", R, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 63>("resident 216 regs | small 64 regs, 0 LDS", R, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 71>("resident 216 regs | small 72 regs, 0 LDS", R, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 79>("resident 216 regs | small 80 regs, 0 LDS", R, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 87>("resident 216 regs | small 88 regs, 0 LDS", R, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 4 KB LDS", R, 4096, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 8 KB LDS", R, 8192, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 8.5 KB LDS", R, 8704, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 9 KB LDS", R, 9216, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 10 KB LDS", R, 10240, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 10.5 KB LDS", R, 10752, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 11 KB LDS", R, 11264, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 11.5 KB LDS", R, 11776, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 12 KB LDS", R, 12288, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("resident 216 regs | small 32 regs, 12.5 KB LDS", R, 12800, 432, sink, a, b, ncu)) return 1;
    if (run<215, 71>("resident 216 regs | small 72 regs, 4 KB LDS", R, 4096, 432, sink, a, b, ncu)) return 1;
    if (run<255, 31>("resident 256 regs | small 32 regs, 0 LDS", R, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 31>("no LDS in the resident, 216 regs | small 32 regs, 0 LDS", 2048, 0, 432, sink, a, b, ncu)) return 1;
    if (run<215, 127>("no LDS in the resident, 216 regs | small 128 regs, 16 KB LDS", 2048, 16384, 432, sink, a, b, ncu)) return 1;
    return 0;
}
