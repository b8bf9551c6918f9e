// How fast can 256 CUs write a [M][N] fp32 matrix in the patterns a GEMM epilogue produces?  (round 6 probe; stand-alone:
//   hipcc --offload-arch=gfx950 -O3 -o store_pattern_probe.bin store_pattern_probe.hip && ./store_pattern_probe.bin)
// M = 81920, N = 1024 (335 MB), tiles of 256 x 128 per workgroup of 512 threads unless said otherwise.  Variants:
//   0  the shipped epilogue's pattern: wave (wm, wn) owns 64 x 64, one instruction = 4 rows x 256 B, 16 instructions per lane
//   1  as 0 with the row pitch padded by 128 B
//   2  the workgroup writes whole 512-byte rows: one instruction per wave = 2 rows x 512 B
//   3  128 x 128 tiles, 256 threads (two workgroups per CU fit), pattern of 0
//   4  each workgroup writes one contiguous 128 KB piece (a fill)
//   5  as 0 with non-temporal stores
//   6  as 0, but every workgroup first spins ~6 us (a K loop's worth of time without memory traffic)
//   7  as 4, with the spin of 6
//   8  as 0, grid = 256 persistent workgroups looping over their tiles (no spin)
//   9  as 8 with the spin before every tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void spin_us(int us) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(8);
}

template <int V>
__global__ __launch_bounds__(512) void probe(float* __restrict__ C, int M, int N, int ldc, int tiles_n, int ntiles, int spin) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float4 v = make_float4((float)t, 1.f, 2.f, 3.f);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (spin) spin_us(spin);
        if (V == 4 || V == 7) {
            float4* dst = reinterpret_cast<float4*>(C + (int64_t)tile * 256 * 128);
#pragma unroll
            for (int i = 0; i < 16; ++i) dst[i * 512 + t] = v;
            continue;
        }
        const int tn = tile % tiles_n, tm = tile / tiles_n;
        const int64_t m0 = (int64_t)tm * 256;
        const int n0 = tn * 128;
        if (V == 2) {
            const int r0 = t >> 5, c = (t & 31) * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(C + (m0 + i * 16 + r0) * ldc + n0 + c) = v;
        } else {
            const int wm = wave >> 1, wn = wave & 1;
            const int cq4 = lane & 15, rsub = lane >> 4;
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                float4* dst = reinterpret_cast<float4*>(C + (m0 + wm * 64 + ps * 4 + rsub) * ldc + n0 + wn * 64 + 4 * cq4);
                typedef float f4 __attribute__((ext_vector_type(4)));
                if (V == 5) __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(dst));
                else *dst = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void probe128(float* __restrict__ C, int M, int N, int ldc, int tiles_n) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float4 v = make_float4((float)t, 1.f, 2.f, 3.f);
    const int tile = blockIdx.x;
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int64_t m0 = (int64_t)tm * 128;
    const int n0 = tn * 128;
    const int wm = wave >> 1, wn = wave & 1;
    const int cq4 = lane & 15, rsub = lane >> 4;
#pragma unroll
    for (int ps = 0; ps < 16; ++ps) *reinterpret_cast<float4*>(C + (m0 + wm * 64 + ps * 4 + rsub) * ldc + n0 + wn * 64 + 4 * cq4) = v;
}

int main() {
    const int M = 81920, N = 1024;
    float* C;
    CK(hipMalloc(&C, (size_t)M * (N + 64) * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int tiles_n = N / 128, ntiles = (M / 256) * tiles_n;
    for (int v = 0; v <= 9; ++v) {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            const int ldc = v == 1 ? N + 32 : N;
            switch (v) {
                case 0: hipLaunchKernelGGL(probe<0>, dim3(ntiles), dim3(512), 0, 0, C, M, N, ldc, tiles_n, ntiles, 0); break;
                case 1: hipLaunchKernelGGL(probe<0>, dim3(ntiles), dim3(512), 0, 0, C, M, N, ldc, tiles_n, ntiles, 0); break;
                case 2: hipLaunchKernelGGL(probe<2>, dim3(ntiles), dim3(512), 0, 0, C, M, N, ldc, tiles_n, ntiles, 0); break;
                case 3: hipLaunchKernelGGL(probe128, dim3(ntiles * 2), dim3(256), 0, 0, C, M, N, ldc, tiles_n); break;
                case 4: hipLaunchKernelGGL(probe<4>, dim3(ntiles), dim3(512), 0, 0, C, M, N, ldc, tiles_n, ntiles, 0); break;
                case 5: hipLaunchKernelGGL(probe<5>, dim3(ntiles), dim3(512), 0, 0, C, M, N, ldc, tiles_n, ntiles, 0); break;
                case 6: hipLaunchKernelGGL(probe<0>, dim3(ntiles), dim3(512), 0, 0, C, M, N, ldc, tiles_n, ntiles, 6); break;
                case 7: hipLaunchKernelGGL(probe<4>, dim3(ntiles), dim3(512), 0, 0, C, M, N, ldc, tiles_n, ntiles, 6); break;
                case 8: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, C, M, N, ldc, tiles_n, ntiles, 0); break;
                case 9: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, C, M, N, ldc, tiles_n, ntiles, 6); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("variant %d: %7.1f us  %5.2f TB/s%s\n", v, best * 1e3f, (double)M * N * 4 / (best * 1e-3) / 1e12,
               (v == 6 || v == 7 || v == 9) ? "   (includes 10 rounds x 6 us of spinning = 60 us)" : "");
    }
    return 0;
}
