// Probe of ds_read_b64_tr_b16 lane semantics (gfx950).  LDS holds a [32 rows][64 cols] u16 matrix with value row*256+col.
// Each lane supplies an address; we print what every lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int rowstride_elems) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[32 * 80];
    for (int i = threadIdx.x; i < 32 * 80; i += 64) lds[i] = 0xffff;
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 64; i += 64) { int r = i / 64, c = i % 64; lds[r * rowstride_elems + c] = (unsigned short)(r * 256 + c); }
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    // group g: block rows q (0..3) + 4*(g>>1), columns 16*(g&1) + 4p .. +3
    const unsigned short* addr = lds + (q + 4 * (g >> 1)) * rowstride_elems + 16 * (g & 1) + 4 * p;
    short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)addr);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 80);
    unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l*4+e] >> 8, h[l*4+e] & 255); printf("\n"); }
    return 0;
}
