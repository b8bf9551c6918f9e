// Both gradients of the ConvLSTM's rank-1 gate term from ONE pass over the gate gradient (round 6).
//
// The cell adds  sum_k spcol[b,p,k] * wc[b,n,k]  to the i / f / o gate pre-activations (conv3x3(W, spatial (x) semantic) of the reference,
// AiR/models/baseline_attention.py:40-50, as a 9-tap single-channel conv per stream with the per-sample contracted filter).  Its backward
// needs, per decode step, from the gate gradient dpre [B*P][4C] (first N3 = 3C channels):
//     dsp[b,p,k] = sum_n dpre[b,p,n] * wc[b,n,k]          (contraction over channels)
//     dwc[b,n,k] = sum_p dpre[b,p,n] * spcol[b,p,k]       (contraction over pixels)
// Round 3-5 ran them as two launches of the big GEMM kernels (a batched pointwise GEMM whose 128-column tile carried 20 columns, and a batched
// weight gradient) plus six small launches that padded / scaled / split their small operands: each launch read the 503 MB of dpre's split
// planes, 0.28 ms per decode step in the serial section between two h-gate data gradients.  Here a workgroup owns 160 pixels of one sample
// and streams their plane rows ONCE through LDS (32 pixels x 256 channels per tile, 1040-byte row pitch, two tiles by LDS-DMA); both
// products run on the matrix pipe (v_mfma_f32_16x16x32_f16, three products per fragment pair, smallest first, as the GEMMs they replace):
//   * dsp: A = the tile's pixel rows (ds_read_b128: the pitch puts the 16 rows of a lane group on 16 distinct 16-byte slots), B = the split
//     form of wc^T (rows k, one power-of-two scale per row) straight from L2 in the instruction's own lane layout, per channel chunk;
//   * dwc: A = the SAME tile read channel-major by ds_read_b64_tr_b16 (the hardware transpose: 4 pixels x 16 channels per 16-lane group; 2-way
//     bank conflicts at this pitch, 16 reads per wave and tile), B = the tile's taps, split into two fp16 planes in registers with one
//     power-of-two scale per WORKGROUP (max |tap| of its 160 pixels, found in the prologue): no padded / split copy of spcol exists.
//     (The first version multiplied on the vector pipe, one thread per channel: 5 GFLOP of fp32 FMAs = 120 us of the launch, LDS-broadcast
//     or DPP-broadcast taps alike.)  Partial sums of the 160 pixels go to a slab with their scale; a second tiny kernel adds the slabs of a
//     sample in chunk order (fixed order, no atomics).
// HBM-bound on the planes: 503 MB per launch.  Samples behind their last loss step (row_last) get zeros without their rows being read.
#include "common.h"

typedef _Float16 r1_f16x8 __attribute__((ext_vector_type(8)));
typedef short r1_s4 __attribute__((ext_vector_type(4)));
typedef short r1_s8 __attribute__((ext_vector_type(8)));

namespace {

#define R1_GLDS16(src, dst)                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                       \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define R1_GLDS4(src, dst)                                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                       \
                                     (__attribute__((address_space(3))) void*)(dst), 4, 0, 0)

constexpr int R1_PX = 32;                     // pixels per LDS tile
constexpr int R1_CH = 256;                    // channels per LDS tile: 64 per wave for dwc
constexpr int R1_NT = 5;                      // pixel tiles per workgroup: 160 pixels
constexpr int R1_PITCH = R1_CH * 4 + 16;      // 1040 bytes per pixel row: the 16 rows of a ds_read_b128 lane group fall on 16 distinct 16-byte slots
constexpr int R1_STAGE = R1_PX * R1_PITCH;    // 33280
constexpr int R1_SP = 4096;                   // the tile's taps [32 pixels][KP] fp32 (768 floats moved: 12 wave pieces), + 16 bytes of scratch at 3072
constexpr int R1_LDS = 2 * R1_STAGE + 2 * R1_SP;      // 74752: two workgroups per CU

struct R1Args {
    const unsigned char* Y;      // dpre planes [B*P][ldy/16][2][16] fp16
    const float* sy;             // its scale (device scalar)
    const unsigned char* Ws;     // wc^T planes [B*KP][N3/16][2][16] fp16
    const float* sw;             // one scale per row of Ws
    const float* spcol;          // [B][P][KP]
    float* dsp;                  // [B][P][KP]
    float* slab;                 // [B][chunks][N3][KP] raw partial sums (scaled by sy * the chunk's tap scale)
    float* slab_inv;             // [B][chunks] 1 / tap scale
    const int* row_last;
    int row_step;
    int B, P, N3, ldy, chunks;
};

__device__ __forceinline__ r1_f16x8 r1_tr_pair(const unsigned char* a0, const unsigned char* a1) {
    const r1_s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) r1_s4*)(a0));
    const r1_s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) r1_s4*)(a1));
    const r1_s8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(r1_f16x8, v);
}

template <int KP>
__global__ __launch_bounds__(256, 2) void rank1_grads_kernel(R1Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int px0 = chunk * (R1_NT * R1_PX);
    const int ntile = min(R1_NT, (p.P - px0) / R1_PX);          // P % 32 == 0 (launcher)
    const int nchunk = p.N3 / R1_CH;
    float* dsp_b = p.dsp + ((int64_t)b * p.P + px0) * KP;
    float* slab_b = p.slab + ((int64_t)b * p.chunks + chunk) * p.N3 * KP;
    if (p.row_last != nullptr && p.row_last[b] < p.row_step) {      // (block-uniform) dpre of this sample is exactly zero at this decode step
        for (int i = t; i < ntile * R1_PX * KP; i += 256) dsp_b[i] = 0.f;
        return;                                                     // (its slabs are not written: the reduce skips the sample too)
    }
    const float isy = 1.f / p.sy[0];
    const int64_t rowbytes = (int64_t)p.ldy * 4;
    const unsigned char* Yb = p.Y + ((int64_t)b * p.P + px0) * rowbytes;
    const float* sp_b = p.spcol + ((int64_t)b * p.P + px0) * KP;

    // ---- the workgroup's tap scale: 2^(14 - exponent of max |tap|) over its pixels, so the taps' high planes stay below 2^15 and the low
    // planes keep 11 more bits.  Ordinary loads: BEFORE the first LDS-DMA piece (inside the loop any use of an ordinary load makes the
    // compiler wait vmcnt(0) while pieces are in flight -- the next tile would be waited for before this one is consumed).
    float tap_scale;
    {
        float m = 0.f;
        for (int i = t; i < ntile * R1_PX * KP; i += 256) m = fmaxf(m, fabsf(sp_b[i]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float* scr = reinterpret_cast<float*>(smem + 2 * R1_STAGE + 3072);
        if (lane == 0) scr[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));
        const int e = (__builtin_bit_cast(int, m) >> 23) & 0xff;
        const int se = min(254, max(1, 268 - e));                   // 127 + 14 - (e - 127)
        tap_scale = __builtin_bit_cast(float, se << 23);
        if (t == 0) p.slab_inv[(int64_t)b * p.chunks + chunk] = 1.f / tap_scale;
    }

    // LDS-DMA: one wave instruction moves one pixel row of the tile (64 lanes x 16 B = 1 KB); wave w issues rows w, w + 4, ...
    // (addresses as wave-uniform base + one 32-bit lane offset: the scalar-base form of the instruction, one VGPR for all pieces)
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue = [&](int tile, int ch, int stage) {
        unsigned char* st = smem + stage * R1_STAGE;
        const unsigned char* src = Yb + ((int64_t)tile * R1_PX + wave) * rowbytes + (int64_t)ch * (R1_CH * 4);
        unsigned o = lane16;
        asm volatile("" : "+v"(o));          // (opaque: or the compiler keeps base + lane as a 64-bit VGPR pair per piece and per call site, and spills them)
#pragma unroll
        for (int j = 0; j < R1_PX / 4; ++j) R1_GLDS16(src + (int64_t)(4 * j) * rowbytes + o, st + (wave + 4 * j) * R1_PITCH);
    };
    // the tile's taps [32][KP] floats, as they lie in memory: 12 pieces of 64 floats (the lanes past 32 * KP fetch float 0; their slots are not read)
    unsigned tap_lane[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int d = (wave * 3 + i) * 64 + lane;
        tap_lane[i] = d < R1_PX * KP ? (unsigned)d * 4u : 0u;
    }
    auto issue_taps = [&](int tile, int stage) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(sp_b + (int64_t)tile * R1_PX * KP);
        unsigned char* dst = smem + 2 * R1_STAGE + stage * R1_SP + wave * 768;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            unsigned o = tap_lane[i];
            asm volatile("" : "+v"(o));
            R1_GLDS4(src + o, dst + i * 256);
        }
    };

    // dsp: wave (rt = wave >> 1, ct = wave & 1) owns the 16 x 16 output tile (pixels rt*16.., columns ct*16..) of every pixel tile
    const int rt = wave >> 1, ct = wave & 1;
    const bool ct_live = ct * 16 < KP;                               // (scalar) KP <= 16: the second column tile does not exist
    const int l16 = lane & 15, g4 = lane >> 4;
    const int krow = min(ct * 16 + l16, KP - 1);                     // rows >= KP: a valid row, their columns are never stored
    const int fragoff = (g4 >> 1) * 64 + (g4 & 1) * 16;              // + plane * 32 + kstep * 128: the lane's 8 k of a 32-k block
    const unsigned char* Wb = p.Ws + (int64_t)b * KP * ((int64_t)p.N3 * 4);              // (uniform) the sample's wc^T rows
    const unsigned wlane = (unsigned)krow * (unsigned)p.N3 * 4u + (unsigned)fragoff;       // KP * N3 * 4 < 2^31 (launcher)
    f32x4 dacc[R1_NT];
#pragma unroll
    for (int i = 0; i < R1_NT; ++i) dacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // dwc: wave w owns channels 64w .. 64w+63 of the chunk (four 16-channel groups) x all KP taps (two column tiles), over the workgroup's pixels.
    // Transposed read: lane 4q + p of a 16-lane group addresses pixel row q (of the group's four), channels 4p .. 4p+3 of the 16-channel plane;
    // group g4 takes pixels 8*g4 .. 8*g4+3 and 8*g4+4 .. +7 = the 8 k of its MFMA operand.
    const int tr_off = (8 * g4 + (l16 >> 2)) * R1_PITCH + wave * 256 + (l16 & 3) * 8;         // + mf * 64 + plane * 32 (+ 4 rows)
    // the lane's taps of a tile: pixels 8*g4 + i, tap column min(nf * 16 + l16, KP - 1)
    const int tap_off0 = (8 * g4 * KP + l16) * 4, tap_off1 = (8 * g4 * KP + min(16 + l16, KP - 1)) * 4;
    constexpr int NF = KP > 16 ? 2 : 1;

    const int total = nchunk * ntile;                                // tiles in (channel chunk, pixel tile) order
    // a chunk's 8 k-steps of the wave's wc^T fragments: straight from L2 into registers, reused by the chunk's pixel tiles.  Loaded inside asm
    // statements (the compiler does not count them, see above) and waited for by the counted wait of the chunk's first tile, which names them
    r1_f16x8 bfr[R1_CH / 32][2];
    auto load_bfr = [&](int ch) {
        const unsigned char* base = Wb + (int64_t)ch * (R1_CH * 4);
#pragma unroll
        for (int ks = 0; ks < R1_CH / 32; ++ks) {
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(bfr[ks][0]) : "v"(wlane), "s"(base), "i"(ks * 128) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(bfr[ks][1]) : "v"(wlane), "s"(base), "i"(ks * 128 + 32) : "memory");
        }
    };
    if (total > 0) {
        if (ct_live) load_bfr(0);
        issue_taps(0, 0);
        issue(0, 0, 0);
    }
    int it = 0;
    for (int ch = 0; ch < nchunk; ++ch) {
        f32x4 wacc[4][NF];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) wacc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tile = 0; tile < R1_NT; ++tile) {
            if (tile >= ntile) break;                                // (scalar)
            const int stage = it & 1;
            const bool more = it + 1 < total;
            if (more) {
                const bool wrap = tile + 1 >= ntile;
                issue_taps(wrap ? 0 : tile + 1, stage ^ 1);
                issue(wrap ? 0 : tile + 1, wrap ? ch + 1 : ch, stage ^ 1);
            }
            // tile `it` (and, on a chunk's first tile, the chunk's fragments, requested before it) has landed; the 11 pieces of `it + 1` may be in flight
            if (tile == 0) {
                if (more)
                    asm volatile("s_waitcnt vmcnt(11)" : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[2][0]), "+v"(bfr[2][1]),
                                 "+v"(bfr[3][0]), "+v"(bfr[3][1]), "+v"(bfr[4][0]), "+v"(bfr[4][1]), "+v"(bfr[5][0]), "+v"(bfr[5][1]), "+v"(bfr[6][0]),
                                 "+v"(bfr[6][1]), "+v"(bfr[7][0]), "+v"(bfr[7][1]) : : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[2][0]), "+v"(bfr[2][1]),
                                 "+v"(bfr[3][0]), "+v"(bfr[3][1]), "+v"(bfr[4][0]), "+v"(bfr[4][1]), "+v"(bfr[5][0]), "+v"(bfr[5][1]), "+v"(bfr[6][0]),
                                 "+v"(bfr[6][1]), "+v"(bfr[7][0]), "+v"(bfr[7][1]) : : "memory");
            } else if (more) {
                asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            const unsigned char* st = smem + stage * R1_STAGE;
            // ---- dsp: 8 k-steps of 32 channels, three products each ----
            if (ct_live) {
                const unsigned char* arow = st + (rt * 16 + l16) * R1_PITCH + fragoff;
                f32x4 acc = dacc[tile];
#pragma unroll
                for (int ks = 0; ks < R1_CH / 32; ++ks) {
                    const r1_f16x8 a0 = *reinterpret_cast<const r1_f16x8*>(arow + ks * 128), a1 = *reinterpret_cast<const r1_f16x8*>(arow + ks * 128 + 32);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bfr[ks][1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bfr[ks][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bfr[ks][0], acc, 0, 0, 0);
                    if (ks & 1) __builtin_amdgcn_sched_barrier(0);      // (two k-steps of fragment reads in flight, not all eight: registers)
                }
                dacc[tile] = acc;
                // the NEXT chunk's fragments, behind this chunk's last use of the registers: their latency hides under the work below
                if (tile + 1 >= ntile && ch + 1 < nchunk) load_bfr(ch + 1);
            }
            // ---- dwc: the wave's 64 channels x the tile's 32 pixels (one k-block) x KP taps ----
            __builtin_amdgcn_sched_barrier(0);
            {
                const unsigned char* tp = smem + 2 * R1_STAGE + stage * R1_SP;
                r1_f16x8 th[NF], tl[NF];
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) {
                    const unsigned char* q = tp + (nf == 0 ? tap_off0 : tap_off1);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float x = *reinterpret_cast<const float*>(q + i * KP * 4) * tap_scale;
                        const _Float16 h = (_Float16)x;
                        th[nf][i] = h;
                        tl[nf][i] = (_Float16)(x - (float)h);
                    }
                }
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    const unsigned char* a = st + tr_off + mf * 64;
                    const r1_f16x8 ah = r1_tr_pair(a, a + 4 * R1_PITCH), al = r1_tr_pair(a + 32, a + 32 + 4 * R1_PITCH);
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf) {
                        f32x4 acc = wacc[mf][nf];
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, th[nf], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, tl[nf], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, th[nf], acc, 0, 0, 0);
                        wacc[mf][nf] = acc;
                    }
                    if (mf & 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_s_barrier();                            // everybody is done with this stage before it is refilled (tile it + 2)
            ++it;
        }
        // this channel chunk is done for the workgroup's pixels.  C/D layout of 16x16: column (tap) = lane & 15, row (channel) = 4 * (lane >> 4) + reg
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const int k = nf * 16 + l16;
                if (k < KP) {
                    float* o = slab_b + ((int64_t)ch * R1_CH + wave * 64 + mf * 16 + 4 * g4) * KP + k;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r * KP] = wacc[mf][nf][r];
                }
            }
    }
    // ---- dsp out: column = lane & 15, row = 4 * (lane >> 4) + reg ----
    if (ct_live) {
        const int k = ct * 16 + l16;
        if (k < KP) {
            const float isw = 1.f / p.sw[(int64_t)b * KP + k];
#pragma unroll
            for (int i = 0; i < R1_NT; ++i)
                if (i < ntile) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dsp_b[(int64_t)(i * R1_PX + rt * 16 + 4 * g4 + r) * KP + k] = (dacc[i][r] * isy) * isw;
                }
        }
    }
}

// dwc[b][n][k] = (sum over the sample's pixel chunks, in chunk order, each divided by its tap scale) / sy; zeros for samples behind their last loss step
__global__ __launch_bounds__(256) void rank1_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ slab_inv, const float* __restrict__ sy,
                                                           int B, int chunks, int64_t per, float* __restrict__ dwc, const int* __restrict__ row_last,
                                                           int row_step) {
    const float isy = 1.f / sy[0];
    const int64_t total = (int64_t)B * per;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / per);
        const int64_t r = i - (int64_t)b * per;
        float s = 0.f;
        if (!(row_last != nullptr && row_last[b] < row_step))
            for (int c = 0; c < chunks; ++c) s += slab[((int64_t)b * chunks + c) * per + r] * slab_inv[b * chunks + c];
        dwc[i] = s * isy;
    }
}

template <int KP>
int launch_rank1(const R1Args& a, hipStream_t s) {
    auto kern = rank1_grads_kernel<KP>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R1_LDS);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)a.chunks, (unsigned)a.B), dim3(256), R1_LDS, s, a);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

int64_t r1_slab_bytes(int B, int P, int N3, int KP) {
    const int chunks = (int)sp_cdiv(P, R1_NT * R1_PX);
    return (int64_t)B * chunks * N3 * KP * (int64_t)sizeof(float);
}

}  // namespace

extern "C" int sp_rank1_grads_applies(int B, int P, int N3, int KP, int ldy) {
    return (KP == 20 || KP == 12) && B >= 1 && B <= 65535 && P >= R1_PX && P % R1_PX == 0 && N3 >= R1_CH && N3 % R1_CH == 0 && ldy >= N3 && ldy % 16 == 0 &&
           4LL * KP * N3 < (1LL << 31) && 4LL * B * P * ldy + 64 < (1LL << 40);
}
extern "C" int64_t sp_rank1_grads_workspace(int B, int P, int N3, int KP) {
    const int chunks = (int)sp_cdiv(P, R1_NT * R1_PX);
    return r1_slab_bytes(B, P, N3, KP) + (int64_t)B * chunks * (int64_t)sizeof(float);
}
extern "C" int sp_rank1_grads_f16x2(const void* dpre_planes, const float* dpre_scale, int ldy, const void* wcT_planes, const float* wcT_row_scale,
                                    const float* spcol, int B, int P, int N3, int KP, float* dsp, float* dwc, void* workspace, const int* row_last,
                                    int row_step, void* stream) {
    if (!dpre_planes || !dpre_scale || !wcT_planes || !wcT_row_scale || !spcol || !dsp || !dwc || !workspace) return SP_ENULL;
    if (!sp_rank1_grads_applies(B, P, N3, KP, ldy)) return SP_EINVAL;
    if (((uintptr_t)dpre_planes | (uintptr_t)wcT_planes | (uintptr_t)workspace) & 15) return SP_EINVAL;
    R1Args a{};
    a.Y = (const unsigned char*)dpre_planes; a.sy = dpre_scale; a.Ws = (const unsigned char*)wcT_planes; a.sw = wcT_row_scale;
    a.spcol = spcol; a.dsp = dsp; a.slab = (float*)workspace;
    a.slab_inv = (float*)((unsigned char*)workspace + r1_slab_bytes(B, P, N3, KP));
    a.row_last = row_last; a.row_step = row_step;
    a.B = B; a.P = P; a.N3 = N3; a.ldy = ldy; a.chunks = (int)sp_cdiv(P, R1_NT * R1_PX);
    hipStream_t s = (hipStream_t)stream;
    const int rc = KP == 20 ? launch_rank1<20>(a, s) : launch_rank1<12>(a, s);
    if (rc != SP_OK) return rc;
    const int64_t per = (int64_t)N3 * KP;
    const int64_t nb = sp_cdiv((int64_t)B * per, 256);
    hipLaunchKernelGGL(rank1_reduce_kernel, dim3((unsigned)(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb))), dim3(256), 0, s, (const float*)workspace,
                       (const float*)a.slab_inv, dpre_scale, B, a.chunks, per, dwc, row_last, row_step);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
