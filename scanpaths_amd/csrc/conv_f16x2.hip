// fp32-faithful implicit-GEMM convolution on the fp16 matrix pipe: "2 x fp16 split, 3 products".
//
// Every fp32 operand tensor is scaled by a per-tensor power of two s (so that max|x*s| lies in [8192, 16384): exact, no
// overflow, the residual plane stays a normal fp16 for everything within 2^-11 of the maximum) and split into two fp16 planes
//     x*s = x1 + x2,   x1 = fp16(x*s),  x2 = fp16(x*s - x1)          |x*s - x1 - x2| <= 2^-22 |x*s|
// and  a*b = (a1b1 + a1b2 + a2b1) / (sa*sb) + O(2^-22 |ab|)  is evaluated with THREE v_mfma_f32_16x16x32_f16 per fragment
// pair (fp16 x fp16 products are exact in the fp32 accumulator; smallest first; two-level accumulation every 256 k as in the
// other GEMM kernels).  The element-wise representation errors are independent, so a length-K dot product is off by
// ~2^-22 rms|ab| sqrt(K) -- measured 7.6e-8 relative to rms(C) for K = 64 .. 18432, i.e. 2-4x CLOSER to the fp64 result than
// a CPU fp32 GEMM (1.2e-7 .. 3.0e-7), at half the MFMA work and 2/3 of the operand bytes of the 3 x bf16 / 6-product scheme
// (conv_bf16x3.hip, 5.8e-9: more exact than the fp32 reference itself can resolve).  tests/test_ops_gpu.py holds all three
// GEMM back-ends to the same fp64 bar.
//
// Operand storage ("split-2 interleaved"): [row][k/16][plane 0..1][16 fp16] = 64 contiguous bytes per (row, 16 k); a K-tile is
// 32 k = 128 bytes per row.  A 64-byte zero block follows the data (masked loader lanes); the scale lives in a device float.
#include "common.h"
#include <algorithm>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

#define SP_GLDS16(src, dst)                                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),                       \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

constexpr int HBM = 256, HBN = 128, HBK = 32;
constexpr int HA_BYTES = HBM * 128;              // 32768: [row][8 x 16 B]
constexpr int HB_BYTES = HBN * 128;              // 16384
constexpr int HSTAGE = HA_BYTES + HB_BYTES;      // 49152
constexpr int HNSTAGE = 3;                       // 2 K-tiles (64 k) in flight, 144 KB LDS
constexpr int HCHUNK_KT = 8;                     // fold acc into tot every 8 K-tiles (256 k)
// HALO build of h2_kernel (3x3 stride-1 "same" convs on 64-pixel-wide maps): per 32-channel block ONE activation block with a
// one-pixel halo -- the tile's 4 image rows plus the row above and below, 66 pixel slots per row -- serves all 9 filter taps
constexpr int HALO_W = 64, HALO_ROWSLOTS = HALO_W + 2, HALO_NSLOT = 6 * HALO_ROWSLOTS;      // 396 pixel slots of 128 B
constexpr int HALO_PIECES = (HALO_NSLOT + 7) / 8;                                         // 50 LDS-DMA pieces of 1 KB
constexpr int HALO_A_BYTES = HALO_PIECES * 1024;                                          // 51200 per block, two blocks in flight
constexpr int HALO_LDS = 2 * HALO_A_BYTES + 3 * HB_BYTES;                                 // 151552

struct H2Args {
    const uint16_t* X;    // [pixels][Kc/16][2][16]
    const uint16_t* W;    // [Nout][K/16][2][16]
    const float* bias;
    float* C;
    const float* sx;      // device scalar: scale of the activation operand
    const float* sw;      // scale of the weight operand: one device scalar, or (sw_rows) one per weight row = output column
    int sw_rows;
    int64_t M;
    int Hi, Wi, Kc;
    int ldx;              // channels per pixel row of X (>= Kc: a GEMM may take the first Kc channels of wider rows)
    int Ho, Wo, Nout, ldc;
    int KH, KW, stride, pad, dil;
    int64_t ldwb;         // bytes per weight row (4 * K)
    int64_t w_bstride;    // tap-major build: bytes between the weight sets of consecutive batch items (0: one shared set)
    int rows_per_batch;   // output rows per batch item (a multiple of the 256-row tile)
    int ncblk, nkt, tiles_n;
    float alpha;
    int beta, relu;
    uint32_t x_bytes, w_bytes;
    // LSTM epilogue (h2_kernel<..., LSTM = true>): the GEMM is the h-gate conv of the ConvLSTM, Nout = 4 * lC gate columns
    const float* l_xg;      // [M][4][lC]  x-gate pre-activations (gate-major columns i, f, o, g)
    const float* l_cprev;   // [M][lC]
    const float* l_spcol;   // [M][lKP]    taps of the spatial memory maps
    const float* l_wc;      // [B][3 lC][lKP] per-sample contracted filter of the rank-1 gate term
    float* l_gates;         // [M][4][lC]  activated gates
    float* l_c;             // [M][lC]
    float* l_h;             // [M][lC]
    unsigned* l_hamax;      // max |h| (float bits), reset by the launcher
    uint16_t* l_hplanes;    // nullable: h as the 2xfp16 split operand of its consumers, scale from l_hbound >= max|h|
    float* l_hscale;        // [2] {scale, bound}
    float l_hbound;
    int lC, lP, lKP;
    int l_probe;            // timing build only (sp_set_tuning("h2_dbg", n) on the fused-cell launch): see the LSTM epilogue
    // BatchNorm batch statistics of the output, fused into the epilogue (forward, 16x16x32 build): per M-tile and output column
    // the sum and the sum of squares (fp64) and min / max (fp32) of the tile's valid rows, in the [G = M-tiles][2][Nout] layout
    // of bn_pool.hip's first reduction stage -- the BatchNorm behind this conv starts at its second stage
    double* st_partial;
    float* st_mm;
    // Row sparsity of a gradient (data gradient, MODE 1): row_last[img] = last decode step at which sample img receives any loss gradient
    // (from the loss masks); at step row_step > row_last[img] every row of that sample is exactly zero, so a tile that lies inside such
    // a sample (Ho * Wo % 256 == 0) is zero: written as zeros (or left alone under beta) without touching the operands.
    const int* row_last;
    int row_step;
    // row_nimg > 0 (data gradient, whole tiles per sample, <= 64 samples): LIVE-FIRST tile order.  The plain order hands every XCD a
    // contiguous range of M-tiles (xcd_remap: 4 samples of the 32 at the benchmark size), so with dead samples the launch lasts as
    // long as the XCD with the most live samples (measured: 2.39 ms at 56 % live against 3.05 ms dense).  Here the first
    // nlive * tiles-per-sample * tiles_n workgroups take the tiles of the live samples, dealt to the XCDs in equal contiguous ranges,
    // and the remaining workgroups write the dead samples' zero tiles.  Which workgroup computes a tile changes, the tile does not.
    int row_nimg;
};

// Block tile 256 x 128 x 32, 512 threads = 8 waves (4 along M x 2 along N, wave tile 64x64), 1 workgroup per CU, LDS-DMA
// staging in a 3-stage ring with counted vmcnt (the structure of b3_kernel, see there for why).  Rows are 128 B = half a
// 256-byte LDS bank row, so the source-side swizzle is an XOR: chunk c of row r is stored at position c ^ ((r>>1)&7); the 16
// rows of every ds_read_b128 lane group ({0-3,12-15,20-27} / {4-11,16-19,28-31}) then hit 16 distinct 16-byte slots.
// Schedule of the two waves that share a SIMD (waves w and w+4; MI355X_MICROARCH.md "Two waves per SIMD" items 1, 9): ping-pong with
// ONE barrier per K-tile -- waves 0-3 run  [read(t)] [issue loads(t+2)] [48 MFMA(t)] [wait] [barrier],  waves 4-7 (s_setprio 1: the
// second-dispatched half loses issue arbitration otherwise) run  [48 MFMA(t-1)] [read(t)] [issue loads(t+2)] [wait] [barrier]:  at
// any time one wave of a SIMD is in its matrix segment while its partner is in its LDS / DMA segment; waves 4-7 finish tile nkt-1
// after the loop.  The tap-major build (CBM = false) issues the early half's loads BEFORE its fragment reads (measured better there).
// Schedules that lost their A/B (lockstep, half-tile stagger, LDS-DMA pieces spread through the matrix segment, a second barrier
// per K-tile, 32x32x16 MFMAs, a 128x128-tile / two-workgroups-per-CU kernel for short K) are in the history of this file and in
// DESIGN.md sections 5, 9g with their measurements; they are not built any more.
// DBG (timing library only, -DSP_TIMING_VARIANTS; wrong results): 1 = no global loads (MFMA + LDS side alone), 2 = no MFMAs (load side
// alone), 3 = MFMAs only (no loads, no fragment reads, no barriers), 9 = LDS-DMA loads only.
// NPROD: 3 = fp32-faithful (the two cross terms, then the main product); 1 = THROUGHPUT MODE: only the main product a1*b1, i.e.
// both operands rounded to ONE fp16 plane (fp16 in / fp32 accumulate) -- same operand storage, a third of the MFMA work.
// MFMA shape: v_mfma_f32_16x16x32_f16 (4x4 tiles of 16x16 per wave, one 32-k block per K-tile): same fragments-per-FLOP from LDS and
// cycles per FLOP as 32x32x16, but the chip holds a higher clock on this shape under MFMA load (MI355X_MICROARCH.md "DVFS give-back").
// CBM: channel-block-major K order (for each 32-channel block: all filter taps) instead of tap-major.  Consecutive K-tiles then
// re-read the SAME pixels shifted by one tap, so the activation panel of an M-tile is served from L2 for 8 of 9 taps instead of
// being re-fetched from the Infinity Cache / HBM side per tap.  Per loader lane: the byte offset of its pixel at tap (0,0) and a
// bit mask of the taps that fall inside the image; the tap's offset is one scalar per K-tile.  Needs KH*KW <= 32 and (dgrad)
// stride 1; the sum order over K differs from the tap-major build, so results agree to rounding, not bitwise.
// LSTM (forward): the GEMM is the h-gate conv of the ConvLSTM and the epilogue is the whole cell.  The 128 weight rows of a
// workgroup are gathered as  4 gates x 32 channels  (tile row r -> gate (r>>4)&3, channel c0 + 16*(r>>6) + (r&15)): no data is
// permuted, only the loader's row address, and a lane then holds the four gate pre-activations of ONE channel for its 16 pixels
// in acc4[i][0..3].  Epilogue: + x-gate term + rank-1 gate term (spcol x wc, staged in the now idle LDS), sigmoid/tanh, cell and
// hidden state, max|h| -- the [M][4C] h-gate tensor (42 MB per decode step at the benchmark size) is never written or re-read.
// Needs P % 256 == 0 (a 256-pixel tile lies inside one sample: one filter slice per workgroup), C % 32 == 0, KP <= 32.
__device__ __forceinline__ float h2_sigmoid(float x) { return sp_sigmoid(x); }      // = decoder.hip sigmoidf_ (common.h)
// Chunk swizzle of the halo activation block.  A ds_read_b128 is served in lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (per
// 32-lane half): of the 16 consecutive pixel slots b .. b+15 a group reads slots {0-3,12-15} at chunk c (k-group g4 even) and slots
// {4-11} at chunk c ^ 1 (g4 odd), or the other way round.  The ring's swizzle  c ^ ((row >> 1) & 7)  is conflict-free only for b = 0
// mod 16; in the halo block b = (image row + tap row) * 66 + tap column is arbitrary (3600 of 4800 groups conflict over b = 0..599).
// XOR-ing ((slot >> 1) & 3) into bits 2:1 ONLY keeps bit 0 = the k-group parity: slots of one parity then differ in bits 2:1 within
// the outer four and within the middle four slot pairs (four consecutive values of slot >> 1 mod 4) and the two sets differ in bit 0
// -- 16 distinct 16-byte bank slots for every b (0 of 4800 conflict; enumerated in tests/test_cpu_host.py).
__device__ __forceinline__ int halo_swz(int slot) { return ((slot >> 1) & 3) << 1; }

template <int MODE, int NPROD, bool CBM, bool LSTM = false, bool HALO = false, int DBG = 0>
__global__ __launch_bounds__(512, 2) void h2_kernel(H2Args p) {
    static_assert(!LSTM || MODE == 0, "the LSTM epilogue belongs to the forward build");
    static_assert(!HALO || (CBM && NPROD == 3 && DBG == 0), "the halo build extends the channel-block-major schedule only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr bool LSTM_T = LSTM;      // the fused-cell build multiplies with the weights as the row operand (transposed tile), see mma_group
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;

#ifdef SP_TIMING_VARIANTS
    // probe (wrong results unless only the stagger bits are set): bit 0 = no epilogue loads, bit 1 = no epilogue stores, bit 2 = no epilogue
    // at all; bits 8.. = first-round stagger: G = (probe >> 8) & 15 groups of CUs start (group index) x 2 us x ((probe >> 12) & 31) late
    const int probe = p.l_probe;      // (launches without the cell epilogue: the stagger bits only)
    if ((probe >> 8) && blockIdx.x < 256) {
        const int G = (probe >> 8) & 15, step_us = 2 * ((probe >> 12) & 31);
        const long long wait = (long long)((blockIdx.x >> 3) % G) * step_us * 100;      // s_memrealtime: 100 MHz
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#else
    constexpr int probe = 0;
#endif
    // the two operand scales (device words written by the split kernels): requested FIRST, used by the epilogue -- behind the K loop
    // their latency would be exposed once per workgroup
    const float sx_dev = p.sx[0], sw_dev = p.sw_rows ? 1.f : p.sw[0];
    int tn, tmi;
    bool mapped = false;
    if constexpr (MODE == 1 && !LSTM) {
        if (p.row_nimg > 0) {                          // live-first order (H2Args::row_nimg); everything here is wave-uniform
            const int mine = lane < p.row_nimg ? p.row_last[lane] : -1;
            const uint64_t lm = __ballot(lane < p.row_nimg && mine >= p.row_step);
            const uint64_t all = p.row_nimg == 64 ? ~0ull : ((1ull << p.row_nimg) - 1ull);
            const int tps = (p.Ho * p.Wo) / HBM;        // tiles per sample
            const int nlive = __popcll(lm);
            const int nl = nlive * tps * p.tiles_n;
            const int bid = blockIdx.x;
            uint64_t mask;
            int tml;
            if (bid < nl) {
                supertile_map(xcd_remap(bid, nl), nlive * tps, p.tiles_n, tml, tn, CBM);
                mask = lm;
            } else {
                tn = (bid - nl) % p.tiles_n;
                tml = (bid - nl) / p.tiles_n;
                mask = all & ~lm;
            }
            for (int k = tml / tps; k > 0; --k) mask &= mask - 1ull;      // drop the k lowest set bits: sample = slot-th live / dead one
            tmi = (__ffsll((unsigned long long)mask) - 1) * tps + tml % tps;
            mapped = true;
        }
    }
    if (!mapped) supertile_map(xcd_remap(blockIdx.x, gridDim.x), gridDim.x / p.tiles_n, p.tiles_n, tmi, tn, CBM);
    const int64_t m0 = (int64_t)tmi * HBM;
    const int n0 = LSTM ? tn * 32 : tn * HBN;          // LSTM: first CHANNEL of the tile
    if constexpr (!LSTM) {
        if (p.row_last != nullptr) {                   // (scalar: the whole workgroup takes the same way, before any barrier)
            // data gradient: the tile's sample; batched forward GEMM (one item per sample, whole tiles per item): the tile's item
            const int howo = MODE == 1 ? p.Ho * p.Wo : p.rows_per_batch;
            const int img = (int)(m0 / howo);
            if (howo % HBM == 0 && m0 < p.M && p.row_last[img] < p.row_step) {
                if (!p.beta) {
                    for (int i = t; i < HBM * (HBN / 4); i += 512) {
                        const int row = i / (HBN / 4), n = n0 + (i % (HBN / 4)) * 4;
                        const int64_t m = m0 + row;
                        if (m >= p.M) continue;
                        float* dst = p.C + m * p.ldc + n;
                        if ((p.ldc & 3) == 0 && n + 3 < p.Nout && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0) {
                            *reinterpret_cast<float4*>(dst) = make_float4(0.f, 0.f, 0.f, 0.f);
                        } else {
                            for (int e = 0; e < 4; ++e)
                                if (n + e < p.Nout) dst[e] = 0.f;
                        }
                    }
                }
                return;
            }
        }
    }
    const int HoWo = p.Ho * p.Wo;

    // ---- loader mapping: LDS chunk g = t + 512 j -> row g/8, position g%8; source chunk = position ^ swizzle(row) ----
    int a_c8[4], a_py[4], a_px[4], a_boff[4];
    bool a_rowok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = t + 512 * j;
        const int row = idx >> 3, pos = idx & 7;
        a_c8[j] = (pos ^ ((row >> 1) & 7)) * 16;
        const int64_t m = m0 + row;
        a_rowok[j] = m < p.M;
        const uint32_t mm = a_rowok[j] ? (uint32_t)m : 0u;        // M < 2^31 (launch_h2): 32-bit divisions, a fifth of the 64-bit sequence --
        const uint32_t b = mm / (uint32_t)HoWo;                   // a short-K tile lives ~15 us, the prologue is a visible part of it
        const uint32_t rem = mm - b * (uint32_t)HoWo;
        const int yo = (int)(rem / (uint32_t)p.Wo), xo = (int)rem - yo * p.Wo;
        a_boff[j] = (int)b * p.Hi * p.Wi;
        if (MODE == 0) {
            a_py[j] = yo * p.stride - p.pad;
            a_px[j] = xo * p.stride - p.pad;
        } else {
            a_py[j] = yo + p.pad;
            a_px[j] = xo + p.pad;
        }
    }
    uint32_t b_voff[2];
    bool b_ok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int idx = t + 512 * j;
        const int row = idx >> 3, pos = idx & 7;
        const int c8 = (pos ^ ((row >> 1) & 7)) * 16;
        if constexpr (LSTM) {
            const int ch = n0 + (row >> 6) * 16 + (row & 15);
            b_ok[j] = ch < p.lC;
            b_voff[j] = b_ok[j] ? (uint32_t)((int64_t)(((row >> 4) & 3) * p.lC + ch) * p.ldwb + c8) : 0u;
        } else {
            b_ok[j] = (n0 + row) < p.Nout;
            b_voff[j] = b_ok[j] ? (uint32_t)((int64_t)(n0 + row) * p.ldwb + c8) : 0u;
        }
    }
    // batched GEMM with one weight set per batch item (tap-major build: the cell backward's rank-1 terms): the tile lies inside one item
    const unsigned char* Wl = reinterpret_cast<const unsigned char*>(p.W);
    uint32_t w_left = p.w_bytes;              // bytes from Wl to the zero block behind the last weight set
    if constexpr (!CBM) {
        if (p.w_bstride) {
            const int64_t off = (m0 / p.rows_per_batch) * p.w_bstride;
            Wl += off;
            w_left -= (uint32_t)off;
        }
    }
    int ld_ky = 0, ld_kx = 0, ld_cblk = 0, ld_kt = 0;
    uint32_t a_voff[4];
    bool a_ok[4];
    const int rowbytes = p.ldx * 4;
    uint32_t a_base0[4], a_mask[4];      // CBM: offset of the lane's pixel at tap (0,0) (wrapping arithmetic), valid-tap bits
    if constexpr (CBM && !HALO) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a_base0[j] = (uint32_t)(a_boff[j] + a_py[j] * p.Wi + a_px[j]) * (uint32_t)rowbytes + (uint32_t)a_c8[j];
            uint32_t msk = 0;
            for (int ky = 0; ky < p.KH; ++ky)
                for (int kx = 0; kx < p.KW; ++kx) {
                    const int iy = MODE == 0 ? a_py[j] + ky * p.dil : a_py[j] - ky * p.dil;
                    const int ix = MODE == 0 ? a_px[j] + kx * p.dil : a_px[j] - kx * p.dil;
                    if (a_rowok[j] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) msk |= 1u << (ky * p.KW + kx);
                }
            a_mask[j] = msk;
        }
    }
    int ld_tap = 0;
    // HALO: the lane's source offsets of the activation block (per-lane constants over the whole K loop; only the channel block,
    // a scalar, changes).  LDS piece q = wave + 8 j holds pixel slots 8 q .. 8 q + 7; slot s = (block row s / 66, column s % 66 - 1),
    // block row 0 = the image row above the tile's first row.  Slots outside the image (halo columns, rows above / below the image,
    // the tail of the last piece) read the zero block.  Chunk swizzle: halo_swz (see there).
    uint32_t h_off[7];
    uint32_t h_valid = 0;
    const int h_npieces = HALO ? (HALO_PIECES - wave + 7) / 8 : 0;          // pieces this wave issues per block (7 or 6; scalar)
    if constexpr (HALO) {
        const int img = (int)((uint32_t)m0 / (uint32_t)HoWo);
        const int y0 = (int)(((uint32_t)m0 - (uint32_t)img * (uint32_t)HoWo) / (uint32_t)p.Wo);
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int slot = (wave + 8 * j) * 8 + (lane >> 3), pos = lane & 7;
            const int br = slot / HALO_ROWSLOTS, sx = slot - br * HALO_ROWSLOTS;
            const int y = y0 - 1 + br, x = sx - 1;
            const bool ok = slot < HALO_NSLOT && (unsigned)y < (unsigned)p.Hi && (unsigned)x < (unsigned)p.Wi && m0 < p.M;
            h_off[j] = ok ? (uint32_t)((img * p.Hi + y) * p.Wi + x) * (uint32_t)rowbytes + (uint32_t)((pos ^ halo_swz(slot)) * 16) : 0u;
            h_valid |= ok ? (1u << j) : 0u;
        }
    }
    int h_cb = 0;                 // HALO: next activation block (channel block) to issue
    auto issue_block = [&]() {    // the whole block at once (prologue)
        if constexpr (HALO) {
            unsigned char* stA = smem + (h_cb & 1) * HALO_A_BYTES;
            const uint32_t koffA = (uint32_t)h_cb * 128u;
            const unsigned char* baseA = reinterpret_cast<const unsigned char*>(p.X) + koffA;
            const uint32_t zrelA = p.x_bytes - koffA;
#pragma unroll
            for (int j = 0; j < 7; ++j)
                if (j < h_npieces) SP_GLDS16(baseA + (((h_valid >> j) & 1u) ? h_off[j] : zrelA), stA + (wave + 8 * j) * 1024);
            ++h_cb;
        }
    };

    auto tap_update = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool ok = a_rowok[j];
            int iy, ix;
            if (MODE == 0) {
                iy = a_py[j] + ld_ky * p.dil;
                ix = a_px[j] + ld_kx * p.dil;
                ok = ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            } else {
                const int ty = a_py[j] - ld_ky * p.dil, tx = a_px[j] - ld_kx * p.dil;
                ok = ok && ty >= 0 && tx >= 0;
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
            a_ok[j] = ok;
            a_voff[j] = ok ? (uint32_t)(a_boff[j] + iy * p.Wi + ix) * (uint32_t)rowbytes + (uint32_t)a_c8[j] : 0u;
        }
    };

    auto issue_tile = [&](int stage) {
        unsigned char* st = smem + stage * HSTAGE;
        if constexpr (HALO) {      // weight tile of K-tile (ld_cblk, ld_tap) into the weight ring; activations come by issue_block
            unsigned char* stB = smem + 2 * HALO_A_BYTES + stage * HB_BYTES;
            const uint32_t koffB = (uint32_t)(ld_tap * p.ncblk + ld_cblk) * 128u;
            const unsigned char* baseB = reinterpret_cast<const unsigned char*>(p.W) + koffB;
            const uint32_t zrelB = p.w_bytes - koffB;
#pragma unroll
            for (int j = 0; j < 2; ++j) SP_GLDS16(baseB + (b_ok[j] ? b_voff[j] : zrelB), stB + (wave + 8 * j) * 1024);
            if (++ld_tap == 9) {
                ld_tap = 0;
                ++ld_cblk;
            }
            return;
        }
        if constexpr (CBM) {
            const int step = (ld_ky * p.Wi + ld_kx) * p.dil;                   // scalar: pixel offset of tap (ld_ky, ld_kx)
            const uint32_t delta = (uint32_t)(MODE == 0 ? step : -step) * (uint32_t)rowbytes;
            const uint32_t koffA = (uint32_t)ld_cblk * 128u;
            const unsigned char* baseA = reinterpret_cast<const unsigned char*>(p.X) + koffA;
            const uint32_t zrelA = p.x_bytes - koffA;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                SP_GLDS16(baseA + (((a_mask[j] >> ld_tap) & 1u) ? a_base0[j] + delta : zrelA), st + (wave + 8 * j) * 1024);
            const uint32_t koffB = (uint32_t)(ld_tap * p.ncblk + ld_cblk) * 128u;
            const unsigned char* baseB = reinterpret_cast<const unsigned char*>(p.W) + koffB;
            const uint32_t zrelB = p.w_bytes - koffB;
#pragma unroll
            for (int j = 0; j < 2; ++j) SP_GLDS16(baseB + (b_ok[j] ? b_voff[j] : zrelB), st + HA_BYTES + (wave + 8 * j) * 1024);
            ++ld_tap;
            if (++ld_kx == p.KW) {
                ld_kx = 0;
                if (++ld_ky == p.KH) {
                    ld_ky = 0;
                    ld_tap = 0;
                    ++ld_cblk;
                }
            }
            return;
        }
        if (ld_cblk == 0) tap_update();
        const uint32_t koffA = (uint32_t)ld_cblk * 128u;                 // scalar: 32-channel block inside the pixel row
        const unsigned char* baseA = reinterpret_cast<const unsigned char*>(p.X) + koffA;
        const uint32_t zrelA = p.x_bytes - koffA;
#pragma unroll
        for (int j = 0; j < 4; ++j) SP_GLDS16(baseA + (a_ok[j] ? a_voff[j] : zrelA), st + (wave + 8 * j) * 1024);
        const uint32_t koffB = (uint32_t)ld_kt * 128u;
        const unsigned char* baseB = Wl + koffB;
        const uint32_t zrelB = w_left - koffB;
#pragma unroll
        for (int j = 0; j < 2; ++j) SP_GLDS16(baseB + (b_ok[j] ? b_voff[j] : zrelB), st + HA_BYTES + (wave + 8 * j) * 1024);
        ++ld_kt;
        if (++ld_cblk == p.ncblk) {
            ld_cblk = 0;
            if (++ld_kx == p.KW) {
                ld_kx = 0;
                ++ld_ky;
            }
        }
    };

    // fragment read offsets: row r, chunk c = (k-group >> 1) * 4 + plane * 2 + (k-group & 1) stored at position c ^ ((r>>1)&7); lane = (row l16,
    // k-group g4) of the 32-k block; the four 16-row tiles of a wave are reached by constant offsets
    const int l16 = lane & 15, g4 = lane >> 4;
    // per-row weight scales of the lane's four output columns: requested HERE, ahead of the K loop's LDS-DMA pieces (loads return in
    // order: the counted waits are unaffected), used by the epilogue -- requested there, their latency was exposed once per tile
    float pre_sw[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (!LSTM && !HALO) {
        if (p.sw_rows) {
            int64_t r0 = 0;
            if constexpr (!CBM) {
                if (p.w_bstride) r0 = (m0 / p.rows_per_batch) * (int64_t)p.Nout;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + l16;
                if (n < p.Nout) pre_sw[j] = p.sw[r0 + n];
            }
        }
    }
    const int rot = (l16 >> 1) & 7;
    int offA[2], offB[2];            // [plane]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const int pos = ((g4 >> 1) * 4 + pl * 2 + (g4 & 1)) ^ rot;
        offA[pl] = (wm * 64 + l16) * 128 + pos * 16;
        offB[pl] = HA_BYTES + (wn * 64 + l16) * 128 + pos * 16;
    }

    // HALO: fragment row i*16 + l16 of wave row-group wm is pixel (image row wm of the tile, column i*16 + l16) -> block slot
    // (wm + dy) * 66 + column + dx with (dy, dx) = (ky, kx) forward, (2 - ky, 2 - kx) data gradient; the slot (hence the swizzle)
    // changes with the tap, 16-slot steps leave the swizzle alone (the four row tiles are reached by immediate offsets)
    const int h_slot0 = wm * HALO_ROWSLOTS + l16;
    const int h_chunk = (g4 >> 1) * 4 + (g4 & 1);
    int rd_tap = 0, rd_cb = 0;           // HALO: coordinates of the K-tile whose fragments are read next
    f32x4 tot4[4][4], acc4[4][4];        // two-level accumulation: chunk accumulator, total
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tot4[i][j][r] = 0.f;
                acc4[i][j][r] = 0.f;
            }

    // prologue: tiles 0 and 1 in flight, tile 0 landed (every thread issues 6 loads per tile)
    constexpr bool do_load = DBG != 1 && DBG != 3, do_mma = DBG != 2 && DBG != 9, do_lds = DBG != 3 && DBG != 9;      // DBG 9: loads only
    constexpr bool do_bar = DBG != 3;
    int issued = 0;
    if constexpr (HALO) issue_block();                        // activation block of channel block 0, ahead of the weight tiles
    if (do_load)
        for (; issued < HNSTAGE - 1 && issued < p.nkt; ++issued) issue_tile(issued);
    if (issued >= 2) {
        if constexpr (HALO) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    f16x8 af[2][2][2], bf[2][2][2];      // [kk][i][plane]: all 16 fragments of a K-tile (64 VGPRs)
    // all 16 fragment reads of the K-tile are issued up front: the reads of the second 16-k group then complete behind the
    // MFMAs of the first (with the reads interleaved per group the MFMA+LDS side alone took 3.4 ms)
    auto read_frags = [&](int stage_, int kt_) {
        const unsigned char* st = smem + stage_ * HSTAGE;
        if constexpr (!do_lds) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        f16x8 z;
#pragma unroll
                        for (int e = 0; e < 8; ++e) z[e] = (_Float16)(float)(kt_ + e + lane);
                        af[kk][i][pl] = z;
                        bf[kk][i][pl] = z;
                    }
        } else if constexpr (HALO) {
            const int ky = rd_tap / 3, kx = rd_tap - 3 * ky;                                 // scalar
            const int slot = h_slot0 + (MODE == 0 ? ky * HALO_ROWSLOTS + kx : (2 - ky) * HALO_ROWSLOTS + 2 - kx);
            const int sw = halo_swz(slot);
            const unsigned char* stA = smem + (rd_cb & 1) * HALO_A_BYTES + slot * 128;
            const unsigned char* stB = smem + 2 * HALO_A_BYTES + stage_ * HB_BYTES - HA_BYTES;
            const int oa0 = ((h_chunk) ^ sw) * 16, oa1 = ((h_chunk + 2) ^ sw) * 16;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[kk][i][0] = *reinterpret_cast<const f16x8*>(stA + oa0 + (2 * kk + i) * 16 * 128);
                    af[kk][i][1] = *reinterpret_cast<const f16x8*>(stA + oa1 + (2 * kk + i) * 16 * 128);
                    bf[kk][i][0] = *reinterpret_cast<const f16x8*>(stB + offB[0] + (2 * kk + i) * 16 * 128);
                    bf[kk][i][1] = *reinterpret_cast<const f16x8*>(stB + offB[1] + (2 * kk + i) * 16 * 128);
                }
            if (++rd_tap == 9) {
                rd_tap = 0;
                ++rd_cb;
            }
        } else {
            // af[kk][i][pl] holds row tile 2*kk + i (16 rows each) of the ONE 32-k block; bf likewise for column tiles
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < (NPROD == 3 ? 2 : 1); ++pl) {
                        af[kk][i][pl] = *reinterpret_cast<const f16x8*>(st + offA[pl] + (2 * kk + i) * 16 * 128);
                        bf[kk][i][pl] = *reinterpret_cast<const f16x8*>(st + offB[pl] + (2 * kk + i) * 16 * 128);
                    }
        }
    };
    auto mma_group = [&](int kk) {
        if constexpr (do_mma) {
            // group kk = row tiles 2kk, 2kk+1 against all four column tiles (24 of the K-tile's 48 MFMAs)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4& a4 = acc4[2 * kk + i][j];
                    if constexpr (LSTM_T) {
                        // transposed tile (weights as the row operand): the lane then holds 4 CONSECUTIVE CHANNELS of one pixel -- see
                        // the cell epilogue; same fragments, same products, same sum order
                        if constexpr (NPROD == 3) {
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j >> 1][j & 1][1], af[kk][i][0], a4, 0, 0, 0);
                            a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j >> 1][j & 1][0], af[kk][i][1], a4, 0, 0, 0);
                        }
                        a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j >> 1][j & 1][0], af[kk][i][0], a4, 0, 0, 0);
                        continue;
                    }
                    if constexpr (NPROD == 3) {
                        a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk][i][0], bf[j >> 1][j & 1][1], a4, 0, 0, 0);
                        a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk][i][1], bf[j >> 1][j & 1][0], a4, 0, 0, 0);
                    }
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk][i][0], bf[j >> 1][j & 1][0], a4, 0, 0, 0);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const f16x8 x0 = af[kk][i][pl], x1 = bf[kk][i][pl];
                    asm volatile("" ::"v"(x0), "v"(x1));
                }
        }
    };
    auto fold = [&](int kt_) {           // two-level accumulation: fold the chunk accumulator into the total every 256 k
        if (((kt_ + 1) & (HCHUNK_KT - 1)) == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tot4[i][j] += acc4[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.f;
                }
        }
    };
    auto wait_barrier = [&](int kt_) {
        // tile kt+1 must have landed: everything but the one younger tile (if it was issued)
        if (kt_ + 2 < p.nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (do_bar) __builtin_amdgcn_s_barrier();
    };
    auto prev_stage = [](int st_) { return st_ == 0 ? HNSTAGE - 1 : st_ - 1; };

    int stage = 0;
    const bool late = wave >= 4;         // the half of the workgroup that runs behind (scalar: uniform branch)
    if (late) __builtin_amdgcn_s_setprio(1);
    if constexpr (HALO) {
        // The schedule above with
        // the activation operand staged as halo blocks: per K-tile every lane issues its 2 weight pieces (tile kt + 2); at the first tap
        // of channel block cb it then issues the 6-7 pieces of activation block cb + 1 (into the block buffer that block cb - 1 left at
        // the barrier before this channel block); the block is complete two barriers later, seven K-tiles before its first use.
        // (One block piece per K-tile instead of the burst measured SLOWER: 3.62 / 3.50 ms against 3.56 / 3.43 without halo blocks.)
        int cur_tap = 0, cur_cb = 0;
        auto halo_issue = [&](int kt_) {                      // weight tile kt + 2; at the first tap of a channel block the next block
            if (kt_ + HNSTAGE - 1 < p.nkt) issue_tile(prev_stage(stage));
            if (cur_tap == 0 && cur_cb + 1 < p.ncblk) issue_block();
        };
        auto halo_wait = [&](int kt_) {
            // weight tile kt + 1 (issued one K-tile ago) must have landed; loads complete in issue order, so exactly the loads issued
            // after it may stay outstanding: this K-tile's 2 weight pieces, and the block issued behind the weight pieces of tap 0
            // (during taps 0 and 1: it sits between weight tiles kt + 1 and kt + 2 of tap 1)
            if (kt_ + 2 < p.nkt) {
                if (cur_tap <= 1 && cur_cb + 1 < p.ncblk) {
                    if (h_npieces == 7) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (++cur_tap == 9) {
                cur_tap = 0;
                ++cur_cb;
            }
        };
        if (!late) {
            for (int kt = 0; kt < p.nkt; ++kt) {
                read_frags(stage, kt);
                halo_issue(kt);
                mma_group(0);
                mma_group(1);
                fold(kt);
                halo_wait(kt);
                stage = (stage == HNSTAGE - 1) ? 0 : stage + 1;
            }
        } else {
            for (int kt = 0; kt < p.nkt; ++kt) {
                if (kt > 0) {
                    mma_group(0);                           // tile kt-1: its 16 fragments were read before the last barrier
                    mma_group(1);
                    fold(kt - 1);
                }
                read_frags(stage, kt);
                halo_issue(kt);
                halo_wait(kt);
                stage = (stage == HNSTAGE - 1) ? 0 : stage + 1;
            }
            mma_group(0);
            mma_group(1);
            fold(p.nkt - 1);
        }
    } else if (!late) {
        for (int kt = 0; kt < p.nkt; ++kt) {
            const bool pre = do_load && kt + HNSTAGE - 1 < p.nkt;
            // prefetch of tile kt+2 into the stage read in iteration kt-1 (its partner wave is in its matrix segment now): the tap-major
            // build issues it ahead of the fragment reads, the channel-block-major build behind them
            if (!CBM && pre) issue_tile(prev_stage(stage));
            read_frags(stage, kt);
            if (CBM && pre) issue_tile(prev_stage(stage));
            mma_group(0);
            mma_group(1);
            fold(kt);
            wait_barrier(kt);
            stage = (stage == HNSTAGE - 1) ? 0 : stage + 1;
        }
    } else {
        for (int kt = 0; kt < p.nkt; ++kt) {
            const bool pre = do_load && kt + HNSTAGE - 1 < p.nkt;
            if (kt > 0) {
                mma_group(0);                           // tile kt-1: its 16 fragments were read before the last barrier
                mma_group(1);
                fold(kt - 1);
            }
            read_frags(stage, kt);
            if (pre) issue_tile(prev_stage(stage));
            wait_barrier(kt);
            stage = (stage == HNSTAGE - 1) ? 0 : stage + 1;
        }
        mma_group(0);
        mma_group(1);
        fold(p.nkt - 1);
    }
    if (late) __builtin_amdgcn_s_setprio(0);

    // the two power-of-two scales are undone one after the other: their product can leave the fp32 range (tiny gradients x
    // ordinary weights) although every intermediate value here is representable
    // Per-row weight scales (sw_rows): a power-of-two scale per OUTPUT column factors out of the contraction exactly, so a weight
    // row that is 2^-30 of the tensor's maximum keeps its 22 bits (the per-tensor scale left it none: VERDICT r3 weak #1).
    const float isx = 1.f / sx_dev, isw = 1.f / sw_dev;
    int64_t swrow0 = 0;                        // first weight row of this tile's batch item in the scale vector
    if constexpr (!CBM) {
        if (p.w_bstride) swrow0 = (m0 / p.rows_per_batch) * (int64_t)p.Nout;
    }
    if constexpr (LSTM_T) {
        // ---- the ConvLSTM cell as the epilogue (AiR/models/baseline_attention.py:37-56), round 5 form: NO LDS, NO barrier ------------
        // The K loop multiplied with the weights as the ROW operand, so in acc4[i][q] the lane holds, for pixel  m0 + wm*64 + i*16 + l16,
        // gate q of the four consecutive channels  chb .. chb + 3,  chb = n0 + wn*16 + 4*g4: everything the cell reads and writes per
        // (pixel, gate) is ONE 16-byte access per lane (x-gates in, activated gates / c / h out: 64-byte runs per pixel, the partner
        // wave wn ^ 1 covers the other half of the 128-byte line) -- round 4 staged 128 KB in and 192 KB out per tile through the LDS
        // ring behind five workgroup barriers (64 ds_read_b32 + 64 ds_write_b32 + 36 b128 per lane), with the matrix pipe idle.  The
        // rank-1 gate term  sum_k spcol[pixel][k] * wc[gate][channel][k]  (k < KP <= 32) runs on the otherwise idle matrix pipe as
        // v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chains in k order, like the VALU loop it replaces: 60 MFMAs instead of 960 v_fma per lane
        // at KP = 20), its operands read straight from global memory (28 KB per tile, L2-resident) in the MFMA's own lane layout.
        // Waves leave the K loop at different times (the late half one K-tile behind) and run their epilogues independently.
        // Shapes: P % 256 == 0 and C % 32 == 0 (launcher) => every pixel row and channel of the tile exists.
        if (probe & 4) return;
        const bool nt_st = probe & 8, nt_ld = probe & 16;               // (A/B: streaming stores / loads)
        auto st4 = [&](float* dst, const float4& v) {
            const f32x4 x = {v.x, v.y, v.z, v.w};
            if (nt_st) __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(dst));
            else *reinterpret_cast<f32x4*>(dst) = x;
        };
        auto ld4 = [&](const float* src) {
            const f32x4 x = nt_ld ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src)) : *reinterpret_cast<const f32x4*>(src);
            return make_float4(x[0], x[1], x[2], x[3]);
        };
        const int KP = p.lKP, C = p.lC;
        const int b = (int)(m0 / p.lP);
        const int chb = n0 + wn * 16 + 4 * g4;
        const int64_t mrow = m0 + wm * 64 + l16;                       // + 16 i
        const int nk4 = (KP + 3) >> 2;                                 // k-steps of the rank-1 term (scalar)
        // rank-1 operands in the 16x16x4 layout: A[row = channel l16 of the wave's 16][k = g4], B[col = pixel l16][k = g4]
        const float* wc_l = p.l_wc + ((int64_t)b * 3 * C + n0 + wn * 16 + l16) * KP + g4;      // + q * C * KP + 4 kk
        const float* sp_l = p.l_spcol + mrow * KP + g4;                                        // + 16 i * KP + 4 kk
        float wa[3][4], sb[4][4];
        auto load_rank1 = [&](int kk0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bool ok = 4 * (kk0 + kk) + g4 < KP;
#pragma unroll
                for (int q = 0; q < 3; ++q) wa[q][kk] = ok ? wc_l[(int64_t)q * C * KP + 4 * (kk0 + kk)] : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) sb[i][kk] = ok ? sp_l[(int64_t)16 * i * KP + 4 * (kk0 + kk)] : 0.f;
            }
        };
        auto mma_rank1 = [&](int kk0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (kk0 + kk < nk4) {                                  // (scalar)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            tot4[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[q][kk], sb[i][kk], tot4[i][q], 0, 0, 0);
                }
        };
        float4 xv[4][4];                                               // x-gate pre-activations [i][gate]
        auto load_xg = [&](int i0, int i1) {
#pragma unroll
            for (int i = i0; i < i1; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    xv[i][q] = (probe & 1) ? make_float4(0.1f * q, 0.2f, 0.3f, 0.1f * i)
                                           : ld4(p.l_xg + (mrow + 16 * i) * 4 * C + q * C + chb);
        };
        // the activation scale is undone first: the chunk accumulators die here and the epilogue's loads get their registers
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) tot4[i][q][r] = (tot4[i][q][r] + acc4[i][q][r]) * isx;
        __builtin_amdgcn_sched_barrier(0);
        // every global read of the epilogue is requested now: weight-row scales, the first rank-1 operands, the x-gate tile, c_prev
        float4 swv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            swv[q] = p.sw_rows ? *reinterpret_cast<const float4*>(p.sw + q * C + chb) : make_float4(sw_dev, sw_dev, sw_dev, sw_dev);
        load_rank1(0);
        float4 cpv[4];
        auto load_cp = [&](int i0, int i1) {
#pragma unroll
            for (int i = i0; i < i1; ++i)
                cpv[i] = (p.l_cprev && !(probe & 1)) ? ld4(p.l_cprev + (mrow + 16 * i) * C + chb)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        load_xg(0, 2);
        load_cp(0, 2);
        __builtin_amdgcn_sched_barrier(0);
        // then the weight scale (the two power-of-two scales one after the other: their product can leave the fp32 range); per-row
        // weight scales: row (gate q, channel) = q * C + channel
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float is[4] = {1.f / swv[q].x, 1.f / swv[q].y, 1.f / swv[q].z, 1.f / swv[q].w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) tot4[i][q][r] *= is[r];
        }
        mma_rank1(0);
        if (nk4 > 4) {
            load_rank1(4);
            mma_rank1(4);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_xg(2, 4);                                     // second half of the tile: in flight under the gates of the first
        load_cp(2, 4);
        __builtin_amdgcn_sched_barrier(0);
        const float hsc = p.l_hplanes ? scale_of(__float_as_uint(p.l_hbound)) : 1.f;
        float hmx = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = mrow + 16 * i;
            float4 gi, gf, go, gg, cn, hn;
#define SP_CELL(e)                                                     \
            gi.e = h2_sigmoid(tot4[i][0][ri] + xv[i][0].e);            \
            gf.e = h2_sigmoid(tot4[i][1][ri] + xv[i][1].e);            \
            go.e = h2_sigmoid(tot4[i][2][ri] + xv[i][2].e);            \
            gg.e = sp_tanh(tot4[i][3][ri] + xv[i][3].e);               \
            cn.e = gf.e * cpv[i].e + gi.e * gg.e;                      \
            hn.e = go.e * cn.e;                                        \
            hmx = fmaxf(hmx, fabsf(hn.e));
            { constexpr int ri = 0; SP_CELL(x) }
            { constexpr int ri = 1; SP_CELL(y) }
            { constexpr int ri = 2; SP_CELL(z) }
            { constexpr int ri = 3; SP_CELL(w) }
#undef SP_CELL
            if (probe & 2) {                                   // (timing probe: keep the values alive, store nothing)
                asm volatile("" ::"v"(gi.x), "v"(gi.y), "v"(gi.z), "v"(gi.w), "v"(gf.x), "v"(gf.y), "v"(gf.z), "v"(gf.w));
                asm volatile("" ::"v"(go.x), "v"(go.y), "v"(go.z), "v"(go.w), "v"(gg.x), "v"(gg.y), "v"(gg.z), "v"(gg.w));
                asm volatile("" ::"v"(cn.x), "v"(cn.y), "v"(cn.z), "v"(cn.w), "v"(hn.x), "v"(hn.y), "v"(hn.z), "v"(hn.w));
                continue;
            }
            const int64_t ms = (probe & 64) ? (m & 1023) : m;        // (timing probe: every tile stores into the same 1024 rows)
            float* gp = p.l_gates + ms * 4 * C + chb;
            st4(gp, gi);
            st4(gp + C, gf);
            if (!(probe & 32)) {                               // (timing probe: half of the gate bytes)
                st4(gp + 2 * C, go);
                st4(gp + 3 * C, gg);
            } else {
                asm volatile("" ::"v"(go.x), "v"(go.y), "v"(go.z), "v"(go.w), "v"(gg.x), "v"(gg.y), "v"(gg.z), "v"(gg.w));
            }
            st4(p.l_c + ms * C + chb, cn);
            st4(p.l_h + ms * C + chb, hn);
            if (p.l_hplanes) {
                // h also as the split operand of its consumers (next step's h-gate conv, the saliency tap GEMM): |h| = |o * c| <= |c| <=
                // t + 1, so the operand scale needs no max|h| pass.  Layout [16-channel group][plane][16]: the lane's four channels are
                // 8 bytes of each plane (the four g4 lanes of a pixel complete the group's two 32-byte plane rows)
                ushort4 pa, pb;
                split2(hn.x, hsc, pa.x, pb.x);
                split2(hn.y, hsc, pa.y, pb.y);
                split2(hn.z, hsc, pa.z, pb.z);
                split2(hn.w, hsc, pa.w, pb.w);
                uint16_t* grp = p.l_hplanes + ((ms * C + chb) >> 4) * 32 + (chb & 15);
                *reinterpret_cast<ushort4*>(grp) = pa;
                *reinterpret_cast<ushort4*>(grp + 16) = pb;
            }
        }
        if (p.l_hplanes && blockIdx.x == 0) {
            if (t < 8) reinterpret_cast<uint2*>(p.l_hplanes + 2 * p.M * C)[t] = make_uint2(0u, 0u);      // 64-byte zero block
            if (t == 0) {
                p.l_hscale[0] = hsc;
                p.l_hscale[1] = p.l_hbound;
            }
        }
        if (p.l_hamax) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) hmx = fmaxf(hmx, __shfl_xor(hmx, off));
            if (lane == 0 && hmx > 0.f) atomicMax(p.l_hamax, __float_as_uint(hmx));
        }
        return;
    }
    {
        // C/D layout of 16x16x32: col = lane & 15, row = 4 * (lane >> 4) + reg.  Stored straight from the registers every instruction
        // writes 4 bytes per lane in 64-byte runs (16 columns of one row): 671 MB of output took ~0.4 ms, and the short-K pointwise
        // convs of the encoder -- whose run time IS their epilogue -- ran at 1.2 TB/s.  The wave's 64 x 64 tile is therefore staged
        // through its private slice of the idle LDS ring (row pitch 68 floats: the four row groups of a write land on banks
        // 0/16/32/48 + column, a 16-lane phase of the float4 read-back covers one whole row: no conflicts either way) and leaves as
        // float4 per lane, 256 contiguous bytes per row, a quarter of the store instructions; beta re-reads the same way.
        const bool stats = MODE == 0 && p.st_partial != nullptr;        // scalar
        const bool wide = (p.ldc & 3) == 0 && (p.Nout & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0;
        float* stg = reinterpret_cast<float*>(smem) + wave * (64 * 68);
        const bool full_m = m0 + HBM <= p.M;                             // (scalar)
        double cs[4], cq[4];
        float cmn[4], cmx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + l16;
            const bool n_ok = n < p.Nout;
            const float bv = (n_ok && p.bias) ? p.bias[n] : 0.f;
            const float iswn = (p.sw_rows && n_ok) ? 1.f / ((!LSTM && !HALO) ? pre_sw[j] : p.sw[swrow0 + n]) : isw;
            cs[j] = 0.0;
            cq[j] = 0.0;
            cmn[j] = INFINITY;
            cmx[j] = -INFINITY;
            // The common case -- float4 stores, every row of the tile inside the matrix -- without per-element control flow (round 6: the general
            // loop below tests `wide`, `stats` and the bounds per element, 2-3 scalar branches for each of a lane's 64 values: 5.8 us of a
            // short-K tile's 21 us went into the epilogue's ARITHMETIC, not its stores; profiles/r06_pointwise_modes.log).  Same values, same
            // order of the statistics' sums.  (Lanes whose column does not exist write nothing: their staging slots are never read.)
            if (wide && full_m) {
                if (n_ok) {
                    if (stats) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float v = p.alpha * (((tot4[i][j][r] + acc4[i][j][r]) * isx) * iswn) + bv;
                                stg[(i * 16 + 4 * g4 + r) * 68 + j * 16 + l16] = v;
                                cs[j] += (double)v;
                                cq[j] += (double)v * (double)v;
                                cmn[j] = fminf(cmn[j], v);
                                cmx[j] = fmaxf(cmx[j], v);
                            }
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                stg[(i * 16 + 4 * g4 + r) * 68 + j * 16 + l16] = p.alpha * (((tot4[i][j][r] + acc4[i][j][r]) * isx) * iswn) + bv;
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = i * 16 + 4 * g4 + r;
                    const int64_t m = m0 + wm * 64 + row;
                    float v = p.alpha * (((tot4[i][j][r] + acc4[i][j][r]) * isx) * iswn) + bv;
                    if (wide) {
                        stg[row * 68 + j * 16 + l16] = v;
                    } else if (n_ok && m < p.M) {
                        float* dst = p.C + m * p.ldc + n;
                        if (p.beta) v += *dst;
                        if (p.relu) v = fmaxf(v, 0.f);
                        *dst = v;
                    }
                    if (stats && n_ok && m < p.M) {          // (statistics are only taken without beta / relu / bias: v is final)
                        cs[j] += (double)v;
                        cq[j] += (double)v * (double)v;
                        cmn[j] = fminf(cmn[j], v);
                        cmx[j] = fmaxf(cmx[j], v);
                    }
                }
        }
        if (wide) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private staging: no barrier
            const int cq4 = lane & 15, rsub = lane >> 4;
            const int n = n0 + wn * 64 + 4 * cq4;
            if (full_m && n0 + HBN <= p.Nout && !p.relu) {                  // (scalar) the whole tile exists: 16 x {read 16 B, store 16 B}
                float* dst0 = p.C + (m0 + wm * 64 + rsub) * p.ldc + n;
                const float* src0 = stg + rsub * 68 + 4 * cq4;
                if (!p.beta) {
#pragma unroll
                    for (int ps = 0; ps < 16; ++ps)
                        *reinterpret_cast<float4*>(dst0 + (int64_t)(ps * 4) * p.ldc) = *reinterpret_cast<const float4*>(src0 + ps * 4 * 68);
                } else {       // accumulate into the tensor that is there (a data gradient merged into the other consumer's): the 16 reads first
                    float4 o[16];
#pragma unroll
                    for (int ps = 0; ps < 16; ++ps) o[ps] = *reinterpret_cast<const float4*>(dst0 + (int64_t)(ps * 4) * p.ldc);
#pragma unroll
                    for (int ps = 0; ps < 16; ++ps) {
                        float4 v = *reinterpret_cast<const float4*>(src0 + ps * 4 * 68);
                        v.x += o[ps].x; v.y += o[ps].y; v.z += o[ps].z; v.w += o[ps].w;
                        *reinterpret_cast<float4*>(dst0 + (int64_t)(ps * 4) * p.ldc) = v;
                    }
                }
            } else
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                const int row = ps * 4 + rsub;
                const int64_t m = m0 + wm * 64 + row;
                if (m < p.M && n < p.Nout) {
                    float4 v = *reinterpret_cast<const float4*>(stg + row * 68 + 4 * cq4);
                    float4* dst = reinterpret_cast<float4*>(p.C + m * p.ldc + n);
                    if (p.beta) {
                        const float4 o = *dst;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    if (p.relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    *dst = v;
                }
            }
        }
        if (stats) {
            // (raw barriers with lgkmcnt(0): the hazards are LDS ones; __syncthreads() would also wait until the tile's stores are acknowledged)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // every wave is done with its staging slice
            double* sh_s = reinterpret_cast<double*>(smem);          // [4 wm][128 col][2]
            float* sh_m = reinterpret_cast<float*>(smem + 4 * HBN * 2 * sizeof(double));
#pragma unroll
            for (int j = 0; j < 4; ++j) {                            // the four 16-lane groups hold rows 4*g4 + r of the same column
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {
                    cs[j] += __shfl_xor(cs[j], off);
                    cq[j] += __shfl_xor(cq[j], off);
                    cmn[j] = fminf(cmn[j], __shfl_xor(cmn[j], off));
                    cmx[j] = fmaxf(cmx[j], __shfl_xor(cmx[j], off));
                }
                if (g4 == 0) {
                    const int col = wn * 64 + j * 16 + l16;
                    sh_s[(wm * HBN + col) * 2 + 0] = cs[j];
                    sh_s[(wm * HBN + col) * 2 + 1] = cq[j];
                    sh_m[(wm * HBN + col) * 2 + 0] = cmn[j];
                    sh_m[(wm * HBN + col) * 2 + 1] = cmx[j];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t < HBN && n0 + t < p.Nout) {
                double a = 0.0, b = 0.0;
                float mn = INFINITY, mx = -INFINITY;
#pragma unroll
                for (int w = 0; w < 4; ++w) {              // fixed order over the four row groups of the tile
                    a += sh_s[(w * HBN + t) * 2 + 0];
                    b += sh_s[(w * HBN + t) * 2 + 1];
                    mn = fminf(mn, sh_m[(w * HBN + t) * 2 + 0]);
                    mx = fmaxf(mx, sh_m[(w * HBN + t) * 2 + 1]);
                }
                const int64_t g = m0 / HBM;
                p.st_partial[(g * 2 + 0) * p.Nout + n0 + t] = a;
                p.st_partial[(g * 2 + 1) * p.Nout + n0 + t] = b;
                p.st_mm[(g * 2 + 0) * p.Nout + n0 + t] = mn;
                p.st_mm[(g * 2 + 1) * p.Nout + n0 + t] = mx;
            }
        }
    }
}

// ================================================================================================================
// Weight gradient:  dW[co][tap][ci] = sum_m dY[m][co] * X[pix(m,tap)][ci]   (K = pixels), the structure of w3_kernel:
// K-tiles of 32 pixels staged as [pixel][channel chunks] rows by LDS-DMA, K-major fragments by ds_read_b64_tr_b16.
// Pixel rows are 1024 B (A: 256 co) / 512 B (B: 128 ci), both = 0 mod 256, so the four pixel rows of a transposed-read block
// would collide; row r is stored rotated by rot(r&3) = {0,2,8,10} chunks, which puts the 4 rows x 2 channel groups of every
// 32-lane half on 8 distinct 32-byte slots of the 256-byte bank row.
struct HWArgs {
    const uint16_t* X;     // split-2 [pixels_in][Ci/16][2][16]
    const uint16_t* dY;    // split-2 [pixels_out][Co/16][2][16]
    float* out;
    const float* sx;       // scale of X: one device scalar, or (sx_vec) one per input channel [Ci]
    const float* sy;       // scale of dY: one device scalar, or (sy_vec) one per output channel [Co]
    int sx_vec, sy_vec;    // K = pixels: a power-of-two scale per CHANNEL of either operand factors out of the contraction exactly
    int64_t M;
    int Hi, Wi, Ci, Ho, Wo, Co;
    int ldy;               // channels per pixel row of dY (>= Co: the GEMM may take the first Co channels of wider rows)
    int Nvalid;            // output columns that exist (<= Ntot = KH*KW*Ci: a batched GEMM whose Ci was padded up to 16 k)
    int batched;           // 1: blockIdx.y is a batch item (rows_per_split pixels each), its result goes to out + item * slab_stride, scaled
    int KH, KW, stride, pad, dil;
    int Ntot, ldo, tiles_n, splits;
    int64_t rows_per_split, slab_stride;
    float alpha;
    int beta;
    uint32_t x_bytes, y_bytes;
    const int* row_last;   // batched form (one item per sample): items with row_last[item] < row_step have an all-zero dY -> zero result, no work
    int row_step;
};

constexpr int HWA_ROW = 1024, HWB_ROW = 512;
constexpr int HWA_BYTES = 32 * HWA_ROW;          // 32768
constexpr int HWB_BYTES = 32 * HWB_ROW;          // 16384
constexpr int HWSTAGE = HWA_BYTES + HWB_BYTES;   // 49152

typedef short short4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f16x8 tr_pair_h(const unsigned char* base, int off0, int off1) {
    const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + off0));
    const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)(base + off1));
    typedef short short8v __attribute__((ext_vector_type(8)));
    short8v v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}

__device__ __forceinline__ int rot4(int q) { return 2 * (q & 1) + 8 * (q >> 1); }
// 16x16x32 layout of hw_kernel: the stored position of source chunk j of pixel row r is  perm(j) ^ swz16(r), where perm moves the
// channel-tile index i to the TOP bits of the position (so the four tiles of a wave sit 256 B apart and are reached through the
// instruction's immediate offset: 2 + 4 address registers instead of 32) and swz16(r) in {0,2,..,14} only touches bits 1-3: the
// 4 rows x 2 k-groups of a 32-lane half still cover all eight 32-byte slots of the bank row (SQ_LDS_BANK_CONFLICT = 0).
__device__ __forceinline__ int swz16(int r) { return rot4(r & 3) + 4 * ((r >> 3) & 1); }
__device__ __forceinline__ int permA16(int j) { return ((j & 0x0c) << 2) | ((j & 0x30) >> 2) | (j & 3); }            // [wm|i|pl|p] <-> [i|wm|pl|p], involution
__device__ __forceinline__ int permB16(int j) { return ((j & 0x0c) << 1) | ((j & 0x10) >> 2) | (j & 3); }            // [wn|i1 i0|pl|p] -> [i1 i0|wn|pl|p]
__device__ __forceinline__ int unpermB16(int x) { return ((x & 0x18) >> 1) | ((x & 0x04) << 2) | (x & 3); }

// Schedule: h2_kernel's ping-pong halves with one barrier per K-tile, fragment reads ahead of the LDS-DMA issue block, no s_setprio
// (measured best here); v_mfma_f32_16x16x32_f16, one 32-pixel k-block per K-tile, 4x4 tiles of 16x16 per wave.  The two 16-lane
// groups of a 32-lane half read pixel rows 8 apart at the SAME channels, so the stored rotation of pixel row r is
// rot4(r&3) + 4*((r>>3)&1) chunks: the 4 rows x 2 k-groups of a half cover all 8 32-byte slots of the bank row.
// DBG (timing library only; wrong results): 1 = no global loads, 2 = no MFMAs, 3 = MFMAs only (no loads, no LDS reads, no barriers),
// 5 = LDS-DMA loads only, 6 = fragment reads only, 7 / 8 = only the dY / only the X loads (+ everything else), 9 = no fragment reads
template <int NPROD, int DBG = 0>
__global__ __launch_bounds__(512, 2) void hw_kernel(HWArgs p) {
    // DBG 5: LDS-DMA loads only (no fragment reads, no MFMAs; barriers kept); DBG 6: fragment reads only (no loads, no MFMAs)
    // DBG 7: dY (A) loads only + everything else; DBG 8: X (B) loads only + everything else; DBG 9: loads + MFMAs, no fragment reads
    constexpr bool do_load = DBG != 1 && DBG != 3 && DBG != 6, do_mma = DBG != 2 && DBG != 5 && DBG != 6, do_lds = DBG != 3 && DBG != 5 && DBG != 9;
    constexpr bool load_a = DBG != 8, load_b = DBG != 7;
    constexpr bool do_bar = DBG != 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int tn, tmi, split;
    {
        const int lid = xcd_remap(blockIdx.x, gridDim.x);
        tn = lid % p.tiles_n;
        tmi = lid / p.tiles_n;
        split = blockIdx.y;
    }
    const float sx_dev = p.sx_vec ? 1.f : p.sx[0], sy_dev = p.sy_vec ? 1.f : p.sy[0];      // requested first, used by the epilogue (see h2_kernel)
    const int co0 = tmi * 256, n0 = tn * 128;
    // Narrow weight gradients (Co <= 128 on the 256-row tile: the encoder's layer-1 / layer-2 convs): the waves whose 64 output rows lie
    // beyond Co multiplied zero blocks.  They now skip their fragment reads and MFMAs (scalar branch; their accumulators stay zero and are
    // never stored).  Waves w and w + 4 share a SIMD and w >> 1 is the row group, so with Co <= 128 every SIMD keeps ONE multiplying wave
    // instead of two: the matrix time of the tile halves (round 6; the idle waves still issue their LDS-DMA pieces and join the barriers).
    const bool rows_live = co0 + wm * 64 < p.Co;
    if (p.batched && p.row_last != nullptr && p.row_last[split] < p.row_step) {      // (scalar) this sample's dY is exactly zero
        if (!p.beta) {
            float* o = p.out + (int64_t)split * p.slab_stride;
            for (int i = t; i < 256 * 128; i += 512) {
                const int co = co0 + i / 128, n = n0 + i % 128;
                if (co < p.Co && n < p.Nvalid) o[(int64_t)co * p.ldo + n] = 0.f;
            }
        }
        return;
    }
    const int64_t m_begin = (int64_t)split * p.rows_per_split;
    const int64_t m_end = min(p.M, m_begin + p.rows_per_split);
    const int nkt = (int)((m_end - m_begin + 31) / 32);
    const int HoWo = p.Ho * p.Wo;

    uint32_t a_voff[4];
    int a_r[4];
    bool a_cok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = t + 512 * j;
        const int r = g >> 6, pos = g & 63;
        const int js = permA16(pos ^ swz16(r));      // which source chunk lands at this position
        a_r[j] = r;
        a_cok[j] = co0 + (js >> 2) * 16 < p.Co;
        a_voff[j] = (uint32_t)r * (uint32_t)(4 * p.ldy) + (uint32_t)(co0 * 4 + js * 16);    // bytes relative to pixel mt
    }
    // a 128-column tile of (tap, ci) may span several filter taps when Ci < 128: every 16-channel chunk group lies inside one
    // tap (Ci % 16 == 0), so the tap is a per-lane constant
    int b_r[2], b_c8[2], b_b[2], b_y[2], b_x[2], b_dy[2], b_dx[2];
    bool b_live[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g = t + 512 * j;
        const int r = g >> 5, pos = g & 31;
        const int js = unpermB16(pos ^ swz16(r));
        const int col = n0 + (js >> 2) * 16;                      // first column of this lane's 16-channel group
        const int tap = col / p.Ci, ci = col - tap * p.Ci;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        b_dy[j] = ky * p.dil - p.pad;
        b_dx[j] = kx * p.dil - p.pad;
        b_r[j] = r;
        b_c8[j] = ci * 4 + (js & 3) * 16;
        b_live[j] = col < p.Ntot;
        const uint32_t m = (uint32_t)(m_begin + r);               // coordinates of this lane's pixel in K-tile 0 (M < 2^31: 32-bit divisions)
        const uint32_t b = m / (uint32_t)HoWo;
        const uint32_t rem = m - b * (uint32_t)HoWo;
        b_b[j] = (int)b;
        b_y[j] = (int)(rem / (uint32_t)p.Wo);
        b_x[j] = (int)rem - b_y[j] * p.Wo;
    }
    const int y_adv = 32 / p.Wo, x_adv = 32 - y_adv * p.Wo;      // advancing 32 output pixels = y_adv rows + x_adv columns
    const uint32_t xrow = (uint32_t)(4 * p.Ci);
    // incremental form of the input pixel of this lane: (iy, ix) and its byte offset advance by scalar constants per K-tile
    int b_iy[2], b_ix[2];
    uint32_t b_off[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        b_iy[j] = b_y[j] * p.stride + b_dy[j];
        b_ix[j] = b_x[j] * p.stride + b_dx[j];
        b_off[j] = (uint32_t)((b_b[j] * p.Hi + b_iy[j]) * p.Wi + b_ix[j]) * xrow + (uint32_t)b_c8[j];
    }
    const bool tiny_map = HoWo < 32;
    const int adv_ix = x_adv * p.stride, adv_iy = y_adv * p.stride;
    const uint32_t adv_off = (uint32_t)(adv_iy * p.Wi + adv_ix) * xrow;
    const int wrapx_ix = p.Wo * p.stride;                                            // x wrapped: ix -= Wo*stride, iy += stride
    const uint32_t wrapx_off = (uint32_t)(p.stride * p.Wi - p.Wo * p.stride) * xrow;
    const int wrapy_iy = p.Ho * p.stride;                                            // y wrapped into the next image
    const uint32_t wrapy_off = (uint32_t)(p.Hi * p.Wi - p.Ho * p.stride * p.Wi) * xrow;
    int ld_kt = 0;
    auto issue_tile = [&](int stage) {
        if constexpr (!do_load) return;
        unsigned char* st = smem + stage * HWSTAGE;
        const int64_t mt = m_begin + (int64_t)ld_kt * 32;
        const unsigned char* baseA = reinterpret_cast<const unsigned char*>(p.dY) + mt * (4 * (int64_t)p.ldy);   // scalar
        const uint32_t zrelA = (uint32_t)((int64_t)p.y_bytes - mt * (4 * (int64_t)p.ldy));
        const int rows_left = (int)min((int64_t)32, m_end - mt);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = a_cok[j] && a_r[j] < rows_left;
            if constexpr (load_a) SP_GLDS16(baseA + (ok ? a_voff[j] : zrelA), st + (wave + 8 * j) * 1024);
        }
        const unsigned char* baseB = reinterpret_cast<const unsigned char*>(p.X);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = b_live[j] && b_r[j] < rows_left && (unsigned)b_iy[j] < (unsigned)p.Hi && (unsigned)b_ix[j] < (unsigned)p.Wi;
            if constexpr (load_b) SP_GLDS16(baseB + (ok ? b_off[j] : p.x_bytes), st + HWA_BYTES + (wave + 8 * j) * 1024);
            // advance this lane's pixel by 32 for the next K-tile (no divisions, no multiplications)
            b_x[j] += x_adv;
            b_y[j] += y_adv;
            b_ix[j] += adv_ix;
            b_iy[j] += adv_iy;
            b_off[j] += adv_off;
            if (b_x[j] >= p.Wo) {
                b_x[j] -= p.Wo;
                ++b_y[j];
                b_ix[j] -= wrapx_ix;
                b_iy[j] += p.stride;
                b_off[j] += wrapx_off;
            }
            if (b_y[j] >= p.Ho) {           // maps of >= 32 pixels cross at most one image boundary per K-tile
                b_y[j] -= p.Ho;
                b_iy[j] -= wrapy_iy;
                b_off[j] += wrapy_off;
            }
            if (tiny_map)                   // scalar condition: pure-GEMM uses of the kernel (Ho*Wo < 32)
                while (b_y[j] >= p.Ho) {
                    b_y[j] -= p.Ho;
                    b_iy[j] -= wrapy_iy;
                    b_off[j] += wrapy_off;
                }
        }
        ++ld_kt;
    };

    // transposed-read offsets: the 16-lane group kg = lane >> 4 is the k-group; lane (q = (lane>>2)&3, pp = lane&3) addresses pixel
    // row 8kg + 4s + q, channels cbase + 4pp .. +3 of the 16-channel tile.
    const int q = (lane >> 2) & 3, pp = lane & 3, kg = lane >> 4;
    constexpr int NT = 4;
    int offA[NT][2][2], offB[NT][2][2];     // [i][plane][s]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // base registers only for i < 2 (A: i = 0; B: i = 0, 1) and s2 = 0; the other tiles / the +4-row group are
                // constant byte offsets from them (A: i * 256, B: (i >> 1) * 256, s2: 4 rows), see swz16
                const int row = 8 * kg + q;
                const int pa = (((wm << 2) | (pl << 1) | (pp >> 1)) ^ swz16(row));
                offA[i][pl][s2] = row * HWA_ROW + pa * 16 + (pp & 1) * 8 + i * 256 + s2 * 4 * HWA_ROW;
                const int pb = ((((i & 1) << 3) | (wn << 2) | (pl << 1) | (pp >> 1)) ^ swz16(row));
                offB[i][pl][s2] = HWA_BYTES + row * HWB_ROW + pb * 16 + (pp & 1) * 8 + (i >> 1) * 256 + s2 * 4 * HWB_ROW;
            }

    f32x4 tot4[4][4], acc4[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tot4[i][j][r] = 0.f;
                acc4[i][j][r] = 0.f;
            }

    {
        int issued = 0;
        for (; issued < HNSTAGE - 1 && issued < nkt; ++issued) issue_tile(issued);
        if (issued >= 2 && DBG == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (issued >= 2 && DBG == 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (issued >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    f16x8 af[2][2][2], bf[2][2][2];      // [kk][i][plane]
    auto read_group = [&](int stage_, int kk) {
        const unsigned char* st = smem + stage_ * HWSTAGE;
        if (!rows_live) return;
        if constexpr (!do_lds) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    f16x8 z;
#pragma unroll
                    for (int e = 0; e < 8; ++e) z[e] = (_Float16)(float)(stage_ + e + lane);
                    af[kk][i][pl] = z;
                    bf[kk][i][pl] = z;
                }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int pl = 0; pl < (NPROD == 3 ? 2 : 1); ++pl) {
                // group kk = 16-channel tiles 2kk, 2kk+1 of the ONE 32-pixel block
                af[kk][i][pl] = tr_pair_h(st, offA[(2 * kk + i) % NT][pl][0], offA[(2 * kk + i) % NT][pl][1]);
                bf[kk][i][pl] = tr_pair_h(st, offB[(2 * kk + i) % NT][pl][0], offB[(2 * kk + i) % NT][pl][1]);
            }
    };
    auto mma_group = [&](int kk) {
        if constexpr (!do_mma) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    const f16x8 x0 = af[kk][i][pl], x1 = bf[kk][i][pl];
                    asm volatile("" ::"v"(x0), "v"(x1));
                }
            return;
        }
        if (!rows_live) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4& a4 = acc4[2 * kk + i][j];
                if constexpr (NPROD == 3) {
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk][i][0], bf[j >> 1][j & 1][1], a4, 0, 0, 0);
                    a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk][i][1], bf[j >> 1][j & 1][0], a4, 0, 0, 0);
                }
                a4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[kk][i][0], bf[j >> 1][j & 1][0], a4, 0, 0, 0);
            }
    };
    auto fold = [&](int kt_) {
        if ((kt_ & (HCHUNK_KT - 1)) == HCHUNK_KT - 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    tot4[i][j] += acc4[i][j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc4[i][j][r] = 0.f;
                }
        }
    };
    auto wait_barrier = [&](int kt_) {
        if (kt_ + 2 < nkt && DBG == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (kt_ + 2 < nkt && DBG == 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (kt_ + 2 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (do_bar) __builtin_amdgcn_s_barrier();
    };
    auto prev_stage = [](int st_) { return st_ == 0 ? HNSTAGE - 1 : st_ - 1; };

    int stage = 0;
    const bool late = wave >= 4;
    if (!late) {
        for (int kt = 0; kt < nkt; ++kt) {
            const bool pre = kt + HNSTAGE - 1 < nkt;
            read_group(stage, 0);
            read_group(stage, 1);
            if (pre) issue_tile(prev_stage(stage));      // fragment reads ahead of the LDS-DMA issue block
            mma_group(0);
            mma_group(1);
            fold(kt);
            wait_barrier(kt);
            stage = (stage == HNSTAGE - 1) ? 0 : stage + 1;
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            const bool pre = kt + HNSTAGE - 1 < nkt;
            if (kt > 0) {
                mma_group(0);
                mma_group(1);
                fold(kt - 1);
            }
            read_group(stage, 0);
            read_group(stage, 1);
            if (pre) issue_tile(prev_stage(stage));
            wait_barrier(kt);
            stage = (stage == HNSTAGE - 1) ? 0 : stage + 1;
        }
        if (nkt > 0) {
            mma_group(0);
            mma_group(1);
            fold(nkt - 1);
        }
    }

    float* out = p.out + (p.splits > 1 ? (int64_t)split * p.slab_stride : 0);
    const bool direct = p.splits == 1 || p.batched;      // the result itself (scaled, beta) rather than a raw slab for the reduce pass
    const float isx = 1.f / sx_dev, isy = 1.f / sy_dev;
    {
        const int l16 = lane & 15;
        // per-channel operand scales (sx_vec / sy_vec): output row co carries 1/sy[co], output column n = (tap, ci) carries 1/sx[ci]
        float isyv[4][4], isxv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * 64 + i * 16 + 4 * kg + r;
                isyv[i][r] = (direct && p.sy_vec && co < p.Co) ? 1.f / p.sy[co] : isy;
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + l16;
            isxv[j] = (direct && p.sx_vec && n < p.Nvalid) ? 1.f / p.sx[n % p.Ci] : isx;
        }
        // float4 stores through a wave-private LDS staging tile (see h2_kernel's epilogue): 256-byte runs instead of 64-byte ones
        const bool wide = (p.ldo & 3) == 0 && (p.Nvalid & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
        if (wide) {
            float* stg = reinterpret_cast<float*>(smem) + wave * (64 * 68);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = tot4[i][j][r] + acc4[i][j][r];
                        stg[(i * 16 + 4 * kg + r) * 68 + j * 16 + l16] = direct ? p.alpha * ((v * isxv[j]) * isyv[i][r]) : v;
                    }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int cq4 = lane & 15, rsub = lane >> 4;
            const int n = n0 + wn * 64 + 4 * cq4;
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                const int row = ps * 4 + rsub;
                const int co = co0 + wm * 64 + row;
                if (co < p.Co && n < p.Nvalid) {
                    float4 v = *reinterpret_cast<const float4*>(stg + row * 68 + 4 * cq4);
                    float4* dst = reinterpret_cast<float4*>(out + (int64_t)co * p.ldo + n);
                    if (direct && p.beta) {
                        const float4 o = *dst;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *dst = v;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + l16;
            if (n >= p.Nvalid) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wm * 64 + i * 16 + 4 * kg + r;
                    if (co < p.Co) {
                        float* dst = out + (int64_t)co * p.ldo + n;
                        const float v = tot4[i][j][r] + acc4[i][j][r];
                        if (direct) {
                            float w = p.alpha * ((v * isxv[j]) * isyv[i][r]);
                            if (p.beta) w += *dst;
                            *dst = w;
                        } else {
                            *dst = v;
                        }
                    }
                }
        }
    }
}

// Slab reduce.  The slabs are RAW accumulator sums [nseg][splits][Co][ldo]; segment g (one application of a weight that is applied
// several times, e.g. one decode step of the h-gate conv) has its own operand scales sx[g] / sy[g], each either one device scalar or a
// per-channel vector (x: [Ci], output column n = (tap, ci); y: [Co], output row).  Fixed summation order: deterministic.
constexpr int HW_MAXSEG = 16;
struct HWScales {
    const float* sx[HW_MAXSEG];
    const float* sy[HW_MAXSEG];
    int sx_vec, sy_vec, nseg, Ci;
};

__global__ void hw_reduce_kernel(const float* slab, float* out, int Co, int Ntot, int ldo, int splits, int64_t slab_stride,
                                 float alpha, HWScales sc, int beta) {
    const int64_t total = (int64_t)Co * Ntot;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i / Ntot), n = (int)(i - (int64_t)co * Ntot);
        const int64_t off = (int64_t)co * ldo + n;
        const int ci = n % sc.Ci;
        float acc = 0.f;
        for (int g = 0; g < sc.nseg; ++g) {
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += slab[(int64_t)(g * splits + k) * slab_stride + off];
            const float isx = 1.f / sc.sx[g][sc.sx_vec ? ci : 0], isy = 1.f / sc.sy[g][sc.sy_vec ? co : 0];
            acc += (s * isx) * isy;
        }
        acc *= alpha;
        if (beta) acc += out[off];
        out[off] = acc;
    }
}

// float4 form (Ntot, ldo, Ci and the slab stride multiples of 4; 16-byte aligned bases): one thread per 4 consecutive columns
__global__ void hw_reduce4_kernel(const float* slab, float* out, int Co, int Ntot4, int ldo4, int splits, int64_t slab_stride4,
                                  float alpha, HWScales sc, int beta) {
    const int64_t total = (int64_t)Co * Ntot4;
    const float4* S4 = reinterpret_cast<const float4*>(slab);
    float4* O4 = reinterpret_cast<float4*>(out);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i / Ntot4), n4 = (int)(i - (int64_t)co * Ntot4);
        const int64_t off = (int64_t)co * ldo4 + n4;
        const int ci = (n4 * 4) % sc.Ci;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < sc.nseg; ++g) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < splits; ++k) {               // same per-element order as the scalar form
                const float4 v = S4[(int64_t)(g * splits + k) * slab_stride4 + off];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const float isy = 1.f / sc.sy[g][sc.sy_vec ? co : 0];
            float4 ix;
            if (sc.sx_vec) {
                const float4 q = *reinterpret_cast<const float4*>(sc.sx[g] + ci);
                ix = make_float4(1.f / q.x, 1.f / q.y, 1.f / q.z, 1.f / q.w);
            } else {
                const float q = 1.f / sc.sx[g][0];
                ix = make_float4(q, q, q, q);
            }
            acc.x += (s.x * ix.x) * isy; acc.y += (s.y * ix.y) * isy; acc.z += (s.z * ix.z) * isy; acc.w += (s.w * ix.w) * isy;
        }
        acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
        if (beta) {
            const float4 o = O4[off];
            acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        O4[off] = acc;
    }
}

// ================================================================================================================
// hw2_kernel (round 4): the weight gradient with a LARGER wave tile, for a weight that is applied several times (the h-gate conv of the
// ConvLSTM: T - 1 applications per training step, AiR/models/baseline_attention.py:37-56) -- all applications ("segments") in ONE
// launch at the end of backpropagation through time instead of one launch per decode step.
//   block 256 co x 256 n x 32 pixels, 8 waves, wave tile 64 co x 128 n (4 x 8 tiles of 16x16), SINGLE-level fp32 accumulation
//   (128 accumulator VGPRs; the pixel range of a workgroup bounds the chain), 2-stage ring of 64 KB (one K-tile in flight),
//   48 ds_read_b64_tr_b16 + 96 MFMA per wave and K-tile (hw_kernel: 32 + 48: -25 % LDS read bytes per MFMA) and 8 LDS-DMA pieces per
//   lane and K-tile for 768 MFMAs per workgroup (hw_kernel: 6 for 384: -33 % L2 -> LDS bytes per MFMA).
// tools/probes/hw2_probe.hip (profiles/r04_hw2_probe.log), random operands, per application of the h-gate shape: this loop 3.0-3.2 ms
// against 3.5-3.7 ms for hw_kernel's; ONE application alone (576 workgroups = 2.25 rounds of 256 CUs) 3.77 ms -- the large tile only
// pays with enough workgroups, hence the deferred launch: grid = (144 tiles, nseg * splits).
// Ping-pong halves with one barrier per K-tile as hw_kernel; the late half issues its loads FIRST (one K-tile in flight: the loads
// need the whole matrix segment to land), the early half between its two B-fragment halves.
// Needs: stride 1, Wo % 32 == 0 (a 32-pixel K-tile is part of ONE image row: image, row and first column are scalars), Co % 256 == 0,
// KH*KW*Ci % 256 == 0, Ci % 16 == 0; every segment the same geometry.  Output: RAW slabs [nseg * splits][Co][ldo] for hw_reduce.
struct HW2Args {
    const uint16_t* X[HW_MAXSEG];
    const uint16_t* dY[HW_MAXSEG];
    float* slab;
    int Hi, Wi, Ci, Ho, Wo, Co, ldy;
    int KH, KW, pad, dil;
    int Ntot, ldo, tiles_n, splits;
    int rows_per_split;            // pixels per split (a multiple of 32), the last split of a segment may be shorter
    int M;                         // pixels per segment
    int64_t slab_stride;
    uint32_t x_bytes;
    // Row sparsity of the output gradient (see H2Args): segment g is decode step seg_step[g]; the pixels of a sample with
    // row_last[img] < seg_step[g] carry an exactly-zero dY and are skipped by the K loop (whole samples: Ho * Wo % 32 == 0).
    const int* row_last;
    int seg_step[HW_MAXSEG];
};
constexpr int HW2_ROW = 1024, HW2_OP = 32 * HW2_ROW, HW2_STAGE = 2 * HW2_OP;      // 65536 per stage
constexpr int HW2_LDS = 8 * 32 * 132 * 4;                                        // 135168: the epilogue's staging > the 2-stage ring
// stored position of source chunk j of a B row: [n-tile pair i>>1 | i&1 | wn | plane | half] (tiles reached by immediate offsets)
__device__ __forceinline__ int unpermB2(int x) { return (((x >> 2) & 1) << 5) | ((((x >> 4) << 1) | ((x >> 3) & 1)) << 2) | (x & 3); }

__global__ __launch_bounds__(512, 2) void hw2_kernel(HW2Args p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = lid % p.tiles_n, tmi = lid / p.tiles_n;
    const int seg = (int)blockIdx.y / p.splits, split = (int)blockIdx.y - seg * p.splits;
    const unsigned char* Xp = reinterpret_cast<const unsigned char*>(p.X[seg]);
    const unsigned char* Yp = reinterpret_cast<const unsigned char*>(p.dY[seg]);
    const int co0 = tmi * 256, n0 = tn * 256;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int PP = p.Ho * p.Wo;                                   // pixels per sample
    // samples of this pixel range whose dY rows are exactly zero at this decode step are skipped (bit b of `skip` = sample b_first + b)
    const int b_first = m_begin / PP;
    uint32_t skip = 0;
    int nkt = (m_end - m_begin) >> 5;
    if (p.row_last != nullptr) {
        const int step = p.seg_step[seg];
        const int b_last = (m_end - 1) / PP;
        if (b_last - b_first < 32) {
            nkt = 0;
            for (int b = b_first; b <= b_last; ++b) {
                const int lo = max(m_begin, b * PP), hi = min(m_end, (b + 1) * PP);
                if (p.row_last[b] < step) skip |= 1u << (b - b_first);
                else nkt += (hi - lo) >> 5;
            }
        }
    }

    // loader: LDS piece (wave + 8 j) = pixel row r = wave + 8 j of the K-tile, lane = 16-byte chunk position 0..63 of its 1 KB
    uint32_t a_voff[4], b_rel[4];
    int b_cxy[4];                       // (dy << 16) | (cx & 0xffff): input row offset of the lane's tap, column offset r + dx
    const uint32_t xrow = (uint32_t)(4 * p.Ci);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wave + 8 * j, pos = lane;
        const int ja = permA16(pos ^ swz16(r));
        a_voff[j] = (uint32_t)r * (uint32_t)(4 * p.ldy) + (uint32_t)(co0 * 4 + ja * 16);
        const int jb = unpermB2(pos ^ swz16(r));
        const int col = n0 + (jb >> 2) * 16;                      // first column of this lane's 16-channel group (inside one tap)
        const int tap = col / p.Ci, ci = col - tap * p.Ci;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int dy = ky * p.dil - p.pad, cx = r + kx * p.dil - p.pad;
        b_cxy[j] = (int)(((unsigned)dy << 16) | ((unsigned)cx & 0xffffu));
        b_rel[j] = (uint32_t)(dy * p.Wi + cx) * xrow + (uint32_t)(ci * 4 + (jb & 3) * 16);      // wrapping arithmetic, added to the tile's base
    }
    // scalar pixel state of the next K-tile to load: image b, output row y, first column x0 (stride 1: input pixel = output pixel + tap)
    int ld_b = m_begin / (p.Ho * p.Wo);
    int ld_y = (m_begin - ld_b * p.Ho * p.Wo) / p.Wo;
    int ld_x0 = m_begin - (ld_b * p.Ho + ld_y) * p.Wo;
    int ld_m = m_begin;
    auto skip_samples = [&]() {          // move the loader to the first pixel of the next sample that is not skipped (scalar)
        while (skip && ld_m < m_end && ((skip >> (ld_b - b_first)) & 1u)) {
            ++ld_b;
            ld_y = 0;
            ld_x0 = 0;
            ld_m = ld_b * PP;
        }
    };
    skip_samples();
    auto issue_tile = [&](int stage) {
        unsigned char* st = smem + stage * HW2_STAGE;
        const unsigned char* baseA = Yp + (int64_t)ld_m * (4 * (int64_t)p.ldy);                          // scalar
#pragma unroll
        for (int j = 0; j < 4; ++j) SP_GLDS16(baseA + a_voff[j], st + (wave + 8 * j) * 1024);
        const uint32_t pix0 = (uint32_t)((ld_b * p.Hi + ld_y) * p.Wi + ld_x0) * xrow;                    // scalar
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iy = ld_y + (b_cxy[j] >> 16), ix = ld_x0 + (int)(short)(b_cxy[j] & 0xffff);
            const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            SP_GLDS16(Xp + (ok ? pix0 + b_rel[j] : p.x_bytes), st + HW2_OP + (wave + 8 * j) * 1024);
        }
        ld_m += 32;
        ld_x0 += 32;
        if (ld_x0 >= p.Wo) {
            ld_x0 = 0;
            if (++ld_y == p.Ho) {
                ld_y = 0;
                ++ld_b;
                skip_samples();
            }
        }
    };

    // transposed-read offsets (see hw_kernel): lane (kg = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3) addresses pixel row 8 kg + 4 s + q
    const int q = (lane >> 2) & 3, pp = lane & 3, kg = lane >> 4;
    int offA[2], offB[2][2];            // [plane], [i & 1][plane]; tile i: A + i * 256, B + (i >> 1) * 256; s: + 4 rows
    {
        const int row = 8 * kg + q;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            offA[pl] = row * HW2_ROW + ((((wm << 2) | (pl << 1) | (pp >> 1)) ^ swz16(row)) * 16) + (pp & 1) * 8;
#pragma unroll
            for (int i1 = 0; i1 < 2; ++i1)
                offB[i1][pl] = HW2_OP + row * HW2_ROW + ((((i1 << 3) | (wn << 2) | (pl << 1) | (pp >> 1)) ^ swz16(row)) * 16) + (pp & 1) * 8;
        }
    }
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    f16x8 af[4][2], bf[8][2];
    auto rdA = [&](int stage) {
        const unsigned char* st = smem + stage * HW2_STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) af[i][pl] = tr_pair_h(st + i * 256, offA[pl], offA[pl] + 4 * HW2_ROW);
    };
    auto rdB = [&](int stage, int j0, int j1) {
        const unsigned char* st = smem + stage * HW2_STAGE;
#pragma unroll
        for (int j = j0; j < j1; ++j)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bf[j][pl] = tr_pair_h(st + (j >> 1) * 256, offB[j & 1][pl], offB[j & 1][pl] + 4 * HW2_ROW);
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
    auto wait_all = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    if (nkt > 0) issue_tile(0);
    wait_all();
    const bool late = wave >= 4;
    if (!late) {
        for (int kt = 0; kt < nkt; ++kt) {
            const int st = kt & 1;
            rdA(st);
            rdB(st, 0, 4);
            if (kt + 1 < nkt) issue_tile(st ^ 1);
            rdB(st, 4, 8);
            mm(0, 4);
            mm(4, 8);
            wait_all();
        }
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            const int st = kt & 1;
            if (kt + 1 < nkt) issue_tile(st ^ 1);
            if (kt > 0) mm(0, 8);
            rdA(st);
            rdB(st, 0, 8);
            wait_all();
        }
        if (nkt > 0) mm(0, 8);
    }
    // raw slab tile: the wave's 64 x 128 result through a private 16.9 KB staging slice (row pitch 132 floats: conflict-free both ways),
    // two passes of 32 rows, float4 stores, 512 contiguous bytes per row.  (Every LDS read of the K loop was waited for before its
    // last barrier, the late half multiplies from registers: the ring is free.)
    {
        float* sg = reinterpret_cast<float*>(smem) + wave * (32 * 132);
        const int l16 = lane & 15;
        float* dst0 = p.slab + (int64_t)blockIdx.y * p.slab_stride + (int64_t)(co0 + wm * 64) * p.ldo + n0 + wn * 128;
        const int c4 = lane & 31, rsub = lane >> 5;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sg[(i2 * 16 + 4 * kg + r) * 132 + j * 16 + l16] = acc[half * 2 + i2][j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ps = 0; ps < 16; ++ps) {
                const int row = ps * 2 + rsub;
                *reinterpret_cast<float4*>(dst0 + (int64_t)(half * 32 + row) * p.ldo + 4 * c4) = *reinterpret_cast<const float4*>(sg + row * 132 + 4 * c4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

int hw_splits(const sp_wgrad_desc* d) {
    const int64_t M = (int64_t)d->N_img * d->Ho * d->Wo;
    const int forced = sp_tuning_get(SP_TUNE_HW_SPLITS, 0);      // timing build only: < 256 = split count, >= 256 = workgroup target
    if (forced > 0 && forced < 256) return (int)std::min<int64_t>(forced, std::max<int64_t>(1, M / 32));
    const int64_t tiles = sp_cdiv(d->Co, 256) * sp_cdiv((int64_t)d->KH * d->KW * d->Ci, 128);
    if (forced >= 256 || sp_tuning_get(SP_TUNE_HW_CAP, 0) == 1) {      // round-3 rule (timing build: A/B against the cost rule below)
        int64_t want = sp_cdiv(forced >= 256 ? forced : 2048, tiles); // 1 workgroup per CU: aim for >= 8 rounds of 256
        want = std::min<int64_t>(want, std::max<int64_t>(1, M / 1024));   // >= 32 K-tiles per split
        want = std::min<int64_t>(want, 64);
        return (int)std::max<int64_t>(1, want);
    }
    // One workgroup per CU: the launch lasts  rounds x (K-tiles of a split + a workgroup's fixed cost),  rounds = ceil(tiles x splits /
    // 256).  The old rule (>= 2048 workgroups, at most 64 splits) ignored the round quantisation that decides the few-tile shapes of
    // the encoder: 5 tiles x 64 splits = 320 workgroups = 2 rounds of 160 K-tiles where 51 splits = 255 workgroups run ONE round of 201
    // (measured on the 64 x 576 weight gradient at 80x128 maps: 393 us with 64 splits, 326 us with 128; tools/encoder_census.py under
    // SP_LIBRARY=timing SP_HW_CAP).  Fixed cost of a workgroup (dispatch, prologue, 128 KB slab tile) ~ 12 K-tiles of this kernel; the
    // slab reduce reads `splits` slabs: half a K-tile per split and round.  Large shapes keep their 8 splits under this rule.
    const int64_t smax = std::min<int64_t>(128, std::max<int64_t>(1, M / 1024));       // >= 32 K-tiles per split
    int64_t best = 1;
    double best_cost = 1e30;
    for (int64_t sp = 1; sp <= smax; ++sp) {
        const double rounds = (double)sp_cdiv(tiles * sp, 256);
        const double cost = rounds * ((double)sp_cdiv(sp_cdiv(M, sp), 32) + 12.0 + 0.5 * (double)sp);
        if (cost < best_cost * 0.999) {
            best_cost = cost;
            best = sp;
        }
    }
    return (int)best;
}

// ---- scale + split kernels ------------------------------------------------------------------------------------
// amax over a tensor: atomicMax on the bit pattern of |x| (non-negative floats order like unsigned ints) -- exact, order-free.
__global__ __launch_bounds__(256) void amax_kernel(const float* x, int64_t n4, int64_t n, unsigned* out) {
    float m = 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (; i + 3 * stride < n4; i += 4 * stride) {          // 4 independent 16-byte loads in flight per thread
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        const float ma = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
        const float mb = fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)));
        const float mc = fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w)));
        const float md = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
        m = fmaxf(m, fmaxf(fmaxf(ma, mb), fmaxf(mc, md)));
    }
    for (; i < n4; i += stride) {
        const float4 v = x4[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0)
        for (int64_t j = n4 * 4 + threadIdx.x; j < n; j += blockDim.x) m = fmaxf(m, fabsf(x[j]));
    // one atomic per block: same-address atomics serialise at the L2 (32k of them cost 0.3 ms)
    __shared__ float sh4[4];
    m = block_max_256(m, sh4);
    if (threadIdx.x == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

// x fp32 [rows][K] (K % 16 == 0)  ->  [rows][K/16][2][16] fp16.  One thread per 4 consecutive k.
__global__ __launch_bounds__(256) void split2_kernel(const float* x, int64_t n4, const unsigned* amax, uint16_t* out,
                                                     float* scale_out) {
    const float s = scale_of(*amax);
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n4; base += stride) {      // wave-uniform
        const int64_t i = base + lane;
        const bool live = i < n4;
        const float4 v = live ? reinterpret_cast<const float4*>(x)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        ushort4 a, b;
        split2(v.x, s, a.x, b.x);
        split2(v.y, s, a.y, b.y);
        split2(v.z, s, a.z, b.z);
        split2(v.w, s, a.w, b.w);
        store_planes_quad(out, i, live, a, b);
    }
    if (blockIdx.x == 0 && threadIdx.x < 8)       // 64-byte zero block right after the data
        reinterpret_cast<uint2*>(out + 8 * n4)[threadIdx.x] = make_uint2(0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
}

// w [Co][taps][Ci] fp32 -> rows n = ci, k = (tap, co):  [Ci][taps*Co/16][2][16]  (the K-contiguous B operand of dgrad)
__global__ __launch_bounds__(256) void split2_wT_kernel(const float* w, int Co, int taps, int Ci, const unsigned* amax,
                                                        uint16_t* out, float* scale_out) {
    const float s = scale_of(*amax);
    const int64_t total = (int64_t)Co * taps * Ci;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Co);
        int64_t r = i / Co;
        const int tap = (int)(r % taps);
        const int ci = (int)(r / taps);
        uint16_t a, b;
        split2(w[((int64_t)co * taps + tap) * Ci + ci], s, a, b);
        const int64_t k = (int64_t)tap * Co + co;
        uint16_t* o = out + ((int64_t)ci * ((int64_t)taps * Co / 16) + (k >> 4)) * 32 + (k & 15);
        o[0] = a;
        o[16] = b;
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) reinterpret_cast<uint2*>(out + 2 * total)[threadIdx.x] = make_uint2(0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = s;
}


// ---- scale VECTORS (round 4) -------------------------------------------------------------------------------
// The per-tensor scale gives an operand 22 bits only within 2^-17 of its maximum (fp16's exponent range); a weight row, a gradient
// channel or an activation channel far below the tensor's maximum lost relative precision, and an OUTPUT row / column that receives
// all of its contributions from such a slice (forward: output channel <- weight row; data gradient: input channel <- weight column;
// weight gradient: dW row <- dY channel, dW column <- X channel) carried that error undiluted.  Power-of-two scales along a dimension
// that is NOT contracted factor out exactly, and a scale along the contracted channel dimension of an ACTIVATION is absorbed exactly by
// the weight operand it meets (x * s_c  times  w / s_c):
//   weights      [rows][K]   : one scale per row (= output column of the GEMM); optional `absorb` [Kc]: w[r][tap][c] / absorb[c]
//   activations  [rows][C]   : one scale per CHANNEL (split2_cols): exact for the weight gradient (K = pixels); forward / data
//                              gradient contract over channels -> the weight operand is split with absorb = that vector.
// one block per weight row: row maximum, then the split (the row is re-read from L2)
__global__ __launch_bounds__(256) void split2_rows_kernel(const float* w, int64_t K, int Kc, const float* absorb, uint16_t* out,
                                                          float* row_scale, int64_t rows) {
    __shared__ float sh4[4];
    const int64_t r = blockIdx.x;
    const float4* w4 = reinterpret_cast<const float4*>(w + r * K);
    const int64_t K4 = K / 4;
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < K4; i += 256) {
        float4 v = w4[i];
        if (absorb) {
            const float4 a = *reinterpret_cast<const float4*>(absorb + (i * 4) % Kc);
            v.x /= a.x; v.y /= a.y; v.z /= a.z; v.w /= a.w;
        }
        m = amax4(m, v.x, v.y, v.z, v.w);
    }
    m = block_max_256(m, sh4);
    const float s = scale_of(__float_as_uint(m));
    const int lane = threadIdx.x & 63;
    for (int64_t base = threadIdx.x & ~63; base < K4; base += 256) {      // wave-uniform (store_planes_quad)
        const int64_t i = base + lane;
        const bool live = i < K4;
        float4 v = live ? w4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (absorb && live) {
            const float4 a = *reinterpret_cast<const float4*>(absorb + (i * 4) % Kc);
            v.x /= a.x; v.y /= a.y; v.z /= a.z; v.w /= a.w;
        }
        ushort4 a, b;
        split2(v.x, s, a.x, b.x);
        split2(v.y, s, a.y, b.y);
        split2(v.z, s, a.z, b.z);
        split2(v.w, s, a.w, b.w);
        store_planes_quad(out, r * K4 + i, live, a, b);
    }
    if (threadIdx.x == 0) row_scale[r] = s;
    if (r == rows - 1 && threadIdx.x < 8) reinterpret_cast<uint2*>(out + 2 * rows * K)[threadIdx.x] = make_uint2(0u, 0u);
}

// w [Co][taps][Ci] -> rows ci, k = (tap, co), value w / absorb[co] (absorb nullable), one scale per row ci.  One block per 4
// consecutive ci (float4 reads of the source); the maxima first, then the split.
// (blockIdx.y: batch item -- its own w, its own Ci rows of out and row_scale; the 64-byte zero block follows the last item)
__global__ __launch_bounds__(256) void split2_wT_rows_kernel(const float* w, int Co, int taps, int Ci, const float* absorb,
                                                             uint16_t* out, float* row_scale) {
    __shared__ float sh4[4];
    const int ci0 = blockIdx.x * 4;
    const int64_t Kt = (int64_t)taps * Co;                         // row length
    w += (int64_t)blockIdx.y * Kt * Ci;
    out += (int64_t)blockIdx.y * 2 * Kt * Ci;
    row_scale += (int64_t)blockIdx.y * Ci;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t k = threadIdx.x; k < Kt; k += 256) {
        const int tap = (int)(k / Co), co = (int)(k - (int64_t)tap * Co);
        const float4 v = *reinterpret_cast<const float4*>(w + ((int64_t)co * taps + tap) * Ci + ci0);
        const float ia = absorb ? 1.f / absorb[co] : 1.f;
        m[0] = fmaxf(m[0], fabsf(v.x * ia)); m[1] = fmaxf(m[1], fabsf(v.y * ia));
        m[2] = fmaxf(m[2], fabsf(v.z * ia)); m[3] = fmaxf(m[3], fabsf(v.w * ia));
    }
    float sc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sc[e] = scale_of(__float_as_uint(block_max_256(m[e], sh4)));
    for (int64_t k = threadIdx.x; k < Kt; k += 256) {
        const int tap = (int)(k / Co), co = (int)(k - (int64_t)tap * Co);
        const float4 v = *reinterpret_cast<const float4*>(w + ((int64_t)co * taps + tap) * Ci + ci0);
        const float ia = absorb ? 1.f / absorb[co] : 1.f;
        const float ve[4] = {v.x * ia, v.y * ia, v.z * ia, v.w * ia};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint16_t a, b;
            split2(ve[e], sc[e], a, b);
            uint16_t* o = out + ((int64_t)(ci0 + e) * (Kt / 16) + (k >> 4)) * 32 + (k & 15);
            o[0] = a;
            o[16] = b;
        }
    }
    if (threadIdx.x < 4) row_scale[ci0 + threadIdx.x] = sc[threadIdx.x];
    if (blockIdx.x == 0 && blockIdx.y == gridDim.y - 1 && threadIdx.x < 8)
        reinterpret_cast<uint2*>(out + 2 * (int64_t)Co * taps * Ci)[threadIdx.x] = make_uint2(0u, 0u);
}

// column maxima of x [rows][C] (C % 4 == 0), two stages without atomics (a first version with one atomicMax per thread and column spent
// 0.7 ms per launch on the 4 cache lines of a 128-channel maximum vector): stage 1, thread = (row within the block's stripe, float4
// column group); the block's row groups are combined in LDS and the block writes ONE partial row [C]; stage 2 reduces the partial
// rows and writes the power-of-two scales.
constexpr int COLAMAX_MAXBLK = 512;
__global__ __launch_bounds__(256) void colamax_partial_kernel(const float* x, int64_t rows, int C, float* partial) {
    __shared__ float4 sh[256];
    const int C4 = C / 4;
    const int cg_per = min(C4, 256), rpb = 256 / cg_per;            // column groups and rows covered by one pass of the block
    const int cgi = threadIdx.x % cg_per, rr = threadIdx.x / cg_per;
    for (int c0 = 0; c0 < C4; c0 += cg_per) {                       // (block-uniform trip count)
        const int c4 = c0 + cgi;
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rr < rpb && c4 < C4)
            for (int64_t r = (int64_t)blockIdx.x * rpb + rr; r < rows; r += (int64_t)gridDim.x * rpb) {
                const float4 v = reinterpret_cast<const float4*>(x + r * C)[c4];
                m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
            }
        __syncthreads();
        sh[threadIdx.x] = m;
        __syncthreads();
        if (rr == 0 && c4 < C4) {
            for (int k = 1; k < rpb; ++k) {
                const float4 o = sh[k * cg_per + cgi];
                m.x = fmaxf(m.x, o.x); m.y = fmaxf(m.y, o.y); m.z = fmaxf(m.z, o.z); m.w = fmaxf(m.w, o.w);
            }
            reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * C)[c4] = m;
        }
    }
}
// stage 2: 32 columns per block, 8 row groups of the partial rows per column (one thread per column walked 512 partial rows
// serially: 120 us per launch), combined in LDS
__global__ __launch_bounds__(256) void colamax_final_kernel(const float* partial, int nblk, int C, float* col_scale) {
    __shared__ float sh[256];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    float m = 0.f;
    if (c < C)
        for (int b = rg; b < nblk; b += 8) m = fmaxf(m, partial[(int64_t)b * C + c]);
    sh[threadIdx.x] = m;
    __syncthreads();
    if (rg == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 8; ++k) m = fmaxf(m, sh[k * 32 + (threadIdx.x & 31)]);
        // An all-zero channel (a dead ReLU channel, the gradient columns of a head no sample selected) gets a HUGE scale, not 1: its
        // planes are zero either way, but the weight operand that absorbs the vector divides its entries for this channel by the scale
        // -- with scale 1 those entries (which multiply zeros) would be 2^13 .. 2^24 times larger than the entries that matter and set
        // the weight row's scale, i.e. take the significant bits away from every useful entry of the row (found as a 5e-4 error of the
        // LSTM bias gradients on the bench path: tests/diagnostics/head_grad_probe.py).
        col_scale[c] = m > 0.f ? scale_of(__float_as_uint(m)) : 0x1p100f;
    }
}

// x [rows][C] (C % 16 == 0) -> planes of x[r][c] * col_scale[c]
__global__ __launch_bounds__(256) void split2_cols_kernel(const float* x, int64_t n4, int C, const float* col_scale, uint16_t* out) {
    const int lane = threadIdx.x & 63;
    const int C4 = C / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n4; base += stride) {      // wave-uniform
        const int64_t i = base + lane;
        const bool live = i < n4;
        const float4 v = live ? reinterpret_cast<const float4*>(x)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 sc = live ? reinterpret_cast<const float4*>(col_scale)[i % C4] : make_float4(1.f, 1.f, 1.f, 1.f);
        ushort4 a, b;
        split2(v.x, sc.x, a.x, b.x);
        split2(v.y, sc.y, a.y, b.y);
        split2(v.z, sc.z, a.z, b.z);
        split2(v.w, sc.w, a.w, b.w);
        store_planes_quad(out, i, live, a, b);
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) reinterpret_cast<uint2*>(out + 8 * n4)[threadIdx.x] = make_uint2(0u, 0u);
}

template <int MODE, int NPROD, bool CBM, bool LSTM = false, bool HALO = false, int DBG = 0>
int launch_h2(const H2Args& a, hipStream_t s) {
    auto kern = h2_kernel<MODE, NPROD, CBM, LSTM, HALO, DBG>;
    constexpr int lds = HALO ? HALO_LDS : HNSTAGE * HSTAGE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    const int64_t grid = sp_cdiv(a.M, HBM) * a.tiles_n;
    if (grid <= 0 || grid > 0x7fffffff) return SP_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, a);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

template <int NPROD, int DBG = 0>
int launch_hw(const HWArgs& a, int Co, hipStream_t s) {
    auto kern = hw_kernel<NPROD, DBG>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, HNSTAGE * HWSTAGE);
        attr_set = true;
    }
    const int64_t grid = sp_cdiv(Co, 256) * a.tiles_n;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)a.splits), dim3(512), HNSTAGE * HWSTAGE, s, a);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

int launch_amax(const float* x, int64_t n, unsigned* amax, hipStream_t s) {
    SP_RESET_AMAX(amax, s);   // scratch word of the caller's scale buffer: zeroed here unless the host hands in zeroed single-use slots (see common.h)
    const int64_t n4 = n / 4;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(n4, 256 * 4), 1024));
    hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, s, x, n4, n, amax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

}  // namespace

// out: 2*n fp16 + 64-byte zero block; scale_amax: two 4-byte device words {scale (float, written), amax bits (scratch)}
extern "C" int sp_split2_f16(const float* x, int64_t n, void* out, float* scale_amax, int have_amax, void* stream) {
    if (!x || !out || !scale_amax) return SP_ENULL;
    if (n % 16) return SP_EINVAL;            // every row must be a multiple of 16 long
    hipStream_t s = (hipStream_t)stream;
    unsigned* amax = reinterpret_cast<unsigned*>(scale_amax + 1);
    if (!have_amax) {                        // else: the producer of x already left max|x| (float bits) in scale_amax[1]
        const int rc = launch_amax(x, n, amax, s);
        if (rc != SP_OK) return rc;
    }
    const int64_t n4 = n / 4;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(n4, 256), 4096));
    hipLaunchKernelGGL(split2_kernel, dim3(blocks), dim3(256), 0, s, x, n4, amax, (uint16_t*)out, scale_amax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_split2_f16_wT(const float* w, int Co, int taps, int Ci, void* out, float* scale_amax, void* stream) {
    if (!w || !out || !scale_amax) return SP_ENULL;
    if (((int64_t)taps * Co) % 16) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    unsigned* amax = reinterpret_cast<unsigned*>(scale_amax + 1);
    const int64_t total = (int64_t)Co * taps * Ci;
    const int rc = launch_amax(w, total, amax, s);
    if (rc != SP_OK) return rc;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(total, 256), 4096));
    hipLaunchKernelGGL(split2_wT_kernel, dim3(blocks), dim3(256), 0, s, w, Co, taps, Ci, amax, (uint16_t*)out, scale_amax);
    SP_LAUNCH_CHECK();
    return SP_OK;
}


// weights [rows][K] (K % 16 == 0), value w[r][tap][c] / absorb[c] (absorb [Kc] nullable, Kc % 4 == 0 divides K): one scale per row
extern "C" int sp_split2_f16_rows(const float* w, int64_t rows, int64_t K, int Kc, const float* absorb, void* out, float* row_scale,
                                  void* stream) {
    if (!w || !out || !row_scale) return SP_ENULL;
    if (rows < 1 || K < 16 || K % 16 || (absorb && (Kc < 4 || Kc % 4 || K % Kc))) return SP_EINVAL;
    if (rows > 0x7fffffff || (((uintptr_t)w | (uintptr_t)absorb) & 15)) return SP_EINVAL;
    hipLaunchKernelGGL(split2_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, w, K, Kc, absorb, (uint16_t*)out,
                       row_scale, rows);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_split2_f16_wT_rows_batched(const float* w, int nbatch, int Co, int taps, int Ci, const float* absorb, void* out,
                                             float* row_scale, void* stream);
// w [Co][taps][Ci] -> rows ci, k = (tap, co): the data gradient's weight operand with one scale per row (input channel);
// absorb [Co] nullable: the channel scales of the gradient operand it meets
extern "C" int sp_split2_f16_wT_rows(const float* w, int Co, int taps, int Ci, const float* absorb, void* out, float* row_scale,
                                     void* stream) {
    return sp_split2_f16_wT_rows_batched(w, 1, Co, taps, Ci, absorb, out, row_scale, stream);
}
// nbatch matrices [Co][taps][Ci] one behind the other -> nbatch * Ci rows (item-major), one scale per row: the rank-1 filters wc [B][3C][KP]
// as the B operand [B*KP][3C] of their data gradient in ONE launch (a transposed copy + sp_split2_f16_rows before)
extern "C" int sp_split2_f16_wT_rows_batched(const float* w, int nbatch, int Co, int taps, int Ci, const float* absorb, void* out,
                                             float* row_scale, void* stream) {
    if (!w || !out || !row_scale) return SP_ENULL;
    if (nbatch < 1 || nbatch > 65535 || ((int64_t)taps * Co) % 16 || Ci % 4 || Ci < 4 || ((uintptr_t)w & 15)) return SP_EINVAL;
    hipLaunchKernelGGL(split2_wT_rows_kernel, dim3(Ci / 4, nbatch), dim3(256), 0, (hipStream_t)stream, w, Co, taps, Ci, absorb, (uint16_t*)out,
                       row_scale);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// activations x [rows][C] (C % 16 == 0) with one scale per channel; scratch: sp_split2_f16_cols_workspace(rows, C) bytes
static int colamax_blocks(int64_t rows, int C_) {
    const int C4 = C_ / 4, cg_per = std::min(C4, 256), rpb = 256 / cg_per;
    return (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(rows, (int64_t)rpb * 4), COLAMAX_MAXBLK));
}
extern "C" int64_t sp_split2_f16_cols_workspace(int64_t rows, int C_) {
    if (rows < 1 || C_ < 16 || C_ % 16) return 0;
    return (int64_t)colamax_blocks(rows, C_) * C_ * (int64_t)sizeof(float);
}
extern "C" int sp_split2_f16_cols(const float* x, int64_t rows, int C_, void* out, float* col_scale, void* scratch, void* stream) {
    if (!x || !out || !col_scale || !scratch) return SP_ENULL;
    if (rows < 1 || C_ < 16 || C_ % 16 || (((uintptr_t)x | (uintptr_t)col_scale | (uintptr_t)scratch) & 15)) return SP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int nblk = colamax_blocks(rows, C_);
    hipLaunchKernelGGL(colamax_partial_kernel, dim3(nblk), dim3(256), 0, s, x, rows, C_, (float*)scratch);
    SP_LAUNCH_CHECK();
    hipLaunchKernelGGL(colamax_final_kernel, dim3((unsigned)sp_cdiv(C_, 32)), dim3(256), 0, s, (const float*)scratch, nblk, C_, col_scale);
    SP_LAUNCH_CHECK();
    const int64_t n4 = rows * (C_ / 4);
    const int blocks2 = (int)std::max<int64_t>(1, std::min<int64_t>(sp_cdiv(n4, 256), 4096));
    hipLaunchKernelGGL(split2_cols_kernel, dim3(blocks2), dim3(256), 0, s, x, n4, C_, (const float*)col_scale, (uint16_t*)out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// the halo build of h2_kernel applies: 3x3, stride 1, dilation 1, "same" padding on a 64-pixel-wide map whose images are whole
// numbers of 256-pixel tiles (so a tile is 4 full image rows of one image)
static bool halo_applies(const sp_conv_desc* d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dil == 1 && d->pad == 1 && d->Wo == HALO_W && d->Wi == HALO_W &&
           d->Hi == d->Ho && (d->Ho * d->Wo) % HBM == 0 && d->Kc % 32 == 0;
}

static int conv_igemm_f16(const sp_conv_desc* d, const void* Xs, const float* x_scale, const void* Ws, const float* w_scale,
                          const float* bias, float* out, void* stream, int nprod, double* st_partial = nullptr,
                          float* st_mm = nullptr) {
    if (!d || !Xs || !Ws || !x_scale || !w_scale || !out) return SP_ENULL;
    if (d->mode != 0 && d->mode != 1) return SP_EINVAL;
    if (d->Kc % 32 || d->ldx < d->Kc || d->ldx % 16) return SP_EINVAL;      // a 32-k K-tile must lie inside one filter tap
    if (((uintptr_t)Xs | (uintptr_t)Ws) & 15) return SP_EINVAL;
    if (d->nbatch < 1 || d->stride < 1 || d->dil < 1) return SP_EINVAL;
    const int64_t rows_b = (int64_t)d->N_img * d->Ho * d->Wo;           // output rows per batch item
    if (d->nbatch > 1) {      // batched pointwise GEMM, one weight set per item: dense, contiguous items, whole tiles per item
        if (d->mode != 0 || d->KH * d->KW != 1 || d->stride != 1 || d->pad != 0 || rows_b % HBM || rows_b >= (1LL << 31)) return SP_EINVAL;
        if (d->strideX != rows_b * d->ldx || d->strideC != rows_b * d->ldc || d->strideW != (int64_t)d->Nout * d->Kc) return SP_EINVAL;
        if (st_partial || nprod != 3) return SP_EINVAL;
    }
    H2Args a{};
    a.X = (const uint16_t*)Xs; a.W = (const uint16_t*)Ws; a.bias = bias; a.C = out;
    a.sx = x_scale; a.sw = w_scale; a.sw_rows = d->w_scale_rows ? 1 : 0;
    // row sparsity: the data gradient of a conv (tiles inside one sample) or a batched forward GEMM with one item per sample
    a.row_last = ((d->mode == 1 && d->nbatch == 1) || (d->mode == 0 && d->nbatch > 1)) ? d->row_last : nullptr; a.row_step = d->row_step;
    a.row_nimg = (a.row_last && d->mode == 1 && d->nbatch == 1 && d->N_img <= 64 && ((int64_t)d->Ho * d->Wo) % HBM == 0 &&
                  sp_tuning_get(SP_TUNE_ROW_ORDER, 1) == 1) ? d->N_img : 0;
    a.M = rows_b * d->nbatch;
    a.Hi = d->Hi; a.Wi = d->Wi; a.Kc = d->Kc; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.Nout = d->Nout; a.ldc = d->ldc;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.w_bstride = d->nbatch > 1 ? 4 * d->strideW : 0;
    a.rows_per_batch = (int)rows_b;
    a.ldwb = 4 * (int64_t)d->KH * d->KW * d->Kc;
    a.ncblk = d->Kc / 32;
    a.nkt = d->KH * d->KW * a.ncblk;
    a.tiles_n = (int)sp_cdiv(d->Nout, HBN);
    a.alpha = d->alpha; a.beta = d->beta; a.relu = d->relu;
    const int64_t xb = 4LL * d->nbatch * d->N_img * d->Hi * d->Wi * d->ldx, wb = 4LL * d->nbatch * d->Nout * d->KH * d->KW * d->Kc;
    if (xb + 64 >= (1LL << 32) || wb + 64 >= (1LL << 32)) return SP_EINVAL;      // 32-bit byte offsets in the loaders
    a.x_bytes = (uint32_t)xb; a.w_bytes = (uint32_t)wb;
    if (a.M <= 0 || a.M >= (1LL << 31) || a.Nout <= 0) return SP_EINVAL;             // 32-bit pixel arithmetic in the kernels
    hipStream_t st = (hipStream_t)stream;
    const bool f = d->mode == 0;
    // channel-block-major K order: only where it is defined (several taps, mask fits, stride-1 data gradient)
    const bool cbm_ok = d->KH * d->KW > 1 && d->KH * d->KW <= 32 && (f || d->stride == 1);
    if (st_partial) {        // fused BatchNorm statistics: forward, fp32-faithful
        if (!f || !st_mm || nprod != 3 || d->beta || d->relu || bias) return SP_EINVAL;
        a.st_partial = st_partial; a.st_mm = st_mm;
        return cbm_ok ? launch_h2<0, 3, true>(a, st) : launch_h2<0, 3, false>(a, st);
    }
#ifdef SP_TIMING_VARIANTS      // wrong-result timing modes of the tap-major build (A/B tools only; sp_set_tuning("h2_dbg", n))
    if (sp_tuning_get(SP_TUNE_H2_DBG, 0) >= 256) a.l_probe = sp_tuning_get(SP_TUNE_H2_DBG, 0) & ~255;      // first-round stagger of the CUs (correct results)
    switch (sp_tuning_get(SP_TUNE_H2_DBG, 0)) {
        case 6: return f ? launch_h2<0, 3, false, false, false, 1>(a, st) : launch_h2<1, 3, false, false, false, 1>(a, st);      // no loads
        case 7: return f ? launch_h2<0, 3, false, false, false, 2>(a, st) : launch_h2<1, 3, false, false, false, 2>(a, st);      // no MFMAs
        case 8: return f ? launch_h2<0, 3, false, false, false, 3>(a, st) : launch_h2<1, 3, false, false, false, 3>(a, st);      // MFMAs only
        case 9: return f ? launch_h2<0, 3, false, false, false, 9>(a, st) : launch_h2<1, 3, false, false, false, 9>(a, st);      // LDS-DMA loads only
        default: break;
    }
#endif
    if (nprod == 1)      // throughput mode (one product, both operands rounded to one fp16 plane): tap-major build
        return f ? launch_h2<0, 1, false>(a, st) : launch_h2<1, 1, false>(a, st);
    // channel-block-major K order where that order is defined (with one halo'd activation block per channel block for all 9 taps
    // where the shape allows it), tap-major otherwise
    if (cbm_ok && halo_applies(d) && sp_tuning_get(SP_TUNE_H2_HALO, 1) == 1)
        return f ? launch_h2<0, 3, true, false, true>(a, st) : launch_h2<1, 3, true, false, true>(a, st);
    if (cbm_ok) return f ? launch_h2<0, 3, true>(a, st) : launch_h2<1, 3, true>(a, st);
    return f ? launch_h2<0, 3, false>(a, st) : launch_h2<1, 3, false>(a, st);
}

extern "C" int sp_conv_igemm_f16x2(const sp_conv_desc* d, const void* Xs, const float* x_scale, const void* Ws,
                                   const float* w_scale, const float* bias, float* out, void* stream) {
    return conv_igemm_f16(d, Xs, x_scale, Ws, w_scale, bias, out, stream, 3);
}

// forward conv + the first reduction stage of the BatchNorm behind it: st_partial [tiles][2][Nout] doubles (sum, sum of squares),
// st_mm [tiles][2][Nout] floats (min, max), tiles = sp_conv_stats_tiles(M) -- the layout sp_bn_fwd_split accepts as pre_partial
extern "C" int64_t sp_conv_stats_tiles(const sp_conv_desc* d) {
    if (!d) return 0;
    return sp_cdiv((int64_t)d->N_img * d->Ho * d->Wo, HBM);
}
extern "C" int sp_conv_igemm_f16x2_stats(const sp_conv_desc* d, const void* Xs, const float* x_scale, const void* Ws,
                                         const float* w_scale, float* out, double* st_partial, float* st_mm, void* stream) {
    if (!st_partial || !st_mm) return SP_ENULL;
    return conv_igemm_f16(d, Xs, x_scale, Ws, w_scale, nullptr, out, stream, 3, st_partial, st_mm);
}

extern "C" int sp_conv_igemm_f16x1(const sp_conv_desc* d, const void* Xs, const float* x_scale, const void* Ws,
                                   const float* w_scale, const float* bias, float* out, void* stream) {
    return conv_igemm_f16(d, Xs, x_scale, Ws, w_scale, bias, out, stream, 1);
}

// ConvLSTM step with the cell fused into the h-gate conv (h2_kernel<..., LSTM>):
//   pre = conv3x3(h_prev, Wh) + xg + [spcol x wc]_{i,f,o};  gates = (sigmoid i, f, o; tanh g);  c = f c_prev + i g;  h = o c
// d describes the conv (mode 0, 3x3, Kc = C, Nout = 4C, ldc ignored).  Reference: models/baseline_attention.py ConvLSTM cell
// (SURVEY.md §8a); replaces sp_conv_igemm_f16x2 + sp_lstm_rank1_fwd for steps t >= 1.
extern "C" int sp_gateconv_lstm_f16x2(const sp_conv_desc* d, const void* Hs, const float* h_scale, const void* Ws,
                                      const float* w_scale, const float* xg, const float* c_prev, const float* spcol,
                                      const float* wc, int P, int KP, float* gates, float* c_out, float* h_out, unsigned* h_amax,
                                      void* hout_planes, float* hout_scale, float hout_bound, void* stream) {
    if (!d || !Hs || !Ws || !h_scale || !w_scale || !xg || !spcol || !wc || !gates || !c_out || !h_out) return SP_ENULL;
    if (d->mode != 0 || d->Kc % 32 || d->ldx != d->Kc || d->nbatch != 1 || d->stride != 1 || d->dil < 1) return SP_EINVAL;
    if (d->Nout != 4 * d->Kc || d->Ho != d->Hi || d->Wo != d->Wi || d->KH * d->KW < 2 || d->KH * d->KW > 32) return SP_EINVAL;
    if (P != d->Ho * d->Wo || P % HBM || KP < 1 || KP > 32) return SP_EINVAL;
    if (((uintptr_t)Hs | (uintptr_t)Ws) & 15) return SP_EINVAL;
    // the register-direct cell epilogue reads the x-gates, c_prev and the per-row weight scales and writes gates / c / h as 16-byte
    // accesses per lane (4 consecutive channels; C % 32 == 0 keeps every row 16-byte aligned once the base is)
    if (((uintptr_t)xg | (uintptr_t)c_prev | (uintptr_t)gates | (uintptr_t)c_out | (uintptr_t)h_out) & 15) return SP_EINVAL;
    if (d->w_scale_rows && ((uintptr_t)w_scale & 15)) return SP_EINVAL;
    H2Args a{};
    a.X = (const uint16_t*)Hs; a.W = (const uint16_t*)Ws; a.bias = nullptr; a.C = nullptr;
    a.sx = h_scale; a.sw = w_scale; a.sw_rows = d->w_scale_rows ? 1 : 0;
    a.M = (int64_t)d->N_img * P;
    a.Hi = d->Hi; a.Wi = d->Wi; a.Kc = d->Kc; a.ldx = d->Kc;
    a.Ho = d->Ho; a.Wo = d->Wo; a.Nout = d->Nout; a.ldc = d->Nout;
    a.KH = d->KH; a.KW = d->KW; a.stride = 1; a.pad = d->pad; a.dil = d->dil;
    a.ldwb = 4 * (int64_t)d->KH * d->KW * d->Kc;
    a.ncblk = d->Kc / 32;
    a.nkt = d->KH * d->KW * a.ncblk;
    a.tiles_n = d->Kc / 32;
    a.alpha = 1.f; a.beta = 0; a.relu = 0;
    const int64_t xb = 4LL * a.M * d->Kc, wb = 4LL * d->Nout * d->KH * d->KW * d->Kc;
    if (xb + 64 >= (1LL << 32) || wb + 64 >= (1LL << 32) || a.M <= 0) return SP_EINVAL;
    a.x_bytes = (uint32_t)xb; a.w_bytes = (uint32_t)wb;
    a.l_xg = xg; a.l_cprev = c_prev; a.l_spcol = spcol; a.l_wc = wc;
    if (hout_planes && (!hout_scale || !(hout_bound > 0.f) || ((uintptr_t)hout_planes & 15))) return SP_EINVAL;
    a.l_gates = gates; a.l_c = c_out; a.l_h = h_out; a.l_hamax = h_amax;
    a.l_hplanes = (uint16_t*)hout_planes; a.l_hscale = hout_scale; a.l_hbound = hout_bound;
    a.lC = d->Kc; a.lP = P; a.lKP = KP;
    a.l_probe = sp_tuning_get(SP_TUNE_H2_DBG, 0);      // (product build: always 0)
    hipStream_t st = (hipStream_t)stream;
    SP_RESET_AMAX(h_amax, st);
    if (halo_applies(d) && sp_tuning_get(SP_TUNE_H2_HALO, 1) == 1) return launch_h2<0, 3, true, true, true>(a, st);
    return launch_h2<0, 3, true, true, false>(a, st);
}

extern "C" int64_t sp_conv_wgrad_f16x2_workspace(const sp_wgrad_desc* d) {
    if (!d) return 0;
    if (d->nbatch > 1) return 0;      // batched: every item writes its own result
    const int sp = hw_splits(d);
    return sp <= 1 ? 0 : (int64_t)sp * d->Co * d->ldo * (int64_t)sizeof(float);
}

static int launch_hw_reduce(const HWScales& sc, const float* slab, float* dW, int Co, int Ntot, int ldo, int splits, int64_t slab_stride,
                            float alpha, int beta, hipStream_t s) {
    const int64_t total = (int64_t)Co * Ntot;
    if (Ntot % 4 == 0 && ldo % 4 == 0 && slab_stride % 4 == 0 && sc.Ci % 4 == 0 && (((uintptr_t)slab | (uintptr_t)dW) & 15) == 0) {
        const int blocks = (int)std::min<int64_t>(sp_cdiv(total / 4, 256), 4096);
        hipLaunchKernelGGL(hw_reduce4_kernel, dim3(blocks), dim3(256), 0, s, slab, dW, Co, Ntot / 4, ldo / 4, splits, slab_stride / 4, alpha, sc, beta);
    } else {
        const int blocks = (int)std::min<int64_t>(sp_cdiv(total, 256), 4096);
        hipLaunchKernelGGL(hw_reduce_kernel, dim3(blocks), dim3(256), 0, s, slab, dW, Co, Ntot, ldo, splits, slab_stride, alpha, sc, beta);
    }
    SP_LAUNCH_CHECK();
    return SP_OK;
}

static int conv_wgrad_f16(const sp_wgrad_desc* d, const void* Xsplit, const float* x_scale, const void* dYsplit,
                          const float* y_scale, float* dW, void* workspace, void* stream, int nprod) {
    if (!d || !Xsplit || !dYsplit || !x_scale || !y_scale || !dW) return SP_ENULL;
    if (d->Ci % 16 || d->Co % 16 || d->ldx != d->Ci || d->ldy < d->Co || d->ldy % 16 || d->nbatch < 1) return SP_EINVAL;
    if (((uintptr_t)Xsplit | (uintptr_t)dYsplit) & 15) return SP_EINVAL;
    const int64_t rows_b = (int64_t)d->N_img * d->Ho * d->Wo;      // pixels per batch item
    const bool batched = d->nbatch > 1;
    if (batched) {      // one TN GEMM per item (the cell backward's rank-1 filter gradient): pointwise, dense contiguous items
        if (d->KH * d->KW != 1 || d->stride != 1 || d->pad != 0 || rows_b % 32 || nprod != 3) return SP_EINVAL;
        if (d->strideX != rows_b * d->Ci || d->strideY != rows_b * d->ldy || d->strideO != (int64_t)d->Co * d->ldo) return SP_EINVAL;
    } else if (d->ldo < d->KH * d->KW * d->Ci) {
        return SP_EINVAL;
    }
    HWArgs a;
    a.X = (const uint16_t*)Xsplit; a.dY = (const uint16_t*)dYsplit;
    a.sx = x_scale; a.sy = y_scale;
    a.sx_vec = d->x_scale_vec ? 1 : 0; a.sy_vec = d->y_scale_vec ? 1 : 0;
    if (batched && (a.sx_vec || a.sy_vec)) return SP_EINVAL;      // per-channel scales: one vector per GEMM
    a.row_last = batched ? d->row_last : nullptr; a.row_step = d->row_step;
    a.M = rows_b * d->nbatch;
    a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.ldy = d->ldy;
    a.KH = d->KH; a.KW = d->KW; a.stride = d->stride; a.pad = d->pad; a.dil = d->dil;
    a.Ntot = d->KH * d->KW * d->Ci;
    a.ldo = d->ldo;
    a.tiles_n = (int)sp_cdiv(a.Ntot, 128);
    a.Nvalid = std::min(a.Ntot, d->ldo);          // batched with Ci padded up to 16 k: only the first ldo columns exist
    a.batched = batched ? 1 : 0;
    a.splits = batched ? d->nbatch : hw_splits(d);
    if (!batched && a.splits > 1 && !workspace) return SP_ENULL;
    a.rows_per_split = batched ? rows_b : sp_cdiv(sp_cdiv(a.M, a.splits), 32) * 32;
    a.slab_stride = (int64_t)d->Co * d->ldo;
    a.out = (!batched && a.splits > 1) ? (float*)workspace : dW;
    a.alpha = d->alpha; a.beta = d->beta;
    if (a.M <= 0) return SP_EINVAL;
    const int64_t xb = 4LL * d->nbatch * d->N_img * d->Hi * d->Wi * d->Ci, yb = 4LL * a.M * d->ldy;
    if (xb + 64 >= (1LL << 32) || yb + 64 >= (1LL << 32) || d->Ho * d->Wo < 1) return SP_EINVAL;
    a.x_bytes = (uint32_t)xb; a.y_bytes = (uint32_t)yb;
    hipStream_t s = (hipStream_t)stream;
    int rc = -1000;
#ifdef SP_TIMING_VARIANTS      // wrong-result timing modes (A/B tools only; sp_set_tuning("hw_dbg", n))
    switch (sp_tuning_get(SP_TUNE_HW_DBG, 0)) {
        case 1: rc = launch_hw<3, 1>(a, d->Co, s); break;
        case 2: rc = launch_hw<3, 2>(a, d->Co, s); break;
        case 3: rc = launch_hw<3, 3>(a, d->Co, s); break;
        case 5: rc = launch_hw<3, 5>(a, d->Co, s); break;
        case 6: rc = launch_hw<3, 6>(a, d->Co, s); break;
        case 7: rc = launch_hw<3, 7>(a, d->Co, s); break;
        case 8: rc = launch_hw<3, 8>(a, d->Co, s); break;
        case 9: rc = launch_hw<3, 9>(a, d->Co, s); break;
        default: break;
    }
#endif
    if (rc == -1000) rc = nprod == 1 ? launch_hw<1>(a, d->Co, s) : launch_hw<3>(a, d->Co, s);      // 1: throughput mode
    if (rc != SP_OK) return rc;
    if (a.splits > 1 && !batched) {
        HWScales sc{};
        sc.sx[0] = x_scale; sc.sy[0] = y_scale; sc.sx_vec = a.sx_vec; sc.sy_vec = a.sy_vec; sc.nseg = 1; sc.Ci = d->Ci;
        rc = launch_hw_reduce(sc, (const float*)workspace, dW, d->Co, a.Ntot, d->ldo, a.splits, a.slab_stride, d->alpha, d->beta, s);
        if (rc != SP_OK) return rc;
    }
    return SP_OK;
}

extern "C" int sp_conv_wgrad_f16x2(const sp_wgrad_desc* d, const void* Xsplit, const float* x_scale, const void* dYsplit,
                                   const float* y_scale, float* dW, void* workspace, void* stream) {
    return conv_wgrad_f16(d, Xsplit, x_scale, dYsplit, y_scale, dW, workspace, stream, 3);
}


// ---- the weight gradient of a weight that was applied nseg times (same geometry), all applications in ONE launch (hw2_kernel) ----
// dW = alpha * sum_g dY_g^T X_g (+ dW if beta).  Per application its own split operands and scales (x_scales[g] / y_scales[g]: one
// device scalar each, or per-channel vectors when d->x_scale_vec / d->y_scale_vec).  Returns SP_EINVAL when hw2_kernel's shape
// constraints do not hold (the caller then issues one sp_conv_wgrad_f16x2 per application).
static int hw2_splits(const sp_wgrad_desc* d, int nseg) {
    const int64_t M = (int64_t)d->N_img * d->Ho * d->Wo;
    const int64_t tiles = (int64_t)(d->Co / 256) * ((int64_t)d->KH * d->KW * d->Ci / 256);
    // chains of at most 20480 pixels (single-level accumulation), >= 64 K-tiles per workgroup (its 256 KB slab tile must stay a small
    // part of its work); between these bounds the split count that minimises  rounds x (K-tiles of a split + ~20 K-tiles of fixed cost
    // per workgroup),  rounds = ceil(tiles x nseg x splits / 256): the round-3 rule (>= 8 rounds) ran the few-tile single launches of the
    // encoder on 1.4-2.5 rounds' worth of short workgroups (36 tiles x 40 splits = 5.6 rounds of 64 K-tiles where 7 splits are one
    // round of 366).  The deferred h-gate launch (2160 tile-segments) keeps its 4 splits.
    const int64_t smin = sp_cdiv(M, 20480), smax = std::max<int64_t>(smin, M / 2048);
    if (sp_tuning_get(SP_TUNE_HW_CAP, 0) == 1) {      // timing build: the round-3 rule, for the A/B
        int64_t want = std::max<int64_t>(sp_cdiv(8 * 256, tiles * nseg), smin);
        want = std::min<int64_t>(want, std::max<int64_t>(1, M / 2048));
        return (int)std::max<int64_t>(1, want);
    }
    int64_t best = smin;
    double best_cost = 1e30;
    for (int64_t sp = smin; sp <= smax; ++sp) {
        const double rounds = (double)sp_cdiv(tiles * nseg * sp, 256);
        const double cost = rounds * ((double)sp_cdiv(sp_cdiv(M, sp), 32) + 20.0 + 0.5 * (double)sp);
        if (cost < best_cost * 0.999) {
            best_cost = cost;
            best = sp;
        }
    }
    return (int)std::max<int64_t>(1, best);
}
static bool hw2_applies(const sp_wgrad_desc* d, int nseg) {
    if (nseg < 1 || nseg > HW_MAXSEG || d->nbatch != 1) return false;
    const int64_t M = (int64_t)d->N_img * d->Ho * d->Wo;
    return d->stride == 1 && d->Wo % 32 == 0 && d->Co % 256 == 0 && d->Ci % 16 == 0 && ((int64_t)d->KH * d->KW * d->Ci) % 256 == 0 &&
           d->ldx == d->Ci && d->ldy >= d->Co && d->ldy % 16 == 0 && d->ldo % 4 == 0 && d->ldo >= d->KH * d->KW * d->Ci && M >= 2048 &&
           M < (1LL << 31) && d->Hi < 32768 && d->Wi < 32768 && 4LL * d->N_img * d->Hi * d->Wi * d->Ci + 64 < (1LL << 32);
}
extern "C" int64_t sp_conv_wgrad_f16x2_multi_workspace(const sp_wgrad_desc* d, int nseg) {
    if (!d || !hw2_applies(d, nseg)) return 0;
    return (int64_t)nseg * hw2_splits(d, nseg) * d->Co * d->ldo * (int64_t)sizeof(float);
}
extern "C" int sp_conv_wgrad_f16x2_multi(const sp_wgrad_desc* d, int nseg, const void* const* Xsplits, const float* const* x_scales,
                                         const void* const* dYsplits, const float* const* y_scales, float* dW, void* workspace,
                                         const int* row_last, const int* seg_steps, void* stream) {
    if (!d || !Xsplits || !x_scales || !dYsplits || !y_scales || !dW || !workspace) return SP_ENULL;
    if (!hw2_applies(d, nseg)) return SP_EINVAL;
    if ((row_last != nullptr) != (seg_steps != nullptr)) return SP_ENULL;      // both or neither
    HW2Args a{};
    a.row_last = row_last;
    if (seg_steps)
        for (int g = 0; g < nseg; ++g) a.seg_step[g] = seg_steps[g];
    HWScales sc{};
    for (int g = 0; g < nseg; ++g) {
        if (!Xsplits[g] || !dYsplits[g] || !x_scales[g] || !y_scales[g]) return SP_ENULL;
        if (((uintptr_t)Xsplits[g] | (uintptr_t)dYsplits[g]) & 15) return SP_EINVAL;
        a.X[g] = (const uint16_t*)Xsplits[g]; a.dY[g] = (const uint16_t*)dYsplits[g];
        sc.sx[g] = x_scales[g]; sc.sy[g] = y_scales[g];
    }
    sc.sx_vec = d->x_scale_vec ? 1 : 0; sc.sy_vec = d->y_scale_vec ? 1 : 0; sc.nseg = nseg; sc.Ci = d->Ci;
    a.slab = (float*)workspace;
    a.Hi = d->Hi; a.Wi = d->Wi; a.Ci = d->Ci; a.Ho = d->Ho; a.Wo = d->Wo; a.Co = d->Co; a.ldy = d->ldy;
    a.KH = d->KH; a.KW = d->KW; a.pad = d->pad; a.dil = d->dil;
    a.Ntot = d->KH * d->KW * d->Ci; a.ldo = d->ldo; a.tiles_n = a.Ntot / 256;
    a.splits = hw2_splits(d, nseg);
    a.M = (int)((int64_t)d->N_img * d->Ho * d->Wo);
    a.rows_per_split = (int)(sp_cdiv(sp_cdiv(a.M, a.splits), 32) * 32);
    a.slab_stride = (int64_t)d->Co * d->ldo;
    a.x_bytes = (uint32_t)(4LL * d->N_img * d->Hi * d->Wi * d->Ci);
    hipStream_t s = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(hw2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HW2_LDS);
        attr_set = true;
    }
    const unsigned tiles = (unsigned)((d->Co / 256) * a.tiles_n);
    hipLaunchKernelGGL(hw2_kernel, dim3(tiles, (unsigned)(nseg * a.splits)), dim3(512), HW2_LDS, s, a);
    SP_LAUNCH_CHECK();
    return launch_hw_reduce(sc, (const float*)workspace, dW, d->Co, a.Ntot, d->ldo, a.splits, a.slab_stride, d->alpha, d->beta, s);
}

extern "C" int sp_conv_wgrad_f16x1(const sp_wgrad_desc* d, const void* Xsplit, const float* x_scale, const void* dYsplit,
                                   const float* y_scale, float* dW, void* workspace, void* stream) {
    return conv_wgrad_f16(d, Xsplit, x_scale, dYsplit, y_scale, dW, workspace, stream, 1);
}
