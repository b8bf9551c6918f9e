/*
 * scanpaths_amd.h -- C ABI of the MI355X (gfx950) hot path of chenxy99/Scanpaths.
 *
 * Drop-in boundary: the reference has no FFI -- its hot path is stock torch.nn
 * modules called from Python (AiR/models/baseline_attention.py, models/resnet.py, models/loss.py,
 * AiR/train.py:188-205).  Every entry point below is the device arithmetic of one group of those
 * calls; the host-side Python mirror (scanpaths_amd/models/) binds them with ctypes and exposes the
 * reference's nn.Module / loss / optimiser surface.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers borrowed from the caller
 *     (must outlive the stream op); no allocation, no global state, never synchronises the device;
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it;
 *   - activations are NHWC fp32 ("channels_last"), conv weights are [Cout][KH][KW][Cin] fp32
 *     (== a torch OIHW tensor in channels_last memory format, so reference checkpoints load as-is);
 *   - return 0 on success, negative SP_E* on a rejected argument, positive hipError_t if a launch
 *     failed (the Python wrapper raises RuntimeError).
 */
#ifndef SCANPATHS_AMD_H
#define SCANPATHS_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_EINVAL (-1)   /* unsupported shape / alignment */
#define SP_ENULL (-2)    /* required pointer is NULL */

#define SP_ABI_VERSION 3
int sp_abi_version(void);

/* The ONE piece of process-wide state in the product library: sp_set_tuning("amax_reset", 1) tells the launchers that the caller
 * hands in ZEROED, single-use max|.| slots, so the one-thread reset kernel in front of a producer is only added while the stream is
 * being captured into a HIP graph (a replay re-uses the slot).  Default 0: always reset.  Set it once before launching; every other
 * name returns SP_EINVAL.
 * Wrong-result timing modes ("h2_dbg", "hw_dbg", "b3_dbg": no loads / no MFMAs / ...) and two A/B switches ("h2_halo", "hw_splits") are
 * compiled ONLY into libscanpaths_amd_timing.so (make -C scanpaths_amd/csrc timing; -DSP_TIMING_VARIANTS), which tools/ load for
 * A/B timing; the product library reads no environment variable and cannot be switched into a mode that changes results. */
int sp_set_tuning(const char* name, int value);
/* 1 in the timing build, 0 in the product library. */
int sp_timing_build(void);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), NHWC.
 * One descriptor serves forward conv, data-gradient, dense/batched GEMM (KH=KW=1, H=W=1).
 * Replaces: F.conv2d / nn.Conv2d.forward and its autograd dgrad for every conv on the path
 *   (models/resnet.py:57-152; baseline_attention.py:19-50,204,212-215,270,304-309) and nn.Linear
 *   (baseline_attention.py:207-208,72-88).
 * ---------------------------------------------------------------------------------------------- */
typedef struct sp_conv_desc {
    /* A operand source: NHWC tensor [batchN, Hi, Wi, Kc] with pixel stride ldx (>= Kc) */
    int N_img, Hi, Wi, Kc, ldx;
    /* output: NHWC [batchN, Ho, Wo, Nout] with pixel stride ldc (>= Nout) */
    int Ho, Wo, Nout, ldc;
    int KH, KW, stride, pad, dil;
    /* mode 0 (forward / NK): out[m][n] = sum_{tap,c} X[pix(m,tap)][c] * W[n][tap][c],  W row stride ldw (>= KH*KW*Kc)
     * mode 1 (dgrad   / KN): out[m][n] = sum_{tap,c} X[pixT(m,tap)][c] * W[c][tap][n], W "row" (c,tap) stride ldw (>= Nout)
     *         pixT gathers dY at (y + pad - ky*dil)/stride when divisible (transposed conv) */
    int mode, ldw;
    /* epilogue: out = relu?( alpha*acc + bias[n] + beta*out ) ; bias may be NULL, beta in {0,1} */
    float alpha;
    int beta, relu;
    /* batched GEMM: grid.y = nbatch, element strides between batch items (0 = shared) */
    int nbatch;
    int64_t strideX, strideW, strideC;
    /* split-K for GEMMs with few output tiles (KH=KW=1, nbatch=1): ksplit > 1 partitions the K-tiles over blockIdx.z,
     * partial tiles go to `workspace` (>= ksplit*M*Nout floats) and are summed in a fixed order; 0/1 = off */
    int ksplit;
    void* workspace;
    /* 2xfp16 kernels only (sp_conv_igemm_f16x2*, sp_gateconv_lstm_f16x2): w_scale is a VECTOR with one power-of-two scale per weight
     * row = output column (nbatch * Nout entries; sp_split2_f16_rows / sp_split2_f16_wT_rows) instead of one device scalar */
    int w_scale_rows;
    /* 2xfp16 data gradient (mode 1, nbatch 1) only -- ROW SPARSITY of a gradient implied by the loss masks: row_last[img] (device, one int per
     * sample, nullable) = the last decode step at which sample img receives any loss gradient; a call made for decode step row_step >
     * row_last[img] knows every row of that sample's output gradient to be exactly zero and writes (beta 0) / leaves (beta 1) its output
     * tiles without reading the operands (tiles inside one sample: Ho * Wo % 256 == 0; else the call is dense).  Also honoured by the
     * batched forward form (mode 0, nbatch > 1: one item per sample).  The reference computes
     * these zeros densely (AiR/train.py:190-202: the loss multiplies by action_masks / duration_masks, AiR/models/loss.py:10-14,27-32). */
    const int* row_last;
    int row_step;
} sp_conv_desc;

int sp_conv_igemm(const sp_conv_desc* d, const float* X, const float* W, const float* bias, float* out, void* stream);

/* fp32-faithful variant on the bf16 matrix pipe: every operand pre-split into three bf16 planes x = x1+x2+x3
 * (sp_split3_bf16), six v_mfma_f32_32x32x16_bf16 products per fragment pair, hi/lo fp32 accumulators -> ~2^-24 relative
 * product error at 2.67x the fp32 MFMA rate.  Same descriptor as sp_conv_igemm (nbatch must be 1, Kc % 32 == 0); the
 * weight operand is ALWAYS [Nout][K] rows (K contiguous): for mode 1 pass the operand made by sp_split3_bf16_wT.
 * ldx must equal Kc (dense NHWC source). */
/* "split-3 interleaved" operand: x fp32 [rows][K] (K % 16 == 0) -> bf16 [rows][K/16][3][16] (96 contiguous bytes per row
 * per 16-k group; 6 bytes per element) followed by a 64-byte zero block (out must hold 6*n + 64 bytes) */
int sp_split3_bf16(const float* x, int64_t n, void* out, void* stream);
/* w [Co][taps][Ci] fp32 -> rows ci, k = (tap, co):  bf16 [Ci][taps*Co/16][3][16]  (K-contiguous B operand of dgrad) */
int sp_split3_bf16_wT(const float* w, int Co, int taps, int Ci, void* out, void* stream);
int sp_conv_igemm_bf16x3(const sp_conv_desc* d, const void* Xsplit, const void* Wsplit, const float* bias, float* out,
                         void* stream);

/* Weight gradient (TN GEMM, reduction over output pixels), deterministic split over pixel ranges.
 *   dW[co][tap][ci] (+)= sum_m dY[m][co] * X[pix(m,tap)][ci]
 * Replaces the wgrad half of conv2d/linear backward (autograd of the modules above).
 * workspace: at least sp_conv_wgrad_workspace(d) bytes (0 allowed when it returns 0). */
typedef struct sp_wgrad_desc {
    int N_img, Hi, Wi, Ci, ldx;      /* X: NHWC input of the conv                       */
    int Ho, Wo, Co, ldy;             /* dY: NHWC output-gradient of the conv            */
    int KH, KW, stride, pad, dil;
    int ldo;                         /* row stride of dW rows (>= KH*KW*Ci)             */
    int beta;                        /* 0: overwrite, 1: accumulate into dW             */
    float alpha;
    int nbatch;                      /* batched TN GEMM (KH=KW=1), strides in elements  */
    int64_t strideX, strideY, strideO;
    /* 2xfp16 kernels only: x_scale / y_scale are per-CHANNEL vectors ([Ci] / [Co]; sp_split2_f16_cols) instead of device scalars;
     * K = pixels, so a power-of-two scale per channel of either operand factors out of the contraction exactly (nbatch must be 1) */
    int x_scale_vec, y_scale_vec;
    /* 2xfp16 batched form (nbatch > 1, one item per sample) only: row sparsity as in sp_conv_desc -- items with row_last[item] < row_step
     * have an all-zero dY: their result is written as zeros (beta 0) / left alone (beta 1) without any work; NULL: dense */
    const int* row_last;
    int row_step;
} sp_wgrad_desc;

int64_t sp_conv_wgrad_workspace(const sp_wgrad_desc* d);
int sp_conv_wgrad(const sp_wgrad_desc* d, const float* X, const float* dY, float* dW, void* workspace, void* stream);

/* weight gradient on the same 3xbf16-split scheme (K = pixels; fragments by ds_read_b64_tr_b16).  X / dY are split-3
 * operands of the dense NHWC tensors; needs Ci % 128 == 0, Co % 16 == 0, nbatch == 1, ldx == Ci, ldy == Co. */
int64_t sp_conv_wgrad_bf16x3_workspace(const sp_wgrad_desc* d);
int sp_conv_wgrad_bf16x3(const sp_wgrad_desc* d, const void* Xsplit, const void* dYsplit, float* dW, void* workspace,
                         void* stream);

/* The same three GEMMs with "2 x fp16 split, 3 products" operands (csrc/conv_f16x2.hip): x * s = x1 + x2 with a per-tensor
 * power-of-two scale s (amax * s in [8192, 16384)), a*b = (a1b1 + a1b2 + a2b1)/(sa*sb); still closer to fp64 than a CPU fp32
 * GEMM (7.6e-8 vs 1.2e-7..3e-7 relative) at half the MFMA work of the bf16x3 scheme.  sp_split2_f16: fp32 [rows][K]
 * (K % 16 == 0) -> [rows][K/16][2][16] fp16 + 64-byte zero block (2n+32 halfs); scale_amax = 2 device words {scale (written),
 * scratch: zeroed by the launcher -- after sp_set_tuning("amax_reset", 1) only under stream capture, the caller then passes a ZERO word}.  igemm: Kc % 32 == 0 (a 32-k K-tile must lie inside one tap); wgrad: Ci % 16 == 0, Co % 16 == 0.
 * Row strides: ldx >= Kc (igemm) / ldy >= Co (wgrad), multiples of 16 -- a GEMM may take the first channels of wider rows (the
 * i, f, o columns of the [pixels][4C] gate gradient).  Batched forms (the two rank-1 gradients of the ConvLSTM cell's backward,
 * AiR/models/baseline_attention.py:37-56; they replace the fp32 batched sp_conv_igemm / sp_rank1_dwc on that path):
 *   igemm, nbatch > 1: mode 0, 1x1, stride 1, pad 0; one weight set [Nout][Kc] per item (strideW = Nout*Kc), items contiguous in X and
 *     out (strideX = rows*ldx, strideC = rows*ldc), rows per item a multiple of 256;
 *   wgrad, nbatch > 1: 1x1, stride 1, pad 0; one result [Co][ldo] per item (strideO = Co*ldo, no workspace), items contiguous in X
 *     (ldx == Ci) and dY, rows per item a multiple of 32; ldo < Ci means Ci was padded up to a multiple of 16 and only the first ldo
 *     columns exist.
 * Anything else returns SP_EINVAL. */
int sp_split2_f16(const float* x, int64_t n, void* out, float* scale_amax,
                  int have_amax /* scale_amax[1] already holds max|x| as float bits (fused into the producer: *_amax outputs) */,
                  void* stream);
int sp_split2_f16_wT(const float* w, int Co, int taps, int Ci, void* out, float* scale_amax, void* stream);
/* Scale VECTORS (round 4).  The per-tensor scale keeps 22 bits only within 2^-17 of the tensor's maximum; an output row / column whose
 * contributions ALL come from a far smaller slice of an operand (forward: output channel <- weight row; data gradient: input channel <-
 * weight column; weight gradient: dW row <- dY channel, dW column <- X channel) carried that slice's error undiluted.  Power-of-two
 * scales along a non-contracted dimension factor out exactly; a per-channel scale of an activation that IS contracted (forward / data
 * gradient) is absorbed exactly by the weight operand it meets (x * s_c times w / s_c):
 *   sp_split2_f16_rows     weights [rows][K] (K = taps * Kc, K % 16 == 0) -> planes of w[r][tap][c] / absorb[c] (absorb [Kc] nullable) with
 *                          one scale per row, row_scale [rows]               (forward operand of F.conv2d: AiR/models/resnet.py:57-93)
 *   sp_split2_f16_wT_rows  w [Co][taps][Ci] -> rows ci, k = (tap, co), value w / absorb[co], one scale per row ci (data-gradient operand)
 *   sp_split2_f16_cols     activations / gradients x [rows][C] (C % 16 == 0) -> planes of x * s_c, col_scale [C]; scratch:
 *                          sp_split2_f16_cols_workspace(rows, C) bytes (partial column maxima, two stages, no atomics)
 * Consumers: sp_conv_desc.w_scale_rows, sp_wgrad_desc.x_scale_vec / y_scale_vec; an activation split by _cols enters the igemm with
 * x_scale -> a device 1.0f and a weight operand split with absorb = its col_scale. */
int sp_split2_f16_rows(const float* w, int64_t rows, int64_t K, int Kc, const float* absorb, void* out, float* row_scale, void* stream);
int sp_split2_f16_wT_rows(const float* w, int Co, int taps, int Ci, const float* absorb, void* out, float* row_scale, void* stream);
/* nbatch matrices [Co][taps][Ci] one behind the other -> nbatch * Ci rows (item-major), one scale per row (round 6: the rank-1 filters wc) */
int sp_split2_f16_wT_rows_batched(const float* w, int nbatch, int Co, int taps, int Ci, const float* absorb, void* out, float* row_scale,
                                  void* stream);
int64_t sp_split2_f16_cols_workspace(int64_t rows, int C);
int sp_split2_f16_cols(const float* x, int64_t rows, int C, void* out, float* col_scale, void* scratch, void* stream);
int sp_conv_igemm_f16x2(const sp_conv_desc* d, const void* Xsplit, const float* x_scale, const void* Wsplit, const float* w_scale,
                        const float* bias, float* out, void* stream);
int64_t sp_conv_wgrad_f16x2_workspace(const sp_wgrad_desc* d);
int sp_conv_wgrad_f16x2(const sp_wgrad_desc* d, const void* Xsplit, const float* x_scale, const void* dYsplit,
                        const float* y_scale, float* dW, void* workspace, void* stream);
/* The weight gradient of a weight that was applied nseg (<= 16) times with the same geometry -- the h-gate conv of the ConvLSTM, one
 * application per decode step (AiR/models/baseline_attention.py:37-56, 303-336) -- as ONE launch at the end of backpropagation through
 * time: dW = alpha * sum_g dY_g^T X_g (+ dW).  hw2_kernel: 256 x 256 block tile, 64 x 128 wave tiles, single-level accumulation over
 * pixel ranges of <= 20480, raw slabs [nseg * splits][Co][ldo] in `workspace` (sp_conv_wgrad_f16x2_multi_workspace bytes), reduced in a
 * fixed order with each application's own scales.  Shape constraints: stride 1, Wo % 32 == 0, Co % 256 == 0, KH*KW*Ci % 256 == 0;
 * the workspace query returns 0 and the launch SP_EINVAL when they do not hold (issue one sp_conv_wgrad_f16x2 per application then). */
int64_t sp_conv_wgrad_f16x2_multi_workspace(const sp_wgrad_desc* d, int nseg);
/* row_last / seg_steps (both or neither; seg_steps is a HOST array of nseg ints): application g belongs to decode step seg_steps[g]; the
 * pixels of a sample with row_last[img] < seg_steps[g] carry an exactly-zero output gradient (sp_conv_desc.row_last) and are skipped. */
int sp_conv_wgrad_f16x2_multi(const sp_wgrad_desc* d, int nseg, const void* const* Xsplits, const float* const* x_scales,
                              const void* const* dYsplits, const float* const* y_scales, float* dW, void* workspace,
                              const int* row_last, const int* seg_steps, void* stream);
/* forward conv whose epilogue also writes the first reduction stage of the BatchNorm behind it (per 256-row tile and output column:
 * sum, sum of squares in fp64; min, max in fp32): st_partial [tiles][2][Nout], st_mm [tiles][2][Nout], tiles = sp_conv_stats_tiles(d).
 * No bias / relu / beta.  models/resnet.py:57-93 (conv -> bn). */
int64_t sp_conv_stats_tiles(const sp_conv_desc* d);      /* M-tiles of the kernel that will run d: 256 rows, 128 for the short-K pointwise kernel */
int sp_conv_igemm_f16x2_stats(const sp_conv_desc* d, const void* Xsplit, const float* x_scale, const void* Wsplit,
                              const float* w_scale, float* out, double* st_partial, float* st_mm, void* stream);
/* THROUGHPUT MODE (SURVEY.md §7 hard part 1 / BASELINE.md §4 "bf16-MFMA mode", reported separately from the fp32-faithful
 * headline): the same kernels on the same split-2 operands with only the main product a1*b1, i.e. both operands rounded to ONE
 * fp16 plane (per-tensor power-of-two scale), fp32 accumulation, a third of the MFMA work.  Does NOT meet the 1e-4 parity bar
 * (relative GEMM error ~2^-11); never selected unless SP_SPLIT_SCHEME=f16x1. */
int sp_conv_igemm_f16x1(const sp_conv_desc* d, const void* Xsplit, const float* x_scale, const void* Wsplit, const float* w_scale,
                        const float* bias, float* out, void* stream);
int sp_conv_wgrad_f16x1(const sp_wgrad_desc* d, const void* Xsplit, const float* x_scale, const void* dYsplit,
                        const float* y_scale, float* dW, void* workspace, void* stream);
/* One ConvLSTM step (t >= 1) with the cell fused into the epilogue of the h-gate conv (reference: models/baseline_attention.py
 * ConvLSTM cell :88-118 as restated in SURVEY.md §8a; replaces sp_conv_igemm_f16x2 + sp_lstm_rank1_fwd):
 *   pre = conv(h_prev, Wh) + xg + [spcol x wc] (i, f, o gates);  gates = sigmoid(i, f, o), tanh(g);  c = f*c_prev + i*g;  h = o*c
 * d: mode 0, stride 1, Kc = C, Nout = 4C (gate-major weight rows i, f, o, g), Ho x Wo = Hi x Wi = P pixels, P % 256 == 0, KP <= 32.
 * xg / gates [B*P][4C], c_prev / c_out / h_out [B*P][C], spcol [B*P][KP], wc [B][3C][KP]; h_amax may be NULL.
 * Alignment: C % 32 == 0 and xg, c_prev, gates, c_out, h_out (and w_scale when d->w_scale_rows) 16-byte aligned -- the cell epilogue
 * reads and writes 4 consecutive channels per lane as one 16-byte access; SP_EINVAL otherwise. */
int sp_gateconv_lstm_f16x2(const sp_conv_desc* d, const void* Hsplit, const float* h_scale, const void* Wsplit, const float* w_scale,
                           const float* xg, const float* c_prev, const float* spcol, const float* wc, int P, int KP, float* gates,
                           float* c_out, float* h_out, unsigned* h_amax,
                           void* hout_planes /* nullable: h_out also as a split operand (2*B*P*C fp16 + 64 zero bytes), scale from */,
                           float* hout_scale /* [2] {scale, bound} */, float hout_bound /* >= max|h_out|: t + 1 after t + 1 steps */,
                           void* stream);
/* Both gradients of the ConvLSTM's rank-1 gate term from ONE pass over the gate gradient (round 6; AiR/models/baseline_attention.py:40-50):
 *   dsp[b][p][k] = sum_{n < N3} dpre[b][p][n] * wc[b][n][k]        dwc[b][n][k] = sum_p dpre[b][p][n] * spcol[b][p][k]
 * dpre_planes: the 2xfp16 split operand of the gate gradient [B*P][ldy] (ldy >= N3 channels per row) with its scalar scale; wcT_planes: the
 * split operand of wc transposed, rows (b, k) = [B*KP][N3], one scale per row (sp_split2_f16_rows); spcol fp32 [B][P][KP]; outputs fp32.
 * KP in {12, 20}, P % 32 == 0, N3 % 256 == 0, B <= 65535 (sp_rank1_grads_applies); workspace >= sp_rank1_grads_workspace bytes (pixel-chunk partials of dwc,
 * reduced in chunk order: run-to-run identical).  row_last (nullable): samples with row_last[b] < row_step have an exactly-zero dpre: zeros, rows not read.
 * Replaces one batched sp_conv_igemm_f16x2 + one batched sp_conv_wgrad_f16x2 launch (each re-read the planes) and their operand preparation. */
int sp_rank1_grads_applies(int B, int P, int N3, int KP, int ldy);
int64_t sp_rank1_grads_workspace(int B, int P, int N3, int KP);
int sp_rank1_grads_f16x2(const void* dpre_planes, const float* dpre_scale, int ldy, const void* wcT_planes, const float* wcT_row_scale,
                         const float* spcol, int B, int P, int N3, int KP, float* dsp, float* dwc, void* workspace, const int* row_last,
                         int row_step, void* stream);


/* column sums of a row-major [M][C] matrix (ld = row stride): out[c] = beta*out[c] + sum_m x[m][c]
 * (bias gradients).  workspace >= sp_colsum_workspace(M, C) bytes. */
int64_t sp_colsum_workspace(int64_t M, int C);
int sp_colsum(const float* x, int64_t M, int C, int ld, float* out, int beta, void* workspace, void* stream);
/* row means: out[m] = scale * sum_c x[m][c]  (mean_c(vf): baseline_attention.py:240-244) and its backward
 * dx[m][c] += scale * dout[m] */
int sp_rowsum(const float* x, int64_t M, int C, float scale, float* out, void* stream);
int sp_rowsum_bwd(const float* dout, int64_t M, int C, float scale, float* dx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (+ReLU, + residual add), NHWC, train and eval mode.  models/resnet.py:29-46,63-90.
 * ---------------------------------------------------------------------------------------------- */
int64_t sp_bn_workspace(int64_t M, int C);
/* train-mode statistics over M=N*H*W rows: mean[C], invstd[C]; updates running stats in place
 * (momentum 0.1, unbiased var) when running_mean != NULL. */
int sp_bn_stats(const float* x, int64_t M, int C, float eps, float momentum, float* mean, float* invstd,
                float* running_mean, float* running_var, void* workspace, void* stream);
/* eval mode: mean/invstd from running stats */
int sp_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                     float* invstd, void* stream);
/* y = relu?( (x-mean)*invstd*gamma + beta + residual? ) */
int sp_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                const float* residual, int relu, int64_t M, int C, float* y,
                unsigned* y_amax /* nullable: atomicMax of the float bits of max|y| (zero it first) */, void* stream);
/* backward.  dy_eff = dy * (y>0 if relu).  train: dx = gamma*invstd*(dy_eff - mean(dy_eff) - xhat*mean(dy_eff*xhat));
 * eval: dx = gamma*invstd*dy_eff.  dgamma/dbeta always.  dres (optional) receives dy_eff. */
int sp_bn_backward(const float* dy, const float* x, const float* y, const float* mean, const float* invstd,
                   const float* gamma, int relu, int training, int64_t M, int C, float* dx, float* dres,
                   float* dgamma, float* dbeta, void* workspace,
                   unsigned* dx_amax /* nullable, as y_amax */, void* stream);

/* Train-mode BatchNorm that also EMITS the 2xfp16 split operand of the consumer conv (layout of sp_split2_f16) and keeps the ReLU
 * mask as one bit per element -- same arithmetic as sp_bn_stats + sp_bn_apply / sp_bn_backward (models/resnet.py:29-46,63-90),
 * one HBM pass fewer per direction.  The operand scale comes from an upper bound of max|output| (per-channel extrema of x through
 * the affine map, + max|residual|; backward: |gamma*invstd|*(max|d| + |k1| + |k2|*max|xhat|)), see csrc/bn_pool.hip.
 *   ext [2][C]: per-channel min / max of x (saved for backward);  planes: 2*M*C fp16 + 64 zero bytes;  y_scale / dx_scale [2]:
 *   {scale, bound} as sp_split2_f16's scale_amax;  bound: 4-byte scratch word (reset by the launcher);  mask: sp_bn_mask_words
 *   uint64 words (required when relu);  y / dx (fp32 copies) may be NULL when nothing reads them;  res_amax: device word holding
 *   the float bits of (a bound of) max|residual|. */
int64_t sp_bn_split_workspace(int64_t M, int C);
int64_t sp_bn_mask_words(int64_t M, int C);
int sp_bn_fwd_split(const float* x, int64_t M, int C, float eps, float momentum, const float* gamma, const float* beta,
                    const float* residual, const unsigned* res_amax, int relu, float* mean, float* invstd, float* running_mean,
                    float* running_var, float* ext, float* y, void* planes, float* y_scale, unsigned* bound,
                    unsigned long long* mask, void* workspace,
                    const double* pre_partial /* nullable: [pre_G][2][C] sums from sp_conv_igemm_f16x2_stats */,
                    const float* pre_mm /* [pre_G][2][C] min / max */, int pre_G, void* stream);
int sp_bn_bwd_split(const float* dy, const float* x, const unsigned long long* mask, const float* mean, const float* invstd,
                    const float* gamma, const float* ext, int64_t M, int C, float* dx, float* dres, void* planes, float* dx_scale,
                    unsigned* bound, float* dgamma, float* dbeta, void* workspace, void* stream);

/* MaxPool2d(3, stride 2, pad 0, ceil_mode=True), NHWC.  models/resnet.py:104. */
int sp_maxpool3s2_fwd(const float* x, int N, int H, int W, int C, float* y, int Ho, int Wo, void* stream);
int sp_maxpool3s2_bwd(const float* dy, const float* x, const float* y, int N, int H, int W, int C, float* dx, int Ho,
                      int Wo, void* stream);
/* the same with the window position (ky*3 + kx) of the first maximum saved by the forward pass (one byte per output element):
 * the backward pass reads neither x nor y and makes no value comparisons */
int sp_maxpool3s2_fwd_idx(const float* x, int N, int H, int W, int C, float* y, unsigned char* argmax /* nullable */, int Ho, int Wo,
                          void* stream);
int sp_maxpool3s2_bwd_idx(const float* dy, const unsigned char* argmax, int N, int H, int W, int C, float* dx, int Ho, int Wo,
                          void* stream);
/* NCHW [N,C,H,W] -> NHWC [N,H,W,Cp] (Cp >= C, extra channels zero) and generic last-dim pad/crop copy */
int sp_nchw_to_nhwc_pad(const float* x, int N, int C, int H, int W, int Cp, float* y, void* stream);
int sp_pad_lastdim(const float* x, int64_t rows, int Cin, int Cout, float* y, void* stream);
/* dx = dy * (y > 0)  (ReLU fused into a conv epilogue, e.g. F.relu(sal_conv(x)) baseline_attention.py:270) */
/* out = inputs[0] + ... + inputs[count-1] in list order (count <= 32 host array of device pointers, n % 4 == 0): the gradient
 * fan-in of a tensor consumed by every decode step (x-gate pre-activations), one pass instead of count-1 adds */
int sp_sum_n(const float* const* inputs, int count, int64_t n, float* out,
             unsigned* out_amax /* nullable: float bits of max|out|, as sp_bn_apply's y_amax */, void* stream);
/* ... with the masked-step sparsity of the backward pass (as sp_sum_n_mixed_rows below): term k belongs to decode step steps[k] (HOST
 * array; -1: never skipped) and is exactly zero, hence not read, for a sample b (n / nsamples contiguous elements, a multiple of 4) with
 * row_last[b] < steps[k] (row_last: device array [nsamples]); both NULL: sp_sum_n. */
int sp_sum_n_rows(const float* const* inputs, int count, int64_t n, float* out, unsigned* out_amax, const int* row_last, const int* steps,
                  int nsamples, void* stream);
/* the same fan-in when some contributions exist only as 2xfp16 split operands (the gate gradient of a ConvLSTM step whose fp32 form
 * was left unwritten, sp_lstm_pointwise_bwd_split with dpre == NULL): inputs[k] != NULL -> fp32 term, else planes[k] / scales[k]
 * (sp_split2_f16 layout, device scale) -> the exact value of the split operand.  n % 16 == 0. */
int sp_sum_n_mixed(const float* const* inputs, const void* const* planes, const float* const* scales, int count, int64_t n, float* out,
                   unsigned* out_amax, void* stream);
/* ... with the masked-step sparsity of the backward pass (AiR/models/loss.py:10-14,27-32: the loss multiplies by the masks): term k is
 * the gate gradient of decode step steps[k] (-1: not tied to a step); for a sample b (n / nsamples contiguous elements each, a multiple of
 * 16) with row_last[b] < steps[k] the term is exactly zero and is not read.  row_last: device array [nsamples], steps: HOST array [count]; both or neither. */
int sp_sum_n_mixed_rows(const float* const* inputs, const void* const* planes, const float* const* scales, int count, int64_t n, float* out,
                        unsigned* out_amax, const int* row_last, const int* steps, int nsamples, void* stream);
int sp_relu_bwd(const float* dy, const float* y, int64_t n, float* dx, void* stream);
/* out = a + b (residual joins in backward), n elements */
int sp_add(const float* a, const float* b, float* out, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Attentive ConvLSTM decoder pieces (AiR/models/baseline_attention.py)
 * ---------------------------------------------------------------------------------------------- */
/* ConvLSTM cell pointwise (:44-54).  pre = xg (+ hg) with gate-major channels [i|f|o|g] of width C each;
 * i,f,o = sigmoid, g = tanh, c' = f*c + i*g, h' = o*c' (no tanh on the cell).  c_prev/hg may be NULL (= 0).
 * gates[rows][4C] keeps the activated gates for backward. */
int sp_lstm_pointwise_fwd(const float* xg, const float* hg, const float* c_prev, int64_t rows, int C, float* gates,
                          float* c_out, float* h_out, void* stream);
/* dpre[rows][4C], dc_prev[rows][C] from dh, dc (either may be NULL = 0) */
/* the same cell with the rank-1 gate terms fused in: pre[b,p,g*C+c] += sum_k spcol[b,p,k] * wc[b,g*C+c,k] for the gates
 * g = i,f,o (spcol [B][P][KP] 9-tap im2col of the spatial memories, wc [B][3C][KP] per-sample contracted filters;
 * baseline_attention.py:40-50).  xg/hg/gates [B*P][4C]; C % 64 == 0, KP <= 64. */
int sp_lstm_rank1_fwd(const float* xg, const float* hg, const float* c_prev, const float* spcol, const float* wc, int B, int P,
                      int C, int KP, float* gates, float* c_out, float* h_out,
                      unsigned* h_amax /* nullable, as y_amax */, void* stream);
/* gradient of the per-sample rank-1 filters: dwc [B][N3][KP] = sum_p dpre[b,p,n] * spcol[b,p,k]; dpre rows have ld floats
 * (= 4C, the first N3 = 3C are used); KP <= 24; workspace >= sp_rank1_dwc_workspace bytes (chunk partials, fixed-order reduce) */
int64_t sp_rank1_dwc_workspace(int B, int P, int N3, int KP);
int sp_rank1_dwc(const float* dpre, const float* spcol, int B, int P, int ld, int N3, int KP, void* workspace, float* dwc,
                 void* stream);
/* get_channel_semantic + ReLU (baseline_attention.py:246-250,284,324): out [B][S][C] = relu(alpha * sum_p a[s][b][p] * vf[b][p][c])
 * with alpha = 1/P (S <= 2 attention streams, C <= 512); backward returns d a [S][B][P] and d vf [B][P][C] in one pass over vf
 * (dout is masked by out > 0 inside).  workspace >= sp_sempool_workspace bytes (chunk partials, fixed-order reduce). */
int64_t sp_sempool_workspace(int S, int B, int P, int C);
int sp_sempool_fwd(const float* a, const float* vf, int S, int B, int P, int C, float alpha, void* workspace, float* out,
                   void* stream);
int sp_sempool_bwd(const float* dout, const float* out, const float* a, const float* vf, int S, int B, int P, int C, float alpha,
                   float* da, float* dvf, void* stream);
/* ... row_last != NULL: samples b with row_last[b] < row_step have an exactly-zero dout (masked-step sparsity of the backward pass): their
 * da / dvf rows are written as zeros, vf is not read */
int sp_sempool_bwd_rows(const float* dout, const float* out, const float* a, const float* vf, int S, int B, int P, int C, float alpha,
                        float* da, float* dvf, const int* row_last, int row_step, void* stream);
/* round 6: sbc != 0: the pooled rows / their gradient lie [S][B][C] (what the embedding behind them reads) instead of [B][S][C]: no transposed copy */
int sp_sempool_fwd_sbc(const float* a, const float* vf, int S, int B, int P, int C, float alpha, void* workspace, float* out, int sbc, void* stream);
int sp_sempool_bwd_rows_sbc(const float* dout, const float* out, const float* a, const float* vf, int S, int B, int P, int C, float alpha,
                            float* da, float* dvf, const int* row_last, int row_step, int sbc, void* stream);
int sp_lstm_pointwise_bwd(const float* dh, const float* dc, const float* gates, const float* c_prev,
                          const float* c_out, int64_t rows, int C, float* dpre, float* dc_prev,
                          unsigned* dpre_amax /* nullable, as y_amax */, void* stream);
/* as above, and (planes != NULL) dpre also as the 2xfp16 split operand of the h-gate conv's backward GEMMs, scale from the bound
 * max(D, max|dh| * c_bound / 4, D * cprev_bound / 4), D = max|dh| + max|dc|  (dh_amax / dc_amax: device words with the float bits
 * of (bounds of) the maxima; c_bound >= max|c_out|, cprev_bound >= max|c_prev|: t + 1 and t after t + 1 steps).  dcp_amax
 * (nullable) receives max|dc_prev|.  planes: 2 * rows * 4C fp16 + 64 zero bytes; dpre_scale [2] = {scale, bound}.  C % 256 == 0.
 * dpre may be NULL when planes is given: the fp32 gate gradient is then not written (every consumer reads the split form). */
int sp_lstm_pointwise_bwd_split(const float* dh, const float* dc, const float* gates, const float* c_prev, const float* c_out,
                                int64_t rows, int C, float* dpre, float* dc_prev, unsigned* dpre_amax, unsigned* dcp_amax,
                                const unsigned* dh_amax, const unsigned* dc_amax, float c_bound, float cprev_bound, void* planes,
                                float* dpre_scale, void* stream);
/* the same with the row sparsity of sp_conv_desc.row_last: samples (rows_per_sample consecutive rows each) with row_last[sample] < row_step
 * get zero outputs without reading the inputs */
int sp_lstm_pointwise_bwd_rows(const float* dh, const float* dc, const float* gates, const float* c_prev, const float* c_out,
                               int64_t rows, int C, float* dpre, float* dc_prev, unsigned* dpre_amax, unsigned* dcp_amax,
                               const unsigned* dh_amax, const unsigned* dc_amax, float c_bound, float cprev_bound, void* planes,
                               float* dpre_scale, const int* row_last, int row_step, int rows_per_sample, void* stream);

/* 3x3 zero-padded im2col of single-channel maps [R][H][W] into columns [koff,koff+9) of col[r][p][ldk], and adjoint.
 * Feeds the rank-1 gate convolutions conv3x3(W, spatial (x) semantic) (baseline_attention.py:40-50) and the
 * spatial attention score map (:111-124) as small GEMMs. */
int sp_im2col3x3_1ch(const float* maps, int R, int H, int W, int koff, int ldk, float* col, void* stream);
int sp_col2im3x3_1ch(const float* dcol, int R, int H, int W, int koff, int ldk, float* dmaps, void* stream);
/* all S streams in ONE launch: maps [S][R][H][W] -> col [R][H*W][ldk] (columns >= 9 S written as zeros), and the adjoint dmaps [S][R][H][W] */
int sp_im2col3x3_multi(const float* maps, int S, int R, int H, int W, int ldk, float* col, void* stream);
int sp_col2im3x3_multi(const float* dcol, int S, int R, int H, int W, int ldk, float* dmaps, void* stream);

/* attention over a growing memory list (semantic_att :77-88 / spatial_att :111-124 after removing the terms that are
 * constant along the softmax axis): score[t][r] = <L[t][r][:], u>; a = softmax_t; mem[r][:] = sum_t a[t][r]*L[t][r][:]
 * L: [T][R][D]; alpha: [T][R]; du_partial: [R][D] (column-sum it for du). T <= 40. */
int sp_listatt_fwd(const float* L, const float* u, int T, int R, int D, float* mem, float* alpha, void* stream);
int sp_listatt_bwd(const float* dmem, const float* L, const float* u, const float* alpha, int T, int R, int D, float* dL,
                   float* du_partial, void* stream);

/* out[i] = relu(a[i]*b[i % nb]) and backward (get_spatial_semantic after hoisting mean_c(vf), :240-244,277-278);
 * db_partial has n elements (sum the n/nb periods for db). */
int sp_mulrelu_fwd(const float* a, const float* b, int64_t n, int64_t nb, float* out, void* stream);
int sp_mulrelu_bwd(const float* dout, const float* a, const float* b, const float* out, int64_t n, int64_t nb, float* da,
                   float* db_partial, void* stream);

/* per-sample good/poor selection (:360-374): out[r][:] = sel[r] ? a[r][:] : b[r][:] , and adjoint */
int sp_select_rows(const float* a, const float* b, const unsigned char* sel, int64_t rows, int64_t len, float* out,
                   void* stream);
int sp_select_rows_bwd(const float* dout, const unsigned char* sel, int64_t rows, int64_t len, float* da, float* db,
                       void* stream);

/* predict_head after head composition (:149-174).  Z[b][p][ldz] holds, per head hd at column base hd*HC (HC >= 52):
 *   col 0 terminate map (sal_layer_2 o 5x5), col 1 action map (sal_layer_3 o 5x5), cols 2..50 the 49 taps of
 *   drt_layer_1 o 5x5.  cb: composed biases [nheads][HC] (cb[.][51] = drt_layer_1.bias); w2/b2: drt_layer_2 [2][dh*dw],[2].
 * Outputs per head: logits [nheads][B][1+P] (col 0 = terminate; probabilities when softmax != 0, i.e. eval mode :161-162),
 * amap [nheads][B][P] (relu action map, always pre-softmax), mu/sigma2 [nheads][B], drt [nheads][B][dh*dw] (post-relu). */
int sp_head_finish_fwd(const float* Z, int B, int Hm, int Wm, int ldz, int nheads, int HC, const float* cb,
                       int cb_per_sample /* 1: cb is [B][nheads][HC] (COCO per-task heads) */, const float* w2, const float* b2, int softmax, float* logits, float* amap, float* mu,
                       float* sigma2, float* drt,
                       const float* dpre /* nullable: [nheads][B][dh*dw] duration-site sums incl. composed tap biases (sp_drt_direct_fwd);
                                            then Z needs only the 2 map columns per head */,
                       int zc /* Z columns per head (<= 0: HC) */, void* stream);
/* dlogits is the gradient w.r.t. the `logits` output (probabilities if softmax).  dZ is fully written for the nheads*HC
 * columns.  Partials are per sample: dcb [B][nheads][HC], dw2 [B][nheads][2][dh*dw], db2 [B][nheads][2]. */
int sp_head_finish_bwd(const float* dlogits, const float* damap /* nullable, [nheads][B][P] */, const float* dmu,
                       const float* dsigma2, const float* logits,
                       const float* amap, const float* sigma2, const float* drt, int B, int Hm, int Wm, int ldz,
                       int nheads, int HC, const float* w2, int softmax, float* dZ, float* dcb_partial,
                       float* dw2_partial, float* db2_partial,
                       float* ddpre /* nullable out, gradient of dpre; tap columns of dZ / dcb are then not produced */, int zc,
                       void* stream);

/* The two halves of the epilogue separately (round 6).  parts: bit 0 = the saliency half (terminate logit, action map, softmax: what the next
 * memory update of the decode loop reads, AiR/models/baseline_attention.py:160-166,311-336), bit 1 = the duration half (mu, sigma2, :155-159:
 * read by nothing inside the recurrence).  The decode loop runs parts = 1 per step and parts = 2 ONCE behind the loop over T x nheads virtual
 * heads (head (t, i) of B rows; dpre = the sites of all steps from one sp_drt_direct_fwd launch over T x B rows).  Arguments of the half that
 * is not asked for may be NULL; parts = 3 is sp_head_finish_fwd / _bwd.  parts = 1 writes dcb[0], dcb[1] (others zero), parts = 2 writes
 * dcb[51] (others zero).  live (nullable, parts & 2): live[hd * B + b] = 1 iff slot hd of row b received a non-zero dmu or dsigma2 -- the
 * duration sites' backward (sp_drt_direct_bwd_*_live) skips the others: their site gradients are exact zeros. */
int sp_head_finish_parts_fwd(const float* Z, int B, int Hm, int Wm, int ldz, int nheads, int HC, const float* cb, int cb_per_sample,
                             const float* w2, const float* b2, int softmax, float* logits, float* amap, float* mu, float* sigma2, float* drt,
                             const float* dpre, int zc, int parts, void* stream);
int sp_head_finish_parts_bwd(const float* dlogits, const float* damap, const float* dmu, const float* dsigma2, const float* logits,
                             const float* amap, const float* sigma2, const float* drt, int B, int Hm, int Wm, int ldz, int nheads, int HC,
                             const float* w2, int softmax, float* dZ, float* dcb_partial, float* dw2_partial, float* db2_partial,
                             float* ddpre, int zc, int parts, int* live, void* stream);
/* round 6: dlogits_ld = elements between the (head, sample) rows of dlogits (0 = Hm*Wm + 1): a slice of the stacked outputs' gradient, read in place */
int sp_head_finish_parts_bwd_ld(const float* dlogits, int64_t dlogits_ld, const float* damap, const float* dmu, const float* dsigma2,
                                const float* logits, const float* amap, const float* sigma2, const float* drt, int B, int Hm, int Wm, int ldz,
                                int nheads, int HC, const float* w2, int softmax, float* dZ, float* dcb_partial, float* dw2_partial,
                                float* db2_partial, float* ddpre, int zc, int parts, int* live, void* stream);

/* ------------------------------------------------------------------------------------------------
 * predict_head without the dense 5x5 GEMM (models/baseline_attention.py:149-158), csrc/head_direct.hip.
 * G physical [nheads*HC][5][5][C] composed head filters (row hd*HC+0/1: sal_layer_2/3 o 5x5, rows +2..+50: the 49 taps of
 * drt_layer_1 o 5x5), cb [nheads][HC] composed biases.  Border classes: a site's class = which of the 7x7 taps land inside
 * the zero-padded intermediate map (per axis: first / interior / last); ncls = sp_head_num_classes(Hm, Wm).
 *   compose11: W11 [nheads][ncls][11*11][C] composite stride-5 windows, cbsum [nheads][ncls] sums of in-range tap biases.
 *   sal_gather: T [B][P][ldt] tap partials (1x1 GEMM of h with G rows 0/1 reshaped [(src*2+o)*25+tap][C]) ->
 *               Z2 [B][P][nsel*2] (5x5 spatial sums, zero padding); hmap [B][nsel] int32 = source head of output slot.
 *   drt_direct: Dpre [nsel][B][dh*dw] = cbsum + <W11[class(site)], 11x11 window of h at 5*site-4>; bwd_data writes (or
 *               accumulates into) dh [B][P][C]; bwd_weight reduces per-sample slabs in fixed order -> dW11, dcbsum.
 * ---------------------------------------------------------------------------------------------- */
int sp_head_num_classes(int Hm, int Wm);
int sp_head_compose11_fwd(const float* G, const float* cb, int nheads, int HC, int C, int Hm, int Wm, float* W11,
                          float* cbsum, void* stream);
int sp_head_compose11_bwd(const float* dW11, const float* dcbsum, int nheads, int HC, int C, int Hm, int Wm, float* dG,
                          float* dcb, void* stream);
int sp_sal_gather_fwd(const float* T, int B, int Hm, int Wm, int ldt, int nsel, const int* hmap, float* Z2, void* stream);
int sp_sal_gather_bwd(const float* dZ2, int B, int Hm, int Wm, int ldt, int nsel, int nsrc, const int* hmap, float* dT,
                      void* stream);
/* ... row_last != NULL: samples b with row_last[b] < row_step have an exactly-zero dZ2: their dT rows are zeros, dZ2 is not read */
int sp_sal_gather_bwd_rows(const float* dZ2, int B, int Hm, int Wm, int ldt, int nsel, int nsrc, const int* hmap, float* dT,
                           const int* row_last, int row_step, void* stream);
int sp_drt_direct_fwd(const float* h, const float* W11, const float* cbsum, const int* hmap, int B, int Hm, int Wm, int C,
                      int nsel, float* Dpre, void* stream);
int sp_drt_direct_bwd_data(const float* dDpre, const float* W11, const int* hmap, int B, int Hm, int Wm, int C, int nsel,
                           int accumulate, float* dh, void* stream);
/* ... with the (row, head slot) pairs whose gradient is exactly zero skipped (bit-identical sums): live (nullable) [nsel][B] from
 * sp_head_finish_parts_bwd; row_last (nullable): row b -- decode step b / rowB of sample b % rowB when one launch covers several steps, rowB = B
 * for a per-step launch -- is dead when row_last[b % rowB] < row_step + b / rowB (B % rowB == 0).  Dead rows of dh are written as zeros. */
int sp_drt_direct_bwd_data_live(const float* dDpre, const float* W11, const int* hmap, int B, int Hm, int Wm, int C, int nsel,
                                int accumulate, float* dh, const int* live, const int* row_last, int row_step, int rowB, void* stream);
int sp_drt_direct_bwd_weight_live(const float* dDpre, const float* h, const int* hmap, int B, int Hm, int Wm, int C, int nsel, int nheads,
                                  void* workspace, float* dW11, float* dcbsum, const int* live, const int* row_last, int row_step, int rowB,
                                  void* stream);
int64_t sp_drt_direct_bwd_weight_workspace(int B, int Hm, int Wm, int C, int nsel);
int sp_drt_direct_bwd_weight(const float* dDpre, const float* h, const int* hmap, int B, int Hm, int Wm, int C, int nsel,
                             int nheads, void* workspace, float* dW11, float* dcbsum, void* stream);
/* ... row_last != NULL: samples b with row_last[b] < row_step have an exactly-zero dDpre: their slabs are zeros, h is not read */
int sp_drt_direct_bwd_weight_rows(const float* dDpre, const float* h, const int* hmap, int B, int Hm, int Wm, int C, int nsel,
                                  int nheads, void* workspace, float* dW11, float* dcbsum, const int* row_last, int row_step, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Skinny fp32 GEMM (round 5): C[M <= 64][N] = relu?(alpha * A[M][K] * op(B) + bias[N]) on fp32 MFMA, the weight matrix streamed once
 * by ~512 workgroups.  The dense layers the decode loop applies to a handful of rows per step -- spatial_embed / semantic_embed
 * (nn.Linear, AiR/models/baseline_attention.py:207-208,279-286,319-326) and the contraction of the semantic memory with the rank-1 gate
 * filters (:40-50) -- and their data gradients.  layout 0: B [N][ldb >= K] (C = A B^T; N % 16 == 0); layout 1: B [K][ldb >= N]
 * (C = A B; N % 64 == 0, ldc % 4 == 0); K % 16 == 0, lda % 4 == 0; A, B and the workspace 16-byte aligned, C and bias too
 * for layout 1 and whenever K slices are used (the callers' Python side falls back to sp_conv_igemm otherwise).  workspace >= sp_gemm_skinny_workspace
 * bytes (K slices, summed in slice order: bitwise reproducible).  sp_gemm_skinny_applies: 1 when the shape is supported (else the
 * caller uses sp_conv_igemm).
 * ---------------------------------------------------------------------------------------------- */
int sp_gemm_skinny_applies(int M, int N, int K, int lda, int ldb, int ldc, int layout);
int64_t sp_gemm_skinny_workspace(int M, int N, int K, int layout);
int sp_gemm_skinny(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc, int layout,
                   float alpha, int relu, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Loss (models/loss.py:10-14,27-32; AiR/train.py:192-197) -- value and gradient in one pass.
 * z [B][T][A] logits, gt soft one-hot, masks [B][T]; mask_sums = {sum(action_mask), sum(duration_mask)} on device
 * (all-reduced by the caller under data parallelism so the loss is normalised by the GLOBAL mask sums).
 * out3 = {loss, loss_actions, loss_duration} (local numerators / given sums); dz, dmu, dsigma2 = d loss / d input.
 * ---------------------------------------------------------------------------------------------- */
int64_t sp_scanpath_loss_workspace(int B, int T);
int sp_scanpath_loss(const float* z, const float* gt, const float* amask, const float* mu, const float* sigma2,
                     const float* dur, const float* dmask, int B, int T, int A, float lambda1, const float* mask_sums,
                     float* out3, float* dz, float* dmu, float* dsigma2, void* workspace, void* stream);
/* RL (self-critical) phase log-probabilities, models/loss.py:34-45 (callers AiR/train.py:288-289).  Value and per-element
 * derivative in one pass; mask_sum = device scalar sum(mask) over the whole tensor (the reference's normaliser).
 *   log_action:   out[b] = sum_t log(p[b,t] + 1e-7) * mask[b,t] / mask_sum;   dcoef[b,t] = d out[b] / d p[b,t]
 *   log_duration: out[b] = sum_t (log(1/(d+1e-7) * 1/sqrt(2 pi s2)) - (log(d+1e-7) - mu)^2 / (2 s2)) * mask / mask_sum
 *   rowscale:     out[b,t] = coef[b,t] * g[b]   (chain rule of the two above) */
int sp_log_action(const float* p, const float* mask, int B, int T, const float* mask_sum, float* out, float* dcoef, void* stream);
int sp_log_duration(const float* d, const float* mu, const float* sigma2, const float* mask, int B, int T, const float* mask_sum,
                    float* out, float* dmu, float* dsigma2, void* stream);
int sp_rowscale(const float* coef, const float* g, int B, int T, float* out, void* stream);
/* Masked-step sparsity of the backward pass, derived from the gradient that actually reaches the model's outputs (the reference
 * multiplies every loss term by action_masks / duration_masks, AiR/models/loss.py:10-14,27-32, AiR/train.py:190-197, and computes the
 * resulting zeros densely).  grads: HOST array of `count` (<= 8) device pointers to contiguous fp32 output gradients
 * [nstack][B][T][row_len[k]] (NULL entry: that gradient never arrived = zero); last[b] (device, [B]) = the last decode step t at
 * which any element of any of them is non-zero (NaN counts), -1 when there is none.  No host synchronisation. */
int sp_rows_last(const float* const* grads, const int64_t* row_len, int count, int nstack, int B, int T, int* last, void* stream);
/* out = x * (*scale) with the scalar on the device (chain-rule factor of the loss, no host sync) */
int sp_scale_by(const float* x, const float* scale, int64_t n, float* out, void* stream);
/* deterministic sums (fp64 accumulation); workspace >= sp_sumsq_workspace(n) bytes for both */
int64_t sp_sumsq_workspace(int64_t n);
int sp_sum(const float* x, int64_t n, float* out, void* workspace, void* stream);
int sp_sumsq(const float* g, int64_t n, double* out, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ScanMatch scoring (utils/evaltools/scanmatch.py:88-197; callers utils/evaluation.py:22-64,198-235,361-559), SURVEY.md §8
 * row f2.  float64 with the reference's operation order: scores are bit-exact with the numpy implementation.
 *   submatrix:  sub [nb*nb] (nb = Xbin*Ybin) = |dist - max| - (max - threshold), maxsub [1] = max(sub)          (:88-103)
 *   sequences:  scanpath k = rows start[k] .. start[k]+count[k]-1 of fix [rows][ncol] (x, y[, duration]) -> symbols
 *               seq [nsp][ld] (first seq_len[k] valid; seq may be NULL to size the buffer): offset, clamp to the screen,
 *               int truncation, bin = int(pixel * bins/res), symbol = ybin*Xbin + xbin (or mask[y][x] when a custom
 *               [Yres][Xres] int32 mask is given, maskFromArray :199-200); with tempbin != 0 each symbol is repeated
 *               round_half_even(trunc(duration)/tempbin) times                                                   (:105-135)
 *   score:      scores[p] for pairs[p] = (index into A, index into B) (pairs NULL: p with p); Needleman-Wunsch with
 *               F[i][0] = gap*(i+1), F[0][j] = gap*(j+1); score = max(F) / (maxsub * max(n, m)); one wavefront per pair;
 *               sequence length <= sp_scanmatch_max_len()                                                        (:137-150,188-193)
 *   align:      single pair: F^T [(m+1)][(n+1)], alignment [nalign][2] (-1 = gap) and score                     (:152-197)
 * ---------------------------------------------------------------------------------------------- */
int sp_scanmatch_max_len(void);
int sp_scanmatch_submatrix(int Xbin, int Ybin, double threshold, double* sub, double* maxsub, void* stream);
int sp_scanmatch_sequences(const double* fix, int ncol, const int64_t* start, const int* count, int nsp, int Xres, int Yres,
                           int Xbin, int Ybin, double off_x, double off_y, double tempbin, const int* mask, int ld, int* seq,
                           int* seq_len, void* stream);
int sp_scanmatch_score(const int* seqA, const int* lenA, int ldA, const int* seqB, const int* lenB, int ldB, const int* pairs,
                       int npairs, const double* sub, int nb, const double* maxsub, double gap, double* scores, void* stream);
/* score_long: the same scores for sequences longer than sp_scanmatch_max_len() (heavy-tailed sampled durations in the RL
 * phase, AiR/train.py:256-275 -- the reference's python DP has no length limit): the strip column lives in
 * workspace [npairs][ldA + 1] doubles (sp_scanmatch_score_long_workspace bytes) instead of LDS. */
int64_t sp_scanmatch_score_long_workspace(int ldA, int npairs);
int sp_scanmatch_score_long(const int* seqA, const int* lenA, int ldA, const int* seqB, const int* lenB, int ldB,
                            const int* pairs, int npairs, const double* sub, int nb, const double* maxsub, double gap,
                            double* scores, void* workspace, void* stream);
int sp_scanmatch_align(const int* A, int n, const int* B, int m, const double* sub, int nb, const double* maxsub, double gap,
                       double* F_work /* [(n+1)*(m+1)] scratch */, double* Ft, double* align /* [(n+m)][2] */, int* nalign,
                       double* score, void* stream);

/* SED / STDE of scanpath pairs (utils/evaltools/visual_attention_metrics.py:205-441; callers utils/evaluation.py:68-72,239-243).
 * fix [rows][ncol] (x, y, ...), scanpath k = rows start[k] .. +count[k] (count <= sp_scan_max_fixations()); pairs[p] =
 * (human index, simulated index).  sed[p] = Levenshtein distance of the ngrid x ngrid cell strings (cell = int32(x) //
 * (width // ngrid) + int32(y) // (height // ngrid) * ngrid), bit-exact; stde[p] = mean over k = 1..min(len) of
 * exp(-mean_s min_h sum_i ||s_i - h_i|| / k) on coordinates / max_dim, numpy summation order, NaN when a scanpath is empty
 * (the reference returns None).  Either output may be NULL. */
int sp_scan_max_fixations(void);
int sp_scan_sed_stde(const double* fix, int ncol, const int64_t* start, const int* count, const int* pairs, int npairs, int height,
                     int width, int ngrid, double max_dim, int* sed, double* stde, void* stream);
/* MultiMatch (the five similarities the reference obtains per pair from multimatch_gaze.docomparison, AiR/utils/evaluation.py:7,44-45,213;
 * multimatch_gaze==0.1.2 is not vendored: the published algorithm, restated on the host in utils/evaltools/multimatch.py, is this
 * kernel's checker): fix [total][ncol >= 3] = (x, y, duration), scanpaths of at most sp_scan_max_fixations() fixations; out [npairs][5]
 * = (vector, direction, length, position, duration), five NaNs for a pair with a scanpath of fewer than 3 fixations. */
int sp_scan_multimatch(const double* fix, int ncol, const int64_t* start, const int* count, const int* pairs, int npairs,
                       double screen_w, double screen_h, double* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Post-hoc sampling (models/sampling.py:16-77), SURVEY.md §8 row f1.
 * sp_sample_actions: per (b,t) masked categorical draw (terminate action 0 excluded for t < min_length), probability of the
 *   chosen action from the unmasked distribution, duration = exp(eps*sigma2 + mu); Philox4x32-10(seed; row).
 * sp_generate_scanpath: first-terminate scan -> length [B] (0 -> T quirk), masks [B][T], fix [B][T][3] = (x, y, duration) in
 *   pixels for the nfix[b] leading fixations.
 * ---------------------------------------------------------------------------------------------- */
int sp_sample_actions(const float* probs, const float* mu, const float* sigma2, int B, int T, int A, int min_length,
                      uint64_t seed, int64_t* actions, float* action_probs, float* durations, void* stream);
int sp_generate_scanpath(const int64_t* actions, const float* durations, int B, int T, int map_w, int map_h, int width, int height,
                         float* length, float* action_masks, float* duration_masks, float* fix, int* nfix, void* stream);
/* Beam search of width K (<= 8) over the per-step action distributions probs [B][T][A] (eval-mode outputs) -- BASELINE.json
 * config 5 "beam-4 scanpath sampling"; build-side decoder, the reference only samples (models/sampling.py:16-46).
 * score = sum_t log p_t(a_t); action 0 (terminate) ends a sequence and is allowed from t >= min_length; actions [B][K][T]
 * (0 after termination), scores [B][K] float64, best first.  T <= 64, A <= 32767. */
int sp_beam_search(const float* probs, int B, int T, int A, int min_length, int K, int64_t* actions, double* scores, void* stream);
/* Training targets of one batch from ragged fixation lists (AiR/dataset/dataset.py:111-147 with blur_sigma = None;
 * collate_func :168-211 stacks them): sample b owns fixations start[b] .. start[b]+count[b]-1 of X / Y (pixels of the
 * origin_w x origin_h image) / T_start / T_end (ms).  target [B][T][1 + map_h*map_w] soft one-hot (index 0 = terminate),
 * duration [B][T] seconds, action_mask, duration_mask [B][T].  f64_div: evaluate the cell index and the ms -> s division in
 * float64 (numpy 1.x value-based casting: the reference's pinned numpy==1.19.2) instead of float32 (numpy >= 2). */
int sp_collate_targets(const float* X, const float* Y, const float* T_start, const float* T_end, const int64_t* start,
                       const int* count, const double* origin_w, const double* origin_h, int B, int T, int map_h, int map_w,
                       int f64_div, float* target, float* duration, float* action_mask, float* duration_mask, void* stream);
/* blur_sigma targets (AiR/dataset/dataset.py:144-147, OSIE/dataset/dataset.py:98-101, COCO_Search18/dataset/dataset.py:122-125):
 * in place on target [rows = B*T][1 + map_h*map_w] after sp_collate_targets -- every row whose target is a map cell becomes
 * scipy.ndimage.gaussian_filter(one-hot map, sigma) (mode reflect, truncate 4) divided by its sum; terminate rows are untouched.
 * float32 results agree with scipy to ~1 ulp (the normaliser is summed in a different order). */
int sp_blur_targets(float* target, int rows, int map_h, int map_w, double sigma, void* stream);

/* ------------------------------------------------------------------------------------------------
 * clip_grad_norm_ + Adam (L2 folded into the gradient) over one flat fp32 buffer.  AiR/train.py:116-117,200-202.
 * g_eff = g*gscale (gscale = 1/world_size after a sum all-reduce); total_norm = sqrt(*sumsq)*gscale;
 * coef = min(1, clip/(total_norm+1e-6)) (clip <= 0: no clipping); bc1 = 1-beta1^t, bc2 = 1-beta2^t from the host.
 * ---------------------------------------------------------------------------------------------- */
int sp_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float gscale, float clip,
                 float lr, float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2, void* stream);

#ifdef __cplusplus
}
#endif
#endif
