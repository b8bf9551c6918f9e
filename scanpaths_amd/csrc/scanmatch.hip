// ScanMatch scoring (Cristino et al. 2010) on the device -- SURVEY.md §8 row f2.
// Reference behaviour: utils/evaltools/scanmatch.py:88-197 (substitution matrix :88-103, grid :105-115, fixations ->
// symbol string :117-135, Needleman-Wunsch :137-197); callers utils/evaluation.py:22-64,198-235,361-559 use the score only.
//
// All arithmetic is IEEE float64 with the reference's operation order, so scores are BIT-EXACT with numpy:
// max() is exact and order-free, every cell is max(diag + S, left + gap, up + gap), the score is max(F)/(max(S)*max(n,m)).
//
// nw_score_kernel: one wavefront per sequence pair.  The DP table is swept in 64-column strips; inside a strip lane l owns
// column base+l+1 and at time t fills row t-l+1 (anti-diagonal wavefront), taking F[i][j-1] / F[i-1][j-1] from lane l-1 with
// two wave shuffles and F[i-1][j] from its own previous step -- the table itself is never materialised.  The strip's last
// column is parked in LDS for lane 0 of the next strip.  Thousands of pairs (validation: every sampled scanpath against
// every human scanpath) run concurrently; HBM traffic is the two symbol strings and one 8-byte score per pair.
#include "common.h"

namespace {

constexpr int SM_MAXLEN = 4096;   // symbols per sequence (LDS: 8 B column cell + 4 B symbol each)

__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void submatrix_kernel(int Xbin, int Ybin, double thr, double* __restrict__ sub,
                                                        double* __restrict__ maxsub) {
    // farthest pair of bins = opposite corners; sqrt is correctly rounded, so this equals numpy.max(mat)
    const double dx = (double)(Xbin - 1), dy = (double)(Ybin - 1);
    const double mx = sqrt(dx * dx + dy * dy);
    const int nb = Xbin * Ybin;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nb * nb; i += gridDim.x * 256) {
        const int a = i / nb, b = i % nb;
        const double ex = (double)(a % Xbin - b % Xbin), ey = (double)(a / Xbin - b / Xbin);
        sub[i] = fabs(sqrt(ex * ex + ey * ey) - mx) - (mx - thr);
    }
    // max(S) = |0 - mx| - (mx - thr) on the diagonal
    if (blockIdx.x == 0 && threadIdx.x == 0) *maxsub = fabs(0.0 - mx) - (mx - thr);
}

__device__ __forceinline__ int symbol_of(double x, double y, int Xres, int Yres, int Xbin, int Ybin, const int* mask) {
    if (x < 0) x = 0;
    if (y < 0) y = 0;
    if (x >= Xres) x = Xres - 1;
    if (y >= Yres) y = Yres - 1;
    const long long xi = (long long)x, yi = (long long)y;       // truncation, as int() on a non-negative float
    if (mask) return mask[yi * Xres + xi];
    const int xb = (int)((double)xi * ((double)Xbin / (double)Xres));
    const int yb = (int)((double)yi * ((double)Ybin / (double)Yres));
    return yb * Xbin + xb;
}

// one thread per scanpath; seq == nullptr: lengths only
__global__ __launch_bounds__(64) void sequences_kernel(const double* __restrict__ fix, int ncol, const int64_t* __restrict__ start,
                                                       const int* __restrict__ count, int nsp, int Xres, int Yres, int Xbin,
                                                       int Ybin, double offx, double offy, double tempbin,
                                                       const int* __restrict__ mask, int ld, int* __restrict__ seq,
                                                       int* __restrict__ seq_len) {
    const int sp = blockIdx.x * 64 + threadIdx.x;
    if (sp >= nsp) return;
    const double* f = fix + start[sp] * ncol;
    long long len = 0;
    for (int k = 0; k < count[sp]; ++k) {
        const int sym = symbol_of(f[k * ncol] - offx, f[k * ncol + 1] - offy, Xres, Yres, Xbin, Ybin, mask);
        long long reps = 1;
        if (tempbin != 0.0) {
            double d = f[k * ncol + 2];
            if (d < 0) d = 0;
            reps = (long long)rint((double)(long long)d / tempbin);     // numpy.round: half to even
            if (reps < 0) reps = 0;
        }
        if (seq) {                                    // write what fits; the count below is exact however long the fixation lasts
            const long long room = len < ld ? min(reps, (long long)ld - len) : 0;
            for (long long r = 0; r < room; ++r) seq[(int64_t)sp * ld + len + r] = sym;
        }
        len += reps;                                  // (a heavy-tailed sampled duration must not spin a thread for 1e10 iterations)
    }
    seq_len[sp] = (int)min(len, (long long)0x7fffffff);
}

// LONG = false: strip column and the A symbols live in LDS (sequences up to SM_MAXLEN symbols, every validation-shaped pair).
// LONG = true : the strip column lives in a per-pair global scratch row (colws [npairs][ldA + 1], volatile accesses: the
//               cross-lane hand-over goes through memory) and A is read in place -- no length limit beyond the workspace; used
//               for the rare heavy-tailed sampled durations of the RL phase (the reference's python DP just gets slower).
template <bool LONG>
__global__ __launch_bounds__(64) void nw_score_kernel(const int* __restrict__ seqA, const int* __restrict__ lenA, int ldA,
                                                      const int* __restrict__ seqB, const int* __restrict__ lenB, int ldB,
                                                      const int* __restrict__ pairs, const double* __restrict__ sub, int nb,
                                                      const double* __restrict__ maxsub, double gap, double* __restrict__ scores,
                                                      double* colws) {
    __shared__ double col_s[LONG ? 1 : SM_MAXLEN + 1];
    __shared__ int asym_s[LONG ? 1 : SM_MAXLEN];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int ia = pairs ? pairs[2 * p] : p, ib = pairs ? pairs[2 * p + 1] : p;
    const int n = lenA[ia], m = lenB[ib];
    const int* A = seqA + (int64_t)ia * ldA;
    const int* B = seqB + (int64_t)ib * ldB;
    volatile double* col = LONG ? colws + (int64_t)p * (ldA + 1) : col_s;
    const int* asym = LONG ? A : asym_s;
    if (!LONG)
        for (int i = lane; i < n; i += 64) asym_s[i] = A[i];
    for (int i = lane; i <= n; i += 64) col[i] = gap * (double)(i + 1);      // F[i][0]
    __syncthreads();
    // borders: F[i][0] = gap*(i+1), F[0][j] = gap*(j+1) -- monotone, so their maximum sits at an end
    double fmx = fmax(gap, fmax(gap * (double)(n + 1), gap * (double)(m + 1)));
    for (int base = 0; base < m; base += 64) {
        const int j = base + lane + 1;
        const bool col_ok = j <= m;
        const double* srow_b = sub + (col_ok ? B[j - 1] : 0);
        double cur = gap * (double)(j + 1);          // F[0][j]
        double prev = cur;
        const int last = min(63, m - base - 1);      // last active lane of this strip
        for (int t = 0; t < n + last + 1; ++t) {
            const int i = t - lane + 1;
            double lc = __shfl_up(cur, 1, 64), lp = __shfl_up(prev, 1, 64);
            const bool row_ok = i >= 1 && i <= n;
            if (lane == 0 && row_ok) {
                lc = col[i];
                lp = col[i - 1];
            }
            if (row_ok && col_ok) {
                const double v = fmax(lp + srow_b[(int64_t)asym[i - 1] * nb], fmax(lc + gap, cur + gap));
                prev = cur;
                cur = v;
                fmx = fmax(fmx, v);
                // becomes F[i][base+64] for the next strip; only full strips have a successor, so lane 63 writes col[i]
                // 62 steps after lane 0 read it
                if (lane == 63 && base + 64 < m) col[i] = v;
            }
        }
        if (lane == 0) col[0] = gap * (double)(base + 64 + 1);   // F[0][base+64]
        __syncthreads();
    }
    fmx = wave_max_d(fmx);
    if (lane == 0) scores[p] = fmx / (*maxsub * (double)max(n, m));
}

// full table + traceback for the single-pair API (scanmatch.py:137-197 returns score, alignment and F^T).  One thread.
__global__ void nw_align_kernel(const int* __restrict__ A, int n, const int* __restrict__ B, int m, const double* __restrict__ sub,
                                int nb, const double* __restrict__ maxsub, double gap, double* __restrict__ F /*[(n+1)][(m+1)]*/,
                                double* __restrict__ Ft /*[(m+1)][(n+1)]*/, double* __restrict__ align /*[(n+m)][2]*/,
                                int* __restrict__ nalign, double* __restrict__ score) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int W = m + 1;
    double fmx = -INFINITY;
    for (int i = 0; i <= n; ++i) F[(int64_t)i * W] = gap * (double)(i + 1);
    for (int j = 0; j <= m; ++j) F[j] = gap * (double)(j + 1);
    for (int i = 1; i <= n; ++i)
        for (int j = 1; j <= m; ++j)
            F[(int64_t)i * W + j] = fmax(F[(int64_t)(i - 1) * W + j - 1] + sub[(int64_t)A[i - 1] * nb + B[j - 1]],
                                         fmax(F[(int64_t)i * W + j - 1] + gap, F[(int64_t)(i - 1) * W + j] + gap));
    for (int i = 0; i <= n; ++i)
        for (int j = 0; j <= m; ++j) {
            const double v = F[(int64_t)i * W + j];
            fmx = fmax(fmx, v);
            Ft[(int64_t)j * (n + 1) + i] = v;
        }
    *score = fmx / (*maxsub * (double)max(n, m));
    // traceback: diagonal, then the row above (gap in B), else the column to the left (gap in A); emitted end-first
    int i = n, j = m, step = 0;
    while (i > 0 && j > 0) {
        const double s = F[(int64_t)i * W + j];
        if (s == F[(int64_t)(i - 1) * W + j - 1] + sub[(int64_t)A[i - 1] * nb + B[j - 1]]) {
            align[2 * step] = A[i - 1]; align[2 * step + 1] = B[j - 1]; --i; --j;
        } else if (s == F[(int64_t)(i - 1) * W + j] + gap) {
            align[2 * step] = A[i - 1]; align[2 * step + 1] = -1.0; --i;
        } else {
            align[2 * step] = -1.0; align[2 * step + 1] = B[j - 1]; --j;
        }
        ++step;
    }
    while (i > 0) { align[2 * step] = A[i - 1]; align[2 * step + 1] = -1.0; --i; ++step; }
    while (j > 0) { align[2 * step] = -1.0; align[2 * step + 1] = B[j - 1]; --j; ++step; }
    // reverse in place -> start-first
    for (int a = 0, b = step - 1; a < b; ++a, --b) {
        const double x0 = align[2 * a], x1 = align[2 * a + 1];
        align[2 * a] = align[2 * b]; align[2 * a + 1] = align[2 * b + 1];
        align[2 * b] = x0; align[2 * b + 1] = x1;
    }
    *nalign = step;
}

}  // namespace

extern "C" int sp_scanmatch_max_len(void) { return SM_MAXLEN; }

extern "C" int sp_scanmatch_submatrix(int Xbin, int Ybin, double threshold, double* sub, double* maxsub, void* stream) {
    if (!sub || !maxsub) return SP_ENULL;
    if (Xbin < 1 || Ybin < 1 || (int64_t)Xbin * Ybin > 32768) return SP_EINVAL;
    const int64_t n = (int64_t)Xbin * Ybin * Xbin * Ybin;
    hipLaunchKernelGGL(submatrix_kernel, dim3((unsigned)std::min<int64_t>(sp_cdiv(n, 256), 4096)), dim3(256), 0,
                       (hipStream_t)stream, Xbin, Ybin, threshold, sub, maxsub);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_scanmatch_sequences(const double* fix, int ncol, const int64_t* start, const int* count, int nsp, int Xres,
                                      int Yres, int Xbin, int Ybin, double off_x, double off_y, double tempbin, const int* mask,
                                      int ld, int* seq, int* seq_len, void* stream) {
    if (!fix || !start || !count || !seq_len) return SP_ENULL;
    if (nsp < 1 || ncol < 2 || (tempbin != 0.0 && ncol < 3) || Xres < 1 || Yres < 1 || Xbin < 1 || Ybin < 1 || (seq && ld < 1))
        return SP_EINVAL;
    hipLaunchKernelGGL(sequences_kernel, dim3((nsp + 63) / 64), dim3(64), 0, (hipStream_t)stream, fix, ncol, start, count, nsp,
                       Xres, Yres, Xbin, Ybin, off_x, off_y, tempbin, mask, ld, seq, seq_len);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_scanmatch_score(const int* seqA, const int* lenA, int ldA, const int* seqB, const int* lenB, int ldB,
                                  const int* pairs, int npairs, const double* sub, int nb, const double* maxsub, double gap,
                                  double* scores, void* stream) {
    if (!seqA || !lenA || !seqB || !lenB || !sub || !maxsub || !scores) return SP_ENULL;
    if (npairs < 1 || nb < 1 || ldA < 1 || ldB < 1 || ldA > SM_MAXLEN || ldB > SM_MAXLEN) return SP_EINVAL;
    hipLaunchKernelGGL(nw_score_kernel<false>, dim3(npairs), dim3(64), 0, (hipStream_t)stream, seqA, lenA, ldA, seqB, lenB, ldB,
                       pairs, sub, nb, maxsub, gap, scores, (double*)nullptr);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int64_t sp_scanmatch_score_long_workspace(int ldA, int npairs) {
    return (ldA < 1 || npairs < 1) ? 0 : (int64_t)npairs * ((int64_t)ldA + 1) * (int64_t)sizeof(double);
}

extern "C" int sp_scanmatch_score_long(const int* seqA, const int* lenA, int ldA, const int* seqB, const int* lenB, int ldB,
                                       const int* pairs, int npairs, const double* sub, int nb, const double* maxsub, double gap,
                                       double* scores, void* workspace, void* stream) {
    if (!seqA || !lenA || !seqB || !lenB || !sub || !maxsub || !scores || !workspace) return SP_ENULL;
    if (npairs < 1 || nb < 1 || ldA < 1 || ldB < 1) return SP_EINVAL;
    hipLaunchKernelGGL(nw_score_kernel<true>, dim3(npairs), dim3(64), 0, (hipStream_t)stream, seqA, lenA, ldA, seqB, lenB, ldB,
                       pairs, sub, nb, maxsub, gap, scores, (double*)workspace);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

extern "C" int sp_scanmatch_align(const int* A, int n, const int* B, int m, const double* sub, int nb, const double* maxsub,
                                  double gap, double* F_work, double* Ft, double* align, int* nalign, double* score,
                                  void* stream) {
    if (!sub || !maxsub || !F_work || !Ft || !align || !nalign || !score || (n > 0 && !A) || (m > 0 && !B)) return SP_ENULL;
    if (n < 0 || m < 0 || nb < 1) return SP_EINVAL;
    hipLaunchKernelGGL(nw_align_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, A, n, B, m, sub, nb, maxsub, gap, F_work, Ft,
                       align, nalign, score);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
