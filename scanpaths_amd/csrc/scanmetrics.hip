// String-edit distance (SED) and scaled time-delay-embedding similarity (STDE) of scanpath pairs on the device --
// SURVEY.md §8 row f2, next to ScanMatch in the reference's evaluation (utils/evaluation.py:68-72,239-243).
// Reference behaviour: utils/evaltools/visual_attention_metrics.py:205-218 (euclidean_distance), :236-318 (grid string +
// Levenshtein), :332-441 (time-delay embedding, 'Mean' mode, all window lengths, mean of exp(-d)).
//
// One thread per pair (scanpaths have <= 64 fixations; validation scores ~10^5 pairs at once).  SED is integer work and
// bit-exact.  STDE follows numpy's float64 evaluation order exactly -- products and sums are NOT contracted into FMAs
// (__dmul_rn / __dadd_rn), a window's distances are added with numpy's pairwise-summation scheme (sequential below 8
// terms, 8 interleaved partial sums above), means are left-to-right python sums -- so the only possible difference to the
// reference is the last bit of exp().
#include "common.h"

namespace {

constexpr int MAXFIX = 64;

__device__ __forceinline__ int floordiv(int a, int b) {
    int q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}

// numpy add.reduce over a contiguous float64 vector of n <= 128 terms; term(i) supplies element i
template <typename F>
__device__ __forceinline__ double numpy_sum(int n, F term) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = __dadd_rn(res, term(i));
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = term(j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __dadd_rn(r[j], term(i + j));
    }
    double res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                           __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    for (; i < n; ++i) res = __dadd_rn(res, term(i));
    return res;
}

__global__ __launch_bounds__(64) void sed_stde_kernel(const double* __restrict__ fix, int ncol, const int64_t* __restrict__ start,
                                                      const int* __restrict__ count, const int* __restrict__ pairs, int npairs,
                                                      int height, int width, int ngrid, double max_dim, int* __restrict__ sed,
                                                      double* __restrict__ stde) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= npairs) return;
    const int ih = pairs[2 * p], is = pairs[2 * p + 1];
    const int nh = count[ih], ns = count[is];
    const double* fh = fix + start[ih] * ncol;
    const double* fs = fix + start[is] * ncol;
    // ---- SED: Levenshtein over the n x n grid strings ----
    if (sed) {
        const int ws = width / ngrid, hs = height / ngrid;
        int symh[MAXFIX], row[MAXFIX + 1];
        for (int i = 0; i < nh; ++i) symh[i] = floordiv((int)fh[i * ncol], ws) + floordiv((int)fh[i * ncol + 1], hs) * ngrid;
        // rows indexed by the simulated string, columns by the human string (the distance is symmetric)
        for (int j = 0; j <= nh; ++j) row[j] = j;
        for (int i = 1; i <= ns; ++i) {
            const int c = floordiv((int)fs[(i - 1) * ncol], ws) + floordiv((int)fs[(i - 1) * ncol + 1], hs) * ngrid;
            int diag = row[0];
            row[0] = i;
            for (int j = 1; j <= nh; ++j) {
                const int up = row[j];
                row[j] = min(min(up + 1, row[j - 1] + 1), diag + (c != symh[j - 1] ? 1 : 0));
                diag = up;
            }
        }
        sed[p] = row[nh];
    }
    // ---- STDE ----
    if (stde) {
        double hx[MAXFIX], hy[MAXFIX], sx[MAXFIX], sy[MAXFIX];
        for (int i = 0; i < nh; ++i) { hx[i] = fh[i * ncol] / max_dim; hy[i] = fh[i * ncol + 1] / max_dim; }
        for (int i = 0; i < ns; ++i) { sx[i] = fs[i * ncol] / max_dim; sy[i] = fs[i * ncol + 1] / max_dim; }
        const int kmax = min(nh, ns);
        if (kmax == 0) {
            stde[p] = NAN;          // the reference returns None
            return;
        }
        double simsum = 0.0;
        for (int k = 1; k <= kmax; ++k) {
            double dsum = 0.0;
            const int nsw = ns - k + 1, nhw = nh - k + 1;
            for (int s0 = 0; s0 < nsw; ++s0) {
                double best = INFINITY;
                for (int h0 = 0; h0 < nhw; ++h0) {
                    const double d = numpy_sum(k, [&](int i) {
                        const double dx = sx[s0 + i] - hx[h0 + i], dy = sy[s0 + i] - hy[h0 + i];
                        return __dsqrt_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)));
                    });
                    best = d < best ? d : best;
                }
                dsum = __dadd_rn(dsum, best / (double)k);
            }
            simsum = __dadd_rn(simsum, exp(-(dsum / (double)nsw)));
        }
        stde[p] = simsum / (double)kmax;
    }
}

// ---- MultiMatch (Jarodzka et al. 2010 / Dewhurst et al. 2012): the five similarities the reference takes from the third-party
// multimatch_gaze.docomparison per (ground truth, other) pair (AiR/utils/evaluation.py:7,44-45,213; 5 of the 10 validation columns).
// One thread per pair: saccade vectors -> matrix of vector differences -> cheapest monotone alignment from the first to the last
// saccade pair (moves right / down / diagonal in THAT order of preference, cost = the entered cell, strict <) -> medians of the
// vector / direction / length / position / duration differences along the path -> normalisation.  Fewer than 3 fixations: five NaNs
// (the rule by which the reference drops a pair).  float64 in numpy's evaluation order (no FMA contraction), so the alignment path
// and four of the five values are bit-identical with the host restatement utils/evaltools/multimatch.py (its checker); the direction
// value depends on atan2's last bit.  The DP keeps one row of costs and 2 bits of back-pointer per cell in per-thread scratch.
constexpr int MMSAC = MAXFIX - 1;                 // saccades per scanpath
__device__ __forceinline__ double mm_hyp(double a, double b) { return __dsqrt_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(b, b))); }
__device__ double mm_median(double* v, int n) {   // numpy.median: sort, mean of the two middle values for an even count
    for (int i = 1; i < n; ++i) {
        const double x = v[i];
        int j = i - 1;
        while (j >= 0 && v[j] > x) {
            v[j + 1] = v[j];
            --j;
        }
        v[j + 1] = x;
    }
    return (n & 1) ? v[n / 2] : __dadd_rn(v[n / 2 - 1], v[n / 2]) / 2.0;
}

__global__ __launch_bounds__(64) void multimatch_kernel(const double* __restrict__ fix, int ncol, const int64_t* __restrict__ start,
                                                        const int* __restrict__ count, const int* __restrict__ pairs, int npairs,
                                                        double screen_w, double screen_h, double* __restrict__ out) {
    const int p = blockIdx.x * 64 + threadIdx.x;
    if (p >= npairs) return;
    const int i1 = pairs[2 * p], i2 = pairs[2 * p + 1];
    const int n1 = count[i1], n2 = count[i2];
    double* o = out + 5 * (int64_t)p;
    if (n1 < 3 || n2 < 3) {
        for (int k = 0; k < 5; ++k) o[k] = NAN;
        return;
    }
    const double* f1 = fix + start[i1] * ncol;
    const double* f2 = fix + start[i2] * ncol;
    const int n = n1 - 1, m = n2 - 1;             // saccades
    auto lenx = [&](const double* f, int i) { return f[(i + 1) * ncol] - f[i * ncol]; };
    auto leny = [&](const double* f, int i) { return f[(i + 1) * ncol + 1] - f[i * ncol + 1]; };
    auto cost = [&](int i, int j) { return mm_hyp(lenx(f1, i) - lenx(f2, j), leny(f1, i) - leny(f2, j)); };
    double row[MMSAC];                            // D of the current row (left part) / the previous row (right part)
    unsigned char prev[MMSAC * MMSAC];            // 0: came from (i, j-1), 1: (i-1, j), 2: (i-1, j-1)
    for (int i = 0; i < n; ++i) {
        double diag = 0.0;                        // D[i-1][j-1]
        for (int j = 0; j < m; ++j) {
            const double up = row[j];             // D[i-1][j] (garbage for i == 0: not read)
            if (i == 0 && j == 0) {
                row[0] = 0.0;
                diag = up;
                continue;
            }
            const double c = cost(i, j);
            double best = INFINITY;
            unsigned char arg = 0;
            if (j > 0) {
                const double v = __dadd_rn(row[j - 1], c);
                if (v < best) { best = v; arg = 0; }
            }
            if (i > 0) {
                const double v = __dadd_rn(up, c);
                if (v < best) { best = v; arg = 1; }
            }
            if (i > 0 && j > 0) {
                const double v = __dadd_rn(diag, c);
                if (v < best) { best = v; arg = 2; }
            }
            row[j] = best;
            prev[i * MMSAC + j] = arg;
            diag = up;
        }
    }
    double vec[2 * MMSAC], ang[2 * MMSAC], ln[2 * MMSAC], pos[2 * MMSAC], dur[2 * MMSAC];
    const double PI = 3.141592653589793;
    int cnt = 0, i = n - 1, j = m - 1;
    while (true) {
        const double ax = lenx(f1, i), ay = leny(f1, i), bx = lenx(f2, j), by = leny(f2, j);
        vec[cnt] = mm_hyp(ax - bx, ay - by);
        double t0 = atan2(ay, ax), t1 = atan2(by, bx);
        t0 = t0 < 0 ? PI + (PI + t0) : t0;
        t1 = t1 < 0 ? PI + (PI + t1) : t1;
        const double d = fabs(t0 - t1);
        ang[cnt] = d > PI ? 2 * PI - d : d;
        ln[cnt] = fabs(mm_hyp(ax, ay) - mm_hyp(bx, by));
        pos[cnt] = mm_hyp(f1[i * ncol] - f2[j * ncol], f1[i * ncol + 1] - f2[j * ncol + 1]);
        const double d1 = f1[i * ncol + 2], d2 = f2[j * ncol + 2];
        dur[cnt] = fabs(d1 - d2) / fmax(d1, d2);
        ++cnt;
        if (i == 0 && j == 0) break;
        const unsigned char a = prev[i * MMSAC + j];
        if (a == 0) --j;
        else if (a == 1) --i;
        else { --i; --j; }
    }
    const double diagl = __dsqrt_rn(__dadd_rn(__dmul_rn(screen_w, screen_w), __dmul_rn(screen_h, screen_h)));
    o[0] = 1 - mm_median(vec, cnt) / (2 * diagl);
    o[1] = 1 - mm_median(ang, cnt) / PI;
    o[2] = 1 - mm_median(ln, cnt) / diagl;
    o[3] = 1 - mm_median(pos, cnt) / diagl;
    o[4] = 1 - mm_median(dur, cnt);
}

}  // namespace

extern "C" int sp_scan_max_fixations(void) { return MAXFIX; }

extern "C" int sp_scan_sed_stde(const double* fix, int ncol, const int64_t* start, const int* count, const int* pairs, int npairs,
                                int height, int width, int ngrid, double max_dim, int* sed, double* stde, void* stream) {
    if (!fix || !start || !count || !pairs || (!sed && !stde)) return SP_ENULL;
    if (npairs < 1 || ncol < 2 || ngrid < 1 || (sed && (width / ngrid < 1 || height / ngrid < 1)) || !(max_dim > 0)) return SP_EINVAL;
    hipLaunchKernelGGL(sed_stde_kernel, dim3((npairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, fix, ncol, start, count, pairs,
                       npairs, height, width, ngrid, max_dim, sed, stde);
    SP_LAUNCH_CHECK();
    return SP_OK;
}

// MultiMatch of npairs scanpath pairs; fix [total fixations][ncol >= 3] = (x, y, duration, ...), out [npairs][5] =
// (vector, direction, length, position, duration) similarity, NaN x 5 for a pair with a scanpath of fewer than 3 fixations
extern "C" int sp_scan_multimatch(const double* fix, int ncol, const int64_t* start, const int* count, const int* pairs, int npairs,
                                  double screen_w, double screen_h, double* out, void* stream) {
    if (!fix || !start || !count || !pairs || !out) return SP_ENULL;
    if (npairs < 1 || ncol < 3 || !(screen_w > 0) || !(screen_h > 0)) return SP_EINVAL;
    hipLaunchKernelGGL(multimatch_kernel, dim3((npairs + 63) / 64), dim3(64), 0, (hipStream_t)stream, fix, ncol, start, count, pairs,
                       npairs, screen_w, screen_h, out);
    SP_LAUNCH_CHECK();
    return SP_OK;
}
