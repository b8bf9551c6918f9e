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
