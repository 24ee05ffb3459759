// Piecewise-linear interpolation of one table at many points: numpy.interp of the reference's 'tabulated' engine (tabulated.py:31-36:
// redshift -> E(z), D_C(z) for catalogues of 1e7-1e9 objects).  One lane per sample: bisection in the table (40 002 rows of the DESI table
// = 320 KB per column, L2-resident), then numpy's own arithmetic, slope * (x - xp[j]) + fp[j] with separately rounded product and sum
// (no FMA contraction), so that results are bit-identical to numpy.interp.  Samples outside [xp[0], xp[n-1]] and NaN give NaN.
#include <hip/hip_runtime.h>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"

// hipcc contracts a * b + c into an fma by default, also through the __dmul_rn / __dadd_rn wrappers of its headers: not in this file
#pragma clang fp contract(off)

namespace {

// Every S-th knot sits in LDS: the first log2(n / S) bisection steps run there, the last log2(S) in one or two cache lines of the table.
__global__ __launch_bounds__(256) void interp_linear_kernel(const double* __restrict__ xp, const double* __restrict__ fp, long long n, int stride,
                                                            int ncoarse, const double* __restrict__ x, double* __restrict__ out, long long nx) {
    extern __shared__ double coarse[];   // coarse[i] = xp[i * stride]
    for (int i = threadIdx.x; i < ncoarse; i += 256) coarse[i] = xp[(long long)i * stride];
    __syncthreads();
    const double x0 = coarse[0], xn = xp[n - 1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += (long long)gridDim.x * blockDim.x) {
        const double v = x[i];
        double r = __builtin_nan("");
        if (v >= x0 && v <= xn) {
            if (v == xn) {
                r = fp[n - 1];
            } else {
                int clo = 0, chi = ncoarse;   // coarse[clo] <= v, and v < coarse[chi] if chi < ncoarse
                while (chi - clo > 1) {
                    const int mid = (clo + chi) >> 1;
                    if (coarse[mid] <= v) clo = mid; else chi = mid;
                }
                long long lo = (long long)clo * stride, hi = lo + stride < n - 1 ? lo + stride : n - 1;  // invariant: xp[lo] <= v < xp[hi]
                while (hi - lo > 1) {
                    const long long mid = (lo + hi) >> 1;
                    if (xp[mid] <= v) lo = mid; else hi = mid;
                }
                if (xp[lo] == v) {  // numpy.interp returns the knot value here ("avoid potential non-finite interpolation")
                    out[i] = fp[lo];
                    continue;
                }
                const double slope = (fp[lo + 1] - fp[lo]) / (xp[lo + 1] - xp[lo]);
                r = slope * (v - xp[lo]) + fp[lo];   // product and sum rounded separately (contract(off) above)
                // numpy.interp: if the result is NaN (slope or difference infinite) it retries from the right knot, then takes the common value
                if (r != r) {
                    r = slope * (v - xp[lo + 1]) + fp[lo + 1];
                    if (r != r && fp[lo] == fp[lo + 1]) r = fp[lo];
                }
            }
        }
        out[i] = r;
    }
}

}  // namespace

extern "C" int cp_interp_linear(const double* d_xp, const double* d_fp, long long n, const double* d_x, double* d_out, long long nx, int device,
                                void* stream) {
    if (n < 1 || nx < 0) return cp::fail(CP_EINVAL, "cp_interp_linear: need at least one table row and a non-negative sample count");
    if (nx == 0) return CP_OK;
    if (!d_xp || !d_fp || !d_x || !d_out) return cp::fail(CP_EINVAL, "cp_interp_linear: null pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_interp_linear: cannot select device %d", device);
    int stride = 32;                                    // at most 4096 coarse knots (32 KB of LDS)
    while ((n + stride - 1) / stride > 4096) stride *= 2;
    const int ncoarse = (int)((n + stride - 1) / stride);
    const long long blocks = (nx + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 256 * 8 ? blocks : 256 * 8);
    hipLaunchKernelGGL(interp_linear_kernel, dim3(grid), dim3(256), ncoarse * sizeof(double), static_cast<hipStream_t>(stream), d_xp, d_fp, n, stride,
                       ncoarse, d_x, d_out, nx);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_interp_linear: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

// ---- cubic splines at many points: one lane per query ------------------------------------------------------------------------------------
// The operator form of cp_spline_apply (a band of weights per query, shared by all rows) suits thousands of rows on a fixed set of queries.
// The opposite shape -- one or a few splines evaluated at 1e6-1e9 points, DistanceToRedshift for a catalogue (reference utils.py:275-316) --
// is a search plus one cubic per point: Hermite form on the interval [x_k, x_k+1] from the values and the knot first derivatives
// (the derivatives are cp_spline_apply with nu = 1 at the knots themselves, so any boundary condition of cp_spline_operator is available).
namespace {

__global__ __launch_bounds__(256) void spline_points_kernel(const double* __restrict__ xk, const double* __restrict__ y, const double* __restrict__ s,
                                                            long long n, int ncol, int stride, int ncoarse, const double* __restrict__ xq,
                                                            double* __restrict__ out, long long nq, int nu, int extrapolate) {
    extern __shared__ double coarse[];   // coarse[i] = xk[i * stride]
    for (int i = threadIdx.x; i < ncoarse; i += 256) coarse[i] = xk[(long long)i * stride];
    __syncthreads();
    const double x0 = coarse[0], xn = xk[n - 1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (long long)gridDim.x * blockDim.x) {
        const double v = xq[i];
        const bool inside = v >= x0 && v <= xn;
        if (!(inside || (extrapolate && v == v))) {   // outside without extrapolation, or NaN
            for (int c = 0; c < ncol; ++c) out[(long long)c * nq + i] = __builtin_nan("");
            continue;
        }
        long long lo;
        if (v <= x0) lo = 0;
        else if (v >= xn) lo = n - 2;
        else {
            int clo = 0, chi = ncoarse;
            while (chi - clo > 1) {
                const int mid = (clo + chi) >> 1;
                if (coarse[mid] <= v) clo = mid; else chi = mid;
            }
            lo = (long long)clo * stride;
            long long hi = lo + stride < n - 1 ? lo + stride : n - 1;
            while (hi - lo > 1) {
                const long long mid = (lo + hi) >> 1;
                if (xk[mid] <= v) lo = mid; else hi = mid;
            }
        }
        const double h = xk[lo + 1] - xk[lo], u = v - xk[lo];
        for (int c = 0; c < ncol; ++c) {
            const double* yc = y + (long long)c * n;
            const double* sc = s + (long long)c * n;
            // PPoly coefficients of the interval as scipy's CubicSpline forms them: t = (s_k + s_k+1 - 2 slope) / h
            const double slope = (yc[lo + 1] - yc[lo]) / h;
            const double t = (sc[lo] + sc[lo + 1] - 2. * slope) / h;
            const double c3 = t / h, c2 = (slope - sc[lo]) / h - t, c1 = sc[lo], c0 = yc[lo];
            double r;
            if (nu == 0) r = c0 + u * (c1 + u * (c2 + u * c3));
            else if (nu == 1) r = c1 + u * (2. * c2 + u * 3. * c3);
            else r = 2. * c2 + 6. * c3 * u;
            out[(long long)c * nq + i] = r;
        }
    }
}

}  // namespace

extern "C" int cp_spline_points(const double* d_xk, const double* d_y, const double* d_s, long long n, int ncol, const double* d_xq, double* d_out,
                                long long nq, int nu, int extrapolate, int device, void* stream) {
    if (n < 2 || ncol < 0 || nq < 0 || nu < 0 || nu > 2) return cp::fail(CP_EINVAL, "cp_spline_points: need n >= 2, nu in {0, 1, 2} and non-negative counts");
    if (nq == 0 || ncol == 0) return CP_OK;
    if (!d_xk || !d_y || !d_s || !d_xq || !d_out) return cp::fail(CP_EINVAL, "cp_spline_points: null pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_points: cannot select device %d", device);
    int stride = 32;
    while ((n + stride - 1) / stride > 4096) stride *= 2;
    const int ncoarse = (int)((n + stride - 1) / stride);
    const long long blocks = (nq + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 256 * 8 ? blocks : 256 * 8);
    hipLaunchKernelGGL(spline_points_kernel, dim3(grid), dim3(256), ncoarse * sizeof(double), static_cast<hipStream_t>(stream), d_xk, d_y, d_s, n, ncol, stride,
                       ncoarse, d_xq, d_out, nq, nu, extrapolate);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_points: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

// ---- cubic splines of many rows, each at ITS OWN queries: one lane per (row, query) --------------------------------------------------------
// peakaverage over a batch of cosmologies (reference bao_filter.py:565-574): the knots of the filter's two splines move with the cosmology's
// rs_drag ratio, so the data -> moved knots step is a spline through SHARED knots (the filter's log10 k) evaluated at queries that differ from row to
// row.  The rows' second derivatives at the knots come from cp_spline_rows_second_derivatives; here the four-term formula
//   a y_j + b y_{j+1} + ((a^3 - a) M_j + (b^3 - b) M_{j+1}) h_j^2 / 6,  a = (x_{j+1} - x) / h_j, b = (x - x_j) / h_j,
// with the interval found from a uniform first guess (the knots are a geomspace in log10: uniform to rounding) and set right against the knots
// themselves.  Outside the knots the cubic of the end interval is continued (scipy's extrapolate=True).
namespace {

#pragma clang fp contract(fast)
__global__ __launch_bounds__(256) void spline_rows_at_queries_kernel(const double* __restrict__ xk, const double* __restrict__ y, const double* __restrict__ m,
                                                                     long long nrows, int n, const double* __restrict__ xq, int nq, double* __restrict__ out,
                                                                     int transposed) {
    const double x0 = xk[0], inv_h = (double)(n - 1) / (xk[n - 1] - x0);
    const long long total = nrows * nq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // transposed: consecutive lanes are consecutive ROWS of one query (the stores are contiguous: knot-major output, the layout cp_spline_columns reads)
        const long long row = transposed ? i % nrows : i / nq;
        const int q = (int)(transposed ? i / nrows : i % nq);
        const double v = xq[row * nq + q];
        double r = __builtin_nan("");
        if (v == v) {
            int j = (int)((v - x0) * inv_h);
            j = j < 0 ? 0 : (j > n - 2 ? n - 2 : j);
            while (j > 0 && v < xk[j]) --j;
            while (j < n - 2 && v >= xk[j + 1]) ++j;
            const double h = xk[j + 1] - xk[j];
            const double a = (xk[j + 1] - v) / h, b = (v - xk[j]) / h;
            const double* yr = y + row * n;
            const double* mr = m + row * n;
            r = a * yr[j] + b * yr[j + 1] + ((a * a * a - a) * mr[j] + (b * b * b - b) * mr[j + 1]) * (h * h) / 6.;
        }
        out[transposed ? (long long)q * nrows + row : row * nq + q] = r;
    }
}

}  // namespace

extern "C" int cp_spline_rows_at_queries(const double* d_xk, const double* d_y, const double* d_m, long long nrows, int n, const double* d_xq, int nq,
                                         double* d_out, int transposed, int device, void* stream) {
    if (nrows < 0 || n < 2 || nq < 0) return cp::fail(CP_EINVAL, "cp_spline_rows_at_queries: bad sizes");
    if (nrows == 0 || nq == 0) return CP_OK;
    if (!d_xk || !d_y || !d_m || !d_xq || !d_out) return cp::fail(CP_EINVAL, "cp_spline_rows_at_queries: null pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_rows_at_queries: cannot select device %d", device);
    const long long blocks = (nrows * nq + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 256 * 16 ? blocks : 256 * 16);
    hipLaunchKernelGGL(spline_rows_at_queries_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), d_xk, d_y, d_m, nrows, n, d_xq, nq, d_out,
                       transposed);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_rows_at_queries: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
