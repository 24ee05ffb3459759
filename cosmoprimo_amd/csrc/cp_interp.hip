// Piecewise-linear interpolation of one table at many points: numpy.interp of the reference's 'tabulated' engine (tabulated.py:31-36:
// redshift -> E(z), D_C(z) for catalogues of 1e7-1e9 objects).  One lane per sample: bisection in the table (40 002 rows of the DESI table
// = 320 KB per column, L2-resident), then numpy's own arithmetic, slope * (x - xp[j]) + fp[j] with separately rounded product and sum
// (no FMA contraction), so that results are bit-identical to numpy.interp.  Samples outside [xp[0], xp[n-1]] and NaN give NaN.
#include <hip/hip_runtime.h>

#include <cmath>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_interp_table.h"

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

// samples outside [x0, xn] (or NaN) raise the plan's flag (cp_interp_table_apply on an irregular table: the bisection kernel has no flag)
__global__ __launch_bounds__(256) void flag_outside_kernel(const double* __restrict__ x, long long nx, double x0, double xn, int* flag) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += (long long)gridDim.x * blockDim.x) bad |= !(x[i] >= x0 && x[i] <= xn);
    if (bad) atomicOr(flag, 1);
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

// ---- the same interpolation on a table kept as a plan: (x, f) pairs, the interval GUESSED from the sample -----------------------------------
// A catalogue's redshifts come in no order: every bisection step above is a dependent load, ~10 of them out of LDS and 5 out of L2 per sample.
// Tables of background quantities are regular -- uniform in x, or, as the reference's data/desi.dat (0, then 40 001 redshifts from 1e-8 to 100),
// uniform in log x behind a few leading knots -- so the interval follows from the sample itself: index = first + (T(x) - a) b with T = identity or
// log2 (exponent + the hardware's single-precision log2 of the mantissa: the guess needs ~1e-5 of an interval, not 53 bits), then a walk to the
// exact interval, xp[j] <= x < xp[j + 1] (numpy's), which is 0 or 1 steps for a table that passed the host's check of the fit.  The two knots
// and two values of the interval are 32 contiguous bytes of the pair table.  Tables that fit neither law take the bisection.  Same arithmetic as
// above, bit-identical to numpy.interp.  Samples outside the table (and NaN) come out NaN and raise the plan's flag.
struct cp_interp_table {
    int device;
    long long n;
    double* d_xf;         // (n, 2): (x_i, f_i)
    double* d_x;          // (n): the knots alone (bisection of the irregular case)
    double* d_f;
    int* d_flag;          // raised by samples outside [x_0, x_{n-1}] / NaN
    int law;              // 0 none (bisection), 1 uniform in x, 2 uniform in log2 x
    long long first;      // the law holds from this knot on (the knots before it: bisection among them)
    double a, b;          // index = first + (T(x) - a) * b
    double x0, xn;        // the range of the table
};

namespace {

template <int LAW, typename real>      // real: the type of the samples and of the results (float: computed in double, rounded once, as the reference's cast of its result)
__global__ __launch_bounds__(256) void interp_table_kernel(const cpit::Pair* __restrict__ xf, long long n, long long first, double a, double b,
                                                           const real* __restrict__ x, real* __restrict__ out, long long nx, int* flag) {
    const double x0 = xf[0].x, xn = xf[n - 1].x;
    const double xfirst = xf[first].x;
    bool outside = false;
    constexpr int U = 4;      // samples of a thread in flight together (cp_interp_table.h: interp_samples)
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += U * stride) {
        double v[U], r[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = i + u * stride < nx;
            v[u] = (double)x[live[u] ? i + u * stride : i];
        }
        cpit::interp_samples<LAW, U>(xf, n, first, a, b, x0, xfirst, xn, v, live, r, &outside);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (live[u]) out[i + u * stride] = (real)r[u];
    }
    if (outside && flag) atomicOr(flag, 1);      // flag: null when the caller did not ask (cp_interp_table_apply with outside == NULL)
}

}  // namespace

extern "C" int cp_interp_table_create(cp_interp_table** table, long long n, const double* x, const double* f, int device) {
    if (!table) return cp::fail(CP_EINVAL, "cp_interp_table_create: null output");
    *table = nullptr;
    if (n < 1 || !x || !f) return cp::fail(CP_EINVAL, "cp_interp_table_create: need at least one (x, f) row");
    if (n > (1LL << 27)) return cp::fail(CP_EUNSUPPORTED, "cp_interp_table_create: %lld rows (at most 2^27: the table is staged as pairs on the host)", n);
    for (long long i = 1; i < n; ++i)
        if (!(x[i] >= x[i - 1])) return cp::fail(CP_EINVAL, "cp_interp_table_create: x must be ascending (row %lld)", i);
    cp_interp_table* t = new (std::nothrow) cp_interp_table();
    if (!t) return cp::fail(CP_ENOMEM, "cp_interp_table_create: out of host memory");
    t->device = device; t->n = n; t->law = 0; t->first = 0; t->a = 0.; t->b = 0.; t->x0 = x[0]; t->xn = x[n - 1];
    t->d_xf = t->d_x = t->d_f = nullptr; t->d_flag = nullptr;
    const cpit::Law law = cpit::find_law(x, n);      // uniform in x, uniform in log x behind a few leading knots, or neither (cp_interp_table.h)
    t->law = law.law; t->first = law.first; t->a = law.a; t->b = law.b;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) { delete t; return cp::fail(CP_EDEVICE, "cp_interp_table_create: cannot select device %d", device); }
    std::vector<double> pairs(2 * (size_t)n);
    for (long long i = 0; i < n; ++i) { pairs[2 * i] = x[i]; pairs[2 * i + 1] = f[i]; }
    const int zero = 0;
    bool ok = hipMalloc(&t->d_xf, 2 * n * sizeof(double)) == hipSuccess && hipMalloc(&t->d_x, n * sizeof(double)) == hipSuccess &&
              hipMalloc(&t->d_f, n * sizeof(double)) == hipSuccess && hipMalloc(&t->d_flag, sizeof(int)) == hipSuccess;
    ok = ok && hipMemcpy(t->d_xf, pairs.data(), 2 * n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(t->d_x, x, n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(t->d_f, f, n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(t->d_flag, &zero, sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (!ok) { (void)cp_interp_table_destroy(t); return cp::fail(CP_ENOMEM, "cp_interp_table_create: device allocation or upload failed"); }
    *table = t;
    return CP_OK;
}

extern "C" int cp_interp_table_law(const cp_interp_table* t, int* law, long long* first) {
    if (!t) return cp::fail(CP_EINVAL, "cp_interp_table_law: null table");
    if (law) *law = t->law;
    if (first) *first = t->first;
    return CP_OK;
}

namespace {

template <typename real>
int interp_table_apply(const cp_interp_table* t, const real* d_x, real* d_out, long long nx, int* outside, void* stream, const char* who) {
    if (!t) return cp::fail(CP_EINVAL, "%s: null table", who);
    if (nx < 0) return cp::fail(CP_EINVAL, "%s: negative sample count", who);
    if (outside) *outside = 0;
    if (nx == 0) return CP_OK;
    if (!d_x || !d_out) return cp::fail(CP_EINVAL, "%s: null pointer", who);
    if (t->law == 0 && !std::is_same<real, double>::value)
        return cp::fail(CP_EUNSUPPORTED, "%s: single-precision samples need a table uniform in x or in log x (this one is bisected: widen the samples)", who);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != t->device && hipSetDevice(t->device) != hipSuccess) return cp::fail(CP_EDEVICE, "%s: cannot select device %d", who, t->device);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    int st = CP_OK;
    // The plan has ONE flag word.  A call that does not ask (outside == NULL) hands the kernels no flag at all, so it leaves nothing behind for a
    // later call to trip over; a call that asks clears the word on its own stream in front of its kernels and reads it back behind them: calls
    // that ask must not run concurrently on two streams of one plan (the header says so), calls that do not ask may.
    int* flag = outside ? t->d_flag : nullptr;
    if (flag && hipMemsetAsync(flag, 0, sizeof(int), hs) != hipSuccess) {
        if (prev >= 0 && prev != t->device) (void)hipSetDevice(prev);
        return cp::fail(CP_EDEVICE, "%s: clearing the range flag failed", who);
    }
    if (t->law == 0) {
        // irregular table: bisection (NaN outside, no flag: raised below)
        st = cp_interp_linear(t->d_x, t->d_f, t->n, reinterpret_cast<const double*>(d_x), reinterpret_cast<double*>(d_out), nx, t->device, stream);
    } else {
        const long long blocks = (nx + 255) / 256;
        const unsigned grid = (unsigned)(blocks < 256 * 16 ? blocks : 256 * 16);
        const cpit::Pair* xf = reinterpret_cast<const cpit::Pair*>(t->d_xf);
        if (t->law == 1) hipLaunchKernelGGL((interp_table_kernel<1, real>), dim3(grid), dim3(256), 0, hs, xf, t->n, t->first, t->a, t->b, d_x, d_out, nx, flag);
        else hipLaunchKernelGGL((interp_table_kernel<2, real>), dim3(grid), dim3(256), 0, hs, xf, t->n, t->first, t->a, t->b, d_x, d_out, nx, flag);
        if (hipGetLastError() != hipSuccess) st = cp::fail(CP_EDEVICE, "%s: launch failed", who);
    }
    if (st == CP_OK && outside) {
        if (t->law == 0) {
            hipLaunchKernelGGL(flag_outside_kernel, dim3(1024), dim3(256), 0, hs, reinterpret_cast<const double*>(d_x), nx, t->x0, t->xn, t->d_flag);
            if (hipGetLastError() != hipSuccess) st = cp::fail(CP_EDEVICE, "%s: launch failed", who);
        }
        // the flag comes back with the stream drained: the caller of the reference gets its exception from the call itself (tabulated.py:33-34)
        int host = 0;
        if (st == CP_OK && (hipMemcpyAsync(&host, t->d_flag, sizeof(int), hipMemcpyDeviceToHost, hs) != hipSuccess || hipStreamSynchronize(hs) != hipSuccess))
            st = cp::fail(CP_EDEVICE, "%s: reading the range flag failed", who);
        if (st == CP_OK && host) *outside = 1;
    }
    if (prev >= 0 && prev != t->device) (void)hipSetDevice(prev);
    return st;
}

}  // namespace

extern "C" int cp_interp_table_apply(const cp_interp_table* t, const double* d_x, double* d_out, long long nx, int* outside, void* stream) {
    return interp_table_apply<double>(t, d_x, d_out, nx, outside, stream, "cp_interp_table_apply");
}

extern "C" int cp_interp_table_apply_f32(const cp_interp_table* t, const float* d_x, float* d_out, long long nx, int* outside, void* stream) {
    return interp_table_apply<float>(t, d_x, d_out, nx, outside, stream, "cp_interp_table_apply_f32");
}

extern "C" int cp_interp_table_destroy(cp_interp_table* t) {
    if (!t) return CP_OK;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != t->device) (void)hipSetDevice(t->device);
    if (t->d_xf) (void)hipFree(t->d_xf);
    if (t->d_x) (void)hipFree(t->d_x);
    if (t->d_f) (void)hipFree(t->d_f);
    if (t->d_flag) (void)hipFree(t->d_flag);
    if (prev >= 0 && prev != t->device) (void)hipSetDevice(prev);
    delete t;
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

// ONE spline of up to SPLINE_LDS_KNOTS knots (the distance tables of a cosmology: 119 and 400 knots; DistanceToRedshift: 2048): the knots and the four
// polynomial coefficients of every interval -- the same quotients as above, formed once per workgroup instead of once per sample (four IEEE divisions,
// two thirds of a sample's instructions) -- sit in LDS; a sample is a bisection there, four LDS reads and the Horner form: no table read from memory.
constexpr int SPLINE_LDS_KNOTS = 2048;

__global__ __launch_bounds__(256) void spline_points_lds_kernel(const double* __restrict__ xk, const double* __restrict__ y, const double* __restrict__ s,
                                                                int n, const double* __restrict__ xq, double* __restrict__ out, long long nq, int nu,
                                                                int extrapolate) {
    extern __shared__ double lds[];      // knots (n), then c0, c1, c2, c3 of the n - 1 intervals
    double* xs = lds;
    double* c0s = lds + n;
    double* c1s = c0s + (n - 1);
    double* c2s = c1s + (n - 1);
    double* c3s = c2s + (n - 1);
    for (int i = threadIdx.x; i < n; i += 256) xs[i] = xk[i];
    for (int i = threadIdx.x; i < n - 1; i += 256) {
        const double h = xk[i + 1] - xk[i];
        const double slope = (y[i + 1] - y[i]) / h;
        const double t = (s[i] + s[i + 1] - 2. * slope) / h;
        c0s[i] = y[i]; c1s[i] = s[i]; c2s[i] = (slope - s[i]) / h - t; c3s[i] = t / h;
    }
    __syncthreads();
    const double x0 = xs[0], xn = xs[n - 1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (long long)gridDim.x * blockDim.x) {
        const double v = xq[i];
        const bool inside = v >= x0 && v <= xn;
        double r = __builtin_nan("");
        if (inside || (extrapolate && v == v)) {
            int lo = 0;
            if (v >= xn) lo = n - 2;
            else if (v > x0) {
                int hi = n - 1;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (xs[mid] <= v) lo = mid; else hi = mid;
                }
            }
            const double u = v - xs[lo];
            const double c0 = c0s[lo], c1 = c1s[lo], c2 = c2s[lo], c3 = c3s[lo];
            if (nu == 0) r = c0 + u * (c1 + u * (c2 + u * c3));
            else if (nu == 1) r = c1 + u * (2. * c2 + u * 3. * c3);
            else r = 2. * c2 + 6. * c3 * u;
        }
        out[i] = r;
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
    if (ncol == 1 && n <= SPLINE_LDS_KNOTS && nq >= 65536) {      // a catalogue through one spline: the spline in LDS (a few samples: not worth filling it per workgroup)
        const size_t lds = (size_t)(n + 4 * (n - 1)) * sizeof(double);
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spline_points_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const unsigned lgrid = (unsigned)(blocks < 256 * 4 ? blocks : 256 * 4);
        hipLaunchKernelGGL(spline_points_lds_kernel, dim3(lgrid), dim3(256), lds, static_cast<hipStream_t>(stream), d_xk, d_y, d_s, (int)n, d_xq, d_out, nq, nu,
                           extrapolate);
        const hipError_t le = hipGetLastError();
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
        if (le != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_points: launch failed: %s", hipGetErrorString(le));
        return CP_OK;
    }
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
            double guess = (v - x0) * inv_h;      // clamped as a double: the conversion of an infinite or huge value to int is undefined
            guess = guess > 0. ? guess : 0.;
            int j = guess < (double)(n - 2) ? (int)guess : n - 2;
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
