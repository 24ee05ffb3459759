// cp_interp_table.h -- the law of a table's knots and the interpolation of ONE sample through it (cp_interp_table, cp_interp.hip): plain C++,
// compiled by hipcc into the kernel and by g++ for the CPU tests (tests/host_emu/emu_interp.cpp, under the sanitizers as well).  numpy.interp's
// arithmetic (slope * (x - xp[j]) + fp[j], product and sum rounded separately: the including file switches floating-point contraction off), the
// interval found from a guess -- index = first + (T(x) - a) b, T = identity or log2 -- and a walk to numpy's interval xp[j] <= x < xp[j + 1].
#pragma once

#include <cmath>

// a * b + c stays a product and a sum here (numpy's arithmetic): no contraction into an fma, whatever the including file allows (g++: -ffp-contract=off)
#if defined(__clang__)
#pragma clang fp contract(off)
#endif

#if defined(__HIPCC__)
#define CPIT_HD __host__ __device__ __forceinline__
#else
#define CPIT_HD inline
#endif

namespace cpit {

struct alignas(16) Pair {
    double x, y;
};

struct Law {
    int law;              // 0 none (bisection), 1 uniform in x, 2 uniform in log2 x
    long long first;      // the law holds from this knot on (the knots before it: bisection among them)
    double a, b;          // index = first + (T(x) - a) * b
};

// log2 of a positive finite double to single precision: exponent + log2 of the mantissa (the guess needs ~1e-5 of an interval, not 53 bits)
CPIT_HD double log2_guess(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int e = __builtin_amdgcn_frexp_exp(v);
    const float m = (float)__builtin_amdgcn_frexp_mant(v);
    return (double)e + (double)__builtin_amdgcn_logf(m);
#else
    int e = 0;
    const float m = (float)std::frexp(v, &e);
    return (double)e + (double)std::log2(m);
#endif
}

// the knots from `first` on against first + (T(x) - a) b: accepted when no knot is guessed further than a quarter of an interval from where it
// is (the single-precision log2 of the device adds ~1e-3 of an interval to that; the walk below makes any guess exact, the check keeps it short)
inline bool fit_law(const double* x, long long n, int law, long long first, double* a, double* b) {
    if (n - first < 3) return false;
    auto T = [&](double v) { return law == 2 ? std::log2(v) : v; };
    if (law == 2 && !(x[first] > 0.)) return false;
    const double t0 = T(x[first]), t1 = T(x[n - 1]);
    if (!(t1 > t0) || !std::isfinite(t0) || !std::isfinite(t1)) return false;
    *a = t0;
    *b = (double)(n - 1 - first) / (t1 - t0);
    for (long long i = first; i < n; ++i) {
        const double g = (T(x[i]) - *a) * *b;
        if (!(std::fabs(g - (double)(i - first)) <= 0.25)) return false;
    }
    return true;
}

// uniform in x, else uniform in log2 x from the first positive knot that starts a regular run (at most 8 leading knots are set aside), else none
inline Law find_law(const double* x, long long n) {
    Law L{0, 0, 0., 0.};
    double a = 0., b = 0.;
    if (fit_law(x, n, 1, 0, &a, &b)) return Law{1, 0, a, b};
    for (long long first = 0; first < 8 && first < n - 3; ++first)
        if (x[first] > 0. && fit_law(x, n, 2, first, &a, &b)) return Law{2, first, a, b};
    return L;
}

// one sample through a table with a law (LAW 1 or 2): numpy.interp(v, x, f), NaN and *outside = true for v outside [x_0, x_{n-1}] or NaN
template <int LAW>
CPIT_HD double interp_sample(const Pair* __restrict__ xf, long long n, long long first, double a, double b, double x0, double xfirst, double xn, double v,
                             bool* outside) {
    if (!(v >= x0 && v <= xn)) {
        *outside = true;
        return __builtin_nan("");
    }
    if (v == xn) return xf[n - 1].y;
    long long lo;
    Pair k0, k1;
    if (first > 0 && v < xfirst) {      // among the leading knots (one, for the DESI table): bisection
        lo = 0;
        long long hi = first;
        while (hi - lo > 1) {
            const long long mid = (lo + hi) >> 1;
            if (xf[mid].x <= v) lo = mid; else hi = mid;
        }
        k0 = xf[lo]; k1 = xf[lo + 1];
    } else {
        const double t = LAW == 2 ? log2_guess(v) : v;
        double g = (t - a) * b;
        const double gmax = (double)(n - 2 - first);
        g = g > 0. ? g : 0.;            // (a NaN guess counts as 0)
        g = g < gmax ? g : gmax;        // clamped as a double: converting a value beyond the integer range is undefined
        lo = first + (long long)g;
        k0 = xf[lo]; k1 = xf[lo + 1];      // the guessed interval: both ends in flight together; the walk (rare) re-uses the end it keeps
        while (lo > first && k0.x > v) { --lo; k1 = k0; k0 = xf[lo]; }
        while (lo < n - 2 && k1.x <= v) { ++lo; k0 = k1; k1 = xf[lo + 1]; }
    }
    if (k0.x == v) return k0.y;  // numpy.interp returns the knot value here ("avoid potential non-finite interpolation")
    const double slope = (k1.y - k0.y) / (k1.x - k0.x);
    double r = slope * (v - k0.x) + k0.y;   // product and sum rounded separately
    // numpy.interp: if the result is NaN (slope or difference infinite) it retries from the right knot, then takes the common value
    if (r != r) {
        r = slope * (v - k1.x) + k1.y;
        if (r != r && k0.y == k1.y) r = k0.y;
    }
    return r;
}

// The same for U samples at once (the grid-stride loop of interp_table_kernel): the U guessed intervals are requested together -- one sample at a time a thread
// waited for its sample, then for the two knots its guess pointed at, then for its store, three memory round trips in a row per sample.  The arithmetic of a
// sample is interp_sample's, statement for statement; samples outside the law's part of the table (leading knots, outside, the last knot) go through it.
template <int LAW, int U>
CPIT_HD void interp_samples(const Pair* __restrict__ xf, long long n, long long first, double a, double b, double x0, double xfirst, double xn, const double* v,
                            const bool* live, double* r, bool* outside) {
    long long lo[U];
    Pair k0[U], k1[U];
    bool fast[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        fast[u] = live[u] && v[u] >= x0 && v[u] < xn && !(first > 0 && v[u] < xfirst) && n >= 2;
        const double t = LAW == 2 ? log2_guess(fast[u] ? v[u] : xfirst) : (fast[u] ? v[u] : xfirst);
        double g = (t - a) * b;
        const double gmax = (double)(n - 2 - first);
        g = g > 0. ? g : 0.;
        g = g < gmax ? g : gmax;
        lo[u] = n >= 2 ? first + (long long)g : 0;
        k0[u] = xf[lo[u]];
        k1[u] = xf[n >= 2 ? lo[u] + 1 : 0];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!live[u]) continue;
        if (!fast[u]) {
            r[u] = interp_sample<LAW>(xf, n, first, a, b, x0, xfirst, xn, v[u], outside);
            continue;
        }
        const double vv = v[u];
        long long l = lo[u];
        Pair p0 = k0[u], p1 = k1[u];
        while (l > first && p0.x > vv) { --l; p1 = p0; p0 = xf[l]; }
        while (l < n - 2 && p1.x <= vv) { ++l; p0 = p1; p1 = xf[l + 1]; }
        if (p0.x == vv) {
            r[u] = p0.y;
            continue;
        }
        const double slope = (p1.y - p0.y) / (p1.x - p0.x);
        double res = slope * (vv - p0.x) + p0.y;
        if (res != res) {
            res = slope * (vv - p1.x) + p1.y;
            if (res != res && p0.y == p1.y) res = p0.y;
        }
        r[u] = res;
    }
}

}  // namespace cpit
