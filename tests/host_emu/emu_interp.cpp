// Host access to the table interpolation of csrc/cp_interp_table.h for the CPU tests (TEST HARNESS ONLY): the law of a table's knots as the library
// finds it, and the kernel's per-sample code run over an array of samples on an exact-size pair table.  tests/test_interp_table_host.py checks both
// against numpy.interp; tests/host_san/san_driver.cpp runs them under the sanitizers.  Build with -ffp-contract=off (numpy's arithmetic).
#include <vector>

#include "../../cosmoprimo_amd/csrc/cp_interp_table.h"

extern "C" int emu_interp_law(long long n, const double* x, int* law, long long* first, double* a, double* b) {
    const cpit::Law L = cpit::find_law(x, n);
    *law = L.law; *first = L.first; *a = L.a; *b = L.b;
    return 0;
}

// out[i] = numpy.interp(xq[i], x, f) through the law (law 1 or 2; given, so that a test may hand over a law the table does not follow: the walk
// must still end on numpy's interval); returns 1 when a sample lies outside the table or is NaN
extern "C" int emu_interp_apply(long long n, const double* x, const double* f, int law, long long first, double a, double b, long long nx, const double* xq,
                                double* out) {
    std::vector<cpit::Pair> xf((size_t)n);      // exactly n pairs: a walk past either end is a heap overflow for the sanitizer
    for (long long i = 0; i < n; ++i) xf[(size_t)i] = cpit::Pair{x[i], f[i]};
    bool outside = false;
    for (long long i = 0; i < nx; ++i)
        out[i] = law == 2 ? cpit::interp_sample<2>(xf.data(), n, first, a, b, x[0], x[first], x[n - 1], xq[i], &outside)
                          : cpit::interp_sample<1>(xf.data(), n, first, a, b, x[0], x[first], x[n - 1], xq[i], &outside);
    return outside ? 1 : 0;
}
