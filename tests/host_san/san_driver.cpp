// Sanitizer driver for the host side of the library (TEST HARNESS ONLY; built with
//   g++ -fsanitize=address,undefined -fno-sanitize-recover=all
// by tests/test_host_sanitizers.py).  It runs, on the CPU, the code of the product that is plain C++:
//   * the special functions and the FFTLog table setup behind the C ABI (cp_special.cpp, cp_fftlog_setup.cpp),
//   * the per-thread phases of the fused FFTLog kernel through the emulator (tests/host_emu/emu_fftlog.cpp): the kernel's LDS is a heap
//     array of exactly the size the launch requests, its rows are heap arrays of exactly n samples, so an index of the kernel that leaves
//     its LDS allocation or its rows is an AddressSanitizer report here (GPU sanitizers are not available on the pool).
// Exit status 0 and a last line "ok" mean no report; numerical checks are the business of tests/test_fftlog_host.py.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "../../include/cosmoprimo_amd.h"

extern "C" int emu_fftlog(int n, int np, int nker, const double* pre, const double* post, const double* u_re_im, const double* in,
                          double* out, long long nbatch, int ext_l, double val_l, int ext_r, double val_r, int keep_padding);

static int failures = 0;
#define EXPECT(cond)                                                       \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "line %d: %s\n", __LINE__, #cond);        \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

static void special_functions() {
    std::vector<double> z, out;
    for (double re = -6.5; re <= 30.; re += 0.73)
        for (double im : {-250., -3.1, 0., 1e-9, 0.4, 17., 1e4}) { z.push_back(re); z.push_back(im); }
    const long long n = (long long)z.size() / 2;
    out.resize(z.size());
    EXPECT(cp_loggamma(z.data(), out.data(), n) == CP_OK);
    EXPECT(cp_gamma(z.data(), out.data(), n) == CP_OK);
    for (int kind : {CP_KERNEL_BESSEL_J, CP_KERNEL_SPHERICAL_BESSEL_J, CP_KERNEL_TOPHAT, CP_KERNEL_TOPHAT_SQ, CP_KERNEL_GAUSSIAN, CP_KERNEL_GAUSSIAN_SQ})
        for (double param : {0., 0.5, 2., 3., 4.})
            EXPECT(cp_kernel_eval(kind, param, z.data(), out.data(), n) == CP_OK);
    EXPECT(cp_kernel_eval(12345, 0., z.data(), out.data(), n) != CP_OK);
    EXPECT(cp_loggamma(z.data(), out.data(), 0) == CP_OK);
}

struct Tables {
    int n, npad, nker;
    std::vector<double> x, delta, lnxy, y, px, py, pre, post, u;
};

static bool make_tables(Tables& t, int n, int nker, int kind, double param, double q, int minfolds, int lowring) {
    t.n = n; t.nker = nker;
    t.npad = cp_fftlog_padded_size(n, minfolds);
    if (t.npad < n) return false;
    t.x.resize((size_t)nker * n);
    for (int k = 0; k < nker; ++k)
        for (int i = 0; i < n; ++i) t.x[(size_t)k * n + i] = std::pow(10., -3. + (5. + k) * i / (n > 1 ? n - 1 : 1));
    std::vector<cp_fftlog_spec> spec(nker);
    for (int k = 0; k < nker; ++k) spec[k] = cp_fftlog_spec{kind, param + 2 * k, q, 1., 3., 0.0635, k % 2 ? -1. : 1.};
    t.delta.resize(nker); t.lnxy.resize(nker); t.y.resize((size_t)nker * n);
    t.px.resize((size_t)nker * t.npad); t.py = t.pre = t.post = t.px;
    t.u.resize((size_t)nker * (t.npad / 2 + 1) * 2);
    return cp_fftlog_tables(n, nker, t.x.data(), spec.data(), minfolds, lowring, 1, nullptr, nullptr, t.delta.data(), t.lnxy.data(), t.y.data(),
                            t.px.data(), t.py.data(), t.pre.data(), t.post.data(), t.u.data()) == CP_OK;
}

static void table_setup_errors() {
    EXPECT(cp_fftlog_padded_size(0, 2) < 0);
    EXPECT(cp_fftlog_padded_size(1000, 2) == 2048);
    Tables t;
    EXPECT(make_tables(t, 64, 2, CP_KERNEL_BESSEL_J, 0., 0., 2, 1));
    // not log-spaced with check_level
    std::vector<double> x(64);
    for (int i = 0; i < 64; ++i) x[i] = 1. + i;
    cp_fftlog_spec spec{CP_KERNEL_BESSEL_J, 0., 0., 1., 0., 1., 1.};
    std::vector<double> d(1), l(1), y(64), a(128), b(128), c(128), e(128), u(130);
    EXPECT(cp_fftlog_tables(64, 1, x.data(), &spec, 2, 1, 1, nullptr, nullptr, d.data(), l.data(), y.data(), a.data(), b.data(), c.data(), e.data(),
                            u.data()) != CP_OK);
    // custom kernel without its samples
    spec.kind = CP_KERNEL_CUSTOM;
    for (int i = 0; i < 64; ++i) x[i] = std::pow(10., -2. + 4. * i / 63.);
    EXPECT(cp_fftlog_tables(64, 1, x.data(), &spec, 2, 1, 0, nullptr, nullptr, d.data(), l.data(), y.data(), a.data(), b.data(), c.data(), e.data(),
                            u.data()) != CP_OK);
    // custom kernel with samples
    std::vector<double> uc(130, 0.5), lr{1., 0.3};
    EXPECT(cp_fftlog_tables(64, 1, x.data(), &spec, 2, 1, 0, uc.data(), lr.data(), d.data(), l.data(), y.data(), a.data(), b.data(), c.data(), e.data(),
                            u.data()) == CP_OK);
}

static void kernel_phases() {
    const double nan = std::numeric_limits<double>::quiet_NaN(), inf = std::numeric_limits<double>::infinity();
    struct Mode { int el; double vl; int er; double vr; int keep; };
    const Mode modes[] = {{0, 0., 0, 0., 0}, {0, 0., 0, 0., 1}, {1, 0., 1, 0., 0}, {0, 1.5, 1, 0., 1}, {2, 0., 2, 0., 0}, {2, 0., 0, 0.3, 1}};
    for (int n : {2, 3, 8, 13, 60, 250, 256, 500, 1000, 1024, 2048, 4096})
        for (int nker : {1, 3}) {
            if (n > 1024 && nker > 1) continue;
            Tables t;
            EXPECT(make_tables(t, n, nker, CP_KERNEL_SPHERICAL_BESSEL_J, 0., 1.5, 2, 1));
            for (long long nbatch : {1LL, 2LL, 5LL}) {
                std::vector<double> in((size_t)nbatch * nker * n), out;
                for (size_t i = 0; i < in.size(); ++i) in[i] = (1. + 0.1 * std::sin(0.37 * i)) * std::pow(t.x[i % n], -1.2);
                // one row 1e-12 of its partner, one row with a NaN, one with an Inf: the screening and scaling branches
                if (nbatch >= 2) for (int i = 0; i < n; ++i) in[(size_t)nker * n + i] *= 1e-12;
                if (nbatch >= 5) { in[(size_t)2 * nker * n + n / 2] = nan; in[(size_t)3 * nker * n] = inf; }
                for (const Mode& m : modes) {
                    out.assign((size_t)nbatch * nker * (m.keep ? t.npad : n), -7.);
                    EXPECT(emu_fftlog(n, t.npad, nker, t.pre.data(), t.post.data(), t.u.data(), in.data(), out.data(), nbatch, m.el, m.vl, m.er, m.vr,
                                      m.keep) == 0);
                    bool written = true;
                    for (double v : out) written = written && v != -7.;
                    EXPECT(written);
                    if (nbatch >= 5 && m.el != 2) {    // row 0 of batch entry 4 is finite, the NaN row is NaN alone
                        EXPECT(std::isfinite(out[(size_t)4 * nker * (m.keep ? t.npad : n)]));
                        EXPECT(std::isnan(out[(size_t)2 * nker * (m.keep ? t.npad : n)]));
                    }
                }
            }
        }
}

extern "C" int emu_splice_build(int n, const double* x, int npieces, const int* piece_first, const int* piece_src, const int* piece_start, int nq, const double* xq,
                                const int* qj, int u0, int u1, int generic_first, int generic_end, int* ints, double* doubles, double* win, int* qe, double* qw);

// the plan builder of the uniform-stretch spliced spline (csrc/cp_splice_uniform_plan.h) on the knots of wallish2018 and on knots it must refuse,
// every array of exactly the size the builder is told
static void splice_plans() {
    for (int variant = 0; variant < 3; ++variant) {
        const int nk = 1024, nlin = variant == 2 ? 8192 : 4096;
        std::vector<double> k(nk), klin(nlin), knots;
        for (int i = 0; i < nk; ++i) k[i] = std::pow(10., -7. + 9. * i / (nk - 1));
        for (int i = 0; i < nlin; ++i) klin[i] = 1e-7 + (2. - 1e-7) * i / (nlin - 1);
        std::vector<int> first(3, 0), src = {0, 1, 0}, start(3, 0);
        int nleft = 0, nmid = 0, nright = 0;
        for (int i = 0; i < nk; ++i) if (k[i] < 5e-4) { knots.push_back(k[i]); ++nleft; }
        for (int i = 0; i < nlin; ++i) if (klin[i] > 1e-2 && klin[i] < 1.5) { if (!nmid) start[1] = i; knots.push_back(klin[i]); ++nmid; }
        for (int i = 0; i < nk; ++i) if (k[i] > 2.) { if (!nright) start[2] = i; knots.push_back(k[i]); ++nright; }
        first[1] = nleft; first[2] = nleft + nmid;
        const int n = (int)knots.size();
        std::vector<int> qj(nk);
        int gf = nk, ge = 0;
        for (int q = 0; q < nk; ++q) {
            int j = (int)(std::upper_bound(knots.begin(), knots.end(), k[q]) - knots.begin()) - 1;
            j = j < 0 ? 0 : (j > n - 2 ? n - 2 : j);
            qj[q] = j;
            if ((k[q] >= 5e-4 && k[q] <= 2.) || (variant == 1 && q >= 100)) { gf = q < gf ? q : gf; ge = q + 1; }      // (1: spline queries inside the left piece)
        }
        std::vector<int> ints(15), qe(512);
        std::vector<double> doubles(2), win(512), qw(2048);
        const int ok = emu_splice_build(n, knots.data(), 3, first.data(), src.data(), start.data(), nk, k.data(), qj.data(), nleft, nleft + nmid - 1, gf, ge, ints.data(),
                                        doubles.data(), win.data(), qe.data(), qw.data());
        EXPECT(ok == (variant == 0 ? 1 : 0));      // 1: spline queries far from the stretch; 2: more knots on it than 64 x 57
        if (ok) EXPECT(ints[1] == nmid && ints[0] * 64 >= nmid);
    }
}

extern "C" int emu_interp_law(long long n, const double* x, int* law, long long* first, double* a, double* b);
extern "C" int emu_interp_apply(long long n, const double* x, const double* f, int law, long long first, double a, double b, long long nx, const double* xq, double* out);

// the table interpolation of the 'tabulated' engine (csrc/cp_interp_table.h): the law of the knots and the per-sample walk on exact-size tables --
// the shape of data/desi.dat, a uniform table, and laws the table does not follow (guesses far outside the table, NaN and infinite guesses)
static void interp_tables() {
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<double> x, f;
        if (variant == 0) { x.push_back(0.); for (int i = 0; i <= 4000; ++i) x.push_back(std::pow(10., -8. + 10. * i / 4000.)); }
        if (variant == 1) for (int i = 0; i < 777; ++i) x.push_back(-3. + 0.01 * i);
        if (variant == 2) for (int i = 0; i < 300; ++i) x.push_back(1. + 0.03 * i + 0.0299 * ((i * 7919) % 13) / 13.);
        for (size_t i = 0; i < x.size(); ++i) f.push_back(std::cos(3. * i));
        const long long n = (long long)x.size();
        int law = 0; long long first = 0; double a = 0., b = 0.;
        emu_interp_law(n, x.data(), &law, &first, &a, &b);
        EXPECT(law == (variant == 0 ? 2 : (variant == 1 ? 1 : 0)) && first == (variant == 0 ? 1 : 0));
        std::vector<double> q, out;
        for (size_t i = 0; i + 1 < x.size(); ++i) { q.push_back(x[i]); q.push_back(0.5 * (x[i] + x[i + 1])); q.push_back(std::nextafter(x[i + 1], -1e308)); }
        q.push_back(x.back()); q.push_back(std::nextafter(x.front(), -1e308)); q.push_back(std::nextafter(x.back(), 1e308)); q.push_back(std::nan(""));
        out.resize(q.size());
        const double laws[6][3] = {{(double)law, a, b}, {1., x[0], 1e300}, {1., -1e300, 1e300}, {2., 1e300, -1e300}, {1., std::nan(""), 1.}, {2., 0., std::nan("")}};
        for (int l = (law ? 0 : 1); l < 6; ++l) {
            if (laws[l][0] == 2. && !(x[first] > 0.)) continue;      // (the logarithm law is for positive knots)
            const int outside = emu_interp_apply(n, x.data(), f.data(), (int)laws[l][0], first, laws[l][1], laws[l][2], (long long)q.size(), q.data(), out.data());
            EXPECT(outside == 1);
            bool same = true;
            for (size_t i = 0; i + 3 < q.size(); ++i) {
                const size_t j = (size_t)(std::upper_bound(x.begin(), x.end(), q[i]) - x.begin()) - 1;
                const size_t k = j + 1 < x.size() ? j : x.size() - 2;
                const double ref = q[i] == x.back() ? f.back() : (q[i] == x[k] ? f[k] : (f[k + 1] - f[k]) / (x[k + 1] - x[k]) * (q[i] - x[k]) + f[k]);
                same = same && out[i] == ref;
            }
            EXPECT(same);
            EXPECT(out[q.size() - 1] != out[q.size() - 1] && out[q.size() - 2] != out[q.size() - 2] && out[q.size() - 3] != out[q.size() - 3]);
        }
    }
}

int main() {
    interp_tables();
    special_functions();
    table_setup_errors();
    kernel_phases();
    splice_plans();
    if (failures) { std::fprintf(stderr, "%d expectation(s) failed\n", failures); return 1; }
    std::puts("ok");
    return 0;
}
