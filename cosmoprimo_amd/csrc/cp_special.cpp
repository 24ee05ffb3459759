// cp_special.cpp -- host special functions for FFTLog table setup: complex log-gamma / gamma and
// the Mellin-transform kernels U_K(z) of the reference (cosmoprimo/fftlog.py:666-766).
//
// The reference calls scipy.special.loggamma / gamma (third-party, unpinned; exercised version
// scipy 1.15.3).  Its published algorithm (Hare 1997, as implemented in scipy's loggamma) is
// restated here: Stirling series for large |z|, Taylor series around z = 1 (and 2 via the
// recurrence), reflection for Re z < 0.1, upward recurrence otherwise, with the principal branch
// of log Gamma (branch cut on the negative real axis, real on the positive real axis).
#include <cmath>
#include <complex>
#include <limits>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"

namespace {

typedef std::complex<double> cd;

const double kPi = 3.141592653589793238462643383279502884;
const double kLogPi = 1.1447298858494001741434262;    // log(pi)
const double kHalfLog2Pi = 0.918938533204672741780329736406;  // log(2 pi) / 2
const double kSmallX = 7., kSmallY = 7.;

double sinpi_real(double x) {
    double sgn = 1.;
    if (x < 0.) {
        x = -x;
        sgn = -1.;
    }
    const double r = std::fmod(x, 2.);
    if (r < 0.5) return sgn * std::sin(kPi * r);
    if (r > 1.5) return sgn * std::sin(kPi * (r - 2.));
    return -sgn * std::sin(kPi * (r - 1.));
}

double cospi_real(double x) {
    const double r = std::fmod(std::fabs(x), 2.);
    if (r == 0.5 || r == 1.5) return 0.;
    if (r < 1.) return -std::sin(kPi * (r - 0.5));
    return std::sin(kPi * (r - 1.5));
}

// sin(pi z) for complex z, safe for large |Im z|
cd sinpi(cd z) {
    const double piy = kPi * z.imag(), abspiy = std::fabs(piy);
    const double s = sinpi_real(z.real()), c = cospi_real(z.real());
    if (abspiy < 700.) return cd(s * std::cosh(piy), c * std::sinh(piy));
    const double inf = std::numeric_limits<double>::infinity();
    const double exph = std::exp(abspiy / 2.);
    if (exph == inf) {
        const double re = (s == 0.) ? std::copysign(0., s) : std::copysign(inf, s);
        const double im = (c == 0.) ? std::copysign(0., c) : std::copysign(inf, c);
        return cd(re, std::copysign(1., piy) * im);
    }
    return cd(0.5 * s * exph * exph, std::copysign(1., piy) * 0.5 * c * exph * exph);
}

// Stirling series, |z| large: (z - 1/2) log z - z + log(2 pi)/2 + sum B_2k / (2k (2k-1) z^(2k-1))
cd loggamma_stirling(cd z) {
    static const double coeffs[] = {-2.955065359477124183e-2, 6.4102564102564102564e-3, -1.9175269175269175269e-3,
                                    8.4175084175084175084e-4, -5.952380952380952381e-4,  7.9365079365079365079e-4,
                                    -2.7777777777777777778e-3, 8.3333333333333333333e-2};
    const cd rz = 1. / z;
    const cd rzz = rz / z;
    cd p = coeffs[0];
    for (int i = 1; i < 8; ++i) p = p * rzz + coeffs[i];
    return (z - 0.5) * std::log(z) - z + kHalfLog2Pi + rz * p;
}

// Taylor series of log Gamma around z = 1 (coefficients (-1)^n zeta(n) / n, n = 2..23; first term -gamma_E)
cd loggamma_taylor(cd z) {
    static const double coeffs[] = {
        -4.3478266053040259361e-2, 4.5454556293204669442e-2, -4.7619070330142227991e-2, 5.000004769810169364e-2,
        -5.2631679379616660734e-2, 5.5555767627403611102e-2, -5.8823978658684582339e-2, 6.2500955141213040742e-2,
        -6.6668705882420468033e-2, 7.1432946295361336059e-2, -7.6932516411352191473e-2, 8.3353840546109004025e-2,
        -9.0954017145829042233e-2, 1.0009945751278180853e-1, -1.1133426586956469049e-1, 1.2550966952474304242e-1,
        -1.4404989676884611812e-1, 1.6955717699740818995e-1, -2.0738555102867398527e-1, 2.7058080842778454788e-1,
        -4.0068563438653142847e-1, 8.2246703342411321824e-1, -5.7721566490153286061e-1};
    const cd w = z - 1.;
    cd p = coeffs[0];
    for (int i = 1; i < 23; ++i) p = p * w + coeffs[i];
    return w * p;
}

// upward recurrence to Re z > kSmallX for Im z >= 0, tracking branch crossings of the running product
cd loggamma_recurrence(cd z) {
    int signflips = 0, sb = 0, nsb;
    cd shiftprod = z;
    z += 1.;
    while (z.real() <= kSmallX) {
        shiftprod *= z;
        nsb = std::signbit(shiftprod.imag());
        signflips += (nsb != 0 && sb == 0) ? 1 : 0;
        sb = nsb;
        z += 1.;
    }
    return loggamma_stirling(z) - std::log(shiftprod) - cd(0., signflips * 2. * kPi);
}

cd loggamma(cd z) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (std::isnan(z.real()) || std::isnan(z.imag())) return cd(nan, nan);
    if (z.real() <= 0. && z == std::floor(z.real())) return cd(nan, nan);  // poles
    if (z.real() > kSmallX || std::fabs(z.imag()) > kSmallY) return loggamma_stirling(z);
    if (std::abs(z - 1.) <= 0.2) return loggamma_taylor(z);
    if (std::abs(z - 2.) <= 0.2) return std::log(z - 1.) + loggamma_taylor(z - 1.);
    if (z.real() < 0.1) {
        // reflection: log Gamma(z) = log pi - log sin(pi z) - log Gamma(1 - z) (+ branch bookkeeping)
        const double tmp = std::copysign(2. * kPi, z.imag()) * std::floor(0.5 * z.real() + 0.25);
        return cd(kLogPi, tmp) - std::log(sinpi(z)) - loggamma(1. - z);
    }
    if (!std::signbit(z.imag())) return loggamma_recurrence(z);
    return std::conj(loggamma_recurrence(std::conj(z)));
}

cd cgamma(cd z) {
    const double nan = std::numeric_limits<double>::quiet_NaN();
    if (z.imag() == 0.) {
        if (z.real() <= 0. && z.real() == std::floor(z.real())) return cd(nan, nan);
        return cd(std::tgamma(z.real()), 0.);
    }
    return std::exp(loggamma(z));
}

// U_K(z): expressions follow the reference's eval() methods term by term
cd kernel_eval(int kind, double p, cd z) {
    const double ln2 = std::log(2.);
    switch (kind) {
        case CP_KERNEL_BESSEL_J:  // fftlog.py:695
            return std::exp(ln2 * (z - 1.) + loggamma(0.5 * (p + z)) - loggamma(0.5 * (2. + p - z)));
        case CP_KERNEL_SPHERICAL_BESSEL_J:  // fftlog.py:705
            return std::exp(ln2 * (z - 1.5) + loggamma(0.5 * (p + z)) - loggamma(0.5 * (3. + p - z)));
        case CP_KERNEL_TOPHAT:  // fftlog.py:726
            return std::exp(ln2 * (z - 1.) + loggamma(cd(1. + 0.5 * p, 0.)) + loggamma(0.5 * z) - loggamma(0.5 * (2. + p - z)));
        case CP_KERNEL_TOPHAT_SQ:  // fftlog.py:739-746
            if (p == 1.) return -0.25 * std::sqrt(kPi) * std::exp(loggamma(0.5 * (z - 2.)) - loggamma(0.5 * (3. - z)));
            if (p == 3.) return 2.25 * std::sqrt(kPi) * (z - 2.) / (z - 6.) * std::exp(loggamma(0.5 * (z - 4.)) - loggamma(0.5 * (5. - z)));
            return std::exp(ln2 * (p - 1.) + 2. * loggamma(cd(1. + 0.5 * p, 0.)) + loggamma(0.5 * (1. + p - z)) + loggamma(0.5 * z) -
                            loggamma(1. + p - 0.5 * z) - loggamma(0.5 * (2. + p - z))) /
                   std::sqrt(kPi);
        case CP_KERNEL_GAUSSIAN:  // fftlog.py:756
            return std::pow(cd(2., 0.), 0.5 * z - 1.) * cgamma(0.5 * z);
        case CP_KERNEL_GAUSSIAN_SQ:  // fftlog.py:766
            return 0.5 * cgamma(0.5 * z);
    }
    const double nan = std::numeric_limits<double>::quiet_NaN();
    return cd(nan, nan);
}

}  // namespace

extern "C" int cp_loggamma(const double* z, double* out, long long n) {
    if ((!z || !out) && n > 0) return cp::fail(CP_EINVAL, "cp_loggamma: null pointer");
    for (long long i = 0; i < n; ++i) {
        const cd r = loggamma(cd(z[2 * i], z[2 * i + 1]));
        out[2 * i] = r.real();
        out[2 * i + 1] = r.imag();
    }
    return CP_OK;
}

extern "C" int cp_gamma(const double* z, double* out, long long n) {
    if ((!z || !out) && n > 0) return cp::fail(CP_EINVAL, "cp_gamma: null pointer");
    for (long long i = 0; i < n; ++i) {
        const cd r = cgamma(cd(z[2 * i], z[2 * i + 1]));
        out[2 * i] = r.real();
        out[2 * i + 1] = r.imag();
    }
    return CP_OK;
}

extern "C" int cp_kernel_eval(int kind, double param, const double* z, double* out, long long n) {
    if ((!z || !out) && n > 0) return cp::fail(CP_EINVAL, "cp_kernel_eval: null pointer");
    if (kind < CP_KERNEL_BESSEL_J || kind > CP_KERNEL_GAUSSIAN_SQ) return cp::fail(CP_EINVAL, "cp_kernel_eval: unknown kernel kind %d", kind);
    for (long long i = 0; i < n; ++i) {
        const cd r = kernel_eval(kind, param, cd(z[2 * i], z[2 * i + 1]));
        out[2 * i] = r.real();
        out[2 * i + 1] = r.imag();
    }
    return CP_OK;
}
