// cp_fftlog_setup.cpp -- host-side construction of the tables of an FFTLog plan (pure C++, no device code).
//
// Replaces FFTlog._setup of the reference (cosmoprimo/fftlog.py:144-184) and the convention factors its subclasses multiply in
// afterwards (fftlog.py:280, 318-330, 368-377, 403-405, 431-433), written from the maths of SURVEY.md appendix C1:
//   delta   = ln(x[n-1] / x[0]) / (n - 1);   npad = 2^bit_length(n minfolds - 1);   pad splits (npad - n) / 2 | rest
//   lnxy    = (delta / pi) arg U(q + i pi / delta)            low-ringing      |  ln(xy) + delta   otherwise
//   y_j     = exp(lnxy - delta) / x[n-1-j]
//   u_m     = U(q + 2 pi i m / (npad delta)) exp(-2 pi i m lnxy / (npad delta)),   m = 0 .. npad/2
//   pre_j   = c x_j^(p - q),   post_j = s y_j^(-q)            on the log-extended grids; (c, p, s) = convention of the transform
#include <cmath>
#include <complex>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"

namespace {
using cd = std::complex<double>;
const double PI = 3.141592653589793238462643383279502884;

// geometric continuation of a log-spaced grid on both sides, the values numpy's `a0 * (a1 / a0) ** e` / `aN / (aN-1 / aN) ** e`
// give (reference pad(..., extrap='log'), fftlog.py:483-498): the same expressions, so the grids agree to the last bit or two
void extend_log(const double* a, int n, int left, int right, double* out) {
    const double r0 = a[1] / a[0], r1 = a[n - 2] / a[n - 1];
    for (int j = 0; j < left; ++j) out[j] = a[0] * std::pow(r0, (double)(j - left));
    for (int j = 0; j < n; ++j) out[left + j] = a[j];
    for (int j = 0; j < right; ++j) out[left + n + j] = a[n - 1] / std::pow(r1, (double)(j + 1));
}
}  // namespace

extern "C" int cp_fftlog_padded_size(int n, int minfolds) {
    if (n < 1 || minfolds < 1) return -1;
    long long v = (long long)n * minfolds - 1;
    int bits = 0;
    while (v > 0) {
        ++bits;
        v >>= 1;
    }
    if (bits > 30) return -1;
    return 1 << bits;
}

extern "C" int cp_fftlog_tables(int n, int nker, const double* x, const cp_fftlog_spec* spec, int minfolds, int lowring, int check_level,
                                const double* u_custom, const double* lowring_custom, double* delta, double* lnxy, double* y, double* padded_x,
                                double* padded_y, double* pre, double* post, double* u) {
    if (!x || !spec || !delta || !lnxy || !y || !padded_x || !padded_y || !pre || !post || !u)
        return cp::fail(CP_EINVAL, "cp_fftlog_tables: null pointer");
    if (n < 2 || nker < 1) return cp::fail(CP_EINVAL, "cp_fftlog_tables: need n >= 2 and nker >= 1 (got n=%d, nker=%d)", n, nker);
    const int npad = cp_fftlog_padded_size(n, minfolds);
    if (npad < n) return cp::fail(CP_EINVAL, "Convolution size must be larger than input x size");
    const int in_left = (npad - n) / 2, in_right = npad - n - in_left;
    const int out_left = in_right, out_right = in_left;  // fftlog.py:152-153
    const int nu = npad / 2 + 1;
    std::vector<double> zbuf(2 * (size_t)nu), ubuf(2 * (size_t)nu);
    for (int k = 0; k < nker; ++k) {
        const double* xk = x + (size_t)k * n;
        const cp_fftlog_spec& s = spec[k];
        for (int j = 0; j < n; ++j)
            if (!(xk[j] > 0.) || !std::isfinite(xk[j])) return cp::fail(CP_EINVAL, "Input x must be positive and finite");
        const double d = std::log(xk[n - 1] / xk[0]) / (n - 1);
        if (check_level) {
            for (int j = 1; j < n; ++j)   // numpy.allclose(log ratio, delta, rtol=1e-3) with its default atol=1e-8
                if (std::fabs(std::log(xk[j] / xk[j - 1]) - d) > 1e-8 + 1e-3 * std::fabs(d)) return cp::fail(CP_EINVAL, "Input x must be log-spaced");
        }
        delta[k] = d;
        // low-ringing condition
        double l;
        if (lowring) {
            cd uk;
            if (s.kind == CP_KERNEL_CUSTOM) {
                if (!lowring_custom) return cp::fail(CP_EINVAL, "cp_fftlog_tables: custom kernel %d without its low-ringing value", k);
                uk = cd(lowring_custom[2 * k], lowring_custom[2 * k + 1]);
            } else {
                const double z[2] = {s.q, PI / d};
                double r[2];
                const int st = cp_kernel_eval(s.kind, s.param, z, r, 1);
                if (st != CP_OK) return st;
                uk = cd(r[0], r[1]);
            }
            l = d / PI * std::arg(uk);
        } else {
            l = std::log(s.xy) + d;
        }
        lnxy[k] = l;
        double* yk = y + (size_t)k * n;
        const double ey = std::exp(l - d);
        for (int j = 0; j < n; ++j) yk[j] = ey / xk[n - 1 - j];
        double* px = padded_x + (size_t)k * npad;
        double* py = padded_y + (size_t)k * npad;
        extend_log(xk, n, in_left, in_right, px);
        extend_log(yk, n, out_left, out_right, py);
        double* prek = pre + (size_t)k * npad;
        double* postk = post + (size_t)k * npad;
        for (int j = 0; j < npad; ++j) {
            // two roundings as in the reference (x^-q first, then the convention factor x^p c)
            prek[j] = std::pow(px[j], -s.q);
            if (s.pre_power != 0. || s.pre_const != 1.) prek[j] *= std::pow(px[j], s.pre_power) * s.pre_const;
            postk[j] = std::pow(py[j], -s.q) * s.post_sign;
        }
        // u_m; identical consecutive kernels (same kind, parameter, q, delta) share the Mellin transform values
        double* uk = u + (size_t)k * 2 * nu;
        const bool same = k > 0 && s.kind != CP_KERNEL_CUSTOM && spec[k - 1].kind == s.kind && spec[k - 1].param == s.param &&
                          spec[k - 1].q == s.q && delta[k - 1] == d;
        if (s.kind == CP_KERNEL_CUSTOM) {
            if (!u_custom) return cp::fail(CP_EINVAL, "cp_fftlog_tables: custom kernel %d without its values", k);
            for (int m = 0; m < 2 * nu; ++m) ubuf[m] = u_custom[(size_t)k * 2 * nu + m];
        } else if (!same) {
            for (int m = 0; m < nu; ++m) {
                zbuf[2 * m] = s.q;
                zbuf[2 * m + 1] = 2. * PI / npad / d * m;
            }
            const int st = cp_kernel_eval(s.kind, s.param, zbuf.data(), ubuf.data(), nu);
            if (st != CP_OK) return st;
        }
        for (int m = 0; m < nu; ++m) {
            const double ang = -2. * PI * l / npad / d * m;
            const cd v = cd(ubuf[2 * m], ubuf[2 * m + 1]) * cd(std::cos(ang), std::sin(ang));
            uk[2 * m] = v.real();
            uk[2 * m + 1] = v.imag();
        }
    }
    return CP_OK;
}
