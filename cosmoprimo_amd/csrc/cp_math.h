// cp_math.h -- short forms of the transcendental functions the ALU-bound kernels spend their time in (device code, gfx950).
// Each is the textbook argument reduction + polynomial of its function, accurate to 1-2 ulp (stated per function, checked against extended
// precision), at a third to a half of the instructions of the library versions, which carry double-double intermediates and the handling of
// subnormal / overflowing arguments that these kernels' arguments never need; out-of-range arguments go to the library functions out of line.
#pragma once
#include <hip/hip_runtime.h>

namespace cpmath {

// 1 / x for finite, normal x: the hardware estimate and two Newton steps (relative error below 2 ulp), a third of the instructions of an
// IEEE division (v_div_scale x 2, v_div_fmas, v_div_fixup around the same estimate and steps)
__device__ __forceinline__ double recip(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.), r, r);
    r = fma(fma(-x, r, 1.), r, r);
    return r;
}

// log(x), fast for positive, finite, normal x (every argument here is a wavenumber or e + a positive term): the classic argument reduction to
// m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1) and a degree-14 odd series in s (the fdlibm scheme and minimax coefficients), below 1 ulp;
// a third of the instructions of the library log, which carries double-double intermediates this kernel has no use for.
static __device__ __attribute__((noinline)) double log_any(double x) { return log(x); }
static __device__ __attribute__((noinline)) double sin_any(double x) { return sin(x); }

__device__ __forceinline__ double log_pos(double x) {
    if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return log_any(x);   // zero, negative, subnormal, Inf, NaN: the library's answers (out of line)
    double m = __builtin_amdgcn_frexp_mant(x);           // [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double k = (double)e;
    const double f = m - 1.;
    const double s = f * recip(2. + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    return k * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, k * 1.90821492927058770002e-10)) - f);
}

// sin(x) for |x| < 1e6 (k rs_drag reaches 1e4 at k = 100 h/Mpc): n = round(x / (pi / 2)), r = x - n pi/2 with pi/2 in two pieces (33 + 53
// bits: n times the first is exact), then the degree-13 / degree-14 polynomials of sin and cos on [-pi/4, pi/4] picked by n mod 4; absolute
// error below 2e-16 (checked against extended precision on 6e5 arguments up to 1e6).  Larger arguments take the library function.
__device__ __forceinline__ double sin_bounded(double x) {
    if (!(fabs(x) < 1e6)) return sin_any(x);      // (also -Inf and arguments below -1e6: the two-piece reduction is exact for |n| < 2^20 only)
    const double n = rint(x * 6.36619772367581382433e-01);
    double r = fma(-n, 1.57079632673412561417e+00, x);
    r = fma(-n, 6.07710050650619224932e-11, r);
    const double z = r * r;
    const double ps = fma(r * z, fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                                     -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01), r);
    const double pc = fma(z * z, fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                                  2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02), fma(-0.5, z, 1.));
    const int q = (int)n;
    const double v = (q & 1) ? pc : ps;
    return (q & 2) ? -v : v;
}

// exp(x) for the 237 ordinates of a distance: round(x / ln 2), ln 2 in two pieces, the degree-13 Taylor polynomial on |r| <= ln(2) / 2 (remainder
// 4e-18), ldexp -- relative error below 2e-16, 20 instructions for the library's 35
__device__ __forceinline__ double exp_mid(double x) {
    x = x < -746. ? -746. : (x > 710. ? 710. : x);   // 0 and Inf beyond the range of double through ldexp; NaN passes
    const double n = rint(x * 1.4426950408889634);
    double r = fma(-n, 0.6931471803691238, x);
    r = fma(-n, 1.9082149292705877e-10, r);
    double p = 1. / 6227020800.;
    p = fma(p, r, 1. / 479001600.);
    p = fma(p, r, 1. / 39916800.);
    p = fma(p, r, 1. / 3628800.);
    p = fma(p, r, 1. / 362880.);
    p = fma(p, r, 1. / 40320.);
    p = fma(p, r, 1. / 5040.);
    p = fma(p, r, 1. / 720.);
    p = fma(p, r, 1. / 120.);
    p = fma(p, r, 1. / 24.);
    p = fma(p, r, 1. / 6.);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.);
    p = fma(p, r, 1.);
    return ldexp(p, (int)n);
}

// 1 / sqrt(x) for positive, finite, normal x: the hardware estimate (2^-23) and one third-order correction y (1 + e / 2 + 3 e^2 / 8), e = 1 - x y^2
__device__ __forceinline__ double rsqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

// 10^x for the epilogues (tables splined in log10 P): round(x log2 10), log10(2) in two pieces, 10^r = e^(r ln 10) by the degree-13 Taylor
// polynomial on |r ln 10| <= ln(2) / 2, ldexp -- relative error 2e-16 (checked against 50-digit arithmetic), half the instructions of the
// library's exp10, which matters where every lane of the GEMM epilogue takes 64 of them per tile
__device__ __forceinline__ double exp10_mid(double x) {
    // branch-free over the whole line: the exponent is clamped (so that the int conversion is defined for any x; arguments that far out are settled by the selects at the end), NaN goes
    // through the arithmetic, and -Inf / +Inf (whose reduced argument is not finite) are settled by the two selects at the end.  An unrolled
    // GEMM epilogue holds 64 copies of this function: a library call for the rare arguments would put 64 call sites, with their register
    // spills, into it.
    double n = rint(x * 3.321928094887362);
    n = fmax(fmin(n, 1100.), -1100.);
    double r = fma(-n, 0.3010299955494702, x);
    r = fma(-n, 1.1451100898021838e-10, r);
    const double y = r * 2.302585092994046;
    double p = 1. / 6227020800.;
    p = fma(p, y, 1. / 479001600.);
    p = fma(p, y, 1. / 39916800.);
    p = fma(p, y, 1. / 3628800.);
    p = fma(p, y, 1. / 362880.);
    p = fma(p, y, 1. / 40320.);
    p = fma(p, y, 1. / 5040.);
    p = fma(p, y, 1. / 720.);
    p = fma(p, y, 1. / 120.);
    p = fma(p, y, 1. / 24.);
    p = fma(p, y, 1. / 6.);
    p = fma(p, y, 0.5);
    p = fma(p, y, 1.);
    p = fma(p, y, 1.);
    double v = ldexp(p, (int)n);
    // the cut-offs sit where the true result leaves the doubles (10^-324 rounds to 0, 10^308.26 overflows): inside them |n| <= 1077, the clamp above
    // never binds and the polynomial only sees reduced arguments; outside, whatever it returned is replaced
    v = x < -324. ? 0. : v;
    v = x > 308.3 ? __builtin_inf() : v;
    return v;
}

// 10^x where every lane of a tile takes sixteen of them (cp_tables_rows_direct): n = round(64 x log2 10), 10^x = 2^(n >> 6) 2^((n & 63) / 64) 10^r
// with r = x - n log10(2) / 64 in two pieces (|r| <= 0.00236), 2^(j / 64) from a table of 64 correctly rounded doubles in LDS and 10^r - 1 by its
// degree-5 series (remainder 4e-17): relative error 2.0e-16 over |x| < 300 (checked against 50-digit arithmetic), 10 double-precision
// instructions where exp10_mid takes 20 and six selects.  NO range handling: the caller sends tiles that hold |x| >= 300, Inf or NaN to exp10_mid.
__device__ const double exp10_table[64] = {
    1, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284, 1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.1023825833078409, 1.1143867425958924, 1.1265216186082418, 1.1387886347566916, 1.1511892299529827, 1.1637248587775775,
    1.1763969916502812, 1.189207115002721, 1.2021567314527031, 1.215247359980469, 1.22848053610687, 1.241857812073484, 1.2553807570246911,
    1.2690509571917332, 1.2828700160787783, 1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.3396675240533029, 1.3542555469368927,
    1.3690024229745905, 1.383909881963832, 1.3989796725383112, 1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384, 1.5422108254079407, 1.5590044002378369, 1.5759808451078865,
    1.593142151342267, 1.6104903319492543, 1.6280274218573478, 1.6457554781539649, 1.6636765803267364, 1.681792830507429, 1.7001063537185235,
    1.7186192981224779, 1.7373338352737062, 1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989, 1.8340080864093424,
    1.8539791250833855, 1.8741676341103, 1.8945759815869656, 1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.9784560263879509};

__device__ __forceinline__ double exp10_tab(double x, const double* lds_table) {
    const double n = rint(x * 212.60339807279118);
    double r = fma(-n, 0.0047035936813699664, x);      // (33 significant bits: exact for |n| < 2^20)
    r = fma(-n, 8.7973981354298402e-13, r);
    double p = 0.5393829291955814;
    p = fma(p, r, 1.1712551489122669);
    p = fma(p, r, 2.034678592293476);
    p = fma(p, r, 2.6509490552391992);
    p = fma(p, r, 2.3025850929940459);
    p *= r;                                             // 10^r - 1
    const int ni = (int)n;
    const double t = lds_table[ni & 63];
    return ldexp(fma(t, p, t), ni >> 6);
}

}  // namespace cpmath
