// cp_math.h -- short forms of the transcendental functions the ALU-bound kernels spend their time in (device code, gfx950).
// Each is the textbook argument reduction + polynomial of its function, accurate to 1-2 ulp (stated per function, checked against extended
// precision), at a third to a half of the instructions of the library versions, which carry double-double intermediates and the handling of
// subnormal / overflowing arguments that these kernels' arguments never need; out-of-range arguments go to the library functions out of line.
#pragma once
#include <hip/hip_runtime.h>

namespace cpmath {

// One Horner step p r + c as ONE instruction.  hipcc selects v_fmac_f64 (d += a b: the addend is the destination) for fma(p, r, c) and, the constant c
// being needed again at the next sample, copies it into the destination first: two instructions per step, a fifth of the vector instructions of the
// polynomial-heavy loops (the EH98 evaluation: 115 copies in 601).  The three-address form takes the constant where it lives: -6 % of that loop's vector
// instructions, -2 % on wallish2018 and the distance quadrature (tools/ab_asm_fma.sh).
#ifndef CP_ASM_FMA      // 0: plain fma() (measurements: tools/ab_asm_fma.sh)
#define CP_ASM_FMA 1
#endif
#ifndef CP_ASM_FMA_POLY      // the same for the polynomial form exp_mid (what sigma_rz_kernel evaluates: its tables are off); 0: plain fma() (measurements).  Round 5
#define CP_ASM_FMA_POLY 1    // measured it slower (18 -> 42 spilled registers in sigma_rz_kernel); without spills, with the merged reciprocals: config 3 -3 % (profiles/r6_eh_variants.txt)
#endif
__device__ __forceinline__ double horner(double p, double r, double c) {
#if CP_ASM_FMA && defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(r), "v"(c));
    return d;
#else
    return fma(p, r, c);
#endif
}

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
    const double ps = fma(r * z, horner(horner(horner(horner(horner(1.58969099521155010221e-10, z, -2.50507602534068634195e-08), z, 2.75573137070700676789e-06), z,
                                                     -1.98412698298579493134e-04), z, 8.33333333332248946124e-03), z, -1.66666666666666324348e-01), r);
    const double pc = fma(z * z, horner(horner(horner(horner(horner(-1.13596475577881948265e-11, z, 2.08757232129817482790e-09), z, -2.75573143513906633035e-07), z,
                                                  2.48015872894767294178e-05), z, -1.38888888888741095749e-03), z, 4.16666666666666019037e-02), fma(-0.5, z, 1.));
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
#if CP_ASM_FMA_POLY
#define CP_POLY_STEP(p, r, c) horner(p, r, c)
#else
#define CP_POLY_STEP(p, r, c) fma(p, r, c)
#endif
    p = CP_POLY_STEP(p, r, 1. / 479001600.);
    p = CP_POLY_STEP(p, r, 1. / 39916800.);
    p = CP_POLY_STEP(p, r, 1. / 3628800.);
    p = CP_POLY_STEP(p, r, 1. / 362880.);
    p = CP_POLY_STEP(p, r, 1. / 40320.);
    p = CP_POLY_STEP(p, r, 1. / 5040.);
    p = CP_POLY_STEP(p, r, 1. / 720.);
    p = CP_POLY_STEP(p, r, 1. / 120.);
    p = CP_POLY_STEP(p, r, 1. / 24.);
    p = CP_POLY_STEP(p, r, 1. / 6.);
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

// ---- table-driven forms: 64 entries in LDS (a gather of 8- or 16-byte entries no two of which share a bank), short polynomials ----
// 10^x where every lane of a tile takes sixteen of them (cp_tables_rows_direct): n = round(64 x log2 10), 10^x = 2^(n >> 6) 2^((n & 63) / 64) 10^r
// with r = x - n log10(2) / 64 in two pieces (|r| <= 0.00236), 2^(j / 64) from a table of 64 correctly rounded doubles in LDS and 10^r - 1 by its
// degree-5 series (remainder 4e-17): relative error 2.4e-16 over |x| < 300 (tests/test_math_gpu.py, against 80-bit arithmetic), 10 double-precision
// instructions where exp10_mid takes 20 and six selects.  NO range handling: the caller sends tiles that hold |x| >= 300, Inf or NaN to exp10_mid.
// 2^(j / 64), j < 64, correctly rounded (shared by exp10_tab and exp_tab)
__device__ const double exp2_table[64] = {
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

// {1 / c_j, -log(1 / c_j)}, c_j = (1 + (j + 1/2) / 64) / 2 the midpoint of the j-th of 64 equal pieces of [1/2, 1): log m = log(m / c_j) + log c_j
__device__ const double log_table[128] = {
    1.9844961240310077, -0.68536504011789035, 1.9541984732824427, -0.66998012127841089, 1.9248120300751879, -0.65482831625780868,
    1.8962962962962964, -0.63990266604113311, 1.8686131386861313, -0.62519651865143755, 1.8417266187050361, -0.6107035113488708,
    1.8156028368794326, -0.59641755410139419, 1.7902097902097902, -0.58233281421965521, 1.7655172413793103, -0.56844370205898809,
    1.7414965986394557, -0.55474485770082615, 1.7181208053691275, -0.54123113853410332, 1.695364238410596, -0.52789760766463811,
    1.673202614379085, -0.514739523087127, 1.6516129032258065, -0.50175232756031585, 1.6305732484076434, -0.48893163913125448,
    1.6100628930817611, -0.47627324225933099, 1.5900621118012421, -0.46377307949509944, 1.5705521472392638, -0.45142724367280018,
    1.5515151515151515, -0.43923197057898189, 1.532934131736527, -0.42718363206280741, 1.514792899408284, -0.41527872955648898,
    1.4970760233918128, -0.40351388797690257, 1.4797687861271676, -0.3918858499817835, 1.4628571428571429, -0.38039147055604844,
    1.4463276836158192, -0.3690277119057333, 1.4301675977653632, -0.35779163863880753, 1.4143646408839778, -0.34668041321373666,
    1.3989071038251366, -0.33569129163814154, 1.3837837837837839, -0.32482161940123772, 1.3689839572192513, -0.31406882762497579,
    1.3544973544973544, -0.30343042941992004, 1.3403141361256545, -0.29290401643293268, 1.3264248704663213, -0.28248725557467697,
    1.3128205128205128, -0.27217788591581565, 1.2994923857868019, -0.26197371574157391, 1.2864321608040201, -0.25187261975507008,
    1.2736318407960199, -0.2418725364204867, 1.2610837438423645, -0.23197146543777517, 1.248780487804878, -0.22216746534115431,
    1.2367149758454106, -0.21245865121419336, 1.2248803827751196, -0.20284319251475144, 1.2132701421800949, -0.19331931100349606,
    1.2018779342723005, -0.18388527877013738, 1.1906976744186046, -0.17453941635189965, 1.1797235023041475, -0.16528009093910292,
    1.1689497716894977, -0.15610571466306161, 1.158371040723982, -0.14701474296180975, 1.147982062780269, -0.13800567301944369,
    1.1377777777777778, -0.12907704227514236, 1.1277533039647578, -0.12022742699815989, 1.1179039301310043, -0.11145544092532278,
    1.1082251082251082, -0.10275973395776894, 1.0987124463519313, -0.094138990913861909, 1.0893617021276596, -0.085591930335403535,
    1.0801687763713079, -0.077117303344431204, 1.0711297071129706, -0.068713892548051728, 1.0622406639004149, -0.060380510988907482,
    1.0534979423868314, -0.052116001139014101, 1.0448979591836736, -0.043919233934835579, 1.0364372469635628, -0.035789107851585289,
    1.0281124497991967, -0.027724548014854768, 1.0199203187250996, -0.019724505347778573, 1.0118577075098814, -0.011787955752042173,
    1.003921568627451, -0.0039138993211363148};

struct MathTables {      // in LDS, one per workgroup
    double exp2[64];
    double logc[128];
};

// The tables of a kernel that has them, as the evaluation functions take them (they accept null: "no tables, the polynomial forms").  Behind the
// address-space cast of a __shared__ variable the compiler does not see that the pointer is not null: without this the kernels carried BOTH forms of every
// logarithm and exponential with a run-time test in front of each (and the polynomial forms' constants in registers).
__device__ __forceinline__ const MathTables* tables_present(const MathTables* t) {
    __builtin_assume(t != nullptr);
    return t;
}

// every thread of the workgroup calls this; a barrier follows at the caller's
__device__ __forceinline__ void fill_math_tables(MathTables* t) {
    if (blockDim.x >= 128) {      // both entries of the thread requested before either is stored: one memory round trip in front of the barrier, not two
        const int i = threadIdx.x;
        const double e = exp2_table[i & 63], l = log_table[i & 127];
        if (i < 64) t->exp2[i] = e;
        if (i < 128) t->logc[i] = l;
        return;
    }
    for (int i = threadIdx.x; i < 64; i += blockDim.x) t->exp2[i] = exp2_table[i];
    for (int i = threadIdx.x; i < 128; i += blockDim.x) t->logc[i] = log_table[i];
}

// e^x: n = round(64 x / ln 2), ln(2) / 64 in two pieces (|r| <= 0.0055), e^r - 1 by its degree-5 series (remainder 4e-17), 2^((n & 63) / 64)
// from the table, ldexp -- relative error 2.3e-16 (tests/test_math_gpu.py, against 80-bit arithmetic over |x| < 700), 11 double-precision instructions where
// exp_mid takes 20.  Arguments below -800 give 0 (as beyond -745 anyway), NaN passes.
// (exp_tab_core: without the clamp of very negative arguments -- for callers whose arguments stay within +-2e7: beyond, the conversion of n saturates,
// the result is still 0 / Inf of the right sign through ldexp, but the clamp keeps the reduction exact down to the underflow)
__device__ __forceinline__ double exp_tab_core(double x, const MathTables* t);
__device__ __forceinline__ double exp_tab(double x, const MathTables* t) {
    return exp_tab_core(x < -800. ? -800. : x, t);
}
__device__ __forceinline__ double exp_tab_core(double x, const MathTables* t) {
    const double n = rint(x * 92.332482616893657);
    double r = fma(-n, 0.010830424695086549, x);      // (33 significant bits: exact for |n| < 2^20)
    r = fma(-n, 1.162596423439437e-12, r);
    double p = 1. / 120.;
    p = horner(p, r, 1. / 24.);
    p = horner(p, r, 1. / 6.);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.);
    p *= r;
    const int ni = (int)n;
    const double e = t->exp2[ni & 63];
    return ldexp(fma(e, p, e), ni >> 6);
}

// log(x) for positive, finite, normal x: x = 2^e m, m in [1/2, 1) split into 64 pieces by its top mantissa bits, r = m / c_j - 1 (|r| < 0.0079, one
// fma on the tabulated reciprocal), log(1 + r) by its degree-7 series (remainder 2e-18): ABSOLUTE error below 2e-16 max(1, |log x|) -- for
// arguments away from 1 (every logarithm of the fits: log(e + ...)) the relative error of log_pos at 17 instructions for its 40; next to 1 the
// result keeps its absolute, not its relative accuracy.  Zero, negative, subnormal, Inf, NaN: not handled (log_pos).
__device__ __forceinline__ double log_tab(double x, const MathTables* t) {
    const double m = __builtin_amdgcn_frexp_mant(x);
    const int e = __builtin_amdgcn_frexp_exp(x);
    const int j = (__double2hiint(m) >> 14) & 63;
    const double inv = t->logc[2 * j], lc = t->logc[2 * j + 1];
    const double r = fma(m, inv, -1.);
    double p = 1. / 7.;
    p = horner(p, r, -1. / 6.);
    p = horner(p, r, 0.2);
    p = horner(p, r, -0.25);
    p = horner(p, r, 1. / 3.);
    p = fma(p, r, -0.5);
    p = fma(p, r, 1.);
    return fma((double)e, 6.93147180559945309417e-01, fma(p, r, lc));
}

// log_tab where the argument may leave its domain (the logarithm of |T(k)|: a transfer function can cross zero): the library's answers there
__device__ __forceinline__ double log_tab_any(double x, const MathTables* t) {
    if (!(x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308)) return log_any(x);
    return log_tab(x, t);
}

}  // namespace cpmath
