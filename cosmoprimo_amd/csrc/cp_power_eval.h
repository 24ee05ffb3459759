// cp_power_eval.h -- the per-cosmology fit coefficients and the per-wavenumber evaluation of the analytic engines (EH98, EH no-wiggle, BBKS) as
// device functions, shared by the P(k) kernels (cp_power.hip) and by the fused sigma(r, z) kernel (cp_sigma.hip), which evaluates the spectra
// in the front end of its FFTLog instead of reading them.  Reference: eisenstein_hu.py:34-92, 189-215, 241-283, 315-324;
// eisenstein_hu_nowiggle.py:21-51; bbks.py:34-64.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/cosmoprimo_amd.h"
#include "cp_cosmo_common.h"
#include "cp_math.h"

namespace cppower {

using namespace cpcosmo;
using namespace cpmath;

struct EhScalars {
    double omega_b, omega_m, frac_b, theta_cmb, z_eq, k_eq, z_drag, r_drag, r_eq, rs_drag, k_silk, alpha_c, beta_c, alpha_b, beta_node, beta_b,
        alpha_gamma;
    double ln_q_over_kh, ln_ksilk_over_kh;  // log(q / kh) = log(h / (13.41 k_eq)) and log((k / k_silk) / kh) = log(h / k_silk): see transfer_eh
    double q108_per_kh, ksilk14_per_kh;     // (q / kh)^1.08 and ((k / k_silk) / kh)^1.4: with tabulated kh^1.08, kh^1.4 the two powers of a sample are products (transfer_eh_powers)
    double growth0;                         // CPT92 growth factor at z = 0 (Background.growth_factor(0, znorm=0)): the pre-kernel's lane evaluates it once per cosmology (coefficients_kernel), the sigma8 normalisation reads it
};

// x^y for positive, finite, normal x as exp(y log x) with the short forms of cp_math.h: relative error (|y log x| + 1) 2e-16 -- 1e-15 at most over the
// powers of the fits (|y log x| < 3.1) -- at 65 instructions for the library's ~200.  For the pre-kernel that forms the evaluation's constants: its one lane
// per cosmology walks ~25 of them in a row, and it runs in front of every evaluation (a fifteenth of brieden2022's kernel time, a twentieth of config 3).
__device__ __forceinline__ double pow_short(double x, double y) { return exp_mid(y * log_pos(x)); }

// eisenstein_hu.py:34-92 (+ eisenstein_hu_nowiggle.py:21), operation for operation.  SHORT: the powers through pow_short (the constants of the evaluating
// kernels: cosmo_consts); the scalars handed to the caller (cp_eh_scalars: rs_drag, z_drag, ...) take the library's pow.
template <bool SHORT = false>
__device__ __forceinline__ EhScalars eh_scalars(double h, double Omega_cdm, double Omega_b, double T_cmb, bool full) {
    auto pow = [](double x, double y) { return SHORT ? pow_short(x, y) : ::pow(x, y); };
    EhScalars s;
    s.growth0 = 0.;      // (set by the caller that has the whole cosmology)
    s.omega_b = Omega_b * (h * h);
    s.omega_m = Omega_cdm * (h * h) + Omega_b * (h * h);
    s.frac_b = s.omega_b / s.omega_m;
    s.theta_cmb = T_cmb / 2.7;
    const double th2 = s.theta_cmb * s.theta_cmb, thm4 = 1. / (th2 * th2), thm2 = 1. / th2;
    s.z_eq = 2.5e4 * s.omega_m * thm4 - 1.;
    s.k_eq = 0.0746 * s.omega_m * thm2;
    const double b1 = 0.313 * pow(s.omega_m, -0.419) * (1 + 0.607 * pow(s.omega_m, 0.674));
    const double b2 = 0.238 * pow(s.omega_m, 0.223);
    s.z_drag = 1345 * pow(s.omega_m, 0.251) / (1. + 0.659 * pow(s.omega_m, 0.828)) * (1. + b1 * pow(s.omega_b, b2));  // HS96 prefactor
    s.r_drag = 31.5 * s.omega_b * thm4 * (1000. / (1 + s.z_drag));
    s.r_eq = 31.5 * s.omega_b * thm4 * (1000. / (1 + s.z_eq));
    s.rs_drag = 2. / (3. * s.k_eq) * sqrt(6. / s.r_eq) * log((sqrt(1 + s.r_drag) + sqrt(s.r_drag + s.r_eq)) / (1 + sqrt(s.r_eq)));
    s.alpha_gamma = 1. - 0.328 * log(431. * s.omega_m) * s.frac_b + 0.38 * log(22.3 * s.omega_m) * (s.frac_b * s.frac_b);
    if (full) {
        s.k_silk = 1.6 * pow(s.omega_b, 0.52) * pow(s.omega_m, 0.73) * (1 + pow(10.4 * s.omega_m, -0.95));
        const double a1 = pow(46.9 * s.omega_m, 0.670) * (1 + pow(32.1 * s.omega_m, -0.532));
        const double a2 = pow(12.0 * s.omega_m, 0.424) * (1 + pow(45.0 * s.omega_m, -0.582));
        s.alpha_c = pow(a1, -s.frac_b) * pow(a2, -(s.frac_b * s.frac_b * s.frac_b));
        const double bb1 = 0.944 / (1 + pow(458 * s.omega_m, -0.708));
        const double bb2 = 0.395 * pow(s.omega_m, -0.0266);
        s.beta_c = 1. / (1 + bb1 * (pow(1 - s.frac_b, bb2)) - 1);
        const double y = (1 + s.z_eq) / (1 + s.z_drag);
        const double G = y * (-6. * sqrt(1 + y) + (2. + 3. * y) * log((sqrt(1 + y) + 1) / (sqrt(1 + y) - 1)));
        s.alpha_b = 2.07 * s.k_eq * s.rs_drag * pow(1 + s.r_drag, -0.75) * G;
        s.beta_node = 8.41 * pow(s.omega_m, 0.435);
        s.beta_b = 0.5 + s.frac_b + (3. - 2. * s.frac_b) * sqrt((17.2 * s.omega_m) * (17.2 * s.omega_m) + 1);
        s.ln_q_over_kh = log(h / (13.41 * s.k_eq));
        s.ln_ksilk_over_kh = log(h / s.k_silk);
        s.q108_per_kh = exp(1.08 * s.ln_q_over_kh);
        s.ksilk14_per_kh = exp(1.4 * s.ln_ksilk_over_kh);
    } else {
        s.k_silk = s.alpha_c = s.beta_c = s.alpha_b = s.beta_node = s.beta_b = 0.;
        s.ln_q_over_kh = s.ln_ksilk_over_kh = s.q108_per_kh = s.ksilk14_per_kh = 0.;
    }
    return s;
}

// What transfer_eh needs of one cosmology, in the units of the loop over wavenumbers (kh in h/Mpc): once per thread
struct EhPerCosmology {
    double q_per_kh;     // q = kh h / (13.41 k_eq)
    double ks_per_kh;    // ks = kh h rs_drag
    double c_alpha0;     // 14.2 / alpha_c
    double beta18, beta_node3, beta_b3, alpha_b, frac_b;
    double ln_q_over_kh, ln_ksilk_over_kh, q108_per_kh, ksilk14_per_kh;
};

__device__ __forceinline__ EhPerCosmology eh_per_cosmology(const EhScalars& s, double h) {
    EhPerCosmology d;
    d.q_per_kh = h * recip(13.41 * s.k_eq);
    d.ks_per_kh = h * s.rs_drag;
    d.c_alpha0 = 14.2 * recip(s.alpha_c);
    d.beta18 = 1.8 * s.beta_c;
    d.beta_node3 = s.beta_node * s.beta_node * s.beta_node;
    d.beta_b3 = s.beta_b * s.beta_b * s.beta_b;
    d.alpha_b = s.alpha_b;
    d.frac_b = s.frac_b;
    d.ln_q_over_kh = s.ln_q_over_kh;
    d.ln_ksilk_over_kh = s.ln_ksilk_over_kh;
    d.q108_per_kh = s.q108_per_kh;
    d.ksilk14_per_kh = s.ksilk14_per_kh;
    return d;
}

// Three of the six quotients of the EH98 fit through ONE reciprocal (1 / X = A B / (X A B): nine instructions fewer per sample, 1e-16 of T).  Round 5
// measured it slower (the longer live ranges cost spilled registers); with the spills gone (round 6) config 3 -1.5 %, wallish2018 +1.5 %
// (profiles/r6_eh_variants.txt).  0: the six reciprocals (measurements: tools/ab_variants.sh)
#ifndef CP_EH_MERGED_RECIP
#define CP_EH_MERGED_RECIP 1
#endif
#ifndef CP_MATH_TABLES_OFF      // 1: the polynomial forms everywhere (measurements)
#define CP_MATH_TABLES_OFF 0
#endif

// q108 = q^1.08, silk14 = (k / k_silk)^1.4: the two powers of the wavenumber the fit takes
__device__ __forceinline__ double transfer_eh_core(const EhPerCosmology& d, double kh, double q108, double silk14, const MathTables* mt) {
    if (CP_MATH_TABLES_OFF) mt = nullptr;
    const double q = kh * d.q_per_kh;
    const double ks = kh * d.ks_per_kh;
    // mt: the table-driven logarithm and exponential (cp_math.h) where the kernel keeps their tables in LDS, else (null) the polynomial forms
    const double ln_beta = mt ? log_tab(kE + d.beta18 * q, mt) : log_pos(kE + d.beta18 * q);      // (arguments >= e: log_tab's relative error is log_pos's there)
    const double ln_nobeta = mt ? log_tab(kE + 1.8 * q, mt) : log_pos(kE + 1.8 * q);
    const double c386 = 386. * recip(1 + 69.9 * q108);
    const double C_alpha = d.c_alpha0 + c386, C_noalpha = 14.2 + c386;
    const double ks54 = ks * (1. / 5.4), ks52 = ks * (1. / 5.2);
    const double f = recip(1. + (ks54 * ks54) * (ks54 * ks54));          // T_c_f
    const double q2 = q * q;
    // T_c = f ln_beta / d1 + (1 - f) ln_beta / d2
    const double d1 = ln_beta + C_noalpha * q2, d2 = ln_beta + C_alpha * q2;
    // 1 + (beta / ks)^3 = (ks^3 + beta^3) / ks^3
    const double ks3 = ks * ks * ks;
    const double ks_tilde = ks * ks * rcbrt(ks3 + d.beta_node3);         // k rs_drag / cbrt(1 + (beta_node / ks)^3)
#if CP_EH_MERGED_RECIP
    // the three quotients 1 / (d1 d2), 1 / A, 1 / B (A, B: the denominators of T_b_1, T_b_2) through ONE reciprocal: 1 / (X A B) = w, 1 / X = w A B, ...
    const double X = d1 * d2, A = (ln_nobeta + C_noalpha * q2) * fma(ks52, ks52, 1.), B = ks3 + d.beta_b3;
    const double AB = A * B, XA = X * A;
    const double w = recip(X * AB);
    const double T_c = ln_beta * fma(f, d2 - d1, d1) * (w * AB);
    const double T_b_1 = ln_nobeta * (w * (X * B));
    const double T_b_2 = d.alpha_b * ks3 * (w * XA) * (mt ? exp_tab(-silk14, mt) : exp_mid(-silk14));
#else
    const double T_c = ln_beta * fma(f, d2 - d1, d1) * recip(d1 * d2);
    const double T_b_1 = ln_nobeta * recip((ln_nobeta + C_noalpha * q2) * fma(ks52, ks52, 1.));
    const double T_b_2 = d.alpha_b * ks3 * recip(ks3 + d.beta_b3) * (mt ? exp_tab(-silk14, mt) : exp_mid(-silk14));
#endif
    const double sinc = ks_tilde == 0. ? 1. : sin_bounded(ks_tilde) * recip(ks_tilde);   // numpy.sinc(x / pi)
    const double T_b = sinc * (T_b_1 + T_b_2);
    return d.frac_b * T_b + (1 - d.frac_b) * T_c;
}

// eisenstein_hu.py:252-283, the same rational functions with their quotients gathered (6 reciprocals instead of 16 divisions) and the cosmology-only
// factors taken out of the loop over wavenumbers; within 1e-14 of the operation-for-operation form (tests/test_cosmology_gpu.py, 1e-11 against
// the reference's numbers).  ln_kh = log(kh): the powers of q and k / k_silk go through it, x^p = exp(p (log kh + log(x / kh))), one log
// shared by the three powers of k of a P(k) evaluation instead of a pow() each; the relative error of exp(p log x) is |p log x| eps < 2e-15 here.
__device__ __forceinline__ double transfer_eh(const EhPerCosmology& d, double kh, double ln_kh, const MathTables* mt) {
    const bool tab = mt && !CP_MATH_TABLES_OFF;
    const double q108 = tab ? exp_tab(1.08 * (ln_kh + d.ln_q_over_kh), mt) : exp_mid(1.08 * (ln_kh + d.ln_q_over_kh));
    const double silk14 = tab ? exp_tab(1.4 * (ln_kh + d.ln_ksilk_over_kh), mt) : exp_mid(1.4 * (ln_kh + d.ln_ksilk_over_kh));
    return transfer_eh_core(d, kh, q108, silk14, mt);
}

// ... on a grid of wavenumbers shared by the cosmologies of a launch: kh^1.08 and kh^1.4 come from tables next to log kh (powers_of_wavenumbers), and
// the two powers are products with the cosmology's own factors -- two exponentials fewer per sample (3e-16 from the form above)
#ifndef CP_K_POWER_TABLES      // 0: the exponentials of log kh instead (measurements)
#define CP_K_POWER_TABLES 1
#endif
__device__ __forceinline__ double transfer_eh_powers(const EhPerCosmology& d, double kh, double ln_kh, double kh108, double kh14, const MathTables* mt) {
    if (!CP_K_POWER_TABLES) return transfer_eh(d, kh, ln_kh, mt);
    return transfer_eh_core(d, kh, kh108 * d.q108_per_kh, kh14 * d.ksilk14_per_kh, mt);
}

// log k, k^1.08, k^1.4 of the n wavenumbers of a launch: tab (3, n)
__device__ __forceinline__ void powers_of_wavenumber(double k, double* tab, int i, int n) {
    const double l = log_pos(k);
    tab[i] = l;
    tab[n + i] = exp(1.08 * l);
    tab[2 * n + i] = exp(1.4 * l);
}

__device__ __forceinline__ double transfer_nowiggle(const EhScalars& s, double h, double kh, const MathTables* mt) {  // eisenstein_hu_nowiggle.py:45-51
    if (CP_MATH_TABLES_OFF) mt = nullptr;
    const double k = kh * h;
    const double ks = k * s.rs_drag;
    const double x = 0.43 * ks;
    // the four quotients as reciprocals (the hardware estimate and two corrections, 1 ulp: a third of the instructions of an IEEE division -- half of what a
    // sample cost was its divisions); every denominator is >= 1 or a positive constant of the cosmology
    const double gamma_eff = s.omega_m * (s.alpha_gamma + (1 - s.alpha_gamma) * recip(1 + (x * x) * (x * x)));
    const double q = k * (s.theta_cmb * s.theta_cmb) * recip(gamma_eff);
    const double L0 = mt ? log_tab(2 * kE + 1.8 * q, mt) : log_pos(2 * kE + 1.8 * q);
    const double C0 = 14.2 + 731.0 * recip(1 + 62.5 * q);
    return L0 * recip(L0 + C0 * (q * q));
}

// bbks.py:38: gamma = omega_m exp(-Omega_b (1 + sqrt(2 h) / Omega_m)) with the cosmology's Omega_m, massive neutrinos included (cosmology.py:381)
__device__ __forceinline__ double bbks_gamma(const Cosmo& c) {
    const double Omega_m = c.Omega_b + c.Omega_cdm + c.Omega_nu_m;
    return Omega_m * (c.h * c.h) * exp(-c.Omega_b * (1. + sqrt(2. * c.h) / Omega_m));
}

// What one P(k) evaluation needs of a cosmology besides its transfer function: primordial tilt and the constant that turns T^2 k into P
// (Primordial.pk_k, eisenstein_hu.py:214-215; pk_callable, eisenstein_hu.py:321-324: potential_to_density^-2 x curvature_to_potential x
// h^3 A_s = kh x a constant of the cosmology -- the h^3 of the primordial spectrum and of curvature_to_potential cancel)
struct PkPerCosmology {
    double ns_m1, half_alpha_s, sixth_beta_s;      // n_s - 1, alpha_s / 2, beta_s / 6 as the exponent of the tilt takes them (the same products, formed once: in the loop
                                                   // their literal factors were vector registers -- two scalar operands do not go into one instruction)
    double ln_kp, pk_unit, h3_A_s;
};

// mt: the kernel's tables for the short logarithm (cp_math.h) -- every thread of a kernel goes through this once per cosmology, and with the library's
// logarithm and IEEE divisions that was 6 % of the instructions of the transform that generates its own spectra (two cosmologies x 32 samples per thread)
__device__ __forceinline__ PkPerCosmology pk_per_cosmology(const Cosmo& c, const double* pw, const MathTables* mt = nullptr) {
    PkPerCosmology p;
    const double A_s = pw[CP_PK_A_S];
    p.ns_m1 = pw[CP_PK_N_S] - 1.;
    p.half_alpha_s = 1. / 2. * pw[CP_PK_ALPHA_S];
    p.sixth_beta_s = 1. / 6. * pw[CP_PK_BETA_S];
    const double Omega0_m = c.Omega_b + c.Omega_cdm + c.Omega_nu_m;  // cosmology.py:381: + Omega_ncdm_tot - Omega_pncdm_tot (ba.Omega0_m of pk_callable, eisenstein_hu.py:322)
    const double p2d_unit = 3. * Omega0_m * (100. * 100.) * (1. / (2. * (kCkms * kCkms)));
    if (mt && !CP_MATH_TABLES_OFF) {
        p.ln_kp = log_tab_any(pw[CP_PK_K_PIVOT] * recip(c.h), mt);
        p.pk_unit = 9. / 25. * 2. * (kPi * kPi) * recip(p2d_unit * p2d_unit) * A_s;
    } else {
        p.ln_kp = log(pw[CP_PK_K_PIVOT] / c.h);
        p.pk_unit = 9. / 25. * 2. * (kPi * kPi) / (p2d_unit * p2d_unit) * A_s;
    }
    p.h3_A_s = (c.h * c.h * c.h) * A_s;
    return p;
}

#ifndef CP_CONSTS_SHORT_POW      // 0: the library's pow in the pre-kernel (measurements)
#define CP_CONSTS_SHORT_POW 1
#endif
// Everything an evaluation of P(k) needs of ONE cosmology, in the units of the loops over wavenumbers: formed once per cosmology by one lane of
// coefficients_kernel (cp_power.hip) and read back by the evaluating kernels with load_uniform -- scalar loads: the values sit in scalar registers, where
// formed per thread (the parameters gathered, Omega_g by two IEEE divisions, two reciprocals and a logarithm: ~250 instructions per thread and
// cosmology, a tenth of a thread's evaluations in the fused sigma(r, z) kernel) they were 36 vector registers of a kernel that has none to spare.
struct CosmoConsts {
    EhScalars s;             // the fit coefficients themselves (the no-wiggle form reads four of them; growth0 for the sigma8 normalisation)
    EhPerCosmology eh;       // EH98 (zeros for the other engines)
    PkPerCosmology pk;       // primordial tilt and the constant that turns T^2 k into P
    double h, bbks_gamma;    // BBKS: q = kh h / gamma
    double ln_pk_unit;       // log(pk.pk_unit): the log(k P) form (CP_PK_LOG_K_MATTER)
    double A_s;
};

// pw: the CP_PK_NPARAMS primordial parameters of the cosmology, or null (the constants of the spectrum are then zero: transfer functions only)
__device__ inline CosmoConsts cosmo_consts(const Cosmo& c, const double* pw, int engine) {
    CosmoConsts K{};
    if (engine != CP_ENGINE_BBKS) {
        K.s = eh_scalars<CP_CONSTS_SHORT_POW != 0>(c.h, c.Omega_cdm, c.Omega_b, c.T_cmb, engine == CP_ENGINE_EH);
        if (engine == CP_ENGINE_EH) K.eh = eh_per_cosmology(K.s, c.h);
    }
    K.s.growth0 = growth_cpt(c, 0.);
    K.h = c.h;
    K.bbks_gamma = bbks_gamma(c);
    if (pw) {
        K.pk = pk_per_cosmology(c, pw, nullptr);
        K.ln_pk_unit = log(K.pk.pk_unit);
        K.A_s = pw[CP_PK_A_S];
    }
    return K;
}

// *p for a wave-uniform p into scalar registers: the loads go through the constant address space (scalar loads; only the fields that are used are
// loaded).  The memory must not be written by the kernel that reads it this way (the scalar cache is not coherent with the vector stores of a launch).
template <class T>
__device__ __forceinline__ T load_uniform(const T* p) {
    static_assert(sizeof(T) % sizeof(double) == 0 && alignof(T) <= alignof(double), "made of doubles");
    const double __attribute__((address_space(4)))* q = (const double __attribute__((address_space(4)))*)(unsigned long long)p;
    T v;
    double* d = reinterpret_cast<double*>(&v);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / sizeof(double)); ++i) d[i] = q[i];
    return v;
}

// the transfer function of engine ENGINE from the cosmology's constants
__device__ __forceinline__ double transfer_bbks(double h, double gamma, double kh) {  // bbks.py:34-38, 62-64
    const double q = kh * h / gamma;
    const double x = 2.34 * q;
    const double a = 16.2 * q, b = 5.47 * q, c = 6.71 * q;
    // as coded in the reference: 3.89 q (16.2 q)^2, not 3.89 q + (16.2 q)^2
    return log(1 + x) / x / sqrt(sqrt(1. + 3.89 * q * (a * a) + b * b * b + (c * c) * (c * c)));
}

template <int ENGINE>
__device__ __forceinline__ double transfer_any(const CosmoConsts& K, double kh, double ln_kh, double kh108, double kh14, const MathTables* mt) {
    if (ENGINE == CP_ENGINE_BBKS) return transfer_bbks(K.h, K.bbks_gamma, kh);
    return ENGINE == CP_ENGINE_EH ? transfer_eh_powers(K.eh, kh, ln_kh, kh108, kh14, mt) : transfer_nowiggle(K.s, K.h, kh, mt);
}

__device__ __forceinline__ double primordial_tilt_exponent(const PkPerCosmology& p, double ln_kh) {
    const double lnkkp = ln_kh - p.ln_kp;
    return (p.ns_m1 + p.half_alpha_s * lnkkp + p.sixth_beta_s * (lnkkp * lnkkp)) * lnkkp;      // eisenstein_hu.py:214: n_s - 1 + alpha_s / 2 ln + beta_s / 6 ln^2
}
__device__ __forceinline__ double primordial_tilt(const PkPerCosmology& p, double ln_kh, const MathTables* mt) {
    return (mt && !CP_MATH_TABLES_OFF) ? exp_tab(primordial_tilt_exponent(p, ln_kh), mt) : exp_mid(primordial_tilt_exponent(p, ln_kh));
}

}  // namespace cppower
