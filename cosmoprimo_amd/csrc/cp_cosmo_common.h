// cp_cosmo_common.h -- constants and the per-cosmology parameter block shared by the background and P(k) kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/cosmoprimo_amd.h"
#include "cp_math.h"
#include "cp_error.h"

namespace cpcosmo {

using cpmath::exp_mid;
using cpmath::rsqrt_pos;

// physical constants as cosmoprimo/constants.py (scipy.constants values, scipy 1.15)
constexpr double kC = 299792458.0;
constexpr double kStefanBoltzmann = 5.6703744191844314e-08;  // scipy.constants.Stefan_Boltzmann (derived to full precision)
constexpr double kParsec = 3.085677581491367e16;
constexpr double kG = 6.6743e-11;
constexpr double kPi = 3.141592653589793;
constexpr double kE = 2.718281828459045;
constexpr double kMpc = 1e6 * kParsec;
constexpr double kMsun = 1.98847 * 1e30;
constexpr double kRhoCritKg = 3.0 * (100. * 1e3 / kMpc) * (100. * 1e3 / kMpc) / (8 * kPi * kG);  // h^2 kg/m^3
constexpr double kRhoCrit = kRhoCritKg / (1e10 * kMsun) * kMpc * kMpc * kMpc;                      // 1e10 Msun/h / (Mpc/h)^3
constexpr double kCkms = kC / 1e3;

struct Param {
    const double* ptr;  // per-cosmology array, or nullptr ->
    double value;       // broadcast value
};

struct Cosmo {
    double h, Omega_cdm, Omega_b, Omega_k, T_cmb, N_ur, w0, wa;  // inputs
    double Omega_g, Omega_ur, Omega_de;                            // derived, cosmology.py:355-383
    double Omega_nu_m;                                             // Omega_ncdm_tot - Omega_pncdm_tot today (0 without massive species): Omega0_m = Omega_b + Omega_cdm + this, cosmology.py:381
    // massive neutrinos: spline tables of this cosmology (nsp, 4, CP_NCDM_NKNOTS) + the shared knots; nsp == 0: none
    const double* ncdm_tab;
    const double* ncdm_knots;
    int nsp;
};

// natural cubic spline (values y, second derivatives m on the knots x) at z inside [x[0], x[n-1]]
__device__ __forceinline__ double spline_m_eval(const double* x, const double* y, const double* m, int n, double z) {
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (z >= x[mid]) lo = mid;
        else hi = mid;
    }
    const double hh = x[lo + 1] - x[lo];
    const double a = (x[lo + 1] - z) / hh, b = (z - x[lo]) / hh;
    return a * y[lo] + b * y[lo + 1] + ((a * a * a - a) * m[lo] + (b * b * b - b) * m[lo + 1]) * (hh * hh) / 6.;
}

// sum over species (or one species) of the interpolated comoving density (which = 0) or pressure (which = 1) of massive neutrinos:
// DefaultBackground.rho_ncdm / p_ncdm (cosmology.py:1961-1998); NaN outside the knots as Interpolator1D without extrapolation
__device__ __forceinline__ double ncdm_eval(const Cosmo& c, double z, int which, int species = -1) {
    if (c.nsp == 0) return 0.;
    if (!(z >= c.ncdm_knots[0] && z <= c.ncdm_knots[CP_NCDM_NKNOTS - 1])) return __builtin_nan("");
    double tot = 0.;
    for (int s = 0; s < c.nsp; ++s) {
        if (species >= 0 && s != species) continue;
        const double* t = c.ncdm_tab + ((long long)s * 4 + 2 * which) * CP_NCDM_NKNOTS;
        tot = tot + spline_m_eval(c.ncdm_knots, t, t + CP_NCDM_NKNOTS, CP_NCDM_NKNOTS, z);
    }
    return tot;
}

// load the 8 background parameters of cosmology ic and derive the radiation / dark-energy densities
__device__ __forceinline__ Cosmo load_cosmo(const Param* p, long long ic, int second_is_omega_m, const double* ncdm_tab = nullptr,
                                            const double* ncdm_knots = nullptr, int nsp = 0) {
    double v[CP_BG_NPARAMS];
#pragma unroll
    for (int k = 0; k < CP_BG_NPARAMS; ++k) v[k] = p[k].ptr ? p[k].ptr[ic] : p[k].value;
    Cosmo c;
    c.h = v[CP_BG_H];
    c.Omega_b = v[CP_BG_OMEGA_B];
    c.Omega_cdm = second_is_omega_m ? v[CP_BG_OMEGA_CDM] - v[CP_BG_OMEGA_B] : v[CP_BG_OMEGA_CDM];  // cosmology.py:1163-1165
    c.Omega_k = v[CP_BG_OMEGA_K];
    c.T_cmb = v[CP_BG_T_CMB];
    c.N_ur = v[CP_BG_N_UR];
    c.w0 = v[CP_BG_W0_FLD];
    c.wa = v[CP_BG_WA_FLD];
    const double h2rc = c.h * c.h * kRhoCritKg;
    c.Omega_g = (c.T_cmb * c.T_cmb * c.T_cmb * c.T_cmb) * 4. / (kC * kC * kC) * kStefanBoltzmann / h2rc;
    const double T_ur = c.T_cmb * 0.7137658555036082;  // (4/11)^(1/3)
    c.Omega_ur = c.N_ur * 7. / 8. * (T_ur * T_ur * T_ur * T_ur) * 4. / (kC * kC * kC) * kStefanBoltzmann / h2rc;
    c.nsp = nsp;
    c.ncdm_knots = ncdm_knots;
    c.ncdm_tab = nsp ? ncdm_tab + ic * (long long)nsp * 4 * CP_NCDM_NKNOTS : nullptr;
    double Omega_ncdm_tot = 0.;
    c.Omega_nu_m = 0.;
    if (nsp) {  // z = 0 is the first knot: the tabulated values are the reference's _get_ncdm(z=0) (cosmology.py:371-376)
        double rho0 = 0., p0 = 0.;
        for (int s = 0; s < nsp; ++s) {
            rho0 = rho0 + c.ncdm_tab[((long long)s * 4 + 0) * CP_NCDM_NKNOTS];
            p0 = p0 + c.ncdm_tab[((long long)s * 4 + 2) * CP_NCDM_NKNOTS];
        }
        Omega_ncdm_tot = rho0 / kRhoCrit;
        c.Omega_nu_m = rho0 / kRhoCrit - 3. * p0 / kRhoCrit;
        if (second_is_omega_m) c.Omega_cdm = c.Omega_cdm - (rho0 - 3 * p0) / kRhoCrit;  // cosmology.py:1163-1165
    }
    c.Omega_de = 1. - (c.Omega_cdm + c.Omega_b + c.Omega_g + c.Omega_ur + Omega_ncdm_tot + c.Omega_k);
    return c;
}

// comoving critical density rho_crit(z) / rho_c0 pieces, operation order of BaseBackground (cosmology.py:1723-1749)
__device__ __forceinline__ double rho_de(const Cosmo& c, double zp1) {
    return c.Omega_de * pow(zp1, 3. * (c.w0 + c.wa)) * exp(3. * c.wa * (1. / zp1 - 1.)) * kRhoCrit;
}

__device__ __forceinline__ double rho_crit(const Cosmo& c, double zp1) {
    const double m = c.Omega_cdm * kRhoCrit + c.Omega_b * kRhoCrit + ncdm_eval(c, zp1 - 1., 0);
    const double r = c.Omega_g * zp1 * kRhoCrit + c.Omega_ur * zp1 * kRhoCrit;
    return (m + r + rho_de(c, zp1)) + c.Omega_k / zp1 * kRhoCrit;
}

// E(z) with log(1 + z) given (tabulated by the caller): (1 + z)^(3 (w0 + wa)) exp(3 wa (1 / (1 + z) - 1)) is one exp(); its relative error,
// |exponent| eps < 1e-14, sits in the dark-energy term only, far below the 1e-10 the background is held to.  A cosmological constant
// (w0 = -1, wa = 0) needs no transcendental function at all.
__device__ __forceinline__ double efunc_ln(const Cosmo& c, double z, double lzp1) {
    const double zp1 = 1. + z;
    const double m = c.Omega_cdm * kRhoCrit + c.Omega_b * kRhoCrit + ncdm_eval(c, z, 0);
    const double r = c.Omega_g * zp1 * kRhoCrit + c.Omega_ur * zp1 * kRhoCrit;
    const double de = (c.w0 == -1. && c.wa == 0.) ? c.Omega_de / (zp1 * zp1 * zp1) * kRhoCrit
                                                  : c.Omega_de * exp(3. * (c.w0 + c.wa) * lzp1 + 3. * c.wa * (1. / zp1 - 1.)) * kRhoCrit;
    const double rc = (m + r + de) + c.Omega_k / zp1 * kRhoCrit;
    return sqrt(rc * (zp1 * zp1 * zp1) / kRhoCrit);
}

// 1 / E(z) with log(1 + z) and 1 / (1 + z) given: rsqrt of E^2 (no division, no sqrt).  mt: the kernel's tables for the table-driven exponential
// (cp_math.h: 11 instructions where the polynomial form takes 20; 2.3e-16), or null
__device__ __forceinline__ double inv_efunc_ln(const Cosmo& c, double z, double lzp1, double izp1, const cpmath::MathTables* mt = nullptr) {
    const double zp1 = 1. + z;
    const double m = c.Omega_cdm * kRhoCrit + c.Omega_b * kRhoCrit + ncdm_eval(c, z, 0);
    const double r = c.Omega_g * zp1 * kRhoCrit + c.Omega_ur * zp1 * kRhoCrit;
    const double arg = 3. * (c.w0 + c.wa) * lzp1 + 3. * c.wa * (izp1 - 1.);
    const double de = (c.w0 == -1. && c.wa == 0.) ? c.Omega_de * (izp1 * izp1 * izp1) * kRhoCrit
                                                  : c.Omega_de * (mt ? cpmath::exp_tab(arg, mt) : exp_mid(arg)) * kRhoCrit;
    const double rc = (m + r + de) + c.Omega_k * izp1 * kRhoCrit;
    const double e2 = rc * (zp1 * zp1 * zp1) * (1. / kRhoCrit);
    return e2 >= 2.2250738585072014e-308 && e2 <= 1.7976931348623157e308 ? rsqrt_pos(e2) : rsqrt(e2);
}

// The same for the quadrature of bg_kernel (two ordinates per interval, ~110 intervals per sample: its whole cost): the cosmology's constants gathered once
// per sample -- densities in units of the critical density (the factor kRhoCrit of every term and the 1 / kRhoCrit at the end cancel), photons and massless
// neutrinos as one term, the two coefficients of the dark-energy exponent -- and the reciprocal square root with one select for its special values.  Half the
// instructions of the E^2 above; the sums associate differently (1e-16 of E^2).
struct GridCosmo {
    double Om, Or, Ode, Ok, ea, eb;      // Omega_cdm + Omega_b, Omega_g + Omega_ur, Omega_de, Omega_k; exponent = ea log(1 + z) + eb (1 / (1 + z) - 1)
    bool lambda;                         // w0 = -1, wa = 0: the dark energy is (1 + z)^-3 in these units, no exponential
};

__device__ __forceinline__ GridCosmo grid_cosmo(const Cosmo& c) {
    return GridCosmo{c.Omega_cdm + c.Omega_b, c.Omega_g + c.Omega_ur, c.Omega_de, c.Omega_k, 3. * (c.w0 + c.wa), 3. * c.wa, c.w0 == -1. && c.wa == 0.};
}

__device__ __forceinline__ double rsqrt_any(double x) {      // 1 / sqrt(x): x = 0 -> Inf, +Inf -> 0, negative / NaN -> NaN, as the library's rsqrt
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.);
    const double r = fma(y * e, fma(0.375, e, 0.5), y);
    return __builtin_isfinite(y) && y != 0. ? r : y;
}

// ... with what a whole wave knows about its samples (bg_kernel's distance sweeps: two ordinates per interval, the kernel's whole cost):
// wave_fld -- some lane of the wave has a dark-energy fluid: the exponential is evaluated (for all lanes, a select keeps the cube for the lanes with a
// cosmological constant: the same values as the per-lane branch, whose two sides a mixed wave ran one after the other behind exec-mask bookkeeping);
// no lane has one: it is not; wave_lambda -- some lane has a cosmological constant.  SAFE -- every density parameter of every lane is >= 0: E^2 is a sum of positive terms, its reciprocal root needs no
// select for zero / infinite / NaN estimates (the same arithmetic otherwise).  Both are wave-uniform: scalar branches.
// zp1 = 1 + z and zp1c = zp1 (zp1 zp1) come from the grid's tables as log(1 + z) and 1 / (1 + z) do (the same sums and products, formed once on the host:
// four vector instructions fewer per ordinate); z itself is only looked at with massive neutrinos.
template <bool SAFE>
__device__ __forceinline__ double inv_efunc_grid_wave(const GridCosmo& g, const Cosmo& c, double z, double lzp1, double izp1, double zp1, double zp1c,
                                                      const cpmath::MathTables* mt, bool wave_fld, bool wave_lambda) {
    double m = g.Om;
    if (c.nsp) m += ncdm_eval(c, z, 0) * (1. / kRhoCrit);
    double growth;
    if (wave_fld) {
        growth = cpmath::exp_tab_core(fma(g.ea, lzp1, g.eb * (izp1 - 1.)), mt);
        if (wave_lambda) {      // (a wave of fluids only forms no cube and selects nothing: the empty statement keeps the compiler from turning the
            asm volatile("");   // branch back into an unconditional cube and select)
            growth = g.lambda ? izp1 * izp1 * izp1 : growth;
        }
    } else {
        growth = izp1 * izp1 * izp1;
    }
    const double rc = fma(g.Ok, izp1, fma(g.Or, zp1, m) + g.Ode * growth);
    const double x = rc * zp1c;
    return SAFE ? rsqrt_pos(x) : rsqrt_any(x);
}

__device__ __forceinline__ double inv_efunc_grid(const GridCosmo& g, const Cosmo& c, double z, double lzp1, double izp1, const cpmath::MathTables* mt) {
    const double zp1 = 1. + z;
    double m = g.Om;
    if (c.nsp) m += ncdm_eval(c, z, 0) * (1. / kRhoCrit);
    const double de = g.Ode * (g.lambda ? izp1 * izp1 * izp1 : cpmath::exp_tab_core(fma(g.ea, lzp1, g.eb * (izp1 - 1.)), mt));      // (|exponent| < 3 |w0 + wa| log(1e4) + 3 |wa|)
    const double rc = fma(g.Ok, izp1, fma(g.Or, zp1, m) + de);
    return rsqrt_any(rc * (zp1 * zp1 * zp1));
}

__device__ __forceinline__ double efunc(const Cosmo& c, double z) {
    const double zp1 = 1. + z;
    return sqrt(rho_crit(c, zp1) * (zp1 * zp1 * zp1) / kRhoCrit);  // cosmology.py:1754
}

// rho_m(z) = cdm + b + ncdm - 3 p_ncdm (cosmology.py:1704-1707)
__device__ __forceinline__ double rho_m(const Cosmo& c, double zp1) {
    return c.Omega_cdm * kRhoCrit + c.Omega_b * kRhoCrit + ncdm_eval(c, zp1 - 1., 0) - 3. * ncdm_eval(c, zp1 - 1., 1);
}

// CPT92 growth(z) of the analytic engines, un-normalised (eisenstein_hu.py:134-136)
__device__ __forceinline__ double growth_cpt(const Cosmo& c, double z) {
    const double zp1 = 1. + z;
    const double rc = rho_crit(c, zp1);
    const double Om = rho_m(c, zp1) / rc;  // rho_m / rho_crit, cosmology.py:1701, 1796
    const double Ode = rho_de(c, zp1) / rc;
    return 1. / zp1 * 5 * Om / 2. / (pow(Om, 4. / 7.) - Ode + (1. + Om / 2.) * (1 + Ode / 70.));
}

// device copy of the CP_NCDM_NKNOTS massive-neutrino knots (cp_background.hip), nullptr on failure
const double* ncdm_knots_device(int device);

// the massive-neutrino tables of a call as the kernels take them (load_cosmo): nothing for ncdm == NULL or nspecies == 0.  Called with the
// call's device current (the knots are allocated on it the first time).
struct NcdmView {
    const double* tab = nullptr;
    const double* knots = nullptr;
    int nsp = 0;
};
inline int ncdm_view(const cp_ncdm* ncdm, int device, const char* who, NcdmView* v) {
    *v = NcdmView{};
    const int nsp = ncdm ? ncdm->nspecies : 0;
    if (nsp < 0 || (nsp > 0 && !ncdm->tab)) return cp::fail(CP_EINVAL, "%s: bad massive-neutrino tables", who);
    if (!nsp) return CP_OK;
    v->knots = ncdm_knots_device(device);
    if (!v->knots) return cp::fail(CP_ENOMEM, "%s: cannot allocate the massive-neutrino knots on device %d", who, device);
    v->tab = ncdm->tab;
    v->nsp = nsp;
    return CP_OK;
}

}  // namespace cpcosmo
