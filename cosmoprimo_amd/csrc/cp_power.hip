// cp_power.hip -- analytic matter power spectra for batches of cosmologies (gfx950) + C ABI.
//
// Replaces, for the engines 'eisenstein_hu', 'eisenstein_hu_nowiggle' and 'bbks':
//   EisensteinHuEngine._set_rsdrag / compute          eisenstein_hu.py:34-92     (scalars per cosmology)
//   Transfer.transfer_k                               eisenstein_hu.py:241-283, eisenstein_hu_nowiggle.py:34-51, bbks.py:50-64
//   Primordial.pk_k                                   eisenstein_hu.py:189-215
//   Fourier.pk_interpolator: pk_callable x growth^2   eisenstein_hu.py:315-324, growth eisenstein_hu.py:115-140
// One thread per (cosmology, k); the ~15 fit coefficients are recomputed per thread (a few pow / log, cheaper than a
// second kernel and a round trip through HBM).  Output layout (ncosmo, nz, nk), k fastest: rows are ready for the FFTLog
// kernel.  Elementwise and ALU-bound (pow, log, exp); HBM traffic is the 8 nk nz output bytes per cosmology.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/cosmoprimo_amd.h"
#include "cp_cosmo_common.h"
#include "cp_math.h"
#include "cp_error.h"
#include "cp_power_eval.h"
#include "cp_internal.h"

namespace {

using namespace cpcosmo;
using namespace cpmath;
using namespace cppower;

struct Args {
    long long ncosmo;
    Param bg[CP_BG_NPARAMS];
    Param pw[CP_PK_NPARAMS];
    int second_is_omega_m;
    int engine, what;
    long long nk, nz;
    const double* k;  // (nk) shared, h/Mpc
    const double* kscale;  // optional (ncosmo): cosmology ic is evaluated at k * kscale[ic] (brieden2022: k_fid / rescale, k_fid * rescale)
    const double* z;  // (nz) shared (what == CP_PK_MATTER with nz > 0), else unused
    double* out;      // (ncosmo, max(nz, 1), nk)
    long long kchunks, kspan;  // a workgroup evaluates kspan consecutive k of ONE cosmology; kchunks = ceil(nk / kspan) workgroups per cosmology
    const CosmoConsts* consts; // (ncosmo) the cosmologies' constants from coefficients_kernel
    // massive neutrinos (cp_ncdm; nsp == 0: none).  The fits themselves know nothing of them (scalars from omega_cdm + omega_b, eisenstein_hu.py:37-38);
    // they enter through the background: Omega0_m of pk_callable (:322), Omega_m(z) / Omega_de(z) of the CPT92 growth (:134-135), Omega_m of BBKS (bbks.py:38)
    const double* ncdm_tab;
    const double* ncdm_knots;
    int nsp;
};

// The ~25 pow() of the EH98 / no-wiggle fit coefficients depend on the cosmology alone: one lane per cosmology here, read back by
// power_kernel through scalar loads.  (Evaluated in every (cosmology, k) lane, as the first version did, they were 3/4 of the time; evaluated
// by one lane of each power_kernel workgroup they still were half of it for 1024 wavenumbers per cosmology, all on one SIMD of the CU.)
__global__ __launch_bounds__(64) void coefficients_kernel(const Args A, CosmoConsts* out, int with_pw, const double* k_grid, double* ln_k_grid, int n_grid) {
    const long long cosmo_blocks = (A.ncosmo + 63) / 64;
    if ((long long)blockIdx.x >= cosmo_blocks) {      // the workgroups behind the cosmologies': the table of the shared wavenumbers (cp_power_coefficients)
        const int i = (int)(((long long)blockIdx.x - cosmo_blocks) * 64 + threadIdx.x);
        if (i < n_grid) powers_of_wavenumber(k_grid[i], ln_k_grid, i, n_grid);
        return;
    }
    const long long ic = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ic >= A.ncosmo) return;
    double pw[CP_PK_NPARAMS];
#pragma unroll
    for (int i = 0; i < CP_PK_NPARAMS; ++i) {      // (one address per parameter -- the array's entry or the scalar among the arguments -- and the loads back to back, in
        const double* q = A.pw[i].ptr ? A.pw[i].ptr + ic : &A.pw[i].value;      // front of the background parameters': behind a test per parameter they were a memory
        const double raw = *q;                                                   // round trip each, a fifth of this kernel's 21 us)
        pw[i] = with_pw ? raw : 0.;
    }
    const Cosmo c = load_cosmo(A.bg, ic, A.second_is_omega_m, A.ncdm_tab, A.ncdm_knots, A.nsp);
    out[ic] = cosmo_consts(c, with_pw ? pw : nullptr, A.engine);
}

// One workgroup = one cosmology x kspan wavenumbers: growth(z)^2 of every output redshift is evaluated once per workgroup (one lane per
// redshift) and shared through LDS; the lanes then walk the wavenumbers with the per-k part only (9 transcendentals for EH98).
template <int ENGINE>   // one instance per engine: the registers of a launch are those of its own transfer function, not the maximum over the three
__global__ __launch_bounds__(256, 3) void power_kernel(const Args A) {   // 3 waves per SIMD (168 registers): 0.204 ms per 10 000 x 1024 EH98 spectra against 0.23 at 2 waves and 0.29 at 4 (spills)
    __shared__ double sh_g2[256];
    __shared__ MathTables mt;
    fill_math_tables(&mt);
    __syncthreads();      // (the per-cosmology part below takes its logarithms from the tables)
    const int tid = threadIdx.x, bs = blockDim.x;      // 256 threads, or ONE wave per cosmology for large batches (cp_power_eval: the per-cosmology part of a thread -- parameters, constants of the fit, ~150 instructions -- is then paid by 64 lanes instead of 256)
    const long long ic = blockIdx.x / A.kchunks;
    const long long k0 = (long long)(blockIdx.x % A.kchunks) * A.kspan;
    const long long k1 = k0 + A.kspan < A.nk ? k0 + A.kspan : A.nk;
    const CosmoConsts K = load_uniform(A.consts + ic);      // (ic follows from the workgroup's index: scalar loads, the constants in scalar registers)
    const PkPerCosmology& pc = K.pk;
    const bool with_z = A.what == CP_PK_MATTER && A.nz > 0;
    const long long nzs = with_z ? A.nz : 1;
    const double kfac = A.kscale ? A.kscale[ic] : 1.;
    const double ln_pk_unit = K.ln_pk_unit;
    for (long long z0 = 0; z0 < nzs; z0 += bs) {
        if (z0) __syncthreads();  // the previous block of redshifts has been written
        if (with_z && z0 + tid < A.nz) {
            const Cosmo c = load_cosmo(A.bg, ic, A.second_is_omega_m, A.ncdm_tab, A.ncdm_knots, A.nsp);      // (the lanes that evaluate a growth factor only)
            const double g = growth_cpt(c, A.z[z0 + tid]);  // growth_factor(z, znorm=0), eisenstein_hu.py:317
            sh_g2[tid] = g * g;
        }
        __syncthreads();
        const int nzi = with_z ? (int)(A.nz - z0 < bs ? A.nz - z0 : bs) : 1;
        for (long long ik = k0 + tid; ik < k1; ik += bs) {
            const double kh = A.kscale ? A.k[ik] * kfac : A.k[ik];
            double* out = A.out + (ic * nzs + z0) * A.nk + ik;
            const double ln_kh = CP_MATH_TABLES_OFF ? log_pos(kh) : log_tab_any(kh, &mt);      // (2e-16 max(1, |log kh|): the tilt and the powers of k take it times O(1) factors)
            double T = 1.;
            if (A.what != CP_PK_PRIMORDIAL) {
                if (ENGINE == CP_ENGINE_BBKS)
                    T = transfer_bbks(K.h, K.bbks_gamma, kh);
                else
                    T = ENGINE == CP_ENGINE_EH ? transfer_eh(K.eh, kh, ln_kh, tables_present(&mt)) : transfer_nowiggle(K.s, K.h, kh, tables_present(&mt));
            }
            if (A.what == CP_PK_TRANSFER) {
                out[0] = T;
                continue;
            }
            if (A.what == CP_PK_LOG_K_MATTER) {
                // log(k P(k)) term by term: log k + log(T^2) + log(k x constant) + the exponent of the tilt -- the logarithm of the transfer
                // function replaces the exponential of the tilt, and the consumer (the DST of wallish2018, bao_filter.py:371) takes no
                // logarithm of 4096 samples per vector
                out[0] = 2. * (ln_kh + (CP_MATH_TABLES_OFF ? log_pos(fabs(T)) : log_tab_any(fabs(T), &mt))) + ln_pk_unit + primordial_tilt_exponent(pc, ln_kh);
                continue;
            }
            const double tilt = primordial_tilt(pc, ln_kh, tables_present(&mt));
            if (A.what == CP_PK_PRIMORDIAL) {
                out[0] = pc.h3_A_s * tilt;
                continue;
            }
            const double p0 = (T * T) * (kh * pc.pk_unit) * tilt;
            if (!with_z) {
                out[0] = p0;
                continue;
            }
            for (int iz = 0; iz < nzi; ++iz) out[iz * A.nk] = p0 * sh_g2[iz];
        }
    }
}

struct ScalArgs {
    long long ncosmo;
    Param bg[CP_BG_NPARAMS];
    int second_is_omega_m;
    const double* ncdm_tab;
    int nsp;
    double* out;  // (ncosmo, CP_EH_NSCALARS)
};

__global__ __launch_bounds__(256) void eh_scalars_kernel(const ScalArgs A) {
    const long long ic = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ic >= A.ncosmo) return;
    const Cosmo c = load_cosmo(A.bg, ic, A.second_is_omega_m, A.ncdm_tab, nullptr, A.nsp);      // (today's densities only: no knots)
    const EhScalars s = eh_scalars(c.h, c.Omega_cdm, c.Omega_b, c.T_cmb, true);
    double* o = A.out + ic * CP_EH_NSCALARS;
    o[CP_EH_RS_DRAG] = s.rs_drag;
    o[CP_EH_Z_DRAG] = s.z_drag;
    o[CP_EH_Z_EQ] = s.z_eq;
    o[CP_EH_K_EQ] = s.k_eq;
    o[CP_EH_R_DRAG] = s.r_drag;
    o[CP_EH_R_EQ] = s.r_eq;
    o[CP_EH_K_SILK] = s.k_silk;
    o[CP_EH_ALPHA_C] = s.alpha_c;
    o[CP_EH_BETA_C] = s.beta_c;
    o[CP_EH_ALPHA_B] = s.alpha_b;
    o[CP_EH_BETA_NODE] = s.beta_node;
    o[CP_EH_BETA_B] = s.beta_b;
    o[CP_EH_ALPHA_GAMMA] = s.alpha_gamma;
    o[CP_EH_BBKS_GAMMA] = bbks_gamma(c);
}

// ---- eisenstein_hu_nowiggle_variants: Eisenstein & Hu 1997 with massive neutrinos (reference eisenstein_hu_nowiggle_variants.py) ----
struct VarArgs {
    long long ncosmo;
    Param bg[CP_BG_NPARAMS];
    Param pw[CP_PK_NPARAMS];
    int second_is_omega_m;
    int what, of;
    long long nk, nz;
    const double* k;
    const double* z;
    double* out;  // (ncosmo, nz, nk)
    const double* ncdm_tab;
    const double* ncdm_knots;
    int nsp;
    long long kchunks, kspan;  // as in power_kernel's Args
};

struct VarScalars {
    double Omega_ncdm, Omega_pncdm, omega_m, frac_cb, frac_ncdm, theta_cmb, z_eq, rs_drag, p_cb, gamma_ncdm, beta_c;
    double omega_b, frac_b, frac_cdm, k_eq, z_drag, p_c;      // not read by variants_kernel: the engine's attributes (cp_variants_scalars)
};

// engine scalars, _set_rsdrag / compute (eisenstein_hu_nowiggle_variants.py:32-76)
__device__ __forceinline__ VarScalars variants_scalars(const Cosmo& c) {
    VarScalars v;
    const double h2 = c.h * c.h;
    v.Omega_ncdm = 0.;
    v.Omega_pncdm = 0.;
    for (int s = 0; s < c.nsp; ++s) {
        v.Omega_ncdm += c.ncdm_tab[((long long)s * 4 + 0) * CP_NCDM_NKNOTS] / kRhoCrit;
        v.Omega_pncdm += 3. * c.ncdm_tab[((long long)s * 4 + 2) * CP_NCDM_NKNOTS] / kRhoCrit;
    }
    const double omega_b = c.Omega_b * h2;
    v.omega_m = c.Omega_cdm * h2 + c.Omega_b * h2 + v.Omega_ncdm * h2 - v.Omega_pncdm * h2;
    const double omega_m = v.omega_m;
    const double frac_b = omega_b / omega_m, frac_cdm = c.Omega_cdm * h2 / omega_m;
    v.frac_cb = frac_cdm + frac_b;
    v.frac_ncdm = 1. - v.frac_cb;
    const double frac_cb = v.frac_cb, frac_ncdm = v.frac_ncdm;
    const double N = (double)c.nsp;
    v.theta_cmb = c.T_cmb / 2.7;
    v.z_eq = 2.5e4 * omega_m * pow(v.theta_cmb, -4.) - 1.;
    const double b1 = 0.313 * pow(omega_m, -0.419) * (1 + 0.607 * pow(omega_m, 0.674));
    const double b2 = 0.238 * pow(omega_m, 0.223);
    const double z_drag = 1291 * pow(omega_m, 0.251) / (1. + 0.659 * pow(omega_m, 0.828)) * (1. + b1 * pow(omega_b, b2));
    v.omega_b = omega_b;
    v.frac_b = frac_b;
    v.frac_cdm = frac_cdm;
    v.k_eq = 0.0746 * omega_m * pow(v.theta_cmb, -2.);
    v.z_drag = z_drag;
    v.rs_drag = 44.5 * log(9.83 / omega_m) / sqrt(1. + 10. * pow(omega_b, 0.75));
    const double fbn = frac_b + frac_ncdm;
    const double p_c = (5. - sqrt(1 + 24 * frac_cdm)) / 4.;
    v.p_c = p_c;
    v.p_cb = (5. - sqrt(1 + 24. * frac_cb)) / 4.;
    const double p_cb = v.p_cb;
    const double y_drag = (1 + v.z_eq) / (1 + z_drag);
    const double alpha = frac_cdm / frac_cb * (5. - 2. * (p_c + p_cb)) / (5. - 4. * p_cb) * pow(1 + y_drag, p_cb - p_c) *
                         (1 + fbn * (-0.553 + 0.126 * (fbn * fbn))) / (1 - 0.193 * sqrt(frac_ncdm * N) + 0.169 * frac_ncdm * pow(N, 0.2)) *
                         (1 + (p_c - p_cb) / 2 * (1 + 1 / (3. - 4. * p_c) / (7. - 4. * p_cb)) / (1 + y_drag));
    v.gamma_ncdm = sqrt(alpha);
    v.beta_c = 1 / (1 - 0.949 * fbn);
    return v;
}

// the engine's attributes for a batch of cosmologies, one lane each: (ncosmo, CP_VAR_NSCALARS) in the order of enum cp_variants_scalar
struct VarScalArgs {
    long long ncosmo;
    Param bg[CP_BG_NPARAMS];
    int second_is_omega_m;
    const double* ncdm_tab;
    const double* ncdm_knots;
    int nsp;
    double* out;
};

__global__ __launch_bounds__(256) void variants_scalars_kernel(const VarScalArgs A) {
    const long long ic = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ic >= A.ncosmo) return;
    const Cosmo c = load_cosmo(A.bg, ic, A.second_is_omega_m, A.ncdm_tab, A.ncdm_knots, A.nsp);
    const VarScalars v = variants_scalars(c);
    double* o = A.out + ic * CP_VAR_NSCALARS;
    o[CP_VAR_OMEGA_B] = v.omega_b;
    o[CP_VAR_OMEGA_M] = v.omega_m;
    o[CP_VAR_FRAC_B] = v.frac_b;
    o[CP_VAR_FRAC_CDM] = v.frac_cdm;
    o[CP_VAR_FRAC_CB] = v.frac_cb;
    o[CP_VAR_FRAC_NCDM] = v.frac_ncdm;
    o[CP_VAR_THETA_CMB] = v.theta_cmb;
    o[CP_VAR_Z_EQ] = v.z_eq;
    o[CP_VAR_K_EQ] = v.k_eq;
    o[CP_VAR_Z_DRAG] = v.z_drag;
    o[CP_VAR_RS_DRAG] = v.rs_drag;
    o[CP_VAR_P_C] = v.p_c;
    o[CP_VAR_P_CB] = v.p_cb;
    o[CP_VAR_GAMMA_NCDM] = v.gamma_ncdm;
    o[CP_VAR_BETA_C] = v.beta_c;
}

// Same shape as power_kernel: one workgroup = one cosmology x kspan wavenumbers; the engine scalars (one lane) and the CPT growth of every
// redshift (one lane each) are evaluated once per workgroup and shared through LDS.
__global__ __launch_bounds__(256) void variants_kernel(const VarArgs A) {
    __shared__ double knots[CP_NCDM_NKNOTS];
    __shared__ VarScalars sh_v;
    __shared__ double sh_g[256];
    const int tid = threadIdx.x;
    if (A.nsp)
        for (int i = tid; i < CP_NCDM_NKNOTS; i += blockDim.x) knots[i] = A.ncdm_knots[i];
    __syncthreads();
    const long long ic = blockIdx.x / A.kchunks;
    const long long k0 = (long long)(blockIdx.x % A.kchunks) * A.kspan;
    const long long k1 = k0 + A.kspan < A.nk ? k0 + A.kspan : A.nk;
    const Cosmo c = load_cosmo(A.bg, ic, A.second_is_omega_m, A.ncdm_tab, knots, A.nsp);
    double pw[CP_PK_NPARAMS];
#pragma unroll
    for (int i = 0; i < CP_PK_NPARAMS; ++i) pw[i] = A.pw[i].ptr ? A.pw[i].ptr[ic] : A.pw[i].value;
    const double N = (double)c.nsp;
    for (long long z0 = 0; z0 < A.nz; z0 += 256) {
        if (z0) __syncthreads();
        if (z0 + tid < A.nz) sh_g[tid] = growth_cpt(c, A.z[z0 + tid]);  // Background.growth_factor(z, znorm): (1 + znorm) x this, eisenstein_hu.py:134-139
        if (z0 == 0 && tid == 255) sh_v = variants_scalars(c);
        __syncthreads();
        const VarScalars& v = sh_v;
        const int nzi = (int)(A.nz - z0 < 256 ? A.nz - z0 : 256);
        for (long long ik = k0 + tid; ik < k1; ik += 256) {
            // transfer_kz (:114-154): the z-independent part
            const double khm = A.k[ik];          // h/Mpc
            const double k = khm * c.h;          // 1/Mpc
            const double q = k / v.omega_m * (v.theta_cmb * v.theta_cmb);
            const double kr = k * v.rs_drag * 0.43;
            const double gamma_eff = v.omega_m * (v.gamma_ncdm + (1 - v.gamma_ncdm) / (1 + (kr * kr) * (kr * kr)));
            const double q_eff = q * v.omega_m / gamma_eff;
            const double TL = log(kE + 1.84 * v.beta_c * v.gamma_ncdm * q_eff);
            const double TC = 14.4 + 325. / (1 + 60.5 * pow(q_eff, 1.08));
            double T_sup = TL / (TL + TC * (q_eff * q_eff));
            double yfs = 0.;
            if (c.nsp) {
                const double qn = 3.92 * q * sqrt(N / v.frac_ncdm);
                T_sup *= 1 + 1.24 * pow(v.frac_ncdm, 0.64) * pow(N, 0.3 + 0.6 * v.frac_ncdm) / (pow(qn, -1.6) + pow(qn, 0.8));
                const double nq = N * q / v.frac_ncdm;
                yfs = 17.2 * v.frac_ncdm * (1 + 0.488 * pow(v.frac_ncdm, -7. / 6.)) * (nq * nq);
            }
            // primordial spectrum and the potential -> density factors (Fourier.pk_interpolator :178-185, eisenstein_hu.py:214-215)
            double pdd = 0.;
            if (A.what == CP_PK_MATTER) {
                const double kp = pw[CP_PK_K_PIVOT] / c.h;
                const double lnkkp = log(khm / kp);
                const double prim = (c.h * c.h * c.h) * pw[CP_PK_A_S] *
                                    exp((pw[CP_PK_N_S] - 1. + 1. / 2. * pw[CP_PK_ALPHA_S] * lnkkp + 1. / 6. * pw[CP_PK_BETA_S] * (lnkkp * lnkkp)) * lnkkp);
                const double Omega0_m = c.Omega_b + c.Omega_cdm + v.Omega_ncdm - v.Omega_pncdm;  // cosmology.py:381
                const double p2d = 3. * Omega0_m * (100. * 100.) / (2. * (kCkms * kCkms) * (khm * khm));
                pdd = 1. / (p2d * p2d) * (9. / 25. * 2. * (kPi * kPi) / (khm * khm * khm) / (c.h * c.h * c.h)) * prim;
            }
            double* out = A.out + (ic * A.nz + z0) * A.nk + ik;
            for (int iz = 0; iz < nzi; ++iz) {
                const double g = sh_g[iz];
                double ratio = 1.;                 // growth / growth_k0 (:119-134)
                if (c.nsp) {
                    const double gk0 = (1. + v.z_eq) * g;
                    const double t1 = pow(gk0, 1. - v.p_cb);
                    const double t2 = pow(gk0 / (1 + yfs), 0.7);
                    const double growth = A.of == 1 ? pow(1. + t2, v.p_cb / 0.7) * t1 : pow(pow(v.frac_cb, 0.7 / v.p_cb) + t2, v.p_cb / 0.7) * t1;
                    ratio = growth / gk0;
                }
                const double T = T_sup * ratio;
                out[iz * A.nk] = A.what == CP_PK_TRANSFER ? T : (T * T) * (g * g) * pdd;
            }
        }
    }
}

int select_device(int device, int* prev) {
    *prev = -1;
    if (hipGetDevice(prev) != hipSuccess) *prev = -1;
    if (*prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cannot select device %d", device);
    return CP_OK;
}

}  // namespace

extern "C" long long cp_power_workspace_bytes(long long ncosmo) { return ncosmo < 0 ? -1 : (long long)sizeof(CosmoConsts) * ncosmo; }

int cp_power_coefficients(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
                          void* d_work, int device, void* stream, const double* d_k, double* d_ln_k, int n) {
    if (ncosmo <= 0) return CP_OK;
    if (!bg_params || !d_work) return cp::fail(CP_EINVAL, "cp_power_coefficients: null pointer");
    int prev;
    int st = select_device(device, &prev);
    if (st != CP_OK) return st;
    Args A{};
    A.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) A.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    if (pk_params)
        for (int i = 0; i < CP_PK_NPARAMS; ++i) A.pw[i] = Param{pk_params[i].ptr, pk_params[i].value};
    A.second_is_omega_m = second_is_omega_m;
    A.engine = engine;
    NcdmView nu;
    st = ncdm_view(ncdm, device, "cp_power_coefficients", &nu);
    if (st != CP_OK) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return st;
    }
    A.ncdm_tab = nu.tab;
    A.ncdm_knots = nu.knots;
    A.nsp = nu.nsp;
    const int n_grid = d_k && d_ln_k ? n : 0;
    hipLaunchKernelGGL(coefficients_kernel, dim3((unsigned)((ncosmo + 63) / 64 + (n_grid + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream), A,
                       static_cast<CosmoConsts*>(d_work), pk_params ? 1 : 0, d_k, d_ln_k, n_grid);
    const hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_power_coefficients: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_power_eval(int engine, int what, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm,
                             const cp_param* pk_params, long long nk, const double* d_k, const double* d_kscale, long long nz, const double* d_z, double* d_out, void* d_work,
                             int device, void* stream) {
    if (engine < CP_ENGINE_EH || engine > CP_ENGINE_BBKS) return cp::fail(CP_EINVAL, "cp_power_eval: unknown engine %d", engine);
    if (what < CP_PK_MATTER || what > CP_PK_LOG_K_MATTER) return cp::fail(CP_EINVAL, "cp_power_eval: unknown quantity %d", what);
    if (what == CP_PK_LOG_K_MATTER && nz > 0) return cp::fail(CP_EINVAL, "cp_power_eval: log(k P) is without growth factor (nz = 0)");
    if (ncosmo < 0 || nk < 0 || nz < 0) return cp::fail(CP_EINVAL, "cp_power_eval: negative size");
    if (ncosmo == 0 || nk == 0) return CP_OK;
    if (!bg_params || !pk_params || !d_k || !d_out || (nz > 0 && !d_z)) return cp::fail(CP_EINVAL, "cp_power_eval: null pointer");
    int prev;
    int st = select_device(device, &prev);
    if (st != CP_OK) return st;
    Args A;
    A.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) A.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    for (int i = 0; i < CP_PK_NPARAMS; ++i) A.pw[i] = Param{pk_params[i].ptr, pk_params[i].value};
    A.second_is_omega_m = second_is_omega_m;
    A.engine = engine;
    A.what = what;
    A.nk = nk;
    A.nz = what == CP_PK_MATTER ? nz : 0;
    A.k = d_k;
    A.kscale = d_kscale;
    A.z = d_z;
    A.out = d_out;
    A.consts = nullptr;
    NcdmView nu;
    st = ncdm_view(ncdm, device, "cp_power_eval", &nu);
    if (st != CP_OK) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return st;
    }
    A.ncdm_tab = nu.tab;
    A.ncdm_knots = nu.knots;
    A.nsp = nu.nsp;
    // wavenumbers per workgroup: all of a cosmology's when the batch alone fills the chip (>= 8 workgroups per CU), fewer for small batches.  A batch
    // that fills the chip with one WAVE per cosmology (12 waves per CU) takes that shape: a thread's per-cosmology part is amortised over nk / 64
    // wavenumbers instead of nk / 256 (brieden2022 evaluates 341 per cosmology: 2 282 -> ~900 instructions per spectrum for the no-wiggle form)
    const long long block = ncosmo >= 256 * 12 ? 64 : 256;
    long long kiter = (nk + block - 1) / block;
    while (kiter > 1 && ncosmo * ((nk + block * kiter - 1) / (block * kiter)) < 2048) kiter = (kiter + 1) / 2;
    A.kspan = block * kiter;
    A.kchunks = (nk + A.kspan - 1) / A.kspan;
    if (ncosmo * A.kchunks > 2147483647LL) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return cp::fail(CP_EUNSUPPORTED, "cp_power_eval: %lld cosmologies x %lld wavenumbers exceed one launch; split the batch", ncosmo, nk);
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    {   // the constants of the cosmologies: caller-owned workspace, nothing is allocated here
        if (!d_work) {
            if (prev >= 0) (void)hipSetDevice(prev);
            return cp::fail(CP_EINVAL, "cp_power_eval: needs a workspace of cp_power_workspace_bytes(ncosmo) bytes");
        }
        CosmoConsts* consts = static_cast<CosmoConsts*>(d_work);
        A.consts = consts;
        hipLaunchKernelGGL(coefficients_kernel, dim3((unsigned)((ncosmo + 63) / 64)), dim3(64), 0, hs, A, consts, 1, static_cast<const double*>(nullptr),
                           static_cast<double*>(nullptr), 0);
    }
    const dim3 grid((unsigned)(ncosmo * A.kchunks)), threads((unsigned)block);
    if (engine == CP_ENGINE_EH) hipLaunchKernelGGL(power_kernel<CP_ENGINE_EH>, grid, threads, 0, hs, A);
    else if (engine == CP_ENGINE_EH_NOWIGGLE) hipLaunchKernelGGL(power_kernel<CP_ENGINE_EH_NOWIGGLE>, grid, threads, 0, hs, A);
    else hipLaunchKernelGGL(power_kernel<CP_ENGINE_BBKS>, grid, threads, 0, hs, A);
    hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_power_eval: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_power_eval_variants(int what, int of, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm,
                                      const cp_param* pk_params, long long nk, const double* d_k, long long nz, const double* d_z, double* d_out,
                                      int device, void* stream) {
    if (what != CP_PK_MATTER && what != CP_PK_TRANSFER) return cp::fail(CP_EINVAL, "cp_power_eval_variants: unknown quantity %d", what);
    if (of != 0 && of != 1) return cp::fail(CP_EINVAL, "cp_power_eval_variants: of must be 0 (delta_m) or 1 (delta_cb)");
    if (ncosmo < 0 || nk < 0 || nz < 0) return cp::fail(CP_EINVAL, "cp_power_eval_variants: negative size");
    if (ncosmo == 0 || nk == 0 || nz == 0) return CP_OK;
    if (!bg_params || !pk_params || !d_k || !d_z || !d_out) return cp::fail(CP_EINVAL, "cp_power_eval_variants: null pointer");
    const int nsp = ncdm ? ncdm->nspecies : 0;
    if (nsp < 0 || (nsp > 0 && !ncdm->tab)) return cp::fail(CP_EINVAL, "cp_power_eval_variants: bad massive-neutrino tables");
    int prev;
    int st = select_device(device, &prev);
    if (st != CP_OK) return st;
    VarArgs A;
    A.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) A.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    for (int i = 0; i < CP_PK_NPARAMS; ++i) A.pw[i] = Param{pk_params[i].ptr, pk_params[i].value};
    A.second_is_omega_m = second_is_omega_m;
    A.what = what;
    A.of = of;
    A.nk = nk;
    A.nz = nz;
    A.k = d_k;
    A.z = d_z;
    A.out = d_out;
    A.nsp = nsp;
    A.ncdm_tab = nsp ? ncdm->tab : nullptr;
    A.ncdm_knots = nsp ? cpcosmo::ncdm_knots_device(device) : nullptr;
    if (nsp && !A.ncdm_knots) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return cp::fail(CP_ENOMEM, "cp_power_eval_variants: cannot allocate the massive-neutrino knots on device %d", device);
    }
    const long long block = 256;
    long long kiter = (nk + block - 1) / block;
    while (kiter > 1 && ncosmo * ((nk + block * kiter - 1) / (block * kiter)) < 2048) kiter = (kiter + 1) / 2;
    A.kspan = block * kiter;
    A.kchunks = (nk + A.kspan - 1) / A.kspan;
    if (ncosmo * A.kchunks > 2147483647LL) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return cp::fail(CP_EUNSUPPORTED, "cp_power_eval_variants: %lld cosmologies x %lld wavenumbers exceed one launch; split the batch", ncosmo, nk);
    }
    hipLaunchKernelGGL(variants_kernel, dim3((unsigned)(ncosmo * A.kchunks)), dim3((unsigned)block), 0, static_cast<hipStream_t>(stream), A);
    hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_power_eval_variants: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_variants_scalars(long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, double* d_out, int device,
                                   void* stream) {
    if (ncosmo < 0) return cp::fail(CP_EINVAL, "cp_variants_scalars: negative size");
    if (ncosmo == 0) return CP_OK;
    if (!bg_params || !d_out) return cp::fail(CP_EINVAL, "cp_variants_scalars: null pointer");
    const int nsp = ncdm ? ncdm->nspecies : 0;
    if (nsp < 0 || (nsp > 0 && !ncdm->tab)) return cp::fail(CP_EINVAL, "cp_variants_scalars: bad massive-neutrino tables");
    int prev;
    int st = select_device(device, &prev);
    if (st != CP_OK) return st;
    VarScalArgs A;
    A.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) A.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    A.second_is_omega_m = second_is_omega_m;
    A.nsp = nsp;
    A.ncdm_tab = nsp ? ncdm->tab : nullptr;
    A.ncdm_knots = nsp ? cpcosmo::ncdm_knots_device(device) : nullptr;
    A.out = d_out;
    if (nsp && !A.ncdm_knots) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return cp::fail(CP_ENOMEM, "cp_variants_scalars: cannot allocate the massive-neutrino knots on device %d", device);
    }
    hipLaunchKernelGGL(variants_scalars_kernel, dim3((unsigned)((ncosmo + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), A);
    hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_variants_scalars: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_eh_scalars(long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, double* d_out, int device,
                             void* stream) {
    if (ncosmo < 0) return cp::fail(CP_EINVAL, "cp_eh_scalars: negative size");
    if (ncosmo == 0) return CP_OK;
    if (!bg_params || !d_out) return cp::fail(CP_EINVAL, "cp_eh_scalars: null pointer");
    const int nsp = ncdm ? ncdm->nspecies : 0;
    if (nsp < 0 || (nsp > 0 && !ncdm->tab)) return cp::fail(CP_EINVAL, "cp_eh_scalars: bad massive-neutrino tables");
    int prev;
    int st = select_device(device, &prev);
    if (st != CP_OK) return st;
    ScalArgs A;
    A.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) A.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    A.second_is_omega_m = second_is_omega_m;
    A.ncdm_tab = nsp ? ncdm->tab : nullptr;
    A.nsp = nsp;
    A.out = d_out;
    hipLaunchKernelGGL(eh_scalars_kernel, dim3((unsigned)((ncosmo + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), A);
    hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_eh_scalars: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
