/* cosmoprimo_amd.h -- C ABI of libcosmoprimo_amd.so, the MI355X (gfx950) hot-path library.
 *
 * The reference (cosmodesi/cosmoprimo) is pure Python; its "FFI" for this path is the set of
 * third-party native routines its Python calls (numpy.fft, scipy.special.loggamma, scipy
 * splines ...).  Each entry point below replaces one such call site wholesale; the reference line
 * it replaces is cited.  All functions are extern "C", take plain pointers and sizes (no torch
 * types), return an int status (0 = CP_OK) and record a message retrievable with cp_last_error().
 *
 * Device pointers are owned by the caller (torch allocates them); `stream` is a hipStream_t passed
 * as void*.  Plans own only their device copies of the tables; execute() allocates nothing and is
 * asynchronous on `stream`.
 */
#ifndef COSMOPRIMO_AMD_H
#define COSMOPRIMO_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define CP_ABI_VERSION 4

enum cp_status {
    CP_OK = 0,
    CP_EINVAL = 1,       /* bad argument (maps to ValueError) */
    CP_EUNSUPPORTED = 2, /* size outside what the LDS-resident kernel handles (NotImplementedError) */
    CP_EDEVICE = 3,      /* HIP runtime error (RuntimeError) */
    CP_ENOMEM = 4        /* device allocation failed (MemoryError) */
};

/* extrapolation modes of FFTlog.__call__(extrap=...) / pad(), reference cosmoprimo/fftlog.py:436-505 */
enum cp_extrap { CP_EXTRAP_CONSTANT = 0, CP_EXTRAP_EDGE = 1, CP_EXTRAP_LOGLOG = 2 };

/* Mellin-transform kernels U_K(z), reference cosmoprimo/fftlog.py:666-766 */
enum cp_kernel {
    CP_KERNEL_BESSEL_J = 0,           /* BesselJKernel(nu)           fftlog.py:688-695 */
    CP_KERNEL_SPHERICAL_BESSEL_J = 1, /* SphericalBesselJKernel(nu)  fftlog.py:698-705 */
    CP_KERNEL_TOPHAT = 2,             /* TophatKernel(ndim)          fftlog.py:719-726 */
    CP_KERNEL_TOPHAT_SQ = 3,          /* TophatSqKernel(ndim)        fftlog.py:729-746 */
    CP_KERNEL_GAUSSIAN = 4,           /* GaussianKernel              fftlog.py:749-756 */
    CP_KERNEL_GAUSSIAN_SQ = 5,        /* GaussianSqKernel            fftlog.py:759-766 */
    CP_KERNEL_CUSTOM = 100            /* values supplied by the caller (any Python callable kernel, fftlog.py:54-56) */
};

int cp_abi_version(void);
/* message of the last failing call on this thread ("" if none) */
const char* cp_last_error(void);
/* number of visible HIP devices, or -1 (never initialises a device context) */
int cp_device_count(void);

/* ---- host special functions (table setup; replaces scipy.special.loggamma / gamma at
 *      fftlog.py:10-11, 695, 705, 726, 740-746, 756, 766).  Arrays are interleaved (re, im). ---- */
int cp_loggamma(const double* z_re_im, double* out_re_im, long long n);
int cp_gamma(const double* z_re_im, double* out_re_im, long long n);
int cp_kernel_eval(int kind, double param, const double* z_re_im, double* out_re_im, long long n);

/* ---- FFTLog plan tables on the host (replaces FFTlog._setup, fftlog.py:144-184, and the convention factors of its
 *      subclasses, fftlog.py:280, 318-330, 368-377, 403-405, 431-433) ---- */
typedef struct cp_fftlog_spec {
    int kind;         /* cp_kernel; CP_KERNEL_CUSTOM: Mellin-transform values supplied in u_custom / lowring_custom */
    double param;     /* nu / ndim of the kernel */
    double q;         /* tilt, including the convention's offset (1.5 + q for P <-> xi and the variances) */
    double xy;        /* x y product, used when lowring == 0 (fftlog.py:165) */
    double pre_power; /* prefactor  = pre_const x^(pre_power) x^(-q)   e.g. 3, (2 pi)^-1.5 for P -> xi (fftlog.py:319) */
    double pre_const;
    double post_sign; /* postfactor = post_sign y^(-q); the unit phase of complex=True transforms stays with the caller */
} cp_fftlog_spec;
/* padded size 2**((n * minfolds - 1).bit_length()) (fftlog.py:149-150), -1 on bad arguments */
int cp_fftlog_padded_size(int n, int minfolds);
/* x : (nker, n) log-spaced coordinates.  Outputs (host): delta, lnxy : (nker); y : (nker, n); padded_x, padded_y, pre, post :
 * (nker, npad); u : (nker, npad/2 + 1) complex128 interleaved.  u_custom : (nker, npad/2 + 1) complex values of custom kernels at
 * z_m = q + 2 pi i m / (npad delta) (rows of other kernels ignored), lowring_custom : (nker) complex values at q + i pi / delta; both
 * may be NULL when no kernel is custom.  check_level != 0 verifies the log spacing (CP_EINVAL, fftlog.py:155-157). */
int cp_fftlog_tables(int n, int nker, const double* x, const cp_fftlog_spec* spec, int minfolds, int lowring, int check_level,
                     const double* u_custom, const double* lowring_custom, double* delta, double* lnxy, double* y, double* padded_x,
                     double* padded_y, double* pre, double* post, double* u);

/* ---- fused FFTLog (replaces FFTlog.__call__, fftlog.py:198-241: pad -> x prefactor -> rfft -> x u ->
 *      irfft(conj) -> x postfactor -> crop, i.e. numpy.fft.rfft/irfft at fftlog.py:540, 544) ---- */
typedef struct cp_fftlog_plan cp_fftlog_plan;

/* Tables are HOST arrays as the reference's FFTlog._setup produces them (fftlog.py:144-184):
 *   pre, post : (nker, npad) float64  padded_prefactor / padded_postfactor (real part)
 *   u_re_im   : (nker, npad/2 + 1) complex128 padded_u, interleaved
 * npad must be the power of two 2**((n*minfolds-1).bit_length()); pad splits follow fftlog.py:152-153.
 * npad <= 8192 (every size the reference's own callers and BASELINE.json's configs use) runs the fused kernel.  Larger sizes, up to 2^24, run the
 * same arithmetic split the four-step way (csrc/cp_fftlog_large.hip: column transforms of npad / 4096 points around a 4096-point row kernel that
 * holds both transforms and the product with u, through a plan-owned scratch of at most 128 MB): hand-written kernels as well, outside the
 * BASELINE configs, and no headline figure of this package refers to them. */
int cp_fftlog_plan_create(cp_fftlog_plan** plan, int n, int npad, int nker, const double* pre, const double* post,
                          const double* u_re_im, int device);
/* d_in : device (nbatch, nker, n) float64 C-contiguous;  d_out : device (nbatch, nker, n or npad).
 * extrap_*: cp_extrap, val_* used for CP_EXTRAP_CONSTANT.  keep_padding as in fftlog.py:233-237.
 * Rows are independent, as in numpy's row-by-row FFTs: batch items 2q and 2q + 1 of a kernel share one complex transform inside the
 * kernel, which itself transforms a row holding NaN / Inf as zeros and stores NaN for it, and rescales the two rows by exact powers
 * of two when their (tilted) magnitudes differ by more than a factor 32, so that rounding is relative to each row's own magnitude. */
int cp_fftlog_execute(const cp_fftlog_plan* plan, const double* d_in, double* d_out, long long nbatch, int extrap_left,
                      double val_left, int extrap_right, double val_right, int keep_padding, void* stream);
/* cp_fftlog_execute for a consumer that reads the columns [out_first, out_first + out_count) of every output row only (the step of sigma_r from the
 * FFTLog grid to the radii, interpolator.py:285-291, reads the third of the grid its radii can see: cp_spline_plan_columns): the default transform
 * (zero padding, n = npad / 2) then stores that window only -- d_out keeps its row length, entries outside the window are left as they were --;
 * every other case stores whole rows. */
int cp_fftlog_execute_window(const cp_fftlog_plan* plan, const double* d_in, double* d_out, long long nbatch, int extrap_left, double val_left,
                             int extrap_right, double val_right, int keep_padding, int out_first, int out_count, void* stream);
int cp_fftlog_plan_destroy(cp_fftlog_plan* plan);
/* introspection for the bench / tests: workgroups launched per execute for `nbatch`, threads per workgroup, LDS bytes */
int cp_fftlog_plan_info(const cp_fftlog_plan* plan, long long nbatch, int* grid, int* block, int* lds_bytes);

/* ---- background E(z) and distances for batches of cosmologies (replaces DefaultBackground.comoving_radial_distance,
 *      cosmology.py:2027-2042 -- jax.odeint rk4 on the 119-knot grid + natural cubic spline, jax.py:672-716, 169-175 -- the
 *      derived distances cosmology.py:1855-1912, efunc :1751-1754 and the derived density parameters :355-397).
 *      Massless neutrinos only (the reference default, m_ncdm empty). ---- */
enum cp_bg_param {   /* index into the cp_param array */
    CP_BG_H = 0, CP_BG_OMEGA_CDM = 1 /* or Omega_m, see second_is_omega_m */, CP_BG_OMEGA_B = 2, CP_BG_OMEGA_K = 3, CP_BG_T_CMB = 4,
    CP_BG_N_UR = 5, CP_BG_W0_FLD = 6, CP_BG_WA_FLD = 7, CP_BG_NPARAMS = 8
};
enum cp_bg_kind {
    CP_BG_COMOVING_RADIAL = 0,     /* comoving_radial_distance      cosmology.py:2027 */
    CP_BG_COMOVING_TRANSVERSE = 1, /* comoving_transverse_distance  cosmology.py:1893 */
    CP_BG_ANGULAR_DIAMETER = 2,    /* angular_diameter_distance     cosmology.py:1855 */
    CP_BG_LUMINOSITY = 3,          /* luminosity_distance           cosmology.py:1904 */
    CP_BG_EFUNC = 4,               /* efunc                         cosmology.py:1751 */
    CP_BG_HUBBLE = 5,              /* hubble_function               cosmology.py:1756 */
    CP_BG_GROWTH_CPT = 6,          /* un-normalised CPT92 growth(z) of the analytic engines, eisenstein_hu.py:134-136 */
    CP_BG_GROWTH_RATE = 7,         /* growth_rate of the analytic engines, eisenstein_hu.py:143-152 */
    CP_BG_RHO_CRIT = 8,            /* rho_crit(z), 1e10 Msun/h / (Mpc/h)^3   cosmology.py:1738-1749 */
    CP_BG_OMEGA_M_Z = 9,           /* Omega_m(z)                    cosmology.py:1796 */
    CP_BG_OMEGA_DE_Z = 10,         /* Omega_de(z)                   cosmology.py:1850 */
    /* comoving densities rho_x(z) in 1e10 Msun/h / (Mpc/h)^3 (BaseBackground.rho_*, cosmology.py:1680-1736); OR-ed with
       CP_BG_AS_FRACTION they are divided by rho_crit(z): the density parameters Omega_x(z) (cosmology.py:1774-1853) */
    CP_BG_RHO_G = 11,              /* photons                       cosmology.py:1680 */
    CP_BG_RHO_B = 12,              /* baryons                       cosmology.py:1685 */
    CP_BG_RHO_UR = 13,             /* massless neutrinos            cosmology.py:1690 */
    CP_BG_RHO_CDM = 14,            /* cold dark matter              cosmology.py:1699 */
    CP_BG_RHO_K = 15,              /* curvature                     cosmology.py:1709 */
    CP_BG_RHO_LAMBDA = 16,         /* Omega0_de / (1+z)^3: the cosmological-constant form (the caller knows whether w = -1)  :1714 */
    CP_BG_RHO_FLD = 17,            /* Omega0_de (1+z)^(3(1+w0+wa)) exp(3 wa (1/(1+z) - 1)) / (1+z)^3: the fluid form          :1719 */
    CP_BG_RHO_DE = 18,             /* total dark energy             cosmology.py:1724 */
    CP_BG_RHO_TOT = 19,            /* matter + radiation + dark energy  cosmology.py:1731 */
    CP_BG_RHO_M = 20,              /* cdm + baryons (no massive neutrinos on this path)  cosmology.py:1704 */
    CP_BG_RHO_R = 21,              /* photons + massless neutrinos  cosmology.py:1694 */
    CP_BG_T_CMB_Z = 22,            /* T0_cmb (1+z), K               cosmology.py:1762 */
    CP_BG_TIME = 23,               /* proper time (age of the universe at z), Gyr   DefaultBackground.time, cosmology.py:2000-2012 */
    CP_BG_AGE = 24,                /* age today, Gyr (z ignored)                     DefaultBackground.age,  cosmology.py:2014-2025 */
    CP_BG_RHO_NCDM = 25,           /* massive neutrinos: comoving density of species cp_ncdm.species (-1: all)  DefaultBackground.rho_ncdm, cosmology.py:1961-1978 */
    CP_BG_P_NCDM = 26,             /* ... and pressure                                                          DefaultBackground.p_ncdm,   cosmology.py:1980-1998 */
    CP_BG_RS = 27,                 /* comoving sound horizon at z, Mpc/h: Romberg (15 levels) of c_s dtau/da from a = 1e-8   BaseBackground.rs, cosmology.py:1914-1933 */
    CP_BG_RS_COSMOMC = 28,         /* the same with CosmoMC's R = 3e4 a omega_b, proper Mpc                       _compute_rs_cosmomc, cosmology.py:202-228 */
    CP_BG_KIND_LAST = 28,
    CP_BG_AS_FRACTION = 32
};
/* a per-cosmology parameter: device array of ncosmo doubles, or (ptr == NULL) one value for all cosmologies */
typedef struct cp_param {
    const double* ptr;
    double value;
} cp_param;
/* Sample (ic, iz) = cosmology ic evaluated at z[ic * nz + iz] (or z[iz] when z_shared); d_out has ncosmo * nz doubles
 * (Mpc/h for distances).  z outside [0, 9999] gives NaN as in the reference.  Asynchronous on `stream` of `device`. */
int cp_background_distance(long long ncosmo, long long nz, const cp_param* params, int second_is_omega_m, const double* d_z, int z_shared,
                           double* d_out, int kind, int device, void* stream);
/* Derived parameters of a batch of cosmologies without massive species (BaseCosmoParams._get_derived, cosmology.py:331-415: Omega_g from T_cmb,
 * Omega_ur from N_ur, Omega_r, Omega_m, Omega_de by closure, the omega_x = Omega_x h^2, H0, K), one lane per cosmology.
 * params: the CP_BG_NPARAMS background parameters; d_out: (CP_DERIVED_NVALUES, ncosmo), one contiguous row per value. */
enum cp_derived_value {
    CP_DERIVED_H2 = 0, CP_DERIVED_H0 = 1, CP_DERIVED_OMEGA_G = 2, CP_DERIVED_T_UR = 3, CP_DERIVED_OMEGA_UR = 4, CP_DERIVED_OMEGA_R = 5,
    CP_DERIVED_OMEGA_M = 6, CP_DERIVED_OMEGA_DE = 7, CP_DERIVED_K = 8, CP_DERIVED_LITTLE_OMEGA_B = 9, CP_DERIVED_LITTLE_OMEGA_CDM = 10,
    CP_DERIVED_LITTLE_OMEGA_M = 11, CP_DERIVED_LITTLE_OMEGA_G = 12, CP_DERIVED_LITTLE_OMEGA_UR = 13, CP_DERIVED_LITTLE_OMEGA_R = 14,
    CP_DERIVED_LITTLE_OMEGA_K = 15, CP_DERIVED_LITTLE_OMEGA_DE = 16, CP_DERIVED_NVALUES = 17
};
int cp_derived_parameters(long long ncosmo, const cp_param* params, double* d_out, int device, void* stream);
/* the 119 interpolation knots (host), get_default_z_interp('comoving_radial_distance'), cosmology.py:1947-1949 (n = 119),
 * or the 400 knots of time / age, cosmology.py:1945-1946 (n = 400) */
int cp_background_knots(double* zc_out, int n);
/* The knot tables the background kernels read (the 119- and 400-knot grids with their elimination factors, the massive-neutrino knots: 60 KB) onto
 * `device`: one hipMalloc and synchronous uploads, once per device and process (idempotent, thread-safe).  A caller that wants every later entry
 * point asynchronous and allocation-free calls this when it sets the device up (cosmoprimo_amd/background.py does, with the first Background of a
 * device); without it the first cp_background_* / cp_ncdm_* / cp_power_* call that needs a table does the same work inside that call. */
int cp_background_init(int device);
/* the derived distances of ONE cosmology from its radial distances: d_chi, d_z, d_out (n) device (d_out may be d_chi), K = -Omega_k (100 / c)^2
 * (cosmology.py:397), kind one of CP_BG_ANGULAR_DIAMETER / CP_BG_COMOVING_TRANSVERSE / CP_BG_LUMINOSITY -- the last lines of the background kernel
 * (cosmology.py:1855-1912) as one pass over a catalogue whose radial distances come from the cosmology's table */
int cp_distance_from_radial(const double* d_chi, const double* d_z, long long n, double K, int kind, double* d_out, int device, void* stream);

/* ---- massive neutrinos (reference cosmology.py:74-137 _compute_ncdm_momenta, :1961-1998 DefaultBackground.rho_ncdm / p_ncdm) ----
 * The reference tabulates, per species, the comoving density and pressure on 119 redshift knots (get_default_z_interp('rho_ncdm'),
 * cosmology.py:1941-1943) by 100-point Gauss-Laguerre quadrature of the frozen Fermi-Dirac distribution and interpolates them with
 * natural cubic splines wherever E(z) is needed.  cp_ncdm_tables builds those tables (values and spline second derivatives) on the
 * device for a batch of cosmologies; cp_background_eval is cp_background_distance with the tables added to every density. */
#define CP_NCDM_NKNOTS 119
typedef struct cp_ncdm {
    int nspecies;       /* number of massive species (0: none) */
    int species;        /* CP_BG_RHO_NCDM / CP_BG_P_NCDM only: which species, -1 = sum over species */
    const double* tab;  /* device, (ncosmo, nspecies, 4, CP_NCDM_NKNOTS): rho, rho'', p, p'' in 1e10 Msun/h / (Mpc/h)^3 */
} cp_ncdm;
int cp_ncdm_knots(double* zc_out, int n);

/* Linear growth from the ODE D'' = f2 D + f1 D' in eta = ln a (reference DefaultBackground.growth_factor / growth_rate,
 * cosmology.py:2044-2093: jax.odeint 'rk4' on eta = linspace(-6, 0, 201), initial conditions D = D' = e^-6): d_tab (ncosmo, 2, 201) receives
 * D and D'/D on the knots z = exp(-eta) - 1 in ASCENDING z (cp_growth_ode_knots), ready for the natural splines the reference builds on them.
 * mass: 0 'm' (Omega_m incl. massive neutrinos), 1 'cb' (Omega_cdm + Omega_b). */
#define CP_GROWTH_NKNOTS 201
int cp_growth_ode_knots(double* zc_out, int n);
int cp_growth_ode_tables(long long ncosmo, const cp_param* params, int second_is_omega_m, const cp_ncdm* ncdm, int mass, double* d_tab, int device,
                         void* stream);
/* m_ncdm[s] (eV) and T_ncdm_over_cmb[s], s < nspecies: per-cosmology parameters like h and T_cmb; nodes / weights: the nq-point
 * Gauss-Laguerre rule (host arrays, read before the call returns and passed to the kernel by value; the reference uses
 * numpy.polynomial.laguerre.laggauss(100)); d_tab as in cp_ncdm.tab.  Like every execute: allocates nothing, asynchronous on `stream`. */
int cp_ncdm_tables(long long ncosmo, int nspecies, cp_param h, cp_param T_cmb, const cp_param* m_ncdm, const cp_param* T_ncdm_over_cmb, int nq,
                   const double* nodes, const double* weights, double* d_tab, int device, void* stream);
/* cp_background_distance with massive neutrinos (ncdm may be NULL or have nspecies == 0).  With `second_is_omega_m` the non-relativistic
 * part of the neutrinos at z = 0 is taken out of Omega_m as well (cosmology.py:1163-1165). */
int cp_background_eval(long long ncosmo, long long nz, const cp_param* params, int second_is_omega_m, const cp_ncdm* ncdm, const double* d_z,
                       int z_shared, double* d_out, int kind, int device, void* stream);

/* ---- analytic matter power spectra for batches of cosmologies (replaces Transfer.transfer_k, Primordial.pk_k and the
 *      pk_callable x growth_factor_sq of Fourier.pk_interpolator in eisenstein_hu.py:189-215, 241-283, 315-324,
 *      eisenstein_hu_nowiggle.py:34-51, bbks.py:50-64, with the per-cosmology fit coefficients of eisenstein_hu.py:34-92) ---- */
enum cp_engine { CP_ENGINE_EH = 0, CP_ENGINE_EH_NOWIGGLE = 1, CP_ENGINE_BBKS = 2 };
enum cp_pk_what {
    CP_PK_MATTER = 0,    /* P(k, z) = T^2 x potential_to_density x curvature_to_potential x P_R x growth(z)^2 ; nz = 0: without growth */
    CP_PK_TRANSFER = 1,  /* transfer_k */
    CP_PK_PRIMORDIAL = 2, /* Primordial.pk_k */
    CP_PK_LOG_K_MATTER = 3 /* log(k P(k)), nz = 0: the input of the sine transform of wallish2018 (bao_filter.py:371), formed term by term
                            * instead of as the logarithm of the evaluated spectrum */
};
enum cp_pk_param { CP_PK_A_S = 0, CP_PK_N_S = 1, CP_PK_ALPHA_S = 2, CP_PK_BETA_S = 3, CP_PK_K_PIVOT = 4 /* 1/Mpc */, CP_PK_NPARAMS = 5 };
/* bg_params: the CP_BG_NPARAMS background parameters (cp_bg_param); pk_params: CP_PK_NPARAMS primordial parameters.
 * d_k : (nk) wavenumbers in h/Mpc shared by the batch; d_kscale : NULL, or (ncosmo) per-cosmology factors applied to d_k
 * (brieden2022 evaluates at k_fid / rescale and k_fid * rescale, bao_filter.py:493-499); d_z : (nz) redshifts shared by the batch
 * (CP_PK_MATTER only).  d_out : (ncosmo, max(nz, 1), nk), k fastest, (Mpc/h)^3.
 * d_work : device workspace of cp_power_workspace_bytes(ncosmo) bytes for the constants of the cosmologies (fit coefficients, primordial constants:
 * formed by one lane per cosmology in front of the evaluation, read back by it through scalar loads), owned by the caller and free again once the call's
 * kernels have run on `stream`: nothing is allocated inside the call.
 * ncdm (here and in every entry point below that takes one): the massive-neutrino tables of the same cosmologies (cp_ncdm_tables), NULL or nspecies == 0
 * for none.  The fits themselves do not know massive neutrinos (their scalars use omega_cdm + omega_b, eisenstein_hu.py:37-38) and the reference computes
 * with them all the same (its warnings are commented out, eisenstein_hu.py:21-33): the species enter through the background -- Omega0_m of pk_callable
 * (eisenstein_hu.py:322 with cosmology.py:381), Omega_m(z) and Omega_de(z) of the CPT92 growth factor (eisenstein_hu.py:134-135 with cosmology.py:1704-1736),
 * Omega_m / omega_m of the BBKS shape parameter (bbks.py:38), and Omega_cdm when the second background parameter is Omega_m (cosmology.py:1163-1165). */
long long cp_power_workspace_bytes(long long ncosmo);
int cp_power_eval(int engine, int what, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm,
                  const cp_param* pk_params, long long nk, const double* d_k, const double* d_kscale, long long nz, const double* d_z, double* d_out, void* d_work,
                  int device, void* stream);
enum cp_eh_scalar {
    CP_EH_RS_DRAG = 0 /* Mpc */, CP_EH_Z_DRAG = 1, CP_EH_Z_EQ = 2, CP_EH_K_EQ = 3, CP_EH_R_DRAG = 4, CP_EH_R_EQ = 5, CP_EH_K_SILK = 6,
    CP_EH_ALPHA_C = 7, CP_EH_BETA_C = 8, CP_EH_ALPHA_B = 9, CP_EH_BETA_NODE = 10, CP_EH_BETA_B = 11, CP_EH_ALPHA_GAMMA = 12,
    CP_EH_BBKS_GAMMA = 13, CP_EH_NSCALARS = 14
};
/* d_out : (ncosmo, CP_EH_NSCALARS) fit coefficients of eisenstein_hu.py:34-92, eisenstein_hu_nowiggle.py:21, bbks.py:38 */
/* eisenstein_hu_nowiggle_variants (reference eisenstein_hu_nowiggle_variants.py: Eisenstein & Hu 1997 with massive neutrinos,
 * scale-dependent growth): Transfer.transfer_kz (:87-154) or the matter power spectrum of Fourier.pk_interpolator (:159-193),
 * T(k, z)^2 x growth_factor(z, znorm=0)^2 x potential_to_density x curvature_to_potential x P_R(k).
 * what: CP_PK_TRANSFER | CP_PK_MATTER; of: 0 'delta_m', 1 'delta_cb'; ncdm as for cp_background_eval (NULL: no massive species);
 * d_out: (ncosmo, nz, nk), k fastest.  nz >= 1. */
int cp_power_eval_variants(int what, int of, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm,
                           const cp_param* pk_params, long long nk, const double* d_k, long long nz, const double* d_z, double* d_out, int device,
                           void* stream);
int cp_eh_scalars(long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, double* d_out, int device, void* stream);
/* the attributes of the eisenstein_hu_nowiggle_variants engine (EisensteinHuNoWiggleVariantsEngine._set_rsdrag / compute,
 * eisenstein_hu_nowiggle_variants.py:32-76) for a batch of cosmologies, by the device function the evaluation kernel itself uses.
 * ncdm as for cp_power_eval_variants.  d_out : (ncosmo, CP_VAR_NSCALARS). */
enum cp_variants_scalar {
    CP_VAR_OMEGA_B = 0, CP_VAR_OMEGA_M = 1, CP_VAR_FRAC_B = 2, CP_VAR_FRAC_CDM = 3, CP_VAR_FRAC_CB = 4, CP_VAR_FRAC_NCDM = 5, CP_VAR_THETA_CMB = 6,
    CP_VAR_Z_EQ = 7, CP_VAR_K_EQ = 8, CP_VAR_Z_DRAG = 9, CP_VAR_RS_DRAG = 10 /* Mpc */, CP_VAR_P_C = 11, CP_VAR_P_CB = 12, CP_VAR_GAMMA_NCDM = 13,
    CP_VAR_BETA_C = 14, CP_VAR_NSCALARS = 15
};
int cp_variants_scalars(long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, double* d_out, int device,
                        void* stream);

/* ---- cubic splines from fixed knots to fixed queries, applied to batches of rows as a banded linear operator
 *      (replaces scipy.interpolate.CubicSpline / RectBivariateSpline where the grids are shared by the batch:
 *      Interpolator1D jax.py:169-175 as used by integrate_sigma_r2, interpolator.py:285-289; Interpolator2D jax.py:241-271;
 *      the clamped second-derivative splines of wallish2018, bao_filter.py:377-382) ---- */
enum cp_spline_bc { CP_SPLINE_NATURAL = 0 /* y'' = 0 */, CP_SPLINE_CLAMPED = 1 /* y' = 0 */, CP_SPLINE_NOT_A_KNOT = 2 /* scipy default, == FITPACK s=0 */ };
enum cp_spline_post {
    CP_SPLINE_POST_NONE = 0, CP_SPLINE_POST_SQRT = 1, CP_SPLINE_POST_EXP10 = 2 /* 10^x: tables splined in log10 P come out as P */,
    /* OR-ed into post_op of cp_spline_apply, for measurements: force the banded vector-ALU kernel / the dense matrix-core (MFMA f64) kernel; by
     * default operators whose band is wider than half the knots (cp_linop_plan_create: quadrature weights, projectors) take the matrix cores */
    CP_SPLINE_PATH_VALU = 16, CP_SPLINE_PATH_MFMA = 32
};
typedef struct cp_spline_plan cp_spline_plan;
/* x : n strictly increasing knots (host), xq : nq query points (host), nu : derivative order 0..2;
 * extrapolate = 0: queries outside [x[0], x[n-1]] give NaN (Interpolator1D, jax.py:200). */
int cp_spline_plan_create(cp_spline_plan** plan, int n, const double* x, int nq, const double* xq, int bc, int nu, int extrapolate, int device);
/* d_out[row, q] = post_op(scale * sum_j W[q, j] d_y[row, j]);  d_y : (nrows, n), d_out : (nrows, nq), device, row-major */
/* Non-finite knot values: the banded vector kernel confines a NaN / Inf at knot j to the queries whose band covers j; the matrix-core route
 * multiplies the explicit zeros of a 64-query tile's window as well (0 x NaN = NaN), so there the NaN reaches every query of the tiles whose
 * window covers j (the whole row for dense plans).  Callers that rely on containment inside a row (none in this package: rows holding a
 * non-finite sample are NaN throughout in the reference too, jax.py:120-131) force CP_SPLINE_PATH_VALU. */
int cp_spline_apply(const cp_spline_plan* plan, const double* d_y, double* d_out, long long nrows, int post_op, double scale, void* stream);
/* the same with the rows taken in groups of `group` (nrows a multiple of it) and the result stored as (nrows / group, nq, group): the rows of a
 * group -- the redshifts of one table in sigma_rz -- become the fastest axis, i.e. the (nz, nr) -> (nr, nz) transposition (interpolator.py:875)
 * is part of the store instead of a pass of its own.  group = 0: cp_spline_apply. */
int cp_spline_apply_grouped(const cp_spline_plan* plan, const double* d_y, double* d_out, long long nrows, int group, int post_op, double scale, void* stream);
/* the same followed by an outer product with per-row factors, written once: d_out (nrows, nq, nz) = f(scale x spline(d_y)[row, q] x d_g[row, z]),
 * f = sqrt for CP_SPLINE_POST_SQRT, evaluated as sqrt(scale x spline) x sqrt(g): both factors are variances / squared growth factors (a negative one
 * gives NaN).  sigma_rz of separable P(k, z) = P(k) x growth^2(z): PowerSpectrumInterpolator2D.sigma_rz, interpolator.py:846-875 */
int cp_spline_apply_outer(const cp_spline_plan* plan, const double* d_y, const double* d_g, int nz, double* d_out, long long nrows, int post_op,
                          double scale, void* stream);
/* the same machinery for any fixed linear map of rows given densely (w_dense : nq x n row-major, host; a row starting with NaN
 * marks a query that evaluates to NaN): the composite-Simpson weights of integrate_sigma_r2 / integrate_sigma_d2 method 'simpson'
 * (interpolator.py:190-196, 280-284 with jax.py:365-507) are applied this way */
int cp_linop_plan_create(cp_spline_plan** plan, int n, int nq, const double* w_dense, int device);
/* a dense operator along the MIDDLE axis: d_out[b, q, c] = post_op(scale * sum_j W[q, j] d_y[b, j, c]);  d_y : (nbatch, n, ninner), d_out :
 * (nbatch, nq, ninner), c contiguous.  The redshift interpolation of batches of (z, k) tables, written z-major for the FFTLog: the y direction of
 * RectBivariateSpline (jax.py:241-271) in PowerSpectrumInterpolator2D.sigma_rz (interpolator.py:846-875).  Plans of cp_linop_plan_create only. */
int cp_linop_apply_mid(const cp_spline_plan* plan, const double* d_y, double* d_out, long long nbatch, long long ninner, int post_op, double scale,
                       void* stream);
/* Batches of (z, k) tables -> rows of P(k, z) in one kernel: d_out[b, zq, q] = f(scale x sum_zi Wz[zq, zi] sum_j Wk[q, j] d_tables[b, zi, j]),
 * the two passes of RectBivariateSpline (jax.py:241-271) without the k-splined tables in between (the accumulators of the first contraction are
 * the operands of the second).  kplan: spline / operator plan along k (n knots -> nq queries); zplan: cp_linop_plan_create plan of at most 32
 * knots and 64 queries; d_tables : (nbatch, zplan.n, kplan.n), k fastest; d_out : (nbatch, zplan.nq, kplan.nq).  post_op as cp_linop_apply_mid. */
int cp_tables_rows_available(const cp_spline_plan* kplan, const cp_spline_plan* zplan);
int cp_tables_rows(const cp_spline_plan* kplan, const cp_spline_plan* zplan, const double* d_tables, double* d_out, long long nbatch, int post_op, double scale,
                   void* stream);
int cp_spline_plan_destroy(cp_spline_plan* plan);
int cp_spline_plan_info(const cp_spline_plan* plan, int* n, int* nq, int* bandwidth);
/* the entries [first, first + count) of a row of knots that cp_spline_apply / cp_spline_apply_grouped read (whichever kernel they choose): a producer
 * may leave the others unwritten (cp_fftlog_execute_window) */
int cp_spline_plan_columns(const cp_spline_plan* plan, int* first, int* count);
/* the dense operator W (nq x n, row-major, host) and per-query inside-range flags: what the plan is built from */
int cp_spline_operator(int n, const double* x, int nq, const double* xq, int bc, int nu, int extrapolate, double* w_out, int* inside_out);

/* ---- sigma(r, z) of a batch of analytic cosmologies in one call (PowerSpectrumInterpolator2D.sigma_rz, reference interpolator.py:846-875 with
 *      integrate_sigma_r2 :200-292, for the interpolators the analytic engines build from a callable + growth factor, eisenstein_hu.py:295-329):
 *      d_out[c, q, z] = sqrt(spline(TophatVariance FFTLog of P_c(k))(r_q) x growth_sq[c, z]).  The batch is walked in `nblocks` blocks: `stream`
 *      evaluates P(k) (cp_power_eval) and transforms (cp_fftlog_execute) block after block while a second stream owned by the library stores the
 *      (nr x nz) results of the block before (cp_spline_apply_outer) -- ALU-bound and HBM-write-bound kernels side by side; `stream` waits for the
 *      last store before the call's work counts as done on it.  nblocks = 0 (what the package passes): ONE fused kernel instead -- P(k) evaluated
 *      into the row registers, FFTLog in LDS, spline out of LDS, results written once (csrc/cp_sigma.hip) -- when the transform is the default
 *      one (1024 samples padded to 2048: cp_sigma_rz_fused_available), else one block on `stream`.  fftlog: plan of the transform on d_k
 *      (nker = 1, n = nk); spline: plan from the transform's output grid to the radii.  d_work: cp_sigma_rz_workspace_bytes(ncosmo, nk) bytes.
 *      d_pk_out: (ncosmo, nk) or NULL -- the spectra themselves, for a caller that needs them afterwards (the sigma8 normalisation of a batch of
 *      cosmologies: eisenstein_hu.py:94-103 evaluates sigma8 of the fiducial amplitude, the filters then ask for P on the same wavenumbers).
 *      Allocates nothing, asynchronous. ---- */
int cp_sigma_rz_fused_available(const cp_fftlog_plan* fftlog, const cp_spline_plan* spline);
/* FFTLog of (nbatch, n) rows followed by the spline of every output row to the spline plan's queries, root taken for CP_SPLINE_POST_SQRT, as one
 * kernel: integrate_sigma_r2(method='fftlog') for spectra that sit in memory (interpolator.py:285-291); d_out : (nbatch, nq).  The transformed rows
 * are never written.  Plans for which cp_sigma_rz_fused_available() is 1 (CP_EUNSUPPORTED otherwise: make the two calls). */
int cp_fftlog_spline_execute(const cp_fftlog_plan* fftlog, const cp_spline_plan* spline, const double* d_in, double* d_out, long long nbatch, int post_op,
                             void* stream);
/* The same for any number of rows, with the spline SOLVED inside the kernel instead of applied as a banded operator: the output grid of an FFTLog
 * is geometric, the natural spline's tridiagonal system then has constant coefficients (second derivatives scaled by their interval) and its inverse is
 * two geometric tails: a lane owns a run of the knots the queries see (+ a halo of 32 on either side of the stretch, at most 512 knots in all), runs the
 * causal and the anti-causal first-order recursion over them in registers, and takes its neighbours' segment totals by DPP wave shifts -- exact to
 * rounding, no weights to fetch but 12 bytes per query (interpolator.py:285-291, jax.py:169-175 with bc_type='natural').
 * cp_geospline_plan_create: knots (n) a geometric grid (host), queries (nq <= 512, host; outside the knots: NaN); CP_EUNSUPPORTED when the knots
 * are not geometric, the queries span more than 448 knots or come within 32 knots of either end of the grid (take cp_spline_plan + cp_spline_apply).
 * cp_fftlog_geospline_execute: d_in (nbatch, n) rows; group = 0: d_out (nbatch, nq); group > 0 (even, dividing nbatch): rows come in groups
 * (the redshifts of one table) and d_out is (nbatch / group, nq, group), the layout of PowerSpectrumInterpolator2D.sigma_rz (interpolator.py:846-875).
 * d_out = post(spline), post_op CP_SPLINE_POST_NONE or CP_SPLINE_POST_SQRT.  Transform: 1024 samples padded to 2048, one kernel. */
typedef struct cp_geospline_plan cp_geospline_plan;
int cp_geospline_plan_create(cp_geospline_plan** plan, const double* knots, int n, const double* queries, int nq, int device);
/* The same spline with its solve folded into the transform: in the B-spline basis of the geometric knots (scale invariant: B_j(s) = B_0(s / rho^j))
 * the interpolation conditions are a constant-coefficient tridiagonal system, and on FFTLog's periodic padded grid with a power-law postfactor that is
 * a division of u in the frequency domain.  The plan builds and OWNS that transform from the caller's FFTLog tables (pre, post: (npad) of one kernel, u_re_im:
 * (npad / 2 + 1) re / im pairs, as for cp_fftlog_plan_create) and execute then costs four reads and sixteen multiply-adds per query after the FFT;
 * the periodic ends of the padded grid stand for the natural ends of the knots, both forgotten like 0.27^distance: queries within 32 knots of either
 * end of the knots are refused (CP_EUNSUPPORTED, as are a postfactor that is no power law, other sizes than 1024 -> 2048, more than 512 queries or a
 * span of more than 510 knots): take cp_geospline_plan_create.  cp_fftlog_geospline_execute with such a plan takes fftlog = NULL. */
int cp_geospline_plan_create_prefiltered(cp_geospline_plan** plan, int n, int npad, const double* pre, const double* post, const double* u_re_im,
                                         const double* knots, const double* queries, int nq, int device);
/* host only: the B-spline pieces such a plan evaluates with -- basis[4 i + d] = coefficient of x^d, x = (r - s_j) / (s_j+1 - s_j), of the cubic B-spline of the
 * geometric knots (ratio rho > 1) centred on s_(j-1+i); their values at x = 0 are the alpha, beta, gamma of the interpolation conditions */
int cp_geospline_basis(double rho, double* basis);
int cp_geospline_plan_destroy(cp_geospline_plan* plan);
int cp_geospline_plan_info(const cp_geospline_plan* plan, int* first_knot, int* nknots, int* nq);
int cp_fftlog_geospline_execute(const cp_fftlog_plan* fftlog, const cp_geospline_plan* spline, const double* d_in, double* d_out, long long nbatch,
                                int group, int post_op, void* stream);
/* The sigma8 normalisation of a batch of analytic cosmologies (BaseEngine._rescale_sigma8: eisenstein_hu.py:94-103 with Fourier.sigma8_m :331-342) as
 * one kernel: sigma8 at the amplitudes pk_params carry (the reference's first guess _get_A_s_fid, cosmology.py:505-510) from the spectra, the
 * functional d_functional (1, nk) of r = 8 (what transform + spline return for unit spectra: cp_sigma_rz_functional) and the CPT92 growth factor
 * at z = 0; then d_rsigma8[c] = sigma8 / that, d_amplitude[c] = A_s rsigma8^2 (or NULL) and d_pk_out (ncosmo, nk) = the spectra WITHOUT growth at
 * the normalised amplitude (or NULL).  sigma8: one target for all (ptr NULL) or per cosmology.  nk = 1024; d_work as for cp_sigma_rz_functional. */
int cp_sigma8_normalise(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
        int nk,
                        const double* d_k, const double* d_functional, cp_param sigma8, double* d_rsigma8, double* d_amplitude, double* d_pk_out,
                        void* d_work, int device, void* stream);
long long cp_sigma_rz_workspace_bytes(long long ncosmo, int nk);
int cp_sigma_rz_analytic(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
        int nk,
                         const double* d_k, const cp_fftlog_plan* fftlog, const cp_spline_plan* spline, const double* d_growth_sq, int nz,
                         double* d_out, double* d_pk_out, void* d_work, int nblocks, int device, void* stream);
/* the fused kernel of cp_sigma_rz_analytic with the spline evaluated from B-spline coefficients: `spline` is a plan of cp_geospline_plan_create_prefiltered
 * built from the transform's tables (it owns the transform whose u carries the prefilter), so a radius costs four LDS reads and 20 multiply-adds where the
 * banded operator takes ~44 weights from L2.  Radii the plan refuses (near the ends of the output grid): cp_sigma_rz_analytic.  d_work as there. */
int cp_sigma_rz_analytic_prefiltered(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm,
                                     const cp_param* pk_params, int nk, const double* d_k, const cp_geospline_plan* spline, const double* d_growth_sq, int nz,
                                     double* d_out, double* d_pk_out, void* d_work, int device, void* stream);
/* the same for at most 4 radii (the sigma8 normalisation: one) as a linear functional of the spectrum: d_functional (nq, nk) holds the rows F with
 * sigma^2(r_q) = sum_j F[q, j] P(k_j) -- what the caller's transform + spline return for unit spectra --, the kernel evaluates P(k) and the dot
 * products, one wave per cosmology, no transform.  d_work, d_pk_out as for cp_sigma_rz_analytic; d_out (ncosmo, nq, nz). */
int cp_sigma_rz_functional(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
        int nk,
                           const double* d_k, const double* d_functional, int nq, const double* d_growth_sq, int nz, double* d_out, double* d_pk_out,
                           void* d_work, int device, void* stream);


/* clamped cubic spline through uniformly spaced knots (positions 1..n) of x^2-weighted data with the knots of a per-column box
 * [box[2c], box[2c+1]] removed, evaluated at all positions and divided by x^2 (wallish2018 peak removal, bao_filter.py:387-405).
 * d_y, d_out : (ncol, n); d_box : (ncol, 2) int32.  d_out may be d_y (in place: only the boxes are rewritten, nothing is copied). */
int cp_gap_spline(const double* d_y, const int* d_box, double* d_out, long long ncol, int n, int device, void* stream);
/* The box cp_gap_spline removes (bao_filter.py:390-394): d_dd (ncol, n) second derivatives -> d_box (ncol, 2) =
 * [argmax over [margin_first, n - margin_first) + offset_first, argmax over [that argmax + margin_second, n - margin_first) + offset_second],
 * first index of the maximum as ndarray.argmax (index 0 when the second range is empty). */
int cp_wallish_box(const double* d_dd, long long ncol, int n, int margin_first, int margin_second, int offset_first, int offset_second, int* d_box,
                   int device, void* stream);
/* Both of the above steps of wallish2018 that come before cp_gap_spline, as one kernel (bao_filter.py:377-394): the second derivatives at the knots
 * of the clamped CubicSpline through (1 .. n, d_y[row]) -- spline(x, nu=2) -- by a tridiagonal solve in LDS, a wave per sequence, and the box of
 * cp_wallish_box found on them.  d_y : (nrows, n), n in {1024, 2048} (CP_EUNSUPPORTED otherwise: the two calls above); d_box : (nrows, 2) int32; d_dd : (nrows, n) or NULL (the second
 * derivatives are written only on request); d_gap : NULL, or where the sequences are (d_y itself, or a copy of it): cp_gap_spline's rewriting of
 * the box then happens there, in the same kernel. */
int cp_wallish_dd_box(const double* d_y, long long nrows, int n, int margin_first, int margin_second, int offset_first, int offset_second, int* d_box,
                      double* d_dd, double* d_gap, int device, void* stream);

/* ---- the short forms of the transcendental functions the ALU-bound kernels use (csrc/cp_math.h), evaluated on an array ----
 * A diagnostic: what the accuracies stated for them are tested on.  No counterpart in the reference (numpy's libm calls).  d_y[i] = f(d_x[i]). */
enum cp_math_function {
    CP_MATH_EXP_MID = 0,       /* e^x, degree-13 series */
    CP_MATH_EXP_TAB = 1,       /* e^x, 64-entry table + degree-5 series */
    CP_MATH_LOG_POS = 2,       /* log x, degree-14 odd series */
    CP_MATH_LOG_TAB = 3,       /* log x, 64-entry table + degree-7 series (absolute error 2e-16 max(1, |log x|)) */
    CP_MATH_EXP10_MID = 4,
    CP_MATH_EXP10_TAB = 5,     /* |x| < 300 */
    CP_MATH_SIN_BOUNDED = 6,
    CP_MATH_RECIP = 7,         /* finite, normal x */
    CP_MATH_RSQRT_POS = 8      /* positive, finite, normal x */
};
int cp_math_eval(int kind, const double* d_x, double* d_y, long long n, int device, void* stream);

/* ---- clamped cubic spline through knots spliced from contiguous pieces of two row arrays (wallish2018, bao_filter.py:415-431) ----
 * The knots x (nknots, increasing) take their values from up to three pieces: piece p = columns [piece_start[p], piece_start[p] + piece_count[p])
 * of the rows of array piece_src[p] (0 or 1), in this order.  cp_splice_apply solves the tridiagonal system of
 * scipy.interpolate.CubicSpline(x, values, bc_type='clamped') for every row in LDS (a wave per row) and evaluates the spline at the queries xq
 * (NaN outside the knots): d_out (nrows, nq).  With d_tophat (nq; the queries must be the columns of array 0) the damping of wallish2018 follows in
 * the same kernel: d_out = p / ((p / spline - 1) tophat + 1), p = array 0 (bao_filter.py:421-431).  CP_EUNSUPPORTED from plan creation when the
 * knots do not fit the LDS scheme (more than a few thousand knots off a uniform stretch): apply the spline as operators then. */
typedef struct cp_splice_plan cp_splice_plan;
int cp_splice_plan_create(cp_splice_plan** plan, int nknots, const double* x, int npieces, const int* piece_src, const int* piece_start,
                          const int* piece_count, int nq, const double* xq, int device);
int cp_splice_apply(const cp_splice_plan* plan, const double* d_src0, int n0, const double* d_src1, int n1, long long nrows, const double* d_tophat,
                    double* d_out, void* stream);
int cp_splice_plan_destroy(cp_splice_plan* plan);
/* Which kernel the plan runs: 0 the elimination in LDS (any knots); 1 two first-order recursions per lane in registers, for plans most of whose knots
 * lie on a uniform grid with every spline query inside that stretch or in the interval on either side of it (wallish2018: 3 051 of 3 666 knots;
 * csrc/cp_splice_uniform.h) -- the default where it fits.  cp_splice_plan_set_scheme(plan, 1) returns CP_EUNSUPPORTED where it does not. */
int cp_splice_plan_scheme(const cp_splice_plan* plan);
int cp_splice_plan_set_scheme(cp_splice_plan* plan, int scheme);

/* ---- natural / clamped cubic spline of very many rows by elimination in LDS (csrc/cp_spline_rows.hip) ----
 * The same function as cp_spline_plan_create(bc, nu = 0) + cp_spline_apply / cp_spline_apply_grouped
 * (scipy.interpolate.CubicSpline, jax.py:169-175; the FFTLog output -> radii step of sigma_r / sigma_rz, interpolator.py:285-291, 846-876), computed
 * by solving the spline's tridiagonal system per row instead of applying its inverse as a banded operator: ~10 operations per knot the queries
 * can see and 8 per query, against 64 multiply-adds per query (the matrix-core route: whole windows of knots).  Only the knots the queries touch
 * (plus a halo) are read.  d_y : (nrows, n); d_out : (nrows, nq), or (nrows / group, nq, group) for group > 0; post_op : CP_SPLINE_POST_NONE or
 * CP_SPLINE_POST_SQRT of scale x spline; queries outside the knots: NaN, or -- extrapolate -- the cubic of the end interval continued.  bc: any of cp_spline_bc (not-a-knot: the outermost unknowns are
 * eliminated by hand, the system stays tridiagonal).  CP_EUNSUPPORTED from plan creation for windows of more than a few thousand knots: use
 * cp_spline_plan_create. */
typedef struct cp_spline_rows_plan cp_spline_rows_plan;
int cp_spline_rows_plan_create(cp_spline_rows_plan** plan, int n, const double* x, int bc, int extrapolate, int nq, const double* xq, int device);
int cp_spline_rows_plan_info(const cp_spline_rows_plan* plan, int* first_knot, int* nknots, int* rows_per_wave, int* halo);
int cp_spline_rows_apply(const cp_spline_rows_plan* plan, const double* d_y, long long nrows, int post_op, double scale, int group, double* d_out,
                         void* stream);
/* the second derivatives of the spline at its knots, (nrows, n) -- the spline's own representation, from which any query follows by the four-term
 * formula above; needs a plan whose queries span the knots */
int cp_spline_rows_second_derivatives(const cp_spline_rows_plan* plan, const double* d_y, long long nrows, double* d_m, void* stream);
/* the same with the knot values beside them: d_ym (nrows, n, 2) = (y_j, M_j), the four numbers of a query in 32 contiguous bytes */
int cp_spline_rows_pairs(const cp_spline_rows_plan* plan, const double* d_y, long long nrows, double* d_ym, void* stream);
int cp_spline_rows_plan_destroy(cp_spline_rows_plan* plan);
/* cp_tables_rows with the k direction evaluated from the tables' own second derivatives instead of multiplied by an operator: d_m (nbatch, nz, n) =
 * cp_spline_rows_second_derivatives of the tables along k for the knots of kplan (a cp_spline_rows plan, any boundary condition, whose queries --
 * the output wavenumbers -- span the knots); zplan as in cp_tables_rows.  The second derivatives belong to the tables: a caller evaluates them once
 * per table set (what scipy's RectBivariateSpline does when it is built, jax.py:241-271) and re-uses them for every grid of wavenumbers / redshifts.
 * d_m NULL: d_tables holds (nbatch, nz, n, 2) pairs (cp_spline_rows_pairs of the tables) -- one partly used cache line per piece of a row instead of two. */
int cp_tables_rows_direct(const cp_spline_rows_plan* kplan, const cp_spline_plan* zplan, const double* d_tables, const double* d_m, double* d_out,
                          long long nbatch, int post_op, double scale, void* stream);

/* cubic splines of many rows through SHARED knots d_xk (n, ascending; uniformly spaced to rounding: the interval of a query is found from a uniform
 * first guess and set right against the knots), each row evaluated at ITS OWN queries: d_y (nrows, n) values, d_m (nrows, n) second derivatives at the
 * knots (cp_spline_rows_second_derivatives, any boundary condition), d_xq (nrows, nq); outside the knots the cubic of the end interval is continued, NaN
 * queries give NaN.  d_out (nrows, nq), or transposed != 0: (nq, nrows), the knot-major layout cp_spline_columns reads.  peakaverage over a batch of
 * cosmologies (bao_filter.py:565-574: the knots of its two splines move with the rs_drag ratio of the cosmology). */
int cp_spline_rows_at_queries(const double* d_xk, const double* d_y, const double* d_m, long long nrows, int n, const double* d_xq, int nq, double* d_out,
                              int transposed, int device, void* stream);

/* the elementwise stages of the two filters over (nrows, n) batches of spectra, one pass each (csrc/cp_bao.hip):
 * cp_wallish_finish: pknow = d_a (+ d_b when not NULL: the spliced spline applied as two operators), wiggles = (pk / pknow - 1) tophat + 1,
 *   out = pk / wiggles (bao_filter.py:421-431); d_tophat : (n).
 * cp_brieden_ratio: pknow = now x g0[c] x correction[j], ratio = rows / pknow / ratio_fid[j] (bao_filter.py:493-499); (nb, n) rows.
 * cp_brieden_knots: log10 of envelope x pknow x ratio_now_fid against log10(k_fid / rescale[c]) as knot-major (n + 4, nb) arrays with the two
 *   log-log extrapolated knots of _pad_log (interpolator.py:42-87) on either side: the input of cp_spline_columns.
 * cp_brieden_finish: out = pk (nb, nk) with columns [first, first + n) replaced by 10^resampled (knot-major (n, nb)) (bao_filter.py:509). */
int cp_wallish_finish(const double* d_pk, const double* d_a, const double* d_b, const double* d_tophat, double* d_out, long long nrows, int n, int device,
                      void* stream);
int cp_brieden_ratio(const double* d_rows, const double* d_now, const double* d_g0, const double* d_correction, const double* d_ratio_fid, double* d_pknow,
                     double* d_ratio, long long nb, int n, int device, void* stream);
int cp_brieden_knots(const double* d_envelope, const double* d_pknow, const double* d_ratio_now_fid, const double* d_k_fid, const double* d_rescale,
                     double extrap_kmin, double extrap_kmax, double* d_xk, double* d_yk, long long nb, int n, int device, void* stream);
int cp_brieden_finish(const double* d_pk, const double* d_resampled, double* d_out, long long nb, int nk, int first, int n, int device, void* stream);
/* The three steps above as one kernel, a wave per cosmology (no knot-major arrays, no scratch): d_out (nb, nk) = d_pk with the columns [first, first + n)
 * replaced by 10^(natural cubic spline through (log10(k_fid / rescale[c]), log10(envelope pknow ratio_now_fid)) and the two extrapolated knots of _pad_log on
 * either side, evaluated at d_log_k_fid = log10(k_fid)) (bao_filter.py:500-509).  k_fid must be a geometric grid (it is a range of the filter's geomspace):
 * the spline's system then has constant coefficients; for a d_log_k_fid that is not uniformly spaced to 1e-9 of its step the kernel writes NaN over
 * the range instead of a wrong spline).  129 <= n <= 512; CP_EUNSUPPORTED otherwise (the three calls then). */
int cp_brieden_resample(const double* d_envelope, const double* d_pknow, const double* d_ratio_now_fid, const double* d_k_fid, const double* d_log_k_fid,
                        const double* d_rescale, double extrap_kmin, double extrap_kmax, const double* d_pk, double* d_out, long long nb, int n, int nk,
                        int first, int device, void* stream);
/* cp_brieden_ratio + the envelope operator + cp_brieden_resample as one kernel: the envelope (two quadratic splines through the extrema of the fiducial
 * wiggles, bao_filter.py:482-488) is linear in the ratio AT THOSE EXTREMA only, so the spectra are needed at np wavenumbers instead of n, and neither pknow,
 * ratio nor envelope is written to memory.  d_pk_peaks : (nb, np) P_c(k_fid[peaks] / rescale[c]); d_now : (nb, n) the no-wiggle spectra at k_fid x rescale[c];
 * d_g0 : (nb) growth; d_correction, d_ratio_fid : (n); d_peaks : (np) ascending positions in k_fid; d_operator : (np, n), row p = column peaks[p] of the (n x n)
 * operator (its other columns are zero); np <= 64; the rest as cp_brieden_resample (bao_filter.py:493-509). */
int cp_brieden_smooth(const double* d_pk_peaks, const double* d_now, const double* d_g0, const double* d_correction, const double* d_ratio_fid, const int* d_peaks,
                      const double* d_operator, int np, const double* d_ratio_now_fid, const double* d_k_fid, const double* d_log_k_fid, const double* d_rescale,
                      double extrap_kmin, double extrap_kmax, const double* d_pk, double* d_out, long long nb, int n, int nk, int first, int device,
                      void* stream);

/* natural cubic spline per column with per-column knots (brieden2022 re-sampling with one rs_drag ratio per column, bao_filter.py:503-509):
 * d_xk, d_yk : (n, ncol) knot-major; d_xq : (nq) ascending queries shared by all columns; d_out : (nq, ncol);
 * d_scratch : cp_spline_columns_scratch_doubles(ncol, n) doubles (the columns are cut in runs of knots eliminated side by side, each with its own
 * scratch).  Queries outside a column's knots give NaN. */
long long cp_spline_columns_scratch_doubles(long long ncol, int n);
int cp_spline_columns(const double* d_xk, const double* d_yk, long long ncol, int n, const double* d_xq, int nq, double* d_out, double* d_scratch,
                      int device, void* stream);

/* ---- batched real FFTs of rows: the two methods of the reference's FFT engine protocol (BaseFFTEngine, fftlog.py:508-544) as standalone
 *      transforms, for callers that hold a NumpyFFTEngine / FFTWEngine object and call forward / backward themselves (FFTlog.__call__ is
 *      the fused kernel and does not come through here) ---- */
typedef struct cp_rfft_plan cp_rfft_plan;
/* size : real samples per row, a power of two from 8 to 16384 (FFTlog's padded sizes) */
int cp_rfft_plan_create(cp_rfft_plan** plan, int size, int device);
int cp_rfft_plan_destroy(cp_rfft_plan* plan);
/* NumpyFFTEngine.forward (fftlog.py:536-539): d_in (nrows, size) real -> d_out (nrows, size / 2 + 1) complex (re, im interleaved) = rfft */
int cp_rfft_forward(const cp_rfft_plan* plan, const double* d_in, double* d_out, long long nrows, void* stream);
/* NumpyFFTEngine.backward (fftlog.py:541-544): d_in (nrows, size / 2 + 1) complex -> d_out (nrows, size) real = irfft(conj(in), n=size) when
 * conj_input != 0, irfft(in, n=size) otherwise; imaginary parts of the DC and Nyquist bins are ignored, as numpy's c2r does.  Rows are
 * independent (one transform per row).  Not in place. */
int cp_rfft_backward(const cp_rfft_plan* plan, const double* d_in, double* d_out, long long nrows, int conj_input, void* stream);

/* ---- batched orthonormal DST-II / DST-III of rows (replaces scipy.fftpack.dst / idst(type=2, norm='ortho') of the
 *      wallish2018 filter, bao_filter.py:371-372, 412) ---- */
typedef struct cp_dst_plan cp_dst_plan;
/* n in {256, 1024, 4096}; kx : optional (n) host abscissa for the fused maps of cp_dst_execute, or NULL */
int cp_dst_plan_create(cp_dst_plan** plan, int n, const double* kx, int device);
/* d_in, d_out : (nrows, n) device.  inverse = 0: Y = dst2_ortho(x); 1: x = idst2_ortho(Y).  flags:
 * CP_DST_FUSED: forward transforms log(kx_n * in_n) (bao_filter.py:371), inverse returns exp(x_n) / kx_n (bao_filter.py:413);
 * CP_DST_SPLIT: the coefficient side is de-interleaved, Y_0, Y_2, ... in the first half of each row and Y_1, Y_3, ... in the second
 *               (the filter's even / odd sequences, bao_filter.py:373, 408-410, without gather / scatter copies)..
 * Rows are independent, as in scipy's row-by-row transform, although two of them share one complex FFT inside the kernel: a row holding a
 * sample that is not finite (or, with CP_DST_FUSED forward, not positive) is stored as NaN and leaves its partner untouched. */
#define CP_DST_FUSED 1
#define CP_DST_SPLIT 2
int cp_dst_execute(const cp_dst_plan* plan, const double* d_in, double* d_out, long long nrows, int inverse, int flags, void* stream);
int cp_dst_plan_destroy(cp_dst_plan* plan);
/* Forward transform (inverse = 0) of the rows log(kx_n P_c(kx_n)), c < ncosmo, with P_c the spectrum WITHOUT growth of cosmology c of an analytic
 * engine (cp_engine; bg_params / pk_params as for cp_power_eval) evaluated inside the kernel: wallish2018 on a batch of cosmologies
 * (bao_filter.py:371 behind eisenstein_hu.py:315-324) without the (ncosmo, 4096) rows of cp_power_eval(CP_PK_LOG_K_MATTER) in between -- same
 * arithmetic per sample.  Plans of length 4096 made with their abscissa kx; flags: CP_DST_SPLIT.  d_out: (ncosmo, 4096).
 * d_work: cp_dst_forward_analytic_workspace_bytes(ncosmo) bytes, free again once the call's kernels have run on `stream`. */
long long cp_dst_forward_analytic_workspace_bytes(long long ncosmo);
int cp_dst_forward_analytic(const cp_dst_plan* plan, int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m,
                            const cp_ncdm* ncdm, const cp_param* pk_params, double* d_out, void* d_work, int flags, void* stream);
/* The same with the NEXT step of wallish2018 in the kernel's epilogue (cp_wallish_dd_box with d_gap = d_y, bao_filter.py:373-405): the coefficients are
 * written de-interleaved (CP_DST_SPLIT: every row is its even-indexed sequence followed by its odd-indexed one, 2048 knots each), d_box (2 ncosmo, 2)
 * receives the box of every sequence, and the boxes are already rewritten in d_out -- the four sequences of a pair of cosmologies are solved by the
 * four waves of the workgroup that transformed them, without the (2 ncosmo, 2048) coefficients being read again. */
int cp_dst_forward_analytic_box(const cp_dst_plan* plan, int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m,
                                const cp_ncdm* ncdm, const cp_param* pk_params, double* d_out, void* d_work, int* d_box, int margin_first, int margin_second,
                                int offset_first, int offset_second, void* stream);

/* Everything of wallish2018 behind its forward transform as ONE kernel (bao_filter.py:373-431): for every row of d_coef (nrows, 4096: the sine-transform
 * coefficients in the CP_DST_SPLIT layout, as cp_dst_forward_analytic / cp_dst_execute write them) the second derivatives of its two sequences, their boxes
 * (d_box (2 nrows, 2), as cp_wallish_dd_box) and the boxes rewritten IN PLACE in d_coef; the inverse transform with exp(.) / k_lin; the clamped spline through
 * the spliced knots (plan `splice`: cp_splice_plan_create with the stretch of the linear grid taken from the transformed rows, array 1, the knots outside it
 * from the rows d_pk (nrows, npk) of P, array 0) at the filter's npk wavenumbers and the damping with d_tophat (or NULL): d_out (nrows, npk).  What
 * cp_wallish_dd_box + cp_dst_execute(inverse, CP_DST_FUSED | CP_DST_SPLIT) + cp_splice_apply compute with the transformed rows never leaving the CU.
 * CP_EUNSUPPORTED when the plans are not the filter's (transform of length 4096 with its abscissa; a splice plan that runs the uniform-stretch scheme with
 * 49 knots per lane): the caller then takes the three calls. */
int cp_wallish_tail(const cp_dst_plan* dst, const cp_splice_plan* splice, double* d_coef, const double* d_pk, int npk, long long nrows, int margin_first,
                    int margin_second, int offset_first, int offset_second, const double* d_tophat, int* d_box, double* d_out, void* stream);

/* cp_dst_forward_analytic + cp_wallish_tail as ONE kernel: wallish2018 (bao_filter.py:371-431) of a batch of cosmologies of an analytic engine from their
 * parameters (engine, bg_params, ncdm, pk_params as for cp_power_eval; d_work: cp_dst_forward_analytic_workspace_bytes(ncosmo) bytes) and the rows d_pk
 * (ncosmo, npk) of their spectra at the filter's wavenumbers: a workgroup evaluates log(k_lin P_c(k_lin)) of a pair of cosmologies into the forward
 * transform and takes the coefficients through cp_wallish_tail's stages without their leaving the CU.  d_coef: NULL, or (ncosmo, 4096) to receive the
 * coefficient sequences with their boxes rewritten (what the two calls leave in d_coef); d_box, d_out, the plans and CP_EUNSUPPORTED as for cp_wallish_tail. */
int cp_wallish_full(const cp_dst_plan* dst, const cp_splice_plan* splice, int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m,
                    const cp_ncdm* ncdm, const cp_param* pk_params, const double* d_pk, int npk, int margin_first, int margin_second, int offset_first,
                    int offset_second, const double* d_tophat, int* d_box, double* d_coef, double* d_out, void* d_work, void* stream);

/* tensor-product spline at PAIRS of points, RectBivariateSpline(...)(x, y, grid=False) (Interpolator2D, jax.py:241-287):
 * d_out[b, q] = sum_i sum_j d_wx[q, i] d_f[b, i, j] d_wy[q, j]; d_wx (nq, nx), d_wy (nq, ny): rows of the two 1-D spline operators at the queries,
 * d_f (nbatch, nx, ny) tables, d_out (nbatch, nq). */
int cp_bilinear_pairs(const double* d_wx, const double* d_wy, const double* d_f, double* d_out, long long nbatch, int nq, int nx, int ny, int device,
                      void* stream);

/* ---- piecewise-linear interpolation of one table at many points (replaces numpy.interp of the 'tabulated' engine, tabulated.py:31-36) ----
 * d_xp (ascending), d_fp : (n) device table; d_x, d_out : (nx) device.  Bit-identical to numpy.interp inside [xp[0], xp[n-1]];
 * NaN outside (the reference raises CosmologyError there: its caller checks the range) and for NaN samples. */
int cp_interp_linear(const double* d_xp, const double* d_fp, long long n, const double* d_x, double* d_out, long long nx, int device, void* stream);
/* The same on a table kept as a plan (TabulatedEngine, tabulated.py:6-36: one table per column, built once, applied to every catalogue): x, f : (n) HOST
 * arrays, x ascending.  The plan holds (x, f) pairs on the device and the LAW of the knots -- 1 uniform in x, 2 uniform in log x behind at most 8 leading
 * knots (the reference's data/desi.dat: 0, then 40 001 redshifts from 1e-8 to 100), 0 neither: the interval of a sample is then guessed from the sample
 * itself and corrected by a walk (0 or 1 steps) instead of bisected (15 dependent loads per sample); law 0 bisects.  Bit-identical to numpy.interp in all
 * three cases.  outside (host, may be NULL): set to 1 when a sample lies outside [x_0, x_{n-1}] or is NaN (those come out NaN) -- the reference raises
 * there (tabulated.py:33-34); asking for it makes the call wait for the stream, NULL leaves it asynchronous (and raises nothing: a later call that asks
 * reports its own samples only).  The table has one flag word: calls that ask for `outside` must not overlap on two streams of one table; calls that do
 * not may.  cp_interp_table_law reports the law found. */
typedef struct cp_interp_table cp_interp_table;
int cp_interp_table_create(cp_interp_table** table, long long n, const double* x, const double* f, int device);
int cp_interp_table_law(const cp_interp_table* table, int* law, long long* first);
int cp_interp_table_apply(const cp_interp_table* table, const double* d_x, double* d_out, long long nx, int* outside, void* stream);
/* single-precision samples and results (catalogues kept in float32): computed in double as above, the result rounded once -- the reference's cast of its
 * float64 result to the dtype of its input; tables with a law only (CP_EUNSUPPORTED otherwise: the caller widens the samples) */
int cp_interp_table_apply_f32(const cp_interp_table* table, const float* d_x, float* d_out, long long nx, int* outside, void* stream);
int cp_interp_table_destroy(cp_interp_table* table);

/* ---- cubic splines at many points (replaces Interpolator1D.__call__ = CubicSpline(x, fun)(xq) when there are few splines and 1e6-1e9 queries:
 *      DistanceToRedshift, utils.py:275-316; jax.py:169-175) ----
 * d_xk : (n) ascending knots; d_y, d_s : (ncol, n) values and first derivatives at the knots (cp_spline_apply with nu = 1 at the knots);
 * d_xq : (nq) queries; d_out : (ncol, nq).  nu : derivative order 0, 1, 2.  Outside [xk[0], xk[n-1]]: the end polynomials if extrapolate, else NaN. */
int cp_spline_points(const double* d_xk, const double* d_y, const double* d_s, long long n, int ncol, const double* d_xq, double* d_out, long long nq, int nu,
                     int extrapolate, int device, void* stream);

/* ---- row screening utility (cp_fftlog_execute and cp_dst_execute screen their rows themselves; this pass is for callers that want
 *      the flags, e.g. to count or report the rows a batch loses) ----
 * d_x : (nrows, n) device.  d_ok[row] = 1 if every entry of the row is finite (and > 0 if require_positive: the fused log map of
 * cp_dst_execute), else 0.  d_scale : optional (nrows) device, 2^e >= max |row| with e <= 1023 (1 for all-zero and non-finite rows), or NULL. */
int cp_rows_screen(const double* d_x, long long nrows, long long n, int require_positive, unsigned char* d_ok, double* d_scale, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COSMOPRIMO_AMD_H */
