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

#define CP_ABI_VERSION 1

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
    CP_KERNEL_GAUSSIAN_SQ = 5         /* GaussianSqKernel            fftlog.py:759-766 */
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

/* ---- fused FFTLog (replaces FFTlog.__call__, fftlog.py:198-241: pad -> x prefactor -> rfft -> x u ->
 *      irfft(conj) -> x postfactor -> crop, i.e. numpy.fft.rfft/irfft at fftlog.py:540, 544) ---- */
typedef struct cp_fftlog_plan cp_fftlog_plan;

/* Tables are HOST arrays as the reference's FFTlog._setup produces them (fftlog.py:144-184):
 *   pre, post : (nker, npad) float64  padded_prefactor / padded_postfactor (real part)
 *   u_re_im   : (nker, npad/2 + 1) complex128 padded_u, interleaved
 * npad must be the power of two 2**((n*minfolds-1).bit_length()); pad splits follow fftlog.py:152-153. */
int cp_fftlog_plan_create(cp_fftlog_plan** plan, int n, int npad, int nker, const double* pre, const double* post,
                          const double* u_re_im, int device);
/* d_in : device (nbatch, nker, n) float64 C-contiguous;  d_out : device (nbatch, nker, n or npad).
 * extrap_*: cp_extrap, val_* used for CP_EXTRAP_CONSTANT.  keep_padding as in fftlog.py:233-237. */
int cp_fftlog_execute(const cp_fftlog_plan* plan, const double* d_in, double* d_out, long long nbatch, int extrap_left,
                      double val_left, int extrap_right, double val_right, int keep_padding, void* stream);
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
    CP_BG_HUBBLE = 5               /* hubble_function               cosmology.py:1756 */
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
/* the 119 interpolation knots (host), get_default_z_interp('comoving_radial_distance'), cosmology.py:1947-1949 */
int cp_background_knots(double* zc_out, int n);

#ifdef __cplusplus
}
#endif
#endif /* COSMOPRIMO_AMD_H */
