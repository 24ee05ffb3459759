// cp_background.hip -- background E(z) and distances for batches of cosmologies (gfx950) + C ABI.
//
// Replaces DefaultBackground.comoving_radial_distance and the distances derived from it
// (reference cosmoprimo/cosmology.py:2027-2042, 1855-1912) together with everything they call:
// BaseBackground.efunc / rho_* (:1680-1754), jax.odeint 'rk4' on the 119-knot grid (jax.py:672-716,
// cosmology.py:1940-1951), the natural cubic spline of Interpolator1D (jax.py:169-175) and the per-cosmology
// derived density parameters (cosmology.py:355-397).
//
// One thread per (cosmology, z) sample, no per-sample storage:
//  * the RK4 of the reference has a y-independent integrand, i.e. it is Simpson with midpoints on each knot
//    interval: inc_i = h_i/6 (f(z_i) + 2 f(mid) + 2 f(mid) + f(z_{i+1})), T_{i+1} = T_i + inc_i;
//  * the natural-spline system for the knot derivatives is tridiagonal with a matrix that depends on the grid
//    only.  The value at z in [z_k, z_{k+1}] needs just s_k and s_{k+1}: eliminate forward from knot 0 up to k and
//    backward from knot 118 down to k+1 (all pivots are host-precomputed constants) and solve the remaining 2x2.
//    Every interval is integrated exactly once, in the order the thread needs it (forward up to k, then from the
//    top down), so the 237 E(z) evaluations of the reference are all that is computed.
// ALU-bound (pow, exp, sqrt per E(z)); HBM traffic is 8 bytes per parameter array + 8 in + 8 out per sample.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <mutex>
#include <type_traits>
#include <limits>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_cosmo_common.h"
#include "cp_error.h"

namespace {

using namespace cpcosmo;

constexpr int NK_DIST = 119;  // knots of get_default_z_interp('comoving_radial_distance'), cosmology.py:1947-1949
constexpr int NK_TIME = 400;  // knots of get_default_z_interp('time' / 'age'), cosmology.py:1945-1946

template <int NK>
struct TablesN {
    double zc[NK], dx[NK], l[NK], u[NK], idf[NK], idb[NK], cp[NK], bq[NK], ra[NK], rb[NK];
    // log(1 + z) and 1 / (1 + z) at the knots and at the interval midpoints zc + dx / 2: every integrand ordinate is one of them
    double lk[NK], lm[NK], ik[NK], im[NK];
    double pk[NK], pm[NK], ck[NK], cm[NK];      // 1 + z and its cube p (p p) at the knots and at the midpoints, as the ordinate forms them
    double h6[NK];            // dx / 6 (the same IEEE quotient the kernels used to form per interval: a division by 6 is not a multiplication, and it was a sixth of their instructions)
    int reach, reach_pad;     // knots after which the backward elimination has forgotten its start (CP_BG_REACH_LEFT of it left)
};
using Tables = TablesN<NK_DIST>;

struct Args {
    long long ncosmo, nz;
    Param p[CP_BG_NPARAMS];
    int second_is_omega_m;  // parameter 1 holds Omega_m (Omega_cdm = Omega_m - Omega_b, cosmology.py:1163-1165)
    const double* z;
    int z_shared;  // z has nz entries shared by all cosmologies, else ncosmo * nz
    double* out;
    int kind;
    const void* tab;  // TablesN<NK> of the kernel instantiation
    // massive neutrinos (cp_ncdm): spline tables (ncosmo, nsp, 4, CP_NCDM_NKNOTS), their knots (device), species selector
    const double* ncdm_tab;
    const double* ncdm_knots;
    int nsp, species;
};

// TIME: the integrand carries 1 / (1 + z) and the result is (T_last - spline(z)) / h / (Gyr per Mpc): DefaultBackground.time / age
// (cosmology.py:2000-2025), same RK4 == Simpson scan and natural spline as the distances, on the 400-knot grid.
#ifndef CP_BG_SORT_INTERVALS      // 1: the samples of a workgroup dealt to its threads in the order of their intervals (see bg_kernel)
#define CP_BG_SORT_INTERVALS 1
#endif
#ifndef CP_BG_LEAN_ORDINATE      // 0: E^2 term by term in the reference's units (cp_cosmo_common.h: inv_efunc_ln), for measurements
#define CP_BG_LEAN_ORDINATE 1
#endif

template <int NK, bool TIME, bool NCDM>   // NCDM = false: no massive species -- their table look-ups (one waterfall loop per ordinate) are not compiled in
__global__ __launch_bounds__(256) void bg_kernel(const Args A) {
    __shared__ TablesN<NK> T;
    {
        // (the thread's entries requested together, then stored: entry by entry the copy was a memory round trip per 256 doubles in front of the barrier)
        const double* src = reinterpret_cast<const double*>(A.tab);
        double* dst = reinterpret_cast<double*>(&T);
        constexpr int NT = (int)(sizeof(TablesN<NK>) / sizeof(double)), PER = (NT + 255) / 256;
        if (blockDim.x == 256) {
            double tmp[PER];
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int i = (int)threadIdx.x + 256 * j;
                tmp[j] = src[i < NT ? i : NT - 1];
            }
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int i = (int)threadIdx.x + 256 * j;
                if (i < NT) dst[i] = tmp[j];
            }
        } else {
            for (int i = threadIdx.x; i < NT; i += blockDim.x) dst[i] = src[i];
        }
    }
    __shared__ double ncdm_knots[CP_NCDM_NKNOTS];
    if (NCDM && A.nsp)
        for (int i = threadIdx.x; i < CP_NCDM_NKNOTS; i += blockDim.x) ncdm_knots[i] = A.ncdm_knots[i];
    __shared__ cpmath::MathTables mt;      // the table-driven exponential of the dark-energy term (two ordinates per interval)
    cpmath::fill_math_tables(&mt);
    __syncthreads();
    const long long nsamp = A.ncosmo * A.nz;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
#if CP_BG_SORT_INTERVALS
    // The quadrature walks the knots up to a sample's interval k (and a fixed number beyond): a wave is busy for the LARGEST k of its 64 samples.  The
    // workgroup's 256 samples are dealt to its threads in the order of their intervals (a counting sort through LDS: a histogram by atomics, its prefix sum,
    // the ranks), so that a wave holds samples of similar redshift -- for redshifts uniform in (0, 3) the four waves then walk 8, 16, 24 and 32 knots forward
    // instead of 32 each.  A sample is computed as before, by another lane; results land at the samples' own places.
    if (!TIME && blockDim.x == 256 && (A.kind == CP_BG_COMOVING_RADIAL || A.kind == CP_BG_ANGULAR_DIAMETER || A.kind == CP_BG_COMOVING_TRANSVERSE ||
                                       A.kind == CP_BG_LUMINOSITY)) {
        __shared__ int hist[128];
        __shared__ short order[256];
        int bin = 127;      // samples past the end of the batch, outside the knots or NaN: last (they leave the kernel at once)
        if (i < nsamp) {
            const double zn = A.z[A.z_shared ? i % A.nz : i];
            if (zn >= T.zc[0] && zn <= T.zc[NK - 1]) {
                int lo = 0, hi = NK - 1;
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (zn >= T.zc[mid]) lo = mid;
                    else hi = mid;
                }
                bin = lo < 126 ? lo : 126;
            }
        }
        if (threadIdx.x < 128) hist[threadIdx.x] = 0;
        __syncthreads();
        const int rank = atomicAdd(&hist[bin], 1);
        __syncthreads();
        if (threadIdx.x < 64) {      // exclusive prefix sum of the 128 counts by one wave: two bins per lane, a scan over the lanes
            const int c0 = hist[2 * threadIdx.x], c1 = hist[2 * threadIdx.x + 1];
            int sum = c0 + c1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(sum, off);
                if ((int)threadIdx.x >= off) sum += up;
            }
            hist[2 * threadIdx.x] = sum - c0 - c1;
            hist[2 * threadIdx.x + 1] = sum - c1;
        }
        __syncthreads();
        order[hist[bin] + rank] = (short)threadIdx.x;
        __syncthreads();
        i = (long long)blockIdx.x * blockDim.x + order[threadIdx.x];
    }
#endif
    if (i >= nsamp) return;
    const long long ic = i / A.nz, iz = i - ic * A.nz;
    const Cosmo c = load_cosmo(A.p, ic, A.second_is_omega_m, A.ncdm_tab, ncdm_knots, NCDM ? A.nsp : 0);
    const double z = A.z[A.z_shared ? iz : i];
    const double nan = __builtin_nan("");
    // ordinates sit on the fixed grid, where log(1 + z) is tabulated: the dark-energy term of E(z) is then ONE exp() instead of pow() x exp()
    // and 1 / E(z) comes from rsqrt(E^2): no division, no sqrt per ordinate
    const GridCosmo gc = grid_cosmo(c);
    auto integrand = [&](double zz, double lzp1, double izp1) {
        const double ie = CP_BG_LEAN_ORDINATE ? inv_efunc_grid(gc, c, zz, lzp1, izp1, &mt) : inv_efunc_ln(c, zz, lzp1, izp1, &mt);
        return TIME ? (kCkms / 100.) * izp1 * ie : (kCkms / 100.) * ie;
    };
    if (!TIME && (A.kind == CP_BG_EFUNC || A.kind == CP_BG_HUBBLE)) {
        const double e = efunc(c, z);
        A.out[i] = A.kind == CP_BG_EFUNC ? e : e * (c.h * 100.);
        return;
    }
    if (!TIME && A.kind == CP_BG_GROWTH_CPT) {  // eisenstein_hu.py:134-136
        A.out[i] = growth_cpt(c, z);
        return;
    }
    if (!TIME && (A.kind == CP_BG_RHO_CRIT || A.kind == CP_BG_OMEGA_M_Z || A.kind == CP_BG_OMEGA_DE_Z)) {  // cosmology.py:1738-1749, 1796, 1850
        const double zp1 = 1. + z;
        const double rc = rho_crit(c, zp1);
        A.out[i] = A.kind == CP_BG_RHO_CRIT ? rc
                 : (A.kind == CP_BG_OMEGA_M_Z ? rho_m(c, zp1) / rc : rho_de(c, zp1) / rc);
        return;
    }
    if (!TIME && (A.kind & ~CP_BG_AS_FRACTION) >= CP_BG_RHO_G) {  // BaseBackground.rho_x / Omega_x, cosmology.py:1680-1736, 1774-1853
        const double zp1 = 1. + z;
        const int what = A.kind & ~CP_BG_AS_FRACTION;
        const double g = c.Omega_g * zp1 * kRhoCrit, ur = c.Omega_ur * zp1 * kRhoCrit;
        const double b = c.Omega_b * 1. * kRhoCrit, cdm = c.Omega_cdm * 1. * kRhoCrit;
        double v = nan;
        switch (what) {
            case CP_BG_RHO_G: v = g; break;
            case CP_BG_RHO_B: v = b; break;
            case CP_BG_RHO_UR: v = ur; break;
            case CP_BG_RHO_CDM: v = cdm; break;
            case CP_BG_RHO_K: v = c.Omega_k / zp1 * kRhoCrit; break;
            case CP_BG_RHO_LAMBDA: v = c.Omega_de / (zp1 * zp1 * zp1) * kRhoCrit; break;
            case CP_BG_RHO_FLD: v = c.Omega_de * pow(zp1, 3. * (1 + c.w0 + c.wa)) * exp(3. * c.wa * (1. / zp1 - 1)) * kRhoCrit / (zp1 * zp1 * zp1); break;
            case CP_BG_RHO_DE: v = rho_de(c, zp1); break;
            case CP_BG_RHO_TOT: v = (cdm + b + ncdm_eval(c, z, 0)) + (g + ur) + rho_de(c, zp1); break;
            case CP_BG_RHO_M: v = cdm + b + ncdm_eval(c, z, 0) - 3. * ncdm_eval(c, z, 1); break;
            case CP_BG_RHO_R: v = g + ur + 3. * ncdm_eval(c, z, 1); break;
            case CP_BG_T_CMB_Z: v = c.T_cmb * zp1; break;
            case CP_BG_RHO_NCDM: v = ncdm_eval(c, z, 0, A.species); break;
            case CP_BG_P_NCDM: v = ncdm_eval(c, z, 1, A.species); break;
        }
        if (A.kind & CP_BG_AS_FRACTION) v = v / rho_crit(c, zp1);
        A.out[i] = v;
        return;
    }
    if (!TIME && A.kind == CP_BG_GROWTH_RATE) {  // Omega_m(z)^(0.55 + 0.05 (1 + w(z=1))), eisenstein_hu.py:151-152
        const double zp1 = 1. + z;
        const double Om = rho_m(c, zp1) / rho_crit(c, zp1);
        const double wz1 = c.w0 + (1. - 0.5) * c.wa;
        A.out[i] = pow(Om, 0.55 + 0.05 * (1 + wz1));
        return;
    }
    if (!(z >= T.zc[0] && z <= T.zc[NK - 1]) && !(TIME && A.kind == CP_BG_AGE)) {  // NaN outside the interpolation range (jax.py:200), also for NaN input
        A.out[i] = nan;
        return;
    }
    // interval k: zc[k] <= z < zc[k+1] (last interval closed), binary search
    int lo = 0, hi = NK - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (z >= T.zc[mid]) lo = mid;
        else hi = mid;
    }
    const int k = lo;
    const double f_first = integrand(T.zc[0], T.lk[0], T.ik[0]);
    double total = 0.;          // TIME: T_last = sum of all increments
    double fprev = f_first;     // integrand at the shared end of the previous interval
    double tk = 0.;             // T_k = sum of inc_i, i < k, accumulated in knot order like the reference's scan
    double inc_prev = 0.;       // inc of the previously processed interval (idx - 1 going up, idx + 1 going down)
    double inc_k = 0.;          // inc of interval k (needed by both eliminations)
    double dp = 0., dq = 0.;    // running right-hand sides of the forward / backward eliminations
    if (!TIME) {
        // what the wave knows of its samples (cp_cosmo_common.h: inv_efunc_grid_wave): scalar branches instead of per-lane ones
        const bool wave_fld = __any(!gc.lambda), wave_lambda = __any(gc.lambda);
        const bool wave_safe = CP_BG_LEAN_ORDINATE && !NCDM && __all(gc.Om >= 0. && gc.Or >= 0. && gc.Ode >= 0. && gc.Ok >= 0.);
        auto sweeps = [&](auto safe_tag) {
        constexpr bool SAFE = decltype(safe_tag)::value;
        auto knot = [&](int j) {      // the ordinate at knot j / at the midpoint of interval j: every argument from the grid's tables
            if (!CP_BG_LEAN_ORDINATE) return (kCkms / 100.) * inv_efunc_ln(c, T.zc[j], T.lk[j], T.ik[j], &mt);
            return (kCkms / 100.) * inv_efunc_grid_wave<SAFE>(gc, c, NCDM ? T.zc[j] : 0., T.lk[j], T.ik[j], T.pk[j], T.ck[j], &mt, wave_fld, wave_lambda);
        };
        auto middle = [&](int j) {
            if (!CP_BG_LEAN_ORDINATE) return (kCkms / 100.) * inv_efunc_ln(c, T.zc[j] + T.dx[j] / 2, T.lm[j], T.im[j], &mt);
            return (kCkms / 100.) * inv_efunc_grid_wave<SAFE>(gc, c, NCDM ? T.zc[j] + T.dx[j] / 2 : 0., T.lm[j], T.im[j], T.pm[j], T.cm[j], &mt, wave_fld, wave_lambda);
        };
        fprev = knot(0);
        // The intervals 0 .. k in knot order (the integral up to the sample and the forward elimination, exact), then the intervals above the
        // sample from `reach` knots above it downwards: the backward elimination forgets where it started at the rate of its multipliers
        // u_j / pivot_j (T.reach: after that many knots CP_BG_REACH_LEFT = 1e-13 of it is left, build_pivots), so the grid above -- it runs to z = 9999, a sample at
        // z < 3 sits in interval 31 at most -- contributes nothing a double can hold.  65 intervals instead of 118 for config 5, and within a
        // pass all lanes that are still busy walk the same interval: the table reads are broadcasts.
        for (int idx = 0; idx <= k; ++idx) {
            const double fm = middle(idx);
            const double fe = knot(idx + 1);
            const double inc = T.h6[idx] * (fprev + 2 * fm + 2 * fm + fe);  // h / 6 (...), jax.py:709 with k2 == k3
            fprev = fe;
            const double d = T.ra[idx] * inc_prev + T.rb[idx] * inc;      // knot idx: d = ra * inc_{idx-1} + rb * inc_idx
            dp = (d - T.l[idx] * dp) * T.idf[idx];
            if (idx < k) tk = tk + inc;
            inc_k = inc;
            inc_prev = inc;
        }
        const int top = k + 1 + T.reach < NK - 1 ? k + 1 + T.reach : NK - 1;      // intervals k + 1 .. top - 1, visited top down
        fprev = knot(top);
        inc_prev = 0.;
        for (int idx = top - 1; idx > k; --idx) {
            const double fm = middle(idx);
            const double fe = knot(idx);
            const double inc = T.h6[idx] * (fe + 2 * fm + 2 * fm + fprev);
            fprev = fe;
            const double d = T.ra[idx + 1] * inc + T.rb[idx + 1] * inc_prev;      // knot idx + 1: d = ra * inc_idx + rb * inc_{idx+1}
            dq = (d - T.u[idx + 1] * dq) * T.idb[idx + 1];
            inc_prev = inc;
        }
        // The reference splines the WHOLE table: an ordinate that is not finite anywhere on the grid (E^2 <= 0 at some redshift: closed models with
        // little matter, a negative dark-energy density) makes every distance of the cosmology NaN, also those far below it.  The ordinates above
        // `top` are not looked at otherwise; E^2 is a sum of positive terms unless one of its density parameters is negative, so only those
        // cosmologies walk them (no lane of config 5 does).
        if (top < NK - 1 && (c.Omega_k < 0. || c.Omega_de < 0. || c.Omega_cdm + c.Omega_b < 0.)) {
            bool bad = false;
            for (int idx = top; idx < NK - 1; ++idx) {
                const double fm = middle(idx);
                const double fe = knot(idx + 1);
                bad |= !(fabs(fm) <= 1.7976931348623157e308) || !(fabs(fe) <= 1.7976931348623157e308);
            }
            if (bad) dq = nan;
        }
        };
        if (wave_safe) sweeps(std::true_type{});
        else sweeps(std::false_type{});
    } else {
    const double f_last = integrand(T.zc[NK - 1], T.lk[NK - 1], T.ik[NK - 1]);
    for (int it = 0; it < NK - 1; ++it) {
        const bool fwd = it <= k;
        const int idx = fwd ? it : (NK - 1) - it + k;  // k+1 .. 117 visited top down
        if (it == k + 1) {                              // direction switch: restart from the last knot
            fprev = f_last;
            inc_prev = 0.;
        }
        const double x0 = T.zc[idx], x1 = T.zc[idx + 1], h = T.dx[idx];
        const double fm = integrand(x0 + h / 2, T.lm[idx], T.im[idx]);
        const double fe = integrand(fwd ? x1 : x0, fwd ? T.lk[idx + 1] : T.lk[idx], fwd ? T.ik[idx + 1] : T.ik[idx]);
        const double k1 = fwd ? fprev : fe, k4 = fwd ? fe : fprev;
        const double inc = T.h6[idx] * (k1 + 2 * fm + 2 * fm + k4);  // h / 6 (...), jax.py:709 with k2 == k3
        fprev = fe;
        if (TIME) total = total + inc;
        if (fwd) {
            // knot idx: d = ra * inc_{idx-1} + rb * inc_idx
            const double d = T.ra[idx] * inc_prev + T.rb[idx] * inc;
            dp = (d - T.l[idx] * dp) * T.idf[idx];
            if (idx < k) tk = tk + inc;
            if (idx == k) inc_k = inc;
        } else {
            // knot idx + 1: d = ra * inc_idx + rb * inc_{idx+1}
            const double d = T.ra[idx + 1] * inc + T.rb[idx + 1] * inc_prev;
            dq = (d - T.u[idx + 1] * dq) * T.idb[idx + 1];
        }
        inc_prev = inc;
    }
    }
    {   // knot k + 1 closes the backward elimination: intervals k (from the forward sweep) and k + 1
        const double inc_up = (k == NK - 2) ? 0. : inc_prev;  // inc_{k+1}: the last interval visited going down (none above the last interval)
        const double d = T.ra[k + 1] * inc_k + T.rb[k + 1] * inc_up;
        dq = (d - T.u[k + 1] * dq) * T.idb[k + 1];
    }
    // s_k + cp_k s_{k+1} = dp ;  s_{k+1} + bq_{k+1} s_k = dq
    const double cpk = T.cp[k], bqk = T.bq[k + 1];
    const double sk = (dp - cpk * dq) / (1. - cpk * bqk);
    const double sk1 = dq - bqk * sk;
    const double hk = T.dx[k];
    const double slope = inc_k / hk;
    const double tt = (sk + sk1 - 2. * slope) / hk;
    const double c3 = tt / hk, c2 = (slope - sk) / hk - tt;
    const double dz = z - T.zc[k];
    double chi = tk + dz * (sk + dz * (c2 + dz * c3));
    if (TIME) {
        constexpr double kGigayearOverMegaparsec = 3.06601394e2;  // cosmoprimo/constants.py:21
        const double age = total / c.h / kGigayearOverMegaparsec;               // (tmp[-1] - tmp[0]) / h / ..., cosmology.py:2024
        A.out[i] = A.kind == CP_BG_AGE ? age : (total - chi) / c.h / kGigayearOverMegaparsec;   // cosmology.py:2011
        return;
    }
    if (A.kind != CP_BG_COMOVING_RADIAL) {
        const double K = -(100. * 100.) / (kCkms * kCkms) * c.Omega_k;  // (h/Mpc)^2, cosmology.py:397
        if (K > 0.) chi = sin(sqrt(K) * chi) / sqrt(K);
        else if (K < 0.) chi = sinh(sqrt(-K) * chi) / sqrt(-K);
        const double da = chi / (1. + z);  // cosmology.py:1868
        chi = A.kind == CP_BG_ANGULAR_DIAMETER ? da : (A.kind == CP_BG_COMOVING_TRANSVERSE ? da * (1. + z) : da * ((1. + z) * (1. + z)));
    }
    A.out[i] = chi;
}

// What is left of the backward elimination's arbitrary start when it reaches the sample's interval: a relative error of the spline's first derivative
// there, i.e. of the cubic's correction to the Simpson sum inside ONE interval -- 1e-13 of that correction is below 1e-14 of the distance (the 1e-18 of
// the earlier rounds walked 32 knots above the sample where 24 do: a seventh of the kernel's ordinates)
#ifndef CP_BG_REACH_LEFT
#define CP_BG_REACH_LEFT 1e-13
#endif
template <int NK>
void build_pivots(TablesN<NK>& t) {
    const int n = NK;
    for (int i = 0; i < n - 1; ++i) t.dx[i] = t.zc[i + 1] - t.zc[i];
    t.dx[n - 1] = 0.;
    for (int i = 0; i < n; ++i) t.h6[i] = t.dx[i] / 6.;
    for (int i = 0; i < n; ++i) {
        t.lk[i] = std::log1p(t.zc[i]);
        t.lm[i] = std::log1p(t.zc[i] + t.dx[i] / 2);
        t.ik[i] = 1. / (1. + t.zc[i]);
        t.im[i] = 1. / (1. + (t.zc[i] + t.dx[i] / 2));
        t.pk[i] = 1. + t.zc[i];
        t.pm[i] = 1. + (t.zc[i] + t.dx[i] / 2);
        t.ck[i] = t.pk[i] * (t.pk[i] * t.pk[i]);
        t.cm[i] = t.pm[i] * (t.pm[i] * t.pm[i]);
    }
    std::vector<double> b(n);
    t.l[0] = 0.; b[0] = 2. * t.dx[0]; t.u[0] = t.dx[0]; t.ra[0] = 0.; t.rb[0] = 3.;
    for (int i = 1; i < n - 1; ++i) {
        t.l[i] = t.dx[i];
        b[i] = 2. * (t.dx[i - 1] + t.dx[i]);
        t.u[i] = t.dx[i - 1];
        t.ra[i] = 3. * t.dx[i] / t.dx[i - 1];
        t.rb[i] = 3. * t.dx[i - 1] / t.dx[i];
    }
    t.l[n - 1] = t.dx[n - 2]; b[n - 1] = 2. * t.dx[n - 2]; t.u[n - 1] = 0.; t.ra[n - 1] = 3.; t.rb[n - 1] = 0.;
    t.idf[0] = 1. / b[0];
    t.cp[0] = t.u[0] * t.idf[0];
    for (int i = 1; i < n; ++i) {
        t.idf[i] = 1. / (b[i] - t.l[i] * t.cp[i - 1]);
        t.cp[i] = t.u[i] * t.idf[i];
    }
    t.idb[n - 1] = 1. / b[n - 1];
    t.bq[n - 1] = t.l[n - 1] * t.idb[n - 1];
    for (int i = n - 2; i >= 0; --i) {
        t.idb[i] = 1. / (b[i] - t.u[i] * t.bq[i + 1]);
        t.bq[i] = t.l[i] * t.idb[i];
    }
    // dq_j = (d_j - u_j dq_{j+1}) / pivot_j: what a start at knot j + r leaves at knot j is the product of the r multipliers u / pivot in between
    t.reach_pad = 0;
    for (t.reach = 8; t.reach < n; t.reach += 4) {
        double worst = 0.;
        for (int j = 1; j + t.reach < n; ++j) {
            double f = 1.;
            for (int r = 0; r < t.reach; ++r) f *= std::fabs(t.u[j + r] * t.idb[j + r]);
            worst = f > worst ? f : worst;
        }
        if (worst < CP_BG_REACH_LEFT) break;
    }
}

void build_tables(Tables& t) {
    // knots: concatenate(linspace(0, 0.3, 20)[:-1], 1 / geomspace(1e-4, 1/1.3, 100)[::-1] - 1), cosmology.py:1947-1949
    const double zm = 0.3;
    for (int i = 0; i < 19; ++i) t.zc[i] = 0. + i * ((zm - 0.) / 19.);  // numpy.linspace: start + i * step
    const double la = std::log10(1e-4), lb = std::log10(1. / (1. + zm));
    double g[100];
    for (int i = 0; i < 100; ++i) g[i] = std::pow(10., la + i * ((lb - la) / 99.));  // numpy.geomspace via logspace
    g[0] = 1e-4;
    g[99] = 1. / (1. + zm);  // geomspace pins its end points
    for (int i = 0; i < 100; ++i) t.zc[19 + i] = 1. / g[99 - i] - 1.;
    build_pivots(t);
}

void build_tables(TablesN<NK_TIME>& t) {
    // knots: 1 / logspace(-8, 0, 400)[::-1] - 1, cosmology.py:1945-1946
    double g[NK_TIME];
    for (int i = 0; i < NK_TIME; ++i) g[i] = std::pow(10., -8. + i * ((0. - -8.) / (NK_TIME - 1)));  // numpy.logspace: 10 ** linspace
    g[NK_TIME - 1] = std::pow(10., 0.);  // linspace pins its last sample
    for (int i = 0; i < NK_TIME; ++i) t.zc[i] = 1. / g[NK_TIME - 1 - i] - 1.;
    build_pivots(t);
}

// one device copy of the grid tables per device and grid: created by cp_background_init(device), or -- for callers that did not ask -- by the first
// entry point that needs it (a hipMalloc and a synchronous upload inside that one call; the mutex keeps two host threads from both doing it)
std::mutex& table_mutex() {
    static std::mutex m;
    return m;
}

template <int NK>
TablesN<NK>* device_tables(int device) {
    static std::atomic<TablesN<NK>*> cache[64];
    if (device < 0 || device >= 64) return nullptr;
    TablesN<NK>* d = cache[device].load(std::memory_order_acquire);
    if (d) return d;
    std::lock_guard<std::mutex> lock(table_mutex());
    d = cache[device].load(std::memory_order_acquire);
    if (!d) {
        std::vector<TablesN<NK>> h(1);
        build_tables(h[0]);
        if (hipMalloc(&d, sizeof(TablesN<NK>)) != hipSuccess) return nullptr;
        if (hipMemcpy(d, h.data(), sizeof(TablesN<NK>), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d);
            return nullptr;
        }
        cache[device].store(d, std::memory_order_release);
    }
    return d;
}

// ---- sound horizon: the reference's fixed-depth Romberg rule (jax.py:519-660, divmax = 15: 2^15 + 1 integrand evaluations) ----
// One wave per (cosmology, z) sample: the lanes stride the new ordinates of every refinement level, a butterfly reduction adds them.
__global__ __launch_bounds__(64) void rs_kernel(const Args A) {
    __shared__ double ncdm_knots[CP_NCDM_NKNOTS];
    if (A.nsp)
        for (int i = threadIdx.x; i < CP_NCDM_NKNOTS; i += blockDim.x) ncdm_knots[i] = A.ncdm_knots[i];
    __syncthreads();
    const long long i = blockIdx.x;
    const long long ic = i / A.nz, iz = i - ic * A.nz;
    const Cosmo c = load_cosmo(A.p, ic, A.second_is_omega_m, A.ncdm_tab, ncdm_knots, A.nsp);
    const double z = A.z[A.z_shared ? iz : i];
    const bool cosmomc = A.kind == CP_BG_RS_COSMOMC;
    const double omega_b = c.Omega_b * (c.h * c.h);
    auto f = [&](double a) {  // dsoundda (cosmology.py:1920-1927) / dsoundda_approx (:215-222)
        const double dtauda = 1. / (a * a * (efunc(c, 1 / a - 1.) * (c.h * 100.)) / kCkms);
        const double R = cosmomc ? 3e4 * a * omega_b : 3 / 4. * a * c.Omega_b / c.Omega_g;
        return dtauda * pow(3 * (1 + R), -0.5);
    };
    const double lo = 1e-8, hi = 1. / (1 + z);
    const double intrange = hi - lo;
    double ordsum = 0.5 * (f(lo) + f(hi));
    constexpr int DIVMAX = 15;
    double row[DIVMAX + 1];
    row[0] = intrange * ordsum;
    double err = 0.;
    int n = 1;
    for (int lev = 1; lev <= DIVMAX; ++lev) {
        n *= 2;
        const int numtosum = n / 2;
        const double h = (hi - lo) * 1. / numtosum;
        const double lox = lo + 0.5 * h;
        double s = 0.;
        for (int j = threadIdx.x; j < numtosum; j += 64) s += f(lox + h * j);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        ordsum += s;
        double x = intrange * ordsum / n;
        double prev = row[0];  // the new row, left to right: x_{k+1} = (4^(k+1) x_k - y_k) / (4^(k+1) - 1)
        row[0] = x;
        double p4 = 1.;
        for (int k = 0; k < lev; ++k) {
            p4 *= 4.;
            x = (p4 * x - prev) / (p4 - 1.0);
            if (k == lev - 1) err = fabs(prev - x);  // |last_row[lev - 1] - row[lev]| (jax.py:648)
            prev = row[k + 1];
            row[k + 1] = x;
        }
    }
    // precision not achieved (epsabs = epsrel = 1e-7): the reference raises; NaN here, turned into the exception by the caller
    const double res = (err < 1e-7 && err < fabs(row[DIVMAX]) * 1e-7) ? row[DIVMAX] : __builtin_nan("");
    if (threadIdx.x == 0) A.out[i] = cosmomc ? res : res * c.h;
}

// ---- linear growth ODE (DefaultBackground.growth_factor / growth_rate, cosmology.py:2044-2093) -------------------------------------------
struct GrowthArgs {
    long long ncosmo;
    Param p[CP_BG_NPARAMS];
    int second_is_omega_m, mass;
    const double* ncdm_tab;
    const double* ncdm_knots;
    int nsp;
    double* tab;  // (ncosmo, 2, CP_GROWTH_NKNOTS)
};

__device__ __forceinline__ void growth_rhs(const Cosmo& c, int mass, double D, double Dp, double eta, double& dD, double& dDp) {
    const double z = exp(-eta) - 1.;
    const double zp1 = 1. + z;
    const double rc = rho_crit(c, zp1);
    const double Ok = c.Omega_k / zp1 * kRhoCrit / rc;
    const double Or = (c.Omega_g * zp1 * kRhoCrit + c.Omega_ur * zp1 * kRhoCrit + 3. * ncdm_eval(c, z, 1)) / rc;
    const double Ode = rho_de(c, zp1) / rc;
    const double Om = mass == 0 ? rho_m(c, zp1) / rc : c.Omega_cdm * 1. * kRhoCrit / rc + c.Omega_b * 1. * kRhoCrit / rc;
    const double w_fld = c.w0 + z / (1. + z) * c.wa;
    const double f1 = -1. - (-1. / 2. * (1. - Ok + Or + 3 * w_fld * Ode));
    const double f2 = 3. / 2. * Om;
    dD = Dp;
    dDp = f2 * D + f1 * Dp;
}

// one thread per cosmology: 200 RK4 steps (jax.py:700-710), results stored in ascending z
__global__ __launch_bounds__(64) void growth_ode_kernel(const GrowthArgs A) {
    __shared__ double ncdm_knots[CP_NCDM_NKNOTS];
    if (A.nsp)
        for (int i = threadIdx.x; i < CP_NCDM_NKNOTS; i += blockDim.x) ncdm_knots[i] = A.ncdm_knots[i];
    __syncthreads();
    const long long ic = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ic >= A.ncosmo) return;
    const Cosmo c = load_cosmo(A.p, ic, A.second_is_omega_m, A.ncdm_tab, ncdm_knots, A.nsp);
    constexpr int n = CP_GROWTH_NKNOTS;
    double* tD = A.tab + ic * 2 * n;
    double* tF = tD + n;
    const double eta0 = -6., step = (0. - -6.) / (n - 1);
    double D = exp(eta0), Dp = D;
    double t_last = eta0;
    for (int i = 0; i < n; ++i) {
        const double t = i == n - 1 ? 0. : eta0 + i * step;  // numpy.linspace
        const double h = t - t_last;
        double a1, b1, a2, b2, a3, b3, a4, b4;
        growth_rhs(c, A.mass, D, Dp, t_last, a1, b1);
        growth_rhs(c, A.mass, D + h * a1 / 2, Dp + h * b1 / 2, t_last + h / 2, a2, b2);
        growth_rhs(c, A.mass, D + h * a2 / 2, Dp + h * b2 / 2, t_last + h / 2, a3, b3);
        growth_rhs(c, A.mass, D + h * a3, Dp + h * b3, t, a4, b4);
        D = D + h / 6. * (a1 + 2 * a2 + 2 * a3 + a4);
        Dp = Dp + h / 6. * (b1 + 2 * b2 + 2 * b3 + b4);
        t_last = t;
        tD[n - 1 - i] = D;
        tF[n - 1 - i] = Dp / D;
    }
}

// ---- massive neutrinos --------------------------------------------------------------------------------------------------------
// knots: concatenate(linspace(0, 1, 20)[:-1], 1 / geomspace(1e-8, 1/2, 100)[::-1] - 1), cosmology.py:1941-1943
void build_ncdm_knots(double* zc) {
    const double zm = 1.;
    for (int i = 0; i < 19; ++i) zc[i] = 0. + i * ((zm - 0.) / 19.);
    const double la = std::log10(1e-8), lb = std::log10(1. / (1. + zm));
    double g[100];
    for (int i = 0; i < 100; ++i) g[i] = std::pow(10., la + i * ((lb - la) / 99.));
    g[0] = 1e-8;
    g[99] = 1. / (1. + zm);
    for (int i = 0; i < 100; ++i) zc[19 + i] = 1. / g[99 - i] - 1.;
}

double* device_ncdm_knots(int device) {
    static std::atomic<double*> cache[64];
    if (device < 0 || device >= 64) return nullptr;
    double* d = cache[device].load(std::memory_order_acquire);
    if (d) return d;
    std::lock_guard<std::mutex> lock(table_mutex());
    d = cache[device].load(std::memory_order_acquire);
    if (!d) {
        double h[CP_NCDM_NKNOTS];
        build_ncdm_knots(h);
        if (hipMalloc(&d, sizeof(h)) != hipSuccess) return nullptr;
        if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d);
            return nullptr;
        }
        cache[device].store(d, std::memory_order_release);
    }
    return d;
}

constexpr int NCDM_MAX_SPECIES = 8;
constexpr int NCDM_MAX_NQ = 128;

struct NcdmArgs {
    long long ncosmo;
    int nsp, nq;
    Param h, T_cmb, m[NCDM_MAX_SPECIES], T_over[NCDM_MAX_SPECIES];
    const double* knots;
    double* tab;
};
// the Gauss-Laguerre rule, nodes then weights (2 nq), travels by value in the kernel arguments (2 KB): no device buffer to allocate, fill
// and free around the launch, so the call is asynchronous and can be recorded into a HIP graph like every other execute
struct NcdmRule {
    double v[2 * NCDM_MAX_NQ];
};

// _compute_ncdm_momenta (cosmology.py:74-137, method 'laguerre') / (1 + z)^3 / h^2 (_get_ncdm, :441-442) on every knot:
// one thread per (cosmology, species, knot) computes the density and the pressure
__global__ __launch_bounds__(128) void ncdm_momenta_kernel(const NcdmArgs A, const NcdmRule R) {
    __shared__ double rule[2 * NCDM_MAX_NQ];
    for (int i = threadIdx.x; i < 2 * A.nq; i += blockDim.x) rule[i] = R.v[i];
    __syncthreads();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.ncosmo * A.nsp * CP_NCDM_NKNOTS) return;
    const int k = (int)(i % CP_NCDM_NKNOTS);
    const int s = (int)((i / CP_NCDM_NKNOTS) % A.nsp);
    const long long ic = i / ((long long)CP_NCDM_NKNOTS * A.nsp);
    const double h = A.h.ptr ? A.h.ptr[ic] : A.h.value;
    const double T_cmb = A.T_cmb.ptr ? A.T_cmb.ptr[ic] : A.T_cmb.value;
    const double m = A.m[s].ptr ? A.m[s].ptr[ic] : A.m[s].value;
    const double T_eff = T_cmb * (A.T_over[s].ptr ? A.T_over[s].ptr[ic] : A.T_over[s].value);
    constexpr double kEv = 1.602176634e-19, kBoltzmann = 1.380649e-23;  // scipy.constants (exact SI values)
    const double z = A.knots[k];
    const double a = 1. / (1. + z);
    const double over_T = kEv / (kBoltzmann * (T_eff / a));
    const double m2 = (m * over_T) * (m * over_T);
    double rho = 0., pr = 0.;
    for (int q = 0; q < A.nq; ++q) {
        const double t = rule[q], w = rule[A.nq + q];
        const double e = sqrt(t * t + m2), f = 1. + exp(-1. * t);
        rho += t * t * e / f * w;
        pr += 1. / 3. * (t * t) * (t * t) / e / f * w;
    }
    const double Ta = T_eff / a;
    const double pref = 7. / 8. * 4 / (kC * kC * kC) * kStefanBoltzmann * (Ta * Ta * Ta * Ta);
    const double norm = (7. * (kPi * kPi * kPi * kPi) / 120.) * (1e10 * kMsun);
    const double zp1 = 1. + z;
    const double scale = (kMpc * kMpc * kMpc) / (zp1 * zp1 * zp1) / (h * h);
    double* t0 = A.tab + ((ic * A.nsp + s) * 4) * (long long)CP_NCDM_NKNOTS;
    t0[k] = pref * rho / norm * scale;
    t0[2 * CP_NCDM_NKNOTS + k] = pref * pr / norm * scale;
}

// second derivatives of the natural cubic splines through the tabulated values: one thread per (cosmology, species, rho | p)
__global__ __launch_bounds__(64) void ncdm_spline_kernel(const NcdmArgs A) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.ncosmo * A.nsp * 2) return;
    double* y = A.tab + (i / 2 * 4 + (i & 1) * 2) * (long long)CP_NCDM_NKNOTS;
    double* m = y + CP_NCDM_NKNOTS;
    const double* x = A.knots;
    constexpr int n = CP_NCDM_NKNOTS;
    double cp[n], dp[n];  // Thomas algorithm on h_{i-1} m_{i-1} + 2 (h_{i-1} + h_i) m_i + h_i m_{i+1} = 6 (slope_i - slope_{i-1})
    cp[0] = 0.;
    dp[0] = 0.;
    for (int j = 1; j < n - 1; ++j) {
        const double h0 = x[j] - x[j - 1], h1 = x[j + 1] - x[j];
        const double rhs = 6. * ((y[j + 1] - y[j]) / h1 - (y[j] - y[j - 1]) / h0);
        const double den = 2. * (h0 + h1) - h0 * cp[j - 1];
        cp[j] = h1 / den;
        dp[j] = (rhs - h0 * dp[j - 1]) / den;
    }
    m[n - 1] = 0.;
    for (int j = n - 2; j >= 1; --j) m[j] = dp[j] - cp[j] * m[j + 1];
    m[0] = 0.;
}

}  // namespace

const double* cpcosmo::ncdm_knots_device(int device) { return device_ncdm_knots(device); }

extern "C" int cp_background_init(int device) {
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_background_init: cannot select device %d", device);
    const bool ok = device_tables<NK_DIST>(device) && device_tables<NK_TIME>(device) && device_ncdm_knots(device);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (!ok) return cp::fail(CP_ENOMEM, "cp_background_init: cannot allocate the knot tables on device %d", device);
    return CP_OK;
}

extern "C" int cp_growth_ode_knots(double* zc_out, int n) {
    if (!zc_out || n != CP_GROWTH_NKNOTS) return cp::fail(CP_EINVAL, "cp_growth_ode_knots: need a buffer of %d doubles", CP_GROWTH_NKNOTS);
    for (int i = 0; i < n; ++i) {  // z = exp(-eta) - 1, eta = linspace(-6, 0, 201), reversed to ascending z
        const int j = n - 1 - i;
        const double eta = j == n - 1 ? 0. : -6. + j * ((0. - -6.) / (n - 1));
        zc_out[i] = std::exp(-eta) - 1.;
    }
    return CP_OK;
}

extern "C" int cp_growth_ode_tables(long long ncosmo, const cp_param* params, int second_is_omega_m, const cp_ncdm* ncdm, int mass, double* d_tab,
                                    int device, void* stream) {
    if (ncosmo < 0) return cp::fail(CP_EINVAL, "cp_growth_ode_tables: negative size");
    if (ncosmo == 0) return CP_OK;
    if (!params || !d_tab) return cp::fail(CP_EINVAL, "cp_growth_ode_tables: null pointer");
    if (mass != 0 && mass != 1) return cp::fail(CP_EINVAL, "cp_growth_ode_tables: mass must be 0 ('m') or 1 ('cb')");
    const int nsp = ncdm ? ncdm->nspecies : 0;
    if (nsp < 0 || (nsp > 0 && !ncdm->tab)) return cp::fail(CP_EINVAL, "cp_growth_ode_tables: bad massive-neutrino tables");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_growth_ode_tables: cannot select device %d", device);
    GrowthArgs A;
    A.ncosmo = ncosmo;
    for (int k = 0; k < CP_BG_NPARAMS; ++k) {
        A.p[k].ptr = params[k].ptr;
        A.p[k].value = params[k].value;
    }
    A.second_is_omega_m = second_is_omega_m;
    A.mass = mass;
    A.nsp = nsp;
    A.ncdm_tab = nsp ? ncdm->tab : nullptr;
    A.ncdm_knots = nsp ? device_ncdm_knots(device) : nullptr;
    A.tab = d_tab;
    int rc = CP_OK;
    if (nsp && !A.ncdm_knots) {
        rc = cp::fail(CP_ENOMEM, "cp_growth_ode_tables: cannot allocate the massive-neutrino knots on device %d", device);
    } else {
        hipLaunchKernelGGL(growth_ode_kernel, dim3((unsigned)((ncosmo + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream), A);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = cp::fail(CP_EDEVICE, "cp_growth_ode_tables: launch failed: %s", hipGetErrorString(e));
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

extern "C" int cp_ncdm_knots(double* zc_out, int n) {
    if (!zc_out || n != CP_NCDM_NKNOTS) return cp::fail(CP_EINVAL, "cp_ncdm_knots: need a buffer of %d doubles", CP_NCDM_NKNOTS);
    build_ncdm_knots(zc_out);
    return CP_OK;
}

extern "C" int cp_ncdm_tables(long long ncosmo, int nspecies, cp_param h, cp_param T_cmb, const cp_param* m_ncdm, const cp_param* T_ncdm_over_cmb,
                              int nq, const double* nodes, const double* weights, double* d_tab, int device, void* stream) {
    if (ncosmo < 0 || nspecies < 0) return cp::fail(CP_EINVAL, "cp_ncdm_tables: negative size");
    if (ncosmo == 0 || nspecies == 0) return CP_OK;
    if (nspecies > NCDM_MAX_SPECIES) return cp::fail(CP_EUNSUPPORTED, "cp_ncdm_tables: at most %d massive species", NCDM_MAX_SPECIES);
    if (nq < 1 || nq > NCDM_MAX_NQ) return cp::fail(CP_EINVAL, "cp_ncdm_tables: quadrature size must be in [1, %d]", NCDM_MAX_NQ);
    if (!m_ncdm || !T_ncdm_over_cmb || !nodes || !weights || !d_tab) return cp::fail(CP_EINVAL, "cp_ncdm_tables: null pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_ncdm_tables: cannot select device %d", device);
    hipStream_t st = static_cast<hipStream_t>(stream);
    NcdmArgs A;
    A.ncosmo = ncosmo;
    A.nsp = nspecies;
    A.nq = nq;
    A.h.ptr = h.ptr; A.h.value = h.value;
    A.T_cmb.ptr = T_cmb.ptr; A.T_cmb.value = T_cmb.value;
    for (int s = 0; s < nspecies; ++s) {
        A.m[s].ptr = m_ncdm[s].ptr; A.m[s].value = m_ncdm[s].value;
        A.T_over[s].ptr = T_ncdm_over_cmb[s].ptr; A.T_over[s].value = T_ncdm_over_cmb[s].value;
    }
    A.knots = device_ncdm_knots(device);
    A.tab = d_tab;
    int rc = CP_OK;
    if (!A.knots) {
        rc = cp::fail(CP_ENOMEM, "cp_ncdm_tables: cannot allocate the knots on device %d", device);
    } else {
        NcdmRule R;
        for (int q = 0; q < nq; ++q) R.v[q] = nodes[q], R.v[nq + q] = weights[q];
        for (int q = 2 * nq; q < 2 * NCDM_MAX_NQ; ++q) R.v[q] = 0.;
        const long long n1 = ncosmo * nspecies * CP_NCDM_NKNOTS, n2 = ncosmo * nspecies * 2;
        hipLaunchKernelGGL(ncdm_momenta_kernel, dim3((unsigned)((n1 + 127) / 128)), dim3(128), 0, st, A, R);
        hipLaunchKernelGGL(ncdm_spline_kernel, dim3((unsigned)((n2 + 63) / 64)), dim3(64), 0, st, A);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) rc = cp::fail(CP_EDEVICE, "cp_ncdm_tables: launch failed: %s", hipGetErrorString(e));
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

extern "C" int cp_background_knots(double* zc_out, int n) {
    if (zc_out && n == NK_TIME) {  // the 400 knots of time / age
        std::vector<TablesN<NK_TIME>> t(1);
        build_tables(t[0]);
        for (int i = 0; i < NK_TIME; ++i) zc_out[i] = t[0].zc[i];
        return CP_OK;
    }
    if (!zc_out || n != NK_DIST) return cp::fail(CP_EINVAL, "cp_background_knots: need a buffer of %d (distances) or %d (time) doubles", NK_DIST, NK_TIME);
    Tables t;
    build_tables(t);
    for (int i = 0; i < NK_DIST; ++i) zc_out[i] = t.zc[i];
    return CP_OK;
}

// The derived distances of ONE cosmology from its radial distances (cosmology.py:1855-1912): the curvature map and the powers of 1 + z of bg_kernel's
// last lines, as one pass over a catalogue whose radial distances came from the cosmology's table (cp_spline_points) -- the five elementwise passes the
// same arithmetic costs as array operations are 2 / 3 of luminosity_distance(10^7 redshifts) otherwise.
namespace {
__global__ __launch_bounds__(256) void distance_from_radial_kernel(const double* __restrict__ chi_in, const double* __restrict__ z, long long n, double K, int kind,
                                                                   double* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        double chi = chi_in[i];
        const double zi = z[i];
        if (K > 0.) chi = sin(sqrt(K) * chi) / sqrt(K);
        else if (K < 0.) chi = sinh(sqrt(-K) * chi) / sqrt(-K);
        const double da = chi / (1. + zi);  // cosmology.py:1868
        out[i] = kind == CP_BG_ANGULAR_DIAMETER ? da : (kind == CP_BG_COMOVING_TRANSVERSE ? da * (1. + zi) : da * ((1. + zi) * (1. + zi)));
    }
}
}  // namespace

extern "C" int cp_distance_from_radial(const double* d_chi, const double* d_z, long long n, double K, int kind, double* d_out, int device, void* stream) {
    if (n < 0) return cp::fail(CP_EINVAL, "cp_distance_from_radial: negative size");
    if (kind != CP_BG_ANGULAR_DIAMETER && kind != CP_BG_COMOVING_TRANSVERSE && kind != CP_BG_LUMINOSITY)
        return cp::fail(CP_EINVAL, "cp_distance_from_radial: kind %d is not a derived distance", kind);
    if (n == 0) return CP_OK;
    if (!d_chi || !d_z || !d_out) return cp::fail(CP_EINVAL, "cp_distance_from_radial: null pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_distance_from_radial: cannot select device %d", device);
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(distance_from_radial_kernel, dim3((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), dim3(256), 0, static_cast<hipStream_t>(stream), d_chi, d_z,
                       n, K, kind, d_out);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_distance_from_radial: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_background_distance(long long ncosmo, long long nz, const cp_param* params, int second_is_omega_m, const double* d_z,
                                      int z_shared, double* d_out, int kind, int device, void* stream) {
    return cp_background_eval(ncosmo, nz, params, second_is_omega_m, nullptr, d_z, z_shared, d_out, kind, device, stream);
}

extern "C" int cp_background_eval(long long ncosmo, long long nz, const cp_param* params, int second_is_omega_m, const cp_ncdm* ncdm,
                                  const double* d_z, int z_shared, double* d_out, int kind, int device, void* stream) {
    if (ncosmo < 0 || nz < 0) return cp::fail(CP_EINVAL, "cp_background_distance: negative size");
    const int nsp = ncdm ? ncdm->nspecies : 0;
    if (nsp < 0 || (nsp > 0 && !ncdm->tab)) return cp::fail(CP_EINVAL, "cp_background_eval: bad massive-neutrino tables");
    if (nsp > 0 && (ncdm->species < -1 || ncdm->species >= nsp)) return cp::fail(CP_EINVAL, "cp_background_eval: species %d of %d", ncdm->species, nsp);
    if (ncosmo == 0 || nz == 0) return CP_OK;
    if (!params || !d_z || !d_out) return cp::fail(CP_EINVAL, "cp_background_distance: null pointer");
    {
        const int base = kind & ~CP_BG_AS_FRACTION;
        if (kind < 0 || base > CP_BG_KIND_LAST || ((kind & CP_BG_AS_FRACTION) && (base < CP_BG_RHO_G || (base >= CP_BG_T_CMB_Z && base <= CP_BG_AGE) || base >= CP_BG_RS)))
            return cp::fail(CP_EINVAL, "cp_background_distance: unknown kind %d", kind);
    }
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_background_distance: cannot select device %d", device);
    const bool is_time = kind == CP_BG_TIME || kind == CP_BG_AGE;
    const void* tab = is_time ? static_cast<const void*>(device_tables<NK_TIME>(device)) : static_cast<const void*>(device_tables<NK_DIST>(device));
    if (!tab) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return cp::fail(CP_ENOMEM, "cp_background_distance: cannot allocate the knot tables on device %d", device);
    }
    Args A;
    A.ncosmo = ncosmo;
    A.nz = nz;
    for (int k = 0; k < CP_BG_NPARAMS; ++k) {
        A.p[k].ptr = params[k].ptr;
        A.p[k].value = params[k].value;
    }
    A.second_is_omega_m = second_is_omega_m;
    A.z = d_z;
    A.z_shared = z_shared;
    A.out = d_out;
    A.kind = kind;
    A.tab = tab;
    A.nsp = nsp;
    A.species = nsp ? ncdm->species : -1;
    A.ncdm_tab = nsp ? ncdm->tab : nullptr;
    A.ncdm_knots = nsp ? device_ncdm_knots(device) : nullptr;
    if (nsp && !A.ncdm_knots) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return cp::fail(CP_ENOMEM, "cp_background_eval: cannot allocate the massive-neutrino knots on device %d", device);
    }
    const long long nsamp = ncosmo * nz;
    const int block = 256;
    const long long grid = (nsamp + block - 1) / block;
    if (kind == CP_BG_RS || kind == CP_BG_RS_COSMOMC) hipLaunchKernelGGL(rs_kernel, dim3((unsigned)nsamp), dim3(64), 0, static_cast<hipStream_t>(stream), A);
    else if (is_time && A.nsp) hipLaunchKernelGGL((bg_kernel<NK_TIME, true, true>), dim3((unsigned)grid), dim3(block), 0, static_cast<hipStream_t>(stream), A);
    else if (is_time) hipLaunchKernelGGL((bg_kernel<NK_TIME, true, false>), dim3((unsigned)grid), dim3(block), 0, static_cast<hipStream_t>(stream), A);
    else if (A.nsp) hipLaunchKernelGGL((bg_kernel<NK_DIST, false, true>), dim3((unsigned)grid), dim3(block), 0, static_cast<hipStream_t>(stream), A);
    else hipLaunchKernelGGL((bg_kernel<NK_DIST, false, false>), dim3((unsigned)grid), dim3(block), 0, static_cast<hipStream_t>(stream), A);
    hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_background_distance: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

// ---- derived parameters of a batch of cosmologies (BaseCosmoParams.__getitem__ -> _get_derived, cosmology.py:331-415), one launch -------------------
// A batch kept on the device derived each of these with two to five framework kernels on (ncosmo,) arrays -- two dozen launches of 3-5 us per
// chunk of config 4, the cosmology object being made anew for every chunk -- where one lane per cosmology does all of it.
namespace {

struct DerivedArgs {
    long long ncosmo;
    Param p[CP_BG_NPARAMS];
    double* out;      // (CP_DERIVED_NVALUES, ncosmo): one row per derived value, so that each is a contiguous (ncosmo,) array
};

__global__ __launch_bounds__(256) void derived_parameters_kernel(const DerivedArgs A) {
    const long long ic = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (ic >= A.ncosmo) return;
    double v[CP_BG_NPARAMS];
#pragma unroll
    for (int k = 0; k < CP_BG_NPARAMS; ++k) v[k] = A.p[k].ptr ? A.p[k].ptr[ic] : A.p[k].value;
    const double h = v[CP_BG_H], Omega_cdm = v[CP_BG_OMEGA_CDM], Omega_b = v[CP_BG_OMEGA_B], Omega_k = v[CP_BG_OMEGA_K], T_cmb = v[CP_BG_T_CMB], N_ur = v[CP_BG_N_UR];
    // the operation order of the host formulas (cosmology.py:355-383 as restated in BaseCosmoParams._derive): batches and single cosmologies agree to rounding
    const double h2 = h * h;
    const double c3 = kC * kC * kC;
    const double Omega_g = (T_cmb * T_cmb) * (T_cmb * T_cmb) * 4. / c3 * kStefanBoltzmann / (h2 * kRhoCritKg);
    const double T_ur = T_cmb * 0.7137658555036082;      // (4 / 11)^(1 / 3)
    const double Omega_ur = N_ur * 7. / 8. * ((T_ur * T_ur) * (T_ur * T_ur)) * 4. / c3 * kStefanBoltzmann / (h2 * kRhoCritKg);
    const double Omega_r = Omega_g + Omega_ur;
    const double Omega_m = Omega_b + Omega_cdm;
    const double Omega_de = 1. - (Omega_cdm + Omega_b + Omega_g + Omega_ur + Omega_k);
    double* o = A.out + ic;
    const long long n = A.ncosmo;
    o[CP_DERIVED_H2 * n] = h2;
    o[CP_DERIVED_H0 * n] = h * 100;
    o[CP_DERIVED_OMEGA_G * n] = Omega_g;
    o[CP_DERIVED_T_UR * n] = T_ur;
    o[CP_DERIVED_OMEGA_UR * n] = Omega_ur;
    o[CP_DERIVED_OMEGA_R * n] = Omega_r;
    o[CP_DERIVED_OMEGA_M * n] = Omega_m;
    o[CP_DERIVED_OMEGA_DE * n] = Omega_de;
    o[CP_DERIVED_K * n] = -(100. * 100.) / (kCkms * kCkms) * Omega_k;
    o[CP_DERIVED_LITTLE_OMEGA_B * n] = Omega_b * h2;
    o[CP_DERIVED_LITTLE_OMEGA_CDM * n] = Omega_cdm * h2;
    o[CP_DERIVED_LITTLE_OMEGA_M * n] = Omega_m * h2;
    o[CP_DERIVED_LITTLE_OMEGA_G * n] = Omega_g * h2;
    o[CP_DERIVED_LITTLE_OMEGA_UR * n] = Omega_ur * h2;
    o[CP_DERIVED_LITTLE_OMEGA_R * n] = Omega_r * h2;
    o[CP_DERIVED_LITTLE_OMEGA_K * n] = Omega_k * h2;
    o[CP_DERIVED_LITTLE_OMEGA_DE * n] = Omega_de * h2;
}

}  // namespace

extern "C" int cp_derived_parameters(long long ncosmo, const cp_param* params, double* d_out, int device, void* stream) {
    if (ncosmo < 0) return cp::fail(CP_EINVAL, "cp_derived_parameters: negative size");
    if (ncosmo == 0) return CP_OK;
    if (!params || !d_out) return cp::fail(CP_EINVAL, "cp_derived_parameters: null pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_derived_parameters: cannot select device %d", device);
    DerivedArgs A;
    A.ncosmo = ncosmo;
    for (int k = 0; k < CP_BG_NPARAMS; ++k) {
        A.p[k].ptr = params[k].ptr;
        A.p[k].value = params[k].value;
    }
    A.out = d_out;
    hipLaunchKernelGGL(derived_parameters_kernel, dim3((unsigned)((ncosmo + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), A);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_derived_parameters: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
