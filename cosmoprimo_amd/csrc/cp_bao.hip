// cp_bao.hip -- the elementwise stages of the wallish2018 and brieden2022 filters over batches of spectra, each as ONE pass (they were chains of
// 5-14 torch elementwise / gather / concatenation kernels between the transforms and the spline operators: a quarter of the filters' GPU time).
//   cp_wallish_finish          bao_filter.py:421-431  pknow = spliced spline; wiggles = (pk / pknow - 1) tophat + 1; out = pk / wiggles
//   cp_brieden_ratio           bao_filter.py:493-499  pknow = P_nowiggle x growth x correction; ratio = P / pknow / ratio_fid
//   cp_brieden_knots           bao_filter.py:500-509 + interpolator.py:42-87 (_pad_log): envelope x pknow x ratio_now_fid -> log10, knot-major, with the
//                                                      two log-log extrapolated knots on either side, per cosmology
//   cp_brieden_finish          bao_filter.py:509      out = pk with 10^(re-sampled log10 P) written over the k_fid range
//   cp_wallish_dd_box          bao_filter.py:377-394  second derivatives of the clamped spline through a sequence of DST coefficients (tridiagonal
//                                                      solve in LDS, a wave per sequence) and the two arg-max searches that delimit the knots to remove
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <type_traits>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_internal.h"
#include "cp_math.h"
#include "cp_splice_uniform.h"
#include "cp_wallish_dd.h"

namespace {

struct DeviceScope {
    int prev = -1;
    bool ok = true;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceScope() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

__global__ __launch_bounds__(256) void wallish_finish_kernel(const double* __restrict__ pk, const double* __restrict__ a, const double* __restrict__ b,
                                                             const double* __restrict__ tophat, double* __restrict__ out, long long total, int n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const double p = pk[i];
        const double pknow = b ? a[i] + b[i] : a[i];
        const double wiggles = (p / pknow - 1.) * tophat[i % n] + 1.;
        out[i] = p / wiggles;
    }
}

__global__ __launch_bounds__(256) void brieden_ratio_kernel(const double* __restrict__ rows, const double* __restrict__ now, const double* __restrict__ g0,
                                                            const double* __restrict__ correction, const double* __restrict__ ratio_fid,
                                                            double* __restrict__ pknow, double* __restrict__ ratio, long long nb, int n) {
    const long long total = nb * n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long c = i / n;
        const int j = (int)(i - c * n);
        const double pn = now[i] * g0[c] * correction[j];
        pknow[i] = pn;
        ratio[i] = rows[i] / pn / ratio_fid[j];
    }
}

// knot-major (n + 4, nb): rows 0, 1 and n + 2, n + 3 are the padding knots of _pad_log.  The inputs are cosmology-major (nb, n), the outputs
// knot-major: a workgroup takes a tile of 64 cosmologies x 64 knots, reads it along the knots (coalesced), turns it in LDS and writes it along the
// cosmologies (coalesced).  (One thread per output element read its inputs 2.7 KB apart from lane to lane: 0.31 ms per 32 768 x 345, 1.5 TB/s of
// useful traffic; the tiles: 0.1 ms.)  Work items beyond the tiles: the four padding rows, a lane per cosmology.
constexpr int BK_TILE = 64;
__global__ __launch_bounds__(256) void brieden_knots_kernel(const double* __restrict__ envelope, const double* __restrict__ pknow,
                                                            const double* __restrict__ ratio_now_fid, const double* __restrict__ k_fid,
                                                            const double* __restrict__ rescale, double kmin, double kmax, double* __restrict__ xk,
                                                            double* __restrict__ yk, long long nb, int n) {
    __shared__ double tile[BK_TILE][BK_TILE + 1];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long ctiles = (nb + BK_TILE - 1) / BK_TILE;
    const int jtiles = (n + BK_TILE - 1) / BK_TILE;
    const long long ntiles = ctiles * jtiles, npad = (nb + 255) / 256;
    for (long long item = blockIdx.x; item < ntiles + npad; item += gridDim.x) {
        if (item < ntiles) {
            const long long c0 = (item / jtiles) * BK_TILE;
            const int j0 = (int)(item % jtiles) * BK_TILE;
            __syncthreads();      // the tile of the previous item has been written out
            {
                const int j = j0 + tx;
                const double rnf = j < n ? ratio_now_fid[j] : 1.;
#pragma unroll 4
                for (int cc = ty; cc < BK_TILE; cc += 4) {
                    const long long c = c0 + cc;
                    if (c < nb && j < n) tile[cc][tx] = log10(envelope[c * n + j] * pknow[c * n + j] * rnf);
                }
            }
            __syncthreads();
            {
                const long long c = c0 + tx;
                const double r = c < nb ? rescale[c] : 1.;
#pragma unroll 4
                for (int jj = ty; jj < BK_TILE; jj += 4) {
                    const int j = j0 + jj;
                    if (c < nb && j < n) {
                        xk[(long long)(j + 2) * nb + c] = log10(k_fid[j] / r);
                        yk[(long long)(j + 2) * nb + c] = tile[tx][jj];
                    }
                }
            }
        } else {
            const long long c = (item - ntiles) * 256 + threadIdx.x;
            if (c >= nb) continue;
            const double r = rescale[c];
            auto logk = [&](int j) { return log10(k_fid[j] / r); };
            auto logp = [&](int j) { return log10(envelope[c * n + j] * pknow[c * n + j] * ratio_now_fid[j]); };
            {      // two points on the line through the first two knots, at lmin and 0.1 logk[0] + 0.9 lmin
                const double lmin = log10(fmin(kmin, k_fid[0] / r * (1 - 1e-9)));
                const double x0 = logk(0), x1 = logk(1), y0 = logp(0), y1 = logp(1);
                const double slope = (y1 - y0) / (x1 - x0);
                const double xa = lmin, xb = x0 * 0.1 + lmin * 0.9;
                xk[c] = xa;
                yk[c] = y0 + slope * (xa - x0);
                xk[nb + c] = xb;
                yk[nb + c] = y0 + slope * (xb - x0);
            }
            {      // ... and through the last two, at 0.1 logk[-1] + 0.9 lmax and lmax
                const double lmax = log10(fmax(kmax, k_fid[n - 1] / r * (1 + 1e-9)));
                const double x0 = logk(n - 1), x1 = logk(n - 2), y0 = logp(n - 1), y1 = logp(n - 2);
                const double slope = (y0 - y1) / (x0 - x1);
                const double xa = x0 * 0.1 + lmax * 0.9, xb = lmax;
                xk[(long long)(n + 2) * nb + c] = xa;
                yk[(long long)(n + 2) * nb + c] = y0 + slope * (xa - x0);
                xk[(long long)(n + 3) * nb + c] = xb;
                yk[(long long)(n + 3) * nb + c] = y0 + slope * (xb - x0);
            }
        }
    }
}

// out (nb, nk) = pk, with columns first .. first + n - 1 replaced by 10^resampled[j, c] (resampled is knot-major (n, nb)): the replaced block
// through 64 x 64 tiles turned in LDS (read along the cosmologies, written along the wavenumbers), the rest a plain copy
__global__ __launch_bounds__(256) void brieden_finish_kernel(const double* __restrict__ pk, const double* __restrict__ resampled, double* __restrict__ out,
                                                             long long nb, int nk, int first, int n) {
    __shared__ double tile[BK_TILE][BK_TILE + 1];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const long long ctiles = (nb + BK_TILE - 1) / BK_TILE;
    const int jtiles = (n + BK_TILE - 1) / BK_TILE;
    const long long ntiles = ctiles * jtiles;
    for (long long item = blockIdx.x; item < ntiles; item += gridDim.x) {
        const long long c0 = (item / jtiles) * BK_TILE;
        const int j0 = (int)(item % jtiles) * BK_TILE;
        __syncthreads();
        {
            const long long c = c0 + tx;
#pragma unroll 4
            for (int jj = ty; jj < BK_TILE; jj += 4) {
                const int j = j0 + jj;
                if (c < nb && j < n) tile[jj][tx] = exp10(resampled[(long long)j * nb + c]);
            }
        }
        __syncthreads();
        {
            const int j = j0 + tx;
#pragma unroll 4
            for (int cc = ty; cc < BK_TILE; cc += 4) {
                const long long c = c0 + cc;
                if (c < nb && j < n) out[c * nk + first + j] = tile[tx][cc];
            }
        }
    }
    // the columns that are kept
    const int kept = nk - n;
    if (kept > 0) {
        const long long total = nb * kept;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
            const long long c = i / kept;
            int k = (int)(i - c * kept);
            k = k < first ? k : k + n;
            out[c * nk + k] = pk[c * nk + k];
        }
    }
}

unsigned grid_for(long long total) {
    const long long blocks = (total + 255) / 256;
    return (unsigned)(blocks < 256 * 16 ? (blocks < 1 ? 1 : blocks) : 256 * 16);
}

// workgroups for the tile walks of the brieden2022 passes: the tiles of (nb, n) and, for the knots, the blocks of padding rows behind them
unsigned grid_tiles(long long nb, int n) {
    const long long items = ((nb + BK_TILE - 1) / BK_TILE) * ((n + BK_TILE - 1) / BK_TILE) + (nb + 255) / 256;
    return (unsigned)(items < 256 * 16 ? (items < 1 ? 1 : items) : 256 * 16);
}

int finish(const char* what, int status_device_ok) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "%s: launch failed: %s", what, hipGetErrorString(e));
    return CP_OK;
}


// ---- wallish2018: second derivatives at the knots and the box, one kernel ---------------------------------------------------------------------
// The wave-level solve, the two arg-max searches and the removal of the box are device functions of cp_wallish_dd.h (shared with the forward
// transform that produces its sequences itself, cp_dst.hip); here a wave takes a sequence from memory.  The operator route applied the inverse of
// the spline's matrix as a banded GEMM (2048 x 2048 with bands of 64: 0.48 ms per 32 768 sequences, then 0.2 ms for the searches over the 537 MB
// it wrote); M itself is written only on request (tests).
using namespace cpdd;

// S = n / 64 knots per lane; LDS per wave: n + 64 doubles, index i -> i + i / S, the sequence first, then d over it, then M over d.
#ifndef CP_DD_ELIMINATION      // 1: sequences of 2048 knots through the elimination in LDS as well (measurements; the default runs the recursions in registers)
#define CP_DD_ELIMINATION 0
#endif
template <int S>
__global__ __launch_bounds__(256, 2) void wallish_dd_box_kernel(const double* y, long long nrows, int margin_first, int margin_second, int off0, int off1,
                                                             int* __restrict__ box, double* __restrict__ dd_out, double* gap) {
    constexpr int N = 64 * S, STRIDE = N + 64;
    extern __shared__ double dd_lds[];
    double* ctab = dd_lds + 4 * STRIDE;           // c_i, i < DD_NTAB (the last entry stands for every later knot)
    double* gtab = ctab + DD_NTAB;                // the same recurrence started from 0 (the eliminations around the box)
    fill_tables(ctab);
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* buf = dd_lds + wave * STRIDE;
    // persistent workgroups: the next sequence of a wave is fetched into registers while this one is solved
    double nxt[S];
    long long row = (long long)blockIdx.x * 4 + wave;
    if (row < nrows) {
#pragma unroll
        for (int k = 0; k < S; ++k) nxt[k] = y[row * N + lane + 64 * k];
    }
    for (; row < nrows; row += (long long)gridDim.x * 4) {
        const double* src = y + row * N;
#pragma unroll
        for (int k = 0; k < S; ++k) buf[(lane + 64 * k) + (lane + 64 * k) / S] = nxt[k];
        wave_lds_phase();      // the sequence is in LDS before any lane reads its neighbours' knots
        if (row + (long long)gridDim.x * 4 < nrows) {
#pragma unroll
            for (int k = 0; k < S; ++k) nxt[k] = y[(row + (long long)gridDim.x * 4) * N + lane + 64 * k];
        }
        int first, second;
        if constexpr (S == 32 && !CP_DD_ELIMINATION) {
            // 2048 knots: the recursions in registers; the sequence stays in LDS, where the removal of the box reads the knots around it
            double m[S];
            second_derivatives_and_box_recursive<PaddedLayout<S>>(buf, lane, margin_first, margin_second, m, first, second);
            if (dd_out) {
#pragma unroll
                for (int t = 0; t < S; ++t) dd_out[row * N + S * lane + t] = m[t];
            }
            if (lane == 0) {
                box[2 * row] = first + off0;
                box[2 * row + 1] = second + off1;
            }
            if (gap) remove_box_parallel<S, PaddedLayout<S>>(buf, gtab, lane, first + off0, second + off1, gap + row * N);
            wave_lds_phase();      // the last reads of the sequence are done before the next one is staged
            continue;
        }
        second_derivatives_and_box<S>(buf, ctab, lane, margin_first, margin_second, first, second);
        if (dd_out) {
            double* dst = dd_out + row * N;
#pragma unroll 4
            for (int j = lane; j < N; j += 64) dst[j] = buf[j + j / S];
        }
        if (lane == 0) {
            box[2 * row] = first + off0;
            box[2 * row + 1] = second + off1;
        }
        wave_lds_phase();      // the last reads of M are done: the buffer is free for the next phase / the next sequence
        if (!gap) continue;
        // the removal of the box on the sequence in memory: the knots around it come from there (the second derivatives in LDS are not needed any more)
        remove_box<S>(buf, gtab, lane, first + off0, second + off1,
                      [&](int lo, int L, int R, int hi, double* zl, double* zr) {
                          for (int e = lane; e <= L - lo; e += 64) {
                              const double x = (double)(lo + e + 1);
                              zl[e] = src[lo + e] * (x * x);
                          }
                          for (int e = lane; e <= hi - R; e += 64) {
                              const double x = (double)(R + e + 1);
                              zr[e] = src[R + e] * (x * x);
                          }
                      },
                      gap + row * N);
    }
}

}  // namespace

extern "C" int cp_wallish_finish(const double* d_pk, const double* d_a, const double* d_b, const double* d_tophat, double* d_out, long long nrows, int n,
                                 int device, void* stream) {
    if (nrows < 0 || n < 1) return cp::fail(CP_EINVAL, "cp_wallish_finish: bad sizes");
    if (nrows == 0) return CP_OK;
    if (!d_pk || !d_a || !d_tophat || !d_out) return cp::fail(CP_EINVAL, "cp_wallish_finish: null pointer");
    DeviceScope scope(device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_wallish_finish: cannot select device %d", device);
    hipLaunchKernelGGL(wallish_finish_kernel, dim3(grid_for(nrows * n)), dim3(256), 0, static_cast<hipStream_t>(stream), d_pk, d_a, d_b, d_tophat, d_out,
                       nrows * n, n);
    return finish("cp_wallish_finish", 0);
}

extern "C" int cp_brieden_ratio(const double* d_rows, const double* d_now, const double* d_g0, const double* d_correction, const double* d_ratio_fid,
                                double* d_pknow, double* d_ratio, long long nb, int n, int device, void* stream) {
    if (nb < 0 || n < 1) return cp::fail(CP_EINVAL, "cp_brieden_ratio: bad sizes");
    if (nb == 0) return CP_OK;
    if (!d_rows || !d_now || !d_g0 || !d_correction || !d_ratio_fid || !d_pknow || !d_ratio) return cp::fail(CP_EINVAL, "cp_brieden_ratio: null pointer");
    DeviceScope scope(device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_brieden_ratio: cannot select device %d", device);
    hipLaunchKernelGGL(brieden_ratio_kernel, dim3(grid_for(nb * n)), dim3(256), 0, static_cast<hipStream_t>(stream), d_rows, d_now, d_g0, d_correction,
                       d_ratio_fid, d_pknow, d_ratio, nb, n);
    return finish("cp_brieden_ratio", 0);
}

extern "C" int cp_brieden_knots(const double* d_envelope, const double* d_pknow, const double* d_ratio_now_fid, const double* d_k_fid,
                                const double* d_rescale, double extrap_kmin, double extrap_kmax, double* d_xk, double* d_yk, long long nb, int n,
                                int device, void* stream) {
    if (nb < 0 || n < 2) return cp::fail(CP_EINVAL, "cp_brieden_knots: bad sizes");
    if (nb == 0) return CP_OK;
    if (!d_envelope || !d_pknow || !d_ratio_now_fid || !d_k_fid || !d_rescale || !d_xk || !d_yk) return cp::fail(CP_EINVAL, "cp_brieden_knots: null pointer");
    DeviceScope scope(device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_brieden_knots: cannot select device %d", device);
    hipLaunchKernelGGL(brieden_knots_kernel, dim3(grid_tiles(nb, n)), dim3(256), 0, static_cast<hipStream_t>(stream), d_envelope, d_pknow,
                       d_ratio_now_fid, d_k_fid, d_rescale, extrap_kmin, extrap_kmax, d_xk, d_yk, nb, n);
    return finish("cp_brieden_knots", 0);
}

// ---- brieden2022: the re-sampling of the smooth spectrum (bao_filter.py:500-509) as ONE kernel, a wave per cosmology --------------------------
// cp_brieden_knots (cosmology-major -> knot-major, the two extrapolated knots of _pad_log on either side), cp_spline_columns (a natural cubic spline per
// column through its own knots log10(k_fid / rescale), elimination factors through a global scratch) and cp_brieden_finish (10^x, knot-major ->
// cosmology-major into the k_fid range of P) move every value through memory three times and turn the layout twice: 0.52 ms per 32 768 cosmologies.
// The knots of a cosmology are a GEOMETRIC grid (k_fid is a range of the filter's geomspace) shifted by log10(rescale): uniform in log10 k, so the
// system for the second derivatives has constant coefficients and is solved by the two first-order recursions of cp_splice_uniform.h (p = sqrt 3 - 2),
// a lane per S knots in registers, the neighbours' totals by DPP shifts.  What the two extrapolated knots on either side change is A p^i + B p^(n-1-i);
// they lie ON the lines through the first / last two samples, so the right-hand sides of the two rows at each junction vanish and A (B) follows in
// closed form from the first (last) two second derivatives of the recursion and the three spacings there; the queries log10(k_fid) all sit at the same
// fraction of their interval, log10(rescale) / h intervals away from their own knot.  (The same steps with the general elimination in LDS, a lane
// per run of intervals, were measured SLOWER than the three kernels: profiles/r4_kernel_experiments.txt.)
namespace {

#ifndef CP_RS_WAVES
#define CP_RS_WAVES 3      // waves per SIMD the kernel is compiled for
#endif
constexpr int RS_SMIN = 3, RS_SMAX = 8;      // knots per lane: 129 <= n <= 512
constexpr double RS_P = -0.26794919243112270647, RS_KAPPA = 0.28867513459481288225;      // sqrt 3 - 2, 1 / (2 sqrt 3)

struct ResampleArgs {
    const double* envelope;       // (nb, n)
    const double* pknow;          // (nb, n)
    const double* ratio_now_fid;  // (n)
    const double* k_fid;          // (n)
    const double* log_k_fid;      // (n) log10 of it: the queries
    const double* rescale;        // (nb)
    double kmin, kmax;
    const double* pk;             // (nb, nk)
    double* out;                  // (nb, nk)
    long long nb;
    int n, nk, first;
    // the form that builds envelope and pknow itself (cp_brieden_smooth): `envelope` = P at the extrema (nb, np), `pknow` = the no-wiggle spectra (nb, n)
    const double* g0;             // (nb)
    const double* correction;     // (n)
    const double* ratio_fid;      // (n)
    const int* peaks;             // (np) positions of the extrema in k_fid
    const double* op;             // (np, n) the columns of the envelope operator that are not zero
    int np;
};

constexpr double rs_ipow(double x, int k) {
    double v = 1.;
    for (int i = 0; i < k; ++i) v *= x;
    return v;
}

// LDS per wave: Y[n + 4] (log10 of the samples, the extrapolated values at 0, 1, n + 2, n + 3), M[n + 4] (second derivatives in units of 6 kappa / h^2).
// S = knots (= samples = queries) per lane.  What does not depend on the cosmology is set up once: the queries and ratio_now_fid (in LDS, shared by
// the four waves), p^(distance of the lane's knots from either end); the next cosmology's samples are requested before this one's are worked on.
// PEAKS: the envelope is linear in the ratio P / pknow / ratio_fid AT THE EXTREMA of the fiducial wiggles only (bao_filter.py:482-488: two quadratic splines
// through them) -- np = 23 of the 341 columns of the operator are not zero.  A lane p < np forms the ratio at its extremum, the lanes take the np columns
// for their S samples from memory (np x n doubles, L2-resident) and sum; pknow = no-wiggle x growth x correction in registers: neither array exists in memory.
// OPLDS: the np x n columns in LDS (63 KB for 23 extrema of 341 samples) shared by the twelve waves of a workgroup of 768 -- one workgroup per CU; from memory
// the 138 loads per lane and cosmology are what the kernel waits for (0.31 ms per 32 768 against 0.17 without the envelope).
template <int S, bool PEAKS, bool OPLDS>
__global__ __launch_bounds__(OPLDS ? 768 : 256, OPLDS ? 1 : CP_RS_WAVES) void brieden_resample_kernel(const ResampleArgs A) {
    constexpr int W = OPLDS ? 12 : 4;      // waves = cosmologies in flight per workgroup
    extern __shared__ double rs_lds[];
    __shared__ cpmath::MathTables mt;
    cpmath::fill_math_tables(&mt);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = A.n, N = n + 4;
    const int stride = N + 4;
    double* Y = rs_lds + (size_t)wave * 2 * stride;
    double* M = Y + stride;
    double* XQ = rs_lds + (size_t)2 * W * stride;      // the queries and ratio_now_fid: the same for every cosmology, shared by the waves
    double* RNF = XQ + n;
    double* CORR = RNF + n;
    double* OP = CORR + n;
    for (int j = threadIdx.x; j < n; j += 64 * W) {
        XQ[j] = A.log_k_fid[j];
        RNF[j] = A.ratio_now_fid[j];
        if (PEAKS) CORR[j] = A.correction[j];
    }
    if (OPLDS)
        for (int j = threadIdx.x; j < A.np * n; j += 64 * W) OP[j] = A.op[j];
    __syncthreads();
    // what the solve rests on: log10(k_fid) uniformly spaced (k_fid a range of a geometric grid).  Anything else: NaN over the k_fid range of every row
    // (the three-kernel route takes any knots)
    int crooked = 0;
    {
        const double x0 = XQ[0], step_x = (XQ[n - 1] - XQ[0]) / (n - 1);
        for (int j = threadIdx.x; j < n; j += 64 * W) crooked |= !(fabs(XQ[j] - fma((double)j, step_x, x0)) <= 1e-9 * fabs(step_x));
    }
    crooked = __syncthreads_or(crooked);
    constexpr double LOG10E = 0.43429448190325182765;
    constexpr int REACH = (32 + S) / S;      // REACH x S >= 33 knots: p^33 = 1e-19
    constexpr double PS = rs_ipow(RS_P, S);
    auto lg10 = [&](double x) { return cpmath::log_tab_any(x, &mt) * LOG10E; };
    const int own = S * lane;
    // p^i for the lane's first knot, p^(n-1-i) for its last one: beyond 40 knots from an end nothing is left (and the powers would underflow);
    // a lane that straddles the last knot: its knots beyond n - 1 do not exist, the power then belongs to knot n - 1 (its knot tlast)
    const int dr = n - 1 - (own + S - 1);
    const int tlast = dr < 0 ? S - 1 + dr : S - 1;
    const double pa = own <= 40 ? pow(RS_P, (double)own) : 0.;
    const double pb = dr <= 40 && dr > -S ? pow(RS_P, (double)(dr < 0 ? 0 : dr)) : 0.;
    const double kf0 = A.k_fid[0], kf1 = A.k_fid[n - 1], lkf0 = A.log_k_fid[0], lkf1 = A.log_k_fid[n - 1];
    const long long step = (long long)gridDim.x * W;
    long long c = (long long)blockIdx.x * W + wave;
    double e[PEAKS ? 1 : S], pn[S], r = 1., g0 = 1., now_peak = 1.;
    // (PEAKS) what lane p < np divides by besides pknow, where its extremum is
    const int peak = PEAKS && lane < A.np ? A.peaks[lane] : 0;
    const double corr_peak = PEAKS ? A.correction[peak] : 1., inv_fid_peak = PEAKS ? 1. / A.ratio_fid[peak] : 1.;
    const double inv_nm1 = 1. / (n - 1);
    auto request = [&](long long cc) {
        if (cc >= A.nb) return;
        r = A.rescale[cc];
        if (PEAKS) {
            g0 = A.g0[cc];
            e[0] = lane < A.np ? A.envelope[cc * A.np + lane] : 0.;
            now_peak = A.pknow[cc * n + peak];
        }
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const int j = lane + 64 * t;
            if (!PEAKS) e[t] = j < n ? A.envelope[cc * n + j] : 1.;
            pn[t] = j < n ? A.pknow[cc * n + j] : 1.;
        }
    };
    request(c);
    for (; c < A.nb; c += step) {
        // the rest of the row as it is, eight loads in flight per lane: a load and its store one after the other -- what the loop `out[k] = pk[k]` compiles to,
        // the two arrays possibly being one -- was eleven memory round trips in a row per cosmology (0.435 -> 0.364 ms per 65 536; both pieces of the row
        // requested together 0.38, the next row's pieces requested an iteration ahead at two waves per SIMD 0.42: profiles/r4_kernel_experiments.txt)
        {
            double* orow = A.out + c * A.nk;
            const double* prow = A.pk + c * A.nk;
            auto copy = [&](int lo, int hi) {
                for (int k0 = lo + lane; k0 < hi; k0 += 64 * 8) {
                    double v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = k0 + 64 * i < hi ? prow[k0 + 64 * i] : 0.;
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (k0 + 64 * i < hi) orow[k0 + 64 * i] = v[i];
                }
            };
            copy(0, A.first);
            copy(A.first + n, A.nk);
        }
        // the geometry of this cosmology's knots (wave-uniform): the uniform stretch x_first + i h, the two knots of _pad_log on either side (interpolator.py:42-87)
        // (kept in scalar registers: some twenty values that live through the whole iteration)
        const double lr = lg10(r);
        const double x_first = cp::wave_uniform(lkf0 - lr), x_last = cp::wave_uniform(lkf1 - lr);
        // (quotients as reciprocals, cp_math.h: a dozen IEEE divisions per cosmology were a sixth of the instructions of the kernel)
        const double h = cp::wave_uniform((x_last - x_first) * inv_nm1), inv_h = cp::wave_uniform(cpmath::recip(h));
        const double inv_r = cpmath::recip(r);
        const double lmin = lg10(fmin(A.kmin, kf0 * inv_r * (1 - 1e-9))), lmax = lg10(fmax(A.kmax, kf1 * inv_r * (1 + 1e-9)));
        const double xa = cp::wave_uniform(lmin), xb = cp::wave_uniform(x_first * 0.1 + lmin * 0.9), xc = cp::wave_uniform(x_last * 0.1 + lmax * 0.9), xd = cp::wave_uniform(lmax);
        // ---- the samples, read where they are; the next cosmology's requested ----
        if (PEAKS) {
            const double ratio = e[0] * cpmath::recip(now_peak * g0 * corr_peak) * inv_fid_peak;      // P / pknow / ratio_fid (bao_filter.py:493-499), at the lane's extremum
            double env[S];
#pragma unroll
            for (int t = 0; t < S; ++t) env[t] = 0.;
            for (int p = 0; p < A.np; p += 2) {      // two columns at a time (2 S independent loads in flight); an odd count: the last one again, times zero
                const int p1 = p + 1 < A.np ? p + 1 : p;
                const double rp0 = cp::lane_value(ratio, p), rp1 = p + 1 < A.np ? cp::lane_value(ratio, p1) : 0.;
                const double* col0 = (OPLDS ? OP : A.op) + (size_t)p * n + lane;
                const double* col1 = (OPLDS ? OP : A.op) + (size_t)p1 * n + lane;
                double c0[S], c1[S];
#pragma unroll
                for (int t = 0; t < S; ++t) {
                    c0[t] = lane + 64 * t < n ? col0[64 * t] : 0.;
                    c1[t] = lane + 64 * t < n ? col1[64 * t] : 0.;
                }
#pragma unroll
                for (int t = 0; t < S; ++t) env[t] = fma(c1[t], rp1, fma(c0[t], rp0, env[t]));
            }
#pragma unroll
            for (int t = 0; t < S; ++t) {
                const int j = lane + 64 * t;
                if (j < n) Y[2 + j] = lg10(env[t] * (pn[t] * g0 * CORR[j]) * RNF[j]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < S; ++t) {
                const int j = lane + 64 * t;
                if (j < n) Y[2 + j] = lg10(e[t] * pn[t] * RNF[j]);
            }
        }
        request(c + step);
        cp::wave_lds_phase();
        const double y0 = Y[2], y1 = Y[3], yn1 = Y[n + 1], yn2 = Y[n];
        const double slope_l = (y1 - y0) * inv_h, slope_r = (yn1 - yn2) * inv_h;
        if (lane == 0) {
            Y[0] = y0 + slope_l * (xa - x_first);
            Y[1] = y0 + slope_l * (xb - x_first);
            Y[n + 2] = yn1 + slope_r * (xc - x_last);
            Y[n + 3] = yn1 + slope_r * (xd - x_last);
            M[0] = 0.;
            M[n + 3] = 0.;
        }
        // ---- second derivatives on the uniform stretch: the two recursions over the lane's knots, the neighbours' totals ----
        double g[S];
        {
            const int lo = own - 1 < 0 ? 0 : own - 1;
            double ym = Y[2 + lo], yc = Y[2 + (own < n ? own : n - 1)];
#pragma unroll
            for (int t = 0; t < S; ++t) {
                const int i = own + t;
                const double yp = Y[2 + (i + 1 < n ? i + 1 : n - 1)];
                g[t] = (i >= 1 && i <= n - 2) ? (yp - yc) - (yc - ym) : 0.;      // (the rows of the two end knots: their right-hand sides vanish)
                ym = yc;
                yc = yp;
            }
        }
#pragma unroll
        for (int t = S - 2; t >= 0; --t) g[t] = fma(RS_P, g[t + 1], g[t]);
        double gr = g[0];
        double f = 0.;
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const double ee = fma(RS_P, f, g[t]);
            f = t + 1 < S ? fma(-RS_P, g[t + 1], ee) : ee;
            g[t] = ee;
        }
        double fl = f, fin = 0., gin = 0., wgt = 1.;
#pragma unroll
        for (int m = 0; m < REACH; ++m) {
            fl = cpsu::from_left(fl);
            gr = cpsu::from_right(gr);
            fin = fma(wgt, fl, fin);
            gin = fma(wgt, gr, gin);
            wgt *= PS;
        }
        {
            double cg = gin, cf = fin;
#pragma unroll
            for (int t = S - 1; t >= 0; --t) {
                cg *= RS_P;
                g[t] += cg;
            }
#pragma unroll
            for (int t = 0; t < S; ++t) {
                cf *= RS_P;
                g[t] += cf;
            }
        }
        // ---- what the extrapolated knots change: A p^i + B p^(n-1-i), from the rows of the junctions (right-hand sides zero): the first and last two
        // second derivatives of the recursion, by shuffles from the lanes that hold them ----
        const int l_last = (n - 1) / S, t_last = (n - 1) - l_last * S, l_prev = (n - 2) / S, t_prev = (n - 2) - l_prev * S;
        double g_last = g[0], g_prev = g[0];
#pragma unroll
        for (int t = 1; t < S; ++t) {
            g_last = t == t_last ? g[t] : g_last;
            g_prev = t == t_prev ? g[t] : g_prev;
        }
        const double m0 = cp::lane_value(g[0], 0), m1 = cp::lane_value(g[1], 0), mn1 = cp::lane_value(g_last, l_last), mn2 = cp::lane_value(g_prev, l_prev);
        auto amplitude = [&](double ha, double H, double ma, double mb) {
            const double alpha = 2. * (H + h) - H * H * cpmath::recip(2. * (ha + H));
            return -(alpha * ma + h * mb) * cpmath::recip(alpha + h * RS_P);
        };
        const double Hl = x_first - xb, Hr = xc - x_last;
        const double Aamp = amplitude(xb - xa, Hl, m0, m1), Bamp = amplitude(xd - xc, Hr, mn1, mn2);
        {
            double ca = Aamp * pa, cb = Bamp * pb;
#pragma unroll
            for (int t = 0; t < S; ++t) {
                g[t] += ca;
                ca *= RS_P;
            }
#pragma unroll
            for (int t = S - 1; t >= 0; --t) {
                if (t <= tlast) {
                    g[t] += cb;
                    cb *= RS_P;
                }
            }
#pragma unroll
            for (int t = 0; t < S; ++t)
                if (own + t < n) M[2 + own + t] = g[t];
        }
        if (lane == 0) {      // (M[2] is this lane's; M[n + 1] = the last knot's: recursion + the two corrections there)
            M[1] = -Hl * g[0] * cpmath::recip(2. * ((xb - xa) + Hl));
            M[n + 2] = -Hr * (mn1 + Bamp) * cpmath::recip(2. * ((xd - xc) + Hr));
        }
        cp::wave_lds_phase();
        // ---- evaluation at log10(k_fid), 10^x into the k_fid range of the row; the rest of the row as it is ----
        double* orow = A.out + c * A.nk;
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const int q = lane + 64 * t;
            if (q >= n) continue;
            const double x = XQ[q];
            double v = __builtin_nan("");
            if (x >= xa && x <= xd) {
                const double u = (x - x_first) * inv_h;
                const double fi = floor(u);
                if (x >= x_first && fi < (double)(n - 1)) {      // an interval of the uniform stretch: this is where the queries are, but for a few at either end
                    const int j = 2 + (int)fi;
                    const double b = u - fi, a = 1. - b;
                    v = fma(a, Y[j], b * Y[j + 1]) + RS_KAPPA * (fma(a * a, a, -a) * M[j] + fma(b * b, b, -b) * M[j + 1]);
                } else {
                    int j;
                    double xl, hh;
                    if (x < x_first) {
                        j = x >= xb ? 1 : 0;
                        xl = j ? xb : xa;
                        hh = j ? x_first - xb : xb - xa;
                    } else {
                        j = x < xc ? n + 1 : n + 2;
                        xl = j == n + 1 ? x_last : xc;
                        hh = j == n + 1 ? xc - x_last : xd - xc;
                    }
                    const double b = (x - xl) / hh, a = 1. - b;
                    const double ratio = hh * inv_h;
                    v = fma(a, Y[j], b * Y[j + 1]) + RS_KAPPA * (ratio * ratio) * (fma(a * a, a, -a) * M[j] + fma(b * b, b, -b) * M[j + 1]);
                }
            }
            if (crooked) v = __builtin_nan("");
            orow[A.first + q] = fabs(v) < 300. ? cpmath::exp10_tab(v, mt.exp2) : cpmath::exp10_mid(v);      // (NaN and what leaves the doubles: the branch-free general form)
        }
        cp::wave_lds_phase();      // last reads before the next cosmology's samples are staged
    }
}

}  // namespace

namespace {

template <int S, bool PEAKS, bool OPLDS>
int launch_resample_as(const ResampleArgs& A, size_t lds, int ncu, void* stream) {
    constexpr int W = OPLDS ? 12 : 4;
    if (lds > 64 * 1024 && cp::allow_full_lds<brieden_resample_kernel<S, PEAKS, OPLDS>>() != hipSuccess) return CP_EDEVICE;
    const long long blocks = (A.nb + W - 1) / W, resident = (long long)ncu * (OPLDS ? 1 : CP_RS_WAVES);
    hipLaunchKernelGGL((brieden_resample_kernel<S, PEAKS, OPLDS>), dim3((unsigned)(blocks < resident ? blocks : resident)), dim3(64 * W), lds,
                       static_cast<hipStream_t>(stream), A);
    return CP_OK;
}

template <bool PEAKS>
int launch_resample(ResampleArgs& A, const char* who, int device, void* stream) {
    const int n = A.n;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
    // per wave Y and M, per workgroup the queries, ratio_now_fid, correction -- and the operator's columns where twelve waves' arrays leave room for them
    const size_t shared = 3 * (size_t)n * sizeof(double), per_wave = 2 * ((size_t)n + 8) * sizeof(double), op = PEAKS ? (size_t)A.np * n * sizeof(double) : 0;
    const bool oplds = PEAKS && 12 * per_wave + shared + op + 2048 <= 160 * 1024;
    const size_t lds = oplds ? 12 * per_wave + shared + op : 4 * per_wave + shared;
    int st = CP_OK;
    auto go = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (PEAKS) st = oplds ? launch_resample_as<S, true, true>(A, lds, ncu, stream) : launch_resample_as<S, true, false>(A, lds, ncu, stream);
        else st = launch_resample_as<S, false, false>(A, lds, ncu, stream);
    };
    switch ((n + 63) / 64) {      // samples per lane
        case 3: go(std::integral_constant<int, 3>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 5: go(std::integral_constant<int, 5>{}); break;
        case 6: go(std::integral_constant<int, 6>{}); break;
        case 7: go(std::integral_constant<int, 7>{}); break;
        default: go(std::integral_constant<int, 8>{}); break;
    }
    if (st != CP_OK) return cp::fail(st, "%s: cannot configure the kernel's LDS", who);
    return finish(who, 0);
}

bool resample_sizes_ok(int n) { return (n + 63) / 64 >= RS_SMIN && (n + 63) / 64 <= RS_SMAX; }

}  // namespace

extern "C" int cp_brieden_resample(const double* d_envelope, const double* d_pknow, const double* d_ratio_now_fid, const double* d_k_fid, const double* d_log_k_fid,
                                   const double* d_rescale, double extrap_kmin, double extrap_kmax, const double* d_pk, double* d_out, long long nb, int n,
                                   int nk, int first, int device, void* stream) {
    if (nb < 0 || n < 4 || nk < 1 || first < 0 || first + n > nk) return cp::fail(CP_EINVAL, "cp_brieden_resample: bad sizes");
    if (!resample_sizes_ok(n))
        return cp::fail(CP_EUNSUPPORTED, "cp_brieden_resample: %d samples per cosmology (129 ... 512): cp_brieden_knots, cp_spline_columns, cp_brieden_finish", n);
    if (nb == 0) return CP_OK;
    if (!d_envelope || !d_pknow || !d_ratio_now_fid || !d_k_fid || !d_log_k_fid || !d_rescale || !d_pk || !d_out) return cp::fail(CP_EINVAL, "cp_brieden_resample: null pointer");
    DeviceScope scope(device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_brieden_resample: cannot select device %d", device);
    ResampleArgs A = {};
    A.envelope = d_envelope; A.pknow = d_pknow; A.ratio_now_fid = d_ratio_now_fid; A.k_fid = d_k_fid; A.log_k_fid = d_log_k_fid; A.rescale = d_rescale;
    A.kmin = extrap_kmin; A.kmax = extrap_kmax; A.pk = d_pk; A.out = d_out; A.nb = nb; A.n = n; A.nk = nk; A.first = first;
    return launch_resample<false>(A, "cp_brieden_resample", device, stream);
}

extern "C" int cp_brieden_smooth(const double* d_pk_peaks, const double* d_now, const double* d_g0, const double* d_correction, const double* d_ratio_fid,
                                 const int* d_peaks, const double* d_operator, int np, const double* d_ratio_now_fid, const double* d_k_fid,
                                 const double* d_log_k_fid, const double* d_rescale, double extrap_kmin, double extrap_kmax, const double* d_pk, double* d_out,
                                 long long nb, int n, int nk, int first, int device, void* stream) {
    if (nb < 0 || n < 4 || nk < 1 || first < 0 || first + n > nk || np < 1 || np > 64) return cp::fail(CP_EINVAL, "cp_brieden_smooth: bad sizes");
    if (!resample_sizes_ok(n)) return cp::fail(CP_EUNSUPPORTED, "cp_brieden_smooth: %d samples per cosmology (129 ... 512)", n);
    if (nb == 0) return CP_OK;
    if (!d_pk_peaks || !d_now || !d_g0 || !d_correction || !d_ratio_fid || !d_peaks || !d_operator || !d_ratio_now_fid || !d_k_fid || !d_log_k_fid || !d_rescale ||
        !d_pk || !d_out)
        return cp::fail(CP_EINVAL, "cp_brieden_smooth: null pointer");
    DeviceScope scope(device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_brieden_smooth: cannot select device %d", device);
    ResampleArgs A = {};
    A.envelope = d_pk_peaks; A.pknow = d_now; A.ratio_now_fid = d_ratio_now_fid; A.k_fid = d_k_fid; A.log_k_fid = d_log_k_fid; A.rescale = d_rescale;
    A.kmin = extrap_kmin; A.kmax = extrap_kmax; A.pk = d_pk; A.out = d_out; A.nb = nb; A.n = n; A.nk = nk; A.first = first;
    A.g0 = d_g0; A.correction = d_correction; A.ratio_fid = d_ratio_fid; A.peaks = d_peaks; A.op = d_operator; A.np = np;
    return launch_resample<true>(A, "cp_brieden_smooth", device, stream);
}

extern "C" int cp_brieden_finish(const double* d_pk, const double* d_resampled, double* d_out, long long nb, int nk, int first, int n, int device,
                                 void* stream) {
    if (nb < 0 || nk < 1 || n < 0 || first < 0 || first + n > nk) return cp::fail(CP_EINVAL, "cp_brieden_finish: bad sizes");
    if (nb == 0) return CP_OK;
    if (!d_pk || !d_resampled || !d_out) return cp::fail(CP_EINVAL, "cp_brieden_finish: null pointer");
    DeviceScope scope(device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_brieden_finish: cannot select device %d", device);
    hipLaunchKernelGGL(brieden_finish_kernel, dim3(grid_tiles(nb, n > 0 ? n : 1)), dim3(256), 0, static_cast<hipStream_t>(stream), d_pk, d_resampled, d_out, nb, nk,
                       first, n);
    return finish("cp_brieden_finish", 0);
}

extern "C" int cp_wallish_dd_box(const double* d_y, long long nrows, int n, int margin_first, int margin_second, int offset_first, int offset_second,
                                 int* d_box, double* d_dd, double* d_gap, int device, void* stream) {
    if (nrows < 0 || margin_first < 0 || 2 * margin_first >= n || margin_second < 0 || margin_second >= n) return cp::fail(CP_EINVAL, "cp_wallish_dd_box: bad sizes");
    if (n != 2048 && n != 1024) return cp::fail(CP_EUNSUPPORTED, "cp_wallish_dd_box: sequences of %d coefficients (built for 1024 and 2048)", n);
    if (nrows == 0) return CP_OK;
    if (!d_y || !d_box) return cp::fail(CP_EINVAL, "cp_wallish_dd_box: null device pointer");
    if ((nrows + 3) / 4 > 2147483647LL) return cp::fail(CP_EUNSUPPORTED, "cp_wallish_dd_box: too many sequences for one launch");
    DeviceScope scope(device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_wallish_dd_box: cannot select device %d", device);
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
    const size_t lds = ((size_t)4 * (n + 64) + 2 * DD_NTAB) * sizeof(double);
    const long long resident = (long long)ncu * (long long)((160 * 1024) / lds < 1 ? 1 : (160 * 1024) / lds);
    const unsigned grid = (unsigned)((nrows + 3) / 4 < resident ? (nrows + 3) / 4 : resident);
    hipStream_t hs = static_cast<hipStream_t>(stream);
#define CP_DD_LAUNCH(S_)                                                                                                                             \
    do {                                                                                                                                             \
        (void)cp::allow_full_lds<&wallish_dd_box_kernel<S_>>();                                                                                      \
        hipLaunchKernelGGL(wallish_dd_box_kernel<S_>, dim3(grid), dim3(256), lds, hs, d_y, nrows, margin_first, margin_second, offset_first,          \
                           offset_second, d_box, d_dd, d_gap);                                                                                       \
    } while (0)
    if (n == 1024) CP_DD_LAUNCH(16);
    else CP_DD_LAUNCH(32);
#undef CP_DD_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_wallish_dd_box: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

// ---- wallish2018: the clamped spline through the spliced knots, evaluated on the filter's wavenumbers, and the wiggle damping -- one kernel ----
// bao_filter.py:415-431: knots = [k < 5e-4 | the linear grid inside (1e-2, 1.5) | k > 2], values = [P | smoothed P on the linear grid | P],
// CubicSpline(bc_type='clamped') evaluated at the filter's k, then pk / ((pk / pknow - 1) tophat + 1).  As spline operators on the two arrays
// (two block-banded GEMMs of 64-wide bands, then an elementwise pass) this was 1.04 ms per 16 384 vectors.  The spline is a tridiagonal system
// for the second derivatives M:  h_{i-1} M_{i-1} + 2 (h_{i-1} + h_i) M_i + h_i M_{i+1} = 6 (s_i - s_{i-1}),  s_i = (y_{i+1} - y_i) / h_i, with
// s_{-1} = s_{n-1} = 0 for the clamped ends.  Its elimination factors 1 / (2 (h_{i-1} + h_i) - h_{i-1} c_{i-1}) depend on the knots only (host,
// plan creation); a wave takes a vector's knot values into LDS and runs both sweeps as in cp_wallish_dd_box -- a lane per segment of S =
// ceil(n / 64) knots, started `halo` knots outside its segment with d = 0 (M = 0), the halo chosen at plan creation so that what is left of the
// start is below 1e-18 -- then evaluates  A y_j + B y_{j+1} + ((A^3 - A) M_j + (B^3 - B) M_{j+1}) h_j^2 / 6  at the queries (weights from the
// plan).  Most knots lie on the uniform grid, where h and the factor are constants once the elimination has forgotten the junction (the
// spacings of a linspace differ by the rounding of its knots, 1e-12 of h: the constants stand for all of them, 1e-13 on the second derivatives):
// only the knots outside that stretch have table entries (LDS).
namespace {

constexpr int SPLICE_KMAX = 60;      // knots per lane the kernel keeps in registers: at most 64 x 60 knots

struct SpliceTables {
    int n, nq, S, halo;
    int piece_src[3], piece_first[3], piece_start[3];      // knot i of piece p = src[piece_src[p]][piece_start[p] + i - piece_first[p]]; piece_first ascending
    int ntab_left, uniform_end;                             // knots [0, ntab_left) and [uniform_end, n) have table entries, the others the constants
    double h0, rh0, inv0;
    const double* tab;                                      // (nslots, 4): 6 / h_i, factor_i, h_{i-1} factor_i, h_i factor_i per slot (see splice_kernel)
    int generic_first, generic_end;                          // queries outside [generic_first, generic_end) fall on a knot taken from the same column of array 0: the result is that value
    const int* qj;                                          // (nq) interval of each query, -1: outside the knots
    const int* qcol;                                        // (nq, 2) columns of y_j and y_{j+1} in their source rows; bit 30 set: source 1
    const double* qw;                                       // (nq, 4) A, B, (A^3 - A) h^2 / 6, (B^3 - B) h^2 / 6
};

struct SpliceArgs {
    SpliceTables T;
    const double* src0;
    const double* src1;
    int n0, n1;
    long long nrows;
    const double* tophat;      // (nq) or null
    double* out;               // (nrows, nq)
};

// S (odd: 64 lanes x S doubles apart fall on 32 different bank pairs) knots per lane.  LDS per wave: halo + 1 | 64 S | halo + 1 doubles -- the knot
// values with the end values repeated on either side (slopes beyond the ends vanish: the clamped boundary condition, and no index is ever
// clamped), then d, then M in the same places, as in wallish_dd_box_kernel -- and, per workgroup, the coefficient table: slot -> (6 / h, factor,
// h_{i-1} x factor, h_i x factor), the knots left of the uniform stretch, ONE slot for the stretch, the knots right of it.  The dependent
// chain of the forward sweep is one FMA per knot: d_i = P_i - Q_i d_{i-1}, P_i = factor_i (sigma_i - sigma_{i-1}), sigma_i = (y_{i+1} - y_i) 6 / h_i.
#ifndef CP_SPLICE_ABLATE      // diagnostic builds (tools/splice_microbench.hip): 1 no sweeps, 2 no evaluation, 4 no knot loads
#define CP_SPLICE_ABLATE 0
#endif
__global__ __launch_bounds__(256) void splice_kernel(const SpliceArgs A) {
    extern __shared__ __attribute__((aligned(32))) double sp_lds[];
    const SpliceTables& T = A.T;
    const int S = T.S, n = T.n, halo = T.halo, pad = halo + 1, stride = 64 * S + 2 * pad + (halo & 1);
    const int nslots = T.ntab_left + 1 + (64 * S + pad - T.uniform_end);      // (entries past the last knot repeat it)
    double* tab = sp_lds + 4 * stride;
    for (int e = threadIdx.x; e < 4 * nslots; e += 256) tab[e] = T.tab[e];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* buf = sp_lds + wave * stride + pad;      // buf[i], -pad <= i < 64 S + pad
    const int L = T.ntab_left, U = T.uniform_end;
    auto slot_of = [&](int i) {      // i < L: i; L <= i < U: L (the uniform stretch); i >= U: i - U + L + 1
        const int lo = i < L ? i : L, hi = i - U + 1;
        return (lo < 0 ? 0 : lo) + (hi < 0 ? 0 : hi);
    };
    const double4* coef = reinterpret_cast<const double4*>(tab);
    const int own = S * lane;
    // persistent workgroups (one per CU: the buffers fill its LDS): the table is fetched once, a wave walks its share of the rows.  One wave per
    // SIMD leaves it 512 registers: the knot values of the NEXT row are fetched into them (a lane takes the knots lane + 64 k, contiguous in their
    // source rows piece by piece) while this row is solved and evaluated.
    constexpr int KMAX = SPLICE_KMAX;
    const int f1 = T.piece_first[1], f2 = T.piece_first[2];
    auto fetch = [&](long long r, double* v) {
        const double* a = A.src0 + r * A.n0;
        const double* b = A.src1 + r * A.n1;
        const double* p0 = (T.piece_src[0] ? b : a) + T.piece_start[0];
        const double* p1 = (T.piece_src[1] ? b : a) + T.piece_start[1] - f1;
        const double* p2 = (T.piece_src[2] ? b : a) + T.piece_start[2] - f2;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int i = lane + 64 * k;
            const double* src = i < f1 ? p0 : (i < f2 ? p1 : p2);
            v[k] = (i < n && !(CP_SPLICE_ABLATE & 4)) ? src[i] : 0.;
        }
    };
    double knots[KMAX];
    long long row = (long long)blockIdx.x * 4 + wave;
    if (row < A.nrows) fetch(row, knots);
    for (; row < A.nrows; row += (long long)gridDim.x * 4) {
    const double* s0 = A.src0 + row * A.n0;
    const double* s1 = A.src1 + row * A.n1;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int i = lane + 64 * k;
        if (i < n) buf[i] = knots[k];
    }
    wave_lds_phase();
    {
        const double first = buf[0], last = buf[n - 1];      // (LDS takes a wave's accesses in order)
        for (int e = lane; e < pad; e += 64) buf[-1 - e] = first;
        for (int e = n + lane; e < 64 * S + pad; e += 64) buf[e] = last;
    }
    wave_lds_phase();      // knot values staged, ends repeated
    if (row + (long long)gridDim.x * 4 < A.nrows) fetch(row + (long long)gridDim.x * 4, knots);
    if (!(CP_SPLICE_ABLATE & 1)) {
        const double beyond = buf[own + S];      // the next segment's first knot, before its owner writes there
        const int start = own - halo;
        double y0 = buf[start], d = 0.;
        double sigma_m = (y0 - buf[start - 1]) * coef[slot_of(start - 1)].x;
#pragma unroll 4
        for (int t = 0; t < halo; ++t) {             // towards the segment: nothing stored
            const int i = start + t;
            const double4 c = coef[slot_of(i)];
            const double yp = buf[i + 1];
            const double sigma = (yp - y0) * c.x;
            d = fma(-c.z, d, c.y * (sigma - sigma_m));
            sigma_m = sigma;
            y0 = yp;
        }
        wave_lds_phase();      // run-ins done before the neighbours' knot values become d
#pragma unroll 4
        for (int t = 0; t < S - 1; ++t) {
            const int i = own + t;
            const double4 c = coef[slot_of(i)];
            const double yp = buf[i + 1];
            const double sigma = (yp - y0) * c.x;
            d = fma(-c.z, d, c.y * (sigma - sigma_m));
            buf[i] = d;                              // (segments past the last knot: slots nobody uses)
            sigma_m = sigma;
            y0 = yp;
        }
        {
            const int i = own + S - 1;
            const double4 c = coef[slot_of(i)];
            const double sigma = (beyond - y0) * c.x;
            buf[i] = fma(-c.z, d, c.y * (sigma - sigma_m));
        }
    }
    wave_lds_phase();      // d complete
    if (!(CP_SPLICE_ABLATE & 1)) {
        double m = 0.;
#pragma unroll 4
        for (int t = 0; t < halo; ++t) {
            const int i = own + S + halo - 1 - t;
            const double d = buf[i];
            m = i >= n - 1 ? d : fma(-coef[slot_of(i)].w, m, d);      // (beyond the last knot: M_{n-1} = d_{n-1} when the sweep gets there)
        }
        wave_lds_phase();
#pragma unroll 4
        for (int t = 0; t < S; ++t) {
            const int i = own + S - 1 - t;
            const double d = buf[i];
            m = i >= n - 1 ? d : fma(-coef[slot_of(i)].w, m, d);
            buf[i] = m;
        }
    }
    wave_lds_phase();      // M complete: the evaluation gathers across segments
    // Queries that fall on a knot taken from their own column of array 0 (wallish2018: every k below and above the linear grid) return that
    // value: whole blocks of 64 such queries skip the tables.  For the others the table entries and the gathered knot values of four blocks are
    // fetched together (registers are plentiful at one wave per SIMD).
    double* out = A.out + row * T.nq;
    const int nblocks = (CP_SPLICE_ABLATE & 2) ? 1 : (T.nq + 63) / 64;
#pragma unroll 4
    for (int blk = 0; blk < nblocks; ++blk) {
        const int q = 64 * blk + lane;
        if (64 * blk + 64 <= T.generic_first || 64 * blk >= T.generic_end) {
            if (q < T.nq) out[q] = s0[q];
            continue;
        }
        if (q >= T.nq) continue;
        const int j = T.qj[q];
        const int c0 = T.qcol[2 * q], c1 = T.qcol[2 * q + 1];
        const double y0 = ((c0 >> 30) & 1 ? s1 : s0)[c0 & 0x3fffffff], y1 = ((c1 >> 30) & 1 ? s1 : s0)[c1 & 0x3fffffff];
        const double4 w = reinterpret_cast<const double4*>(T.qw)[q];
        const int jj = j < 0 ? 0 : j;
        double v = w.x * y0 + w.y * y1 + (w.z * buf[jj] + w.w * buf[jj + 1]);
        if (j < 0) v = __builtin_nan("");
        if (A.tophat) {      // pk / ((pk / pknow - 1) tophat + 1), bao_filter.py:421-431: the queries are the grid of source 0
            const double p = s0[q];
            v = p / ((p / v - 1.) * A.tophat[q] + 1.);
        }
        out[q] = v;
    }
    wave_lds_phase();      // last reads of M before the next row is staged
    }
}

}  // namespace

struct cp_splice_plan {
    SpliceTables T;
    int device;
    double* d_tab;
    int* d_qj;
    int* d_qcol;
    double* d_qw;
    size_t lds_bytes;
    // the scheme for knots on a long uniform stretch (cp_splice_uniform.h), where it fits
    bool has_uniform, use_uniform;
    cpsu::Tables U;
    double* d_uwin;
    double* d_uqw;
    int* d_uqe;
    size_t uniform_lds_bytes;
};

extern "C" int cp_splice_plan_destroy(cp_splice_plan* p) {
    if (!p) return CP_OK;
    {
        DeviceScope scope(p->device);
        if (p->d_tab) (void)hipFree(p->d_tab);
        if (p->d_qj) (void)hipFree(p->d_qj);
        if (p->d_qcol) (void)hipFree(p->d_qcol);
        if (p->d_qw) (void)hipFree(p->d_qw);
        if (p->d_uwin) (void)hipFree(p->d_uwin);
        if (p->d_uqw) (void)hipFree(p->d_uqw);
        if (p->d_uqe) (void)hipFree(p->d_uqe);
    }
    delete p;
    return CP_OK;
}

extern "C" int cp_splice_plan_create(cp_splice_plan** out, int nknots, const double* x, int npieces, const int* piece_src, const int* piece_start,
                                     const int* piece_count, int nq, const double* xq, int device) {
    if (!out) return cp::fail(CP_EINVAL, "cp_splice_plan_create: null plan pointer");
    *out = nullptr;
    if (nknots < 4 || !x || npieces < 1 || npieces > 3 || !piece_src || !piece_start || !piece_count || nq < 1 || !xq)
        return cp::fail(CP_EINVAL, "cp_splice_plan_create: bad arguments");
    if (nknots > (1 << 24) || nq > (1 << 26)) return cp::fail(CP_EUNSUPPORTED, "cp_splice_plan_create: %d knots, %d queries (at most 2^24 / 2^26)", nknots, nq);
    const int n = nknots;
    int total = 0;
    for (int p = 0; p < npieces; ++p) {
        if (piece_count[p] < 0 || piece_start[p] < 0 || (piece_src[p] != 0 && piece_src[p] != 1)) return cp::fail(CP_EINVAL, "cp_splice_plan_create: bad piece %d", p);
        total += piece_count[p];
    }
    if (total != n) return cp::fail(CP_EINVAL, "cp_splice_plan_create: the pieces hold %d knots, not %d", total, n);
    for (int i = 0; i + 1 < n; ++i)
        if (!(x[i + 1] > x[i])) return cp::fail(CP_EINVAL, "cp_splice_plan_create: knots must increase");
    // elimination factors of the system for the second derivatives (clamped ends)
    std::vector<double> h(n), rh(n), inv(n), c(n);
    for (int i = 0; i + 1 < n; ++i) h[i] = x[i + 1] - x[i];
    h[n - 1] = h[n - 2];      // (never multiplies anything that is used: s_{n-1} = 0, and M_{n-1} has no successor)
    for (int i = 0; i < n; ++i) rh[i] = 1. / h[i];
    inv[0] = 1. / (2. * h[0]);
    c[0] = h[0] * inv[0];
    for (int i = 1; i < n; ++i) {
        const double diag = i < n - 1 ? 2. * (h[i - 1] + h[i]) : 2. * h[n - 2];
        inv[i] = 1. / (diag - h[i - 1] * c[i - 1]);
        c[i] = h[i] * inv[i];
    }
    // how far a sweep remembers its start: the forward one by h_{i-1} inv_i per knot, the backward one by c_i
    int halo = 8;
    for (;; halo += 8) {
        if (halo > 128) return cp::fail(CP_EUNSUPPORTED, "cp_splice_plan_create: the elimination does not forget its start within 128 knots");
        double worst = 0.;
        for (int i = halo; i < n; ++i) {
            double f = 1., b = 1.;
            for (int t = 0; t < halo; ++t) {
                f *= std::fabs(h[i - t - 1] * inv[i - t]);
                b *= std::fabs(c[i - t - 1]);
            }
            worst = f > worst ? f : worst;
            worst = b > worst ? b : worst;
        }
        if (worst < 1e-18) break;
    }
    // the uniform stretch: the longest run of equal spacings, shortened on the left until the factor has converged
    int best_lo = 0, best_hi = 0;
    for (int lo = 0; lo < n - 1;) {
        int hi = lo + 1;
        while (hi < n - 1 && std::fabs(h[hi] - h[lo]) <= 2e-11 * h[lo]) ++hi;      // (a linspace: its spacings differ by the rounding of its knots)
        if (hi - lo > best_hi - best_lo) { best_lo = lo; best_hi = hi; }
        lo = hi;
    }
    int ntab_left = n, uniform_end = n;
    double h0 = 1., inv0 = 1.;
    if (best_hi - best_lo > 256) {
        h0 = h[(best_lo + best_hi) / 2];
        inv0 = inv[(best_lo + best_hi) / 2];
        int lo = best_lo + 1;      // the row of knot best_lo still has the spacing of the piece before it
        while (lo < best_hi && std::fabs(inv[lo] - inv0) > 2e-11 * inv0) ++lo;
        int hi = best_hi;          // rows best_lo + 1 .. best_hi - 1 have both spacings on the uniform grid
        while (hi > lo && std::fabs(inv[hi - 1] - inv0) > 2e-11 * inv0) --hi;
        if (hi - lo > 128) { ntab_left = lo; uniform_end = hi; }
    }
    if (n > 64 * SPLICE_KMAX) return cp::fail(CP_EUNSUPPORTED, "cp_splice_plan_create: %d knots, the kernel takes %d", n, 64 * SPLICE_KMAX);
    const int S = ((n + 63) / 64) | 1;      // odd
    const int pad = halo + 1;
    // slots: the knots left of the uniform stretch, one slot for the stretch, the knots right of it, and the positions a lane's segment may
    // reach past the last knot (they repeat it)
    const int nslots = ntab_left + 1 + (64 * S + pad - uniform_end);
    const size_t lds = ((size_t)4 * (64 * S + 2 * pad + (halo & 1)) + 4 * (size_t)nslots) * sizeof(double);
    if (lds > 160 * 1024) return cp::fail(CP_EUNSUPPORTED, "cp_splice_plan_create: %d knots (%d outside a uniform stretch) exceed the LDS of a CU", n, nslots);
    std::vector<double> tab((size_t)4 * nslots);
    auto fill = [&](int slot, int i) {      // 6 / h_i, factor_i, h_{i-1} factor_i, h_i factor_i
        tab[4 * slot] = 6. * rh[i];
        tab[4 * slot + 1] = inv[i];
        tab[4 * slot + 2] = i > 0 ? h[i - 1] * inv[i] : 0.;
        tab[4 * slot + 3] = c[i];
    };
    for (int i = 0; i < ntab_left; ++i) fill(i, i);
    tab[4 * ntab_left] = 6. / h0;
    tab[4 * ntab_left + 1] = inv0;
    tab[4 * ntab_left + 2] = h0 * inv0;
    tab[4 * ntab_left + 3] = h0 * inv0;
    for (int i = uniform_end; i < 64 * S + pad; ++i) fill(i - uniform_end + ntab_left + 1, i < n ? i : n - 1);
    if (ntab_left == n) {      // (no uniform stretch: its slot is never addressed; keep it finite)
        tab[4 * ntab_left] = tab[4 * ntab_left + 1] = 1.;
        tab[4 * ntab_left + 2] = tab[4 * ntab_left + 3] = 0.;
    }
    // queries: interval, source columns of its two knots, weights
    std::vector<int> first(3, n), qj(nq), qcol((size_t)2 * nq);
    std::vector<double> qw((size_t)4 * nq);
    cp_splice_plan* p = new (std::nothrow) cp_splice_plan();
    if (!p) return cp::fail(CP_ENOMEM, "cp_splice_plan_create: host allocation failed");
    for (int k = 0, f = 0; k < 3; ++k) {
        p->T.piece_src[k] = k < npieces ? piece_src[k] : 0;
        p->T.piece_start[k] = k < npieces ? piece_start[k] : 0;
        p->T.piece_first[k] = f;
        f += k < npieces ? piece_count[k] : 0;
    }
    auto column_of = [&](int i) {
        int k = 2;
        while (k > 0 && i < p->T.piece_first[k]) --k;
        return (p->T.piece_start[k] + i - p->T.piece_first[k]) | (p->T.piece_src[k] << 30);
    };
    for (int q = 0; q < nq; ++q) {
        const double v = xq[q];
        if (!(v >= x[0] && v <= x[n - 1])) {
            qj[q] = -1;
            qcol[2 * q] = qcol[2 * q + 1] = 0;
            continue;
        }
        int j = (int)(std::upper_bound(x, x + n, v) - x) - 1;
        j = j > n - 2 ? n - 2 : j;
        const double a = (x[j + 1] - v) / h[j], b = (v - x[j]) / h[j];
        qj[q] = j;
        qcol[2 * q] = column_of(j);
        qcol[2 * q + 1] = column_of(j + 1);
        qw[4 * q] = a;
        qw[4 * q + 1] = b;
        qw[4 * q + 2] = (a * a * a - a) * (h[j] * h[j]) / 6.;
        qw[4 * q + 3] = (b * b * b - b) * (h[j] * h[j]) / 6.;
    }
    int generic_first = nq, generic_end = 0;
    for (int q = 0; q < nq; ++q) {
        const bool on_first = qj[q] >= 0 && qw[4 * q] == 1. && qw[4 * q + 1] == 0. && qcol[2 * q] == q;
        const bool on_second = qj[q] >= 0 && qw[4 * q] == 0. && qw[4 * q + 1] == 1. && qcol[2 * q + 1] == q;
        if (on_first || on_second) continue;
        generic_first = q < generic_first ? q : generic_first;
        generic_end = q + 1;
    }
    if (generic_end <= generic_first) generic_first = generic_end = 0;
    p->T.generic_first = generic_first;
    p->T.generic_end = generic_end;
    p->device = device;
    p->d_tab = nullptr; p->d_qj = nullptr; p->d_qcol = nullptr; p->d_qw = nullptr;
    p->d_uwin = nullptr; p->d_uqw = nullptr; p->d_uqe = nullptr;
    p->has_uniform = p->use_uniform = false;
    p->uniform_lds_bytes = 0;
    p->lds_bytes = lds;
    p->T.n = n; p->T.nq = nq; p->T.S = S; p->T.halo = halo;
    p->T.ntab_left = ntab_left; p->T.uniform_end = uniform_end;
    p->T.h0 = h0; p->T.rh0 = 1. / h0; p->T.inv0 = inv0;
    DeviceScope scope(device);
    bool ok = scope.ok && hipMalloc(&p->d_tab, tab.size() * sizeof(double)) == hipSuccess && hipMalloc(&p->d_qj, qj.size() * sizeof(int)) == hipSuccess &&
              hipMalloc(&p->d_qcol, qcol.size() * sizeof(int)) == hipSuccess && hipMalloc(&p->d_qw, qw.size() * sizeof(double)) == hipSuccess;
    ok = ok && hipMemcpy(p->d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_qj, qj.data(), qj.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_qcol, qcol.data(), qcol.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_qw, qw.data(), qw.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        cp_splice_plan_destroy(p);
        return cp::fail(CP_ENOMEM, "cp_splice_plan_create: cannot place the tables on device %d", device);
    }
    p->T.tab = p->d_tab; p->T.qj = p->d_qj; p->T.qcol = p->d_qcol; p->T.qw = p->d_qw;
    if (best_hi - best_lo + 1 >= 256) {
        cpsu::Built built = cpsu::build(n, x, p->T.piece_first, p->T.piece_src, p->T.piece_start, npieces, nq, xq, qj.data(), best_lo, best_hi, generic_first,
                                        generic_end);
        if (built.ok) {
            ok = hipMalloc(&p->d_uwin, built.win.size() * sizeof(double)) == hipSuccess && hipMalloc(&p->d_uqw, built.qw.size() * sizeof(double)) == hipSuccess &&
                 hipMalloc(&p->d_uqe, built.qe.size() * sizeof(int)) == hipSuccess &&
                 hipMemcpy(p->d_uwin, built.win.data(), built.win.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(p->d_uqw, built.qw.data(), built.qw.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
                 hipMemcpy(p->d_uqe, built.qe.data(), built.qe.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
            if (!ok) {
                cp_splice_plan_destroy(p);
                return cp::fail(CP_ENOMEM, "cp_splice_plan_create: cannot place the tables on device %d", device);
            }
            p->U = built.T;
            p->U.win = p->d_uwin; p->U.qw = p->d_uqw; p->U.qe = p->d_uqe;
            p->uniform_lds_bytes = built.lds_bytes;
            p->has_uniform = p->use_uniform = true;
        }
    }
    *out = p;
    return CP_OK;
}

bool cp_splice_plan_uniform_view(const cp_splice_plan* p, cpsu::Tables* out, int* device) {
    if (!p || !p->has_uniform || !p->use_uniform) return false;
    *out = p->U;
    *device = p->device;
    return true;
}

// which kernel a plan runs: 0 the elimination in LDS (any knots), 1 the recursions on a uniform stretch (cp_splice_uniform.h; the default where it fits)
extern "C" int cp_splice_plan_scheme(const cp_splice_plan* p) { return p && p->use_uniform ? 1 : 0; }

extern "C" int cp_splice_plan_set_scheme(cp_splice_plan* p, int scheme) {
    if (!p) return cp::fail(CP_EINVAL, "cp_splice_plan_set_scheme: null plan");
    if (scheme != 0 && scheme != 1) return cp::fail(CP_EINVAL, "cp_splice_plan_set_scheme: unknown scheme %d", scheme);
    if (scheme == 1 && !p->has_uniform) return cp::fail(CP_EUNSUPPORTED, "cp_splice_plan_set_scheme: the knots and queries of this plan do not fit the uniform-stretch kernel");
    p->use_uniform = scheme == 1;
    return CP_OK;
}

extern "C" int cp_splice_apply(const cp_splice_plan* p, const double* d_src0, int n0, const double* d_src1, int n1, long long nrows, const double* d_tophat,
                               double* d_out, void* stream) {
    if (!p) return cp::fail(CP_EINVAL, "cp_splice_apply: null plan");
    if (nrows < 0) return cp::fail(CP_EINVAL, "cp_splice_apply: negative batch");
    if (nrows == 0) return CP_OK;
    if (!d_src0 || !d_out) return cp::fail(CP_EINVAL, "cp_splice_apply: null device pointer");
    for (int k = 0; k < 3; ++k) {
        const int count = (k < 2 ? p->T.piece_first[k + 1] : p->T.n) - p->T.piece_first[k];
        if (count == 0) continue;
        if (p->T.piece_src[k] == 1 && !d_src1) return cp::fail(CP_EINVAL, "cp_splice_apply: the plan takes knots from a second array");
        if (p->T.piece_start[k] + count > (p->T.piece_src[k] ? n1 : n0)) return cp::fail(CP_EINVAL, "cp_splice_apply: piece %d does not fit its source rows", k);
    }
    if (d_tophat && p->T.nq != n0) return cp::fail(CP_EINVAL, "cp_splice_apply: the damping step needs one query per column of the first array");
    if ((nrows + 3) / 4 > 2147483647LL) return cp::fail(CP_EUNSUPPORTED, "cp_splice_apply: too many rows for one launch");
    DeviceScope scope(p->device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_splice_apply: cannot select device %d", p->device);
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || ncu <= 0) ncu = 256;
    const long long blocks = (nrows + 3) / 4;
    if (p->use_uniform) {
        cpsu::Args U;
        U.T = p->U;
        U.src0 = d_src0; U.src1 = d_src1 ? d_src1 : d_src0; U.n0 = n0; U.n1 = d_src1 ? n1 : n0; U.nrows = nrows; U.tophat = d_tophat; U.out = d_out;
        const hipError_t e = cpsu::launch(U, (unsigned)(blocks < ncu ? blocks : ncu), p->uniform_lds_bytes, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_splice_apply: launch failed: %s", hipGetErrorString(e));
        return CP_OK;
    }
    SpliceArgs A;
    A.T = p->T;
    A.src0 = d_src0; A.src1 = d_src1 ? d_src1 : d_src0; A.n0 = n0; A.n1 = d_src1 ? n1 : n0; A.nrows = nrows; A.tophat = d_tophat; A.out = d_out;
    (void)cp::allow_full_lds<&splice_kernel>();
    const long long per_cu = p->lds_bytes ? (160 * 1024) / (long long)p->lds_bytes : 1;
    const long long resident = (long long)ncu * (per_cu < 1 ? 1 : per_cu);
    hipLaunchKernelGGL(splice_kernel, dim3((unsigned)(blocks < resident ? blocks : resident)), dim3(256), p->lds_bytes, static_cast<hipStream_t>(stream), A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_splice_apply: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
