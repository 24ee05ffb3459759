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

#include <cmath>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"

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

// knot-major (n + 4, nb): rows 0, 1 and n + 2, n + 3 are the padding knots of _pad_log; one thread per (knot, cosmology), cosmology fastest
__global__ __launch_bounds__(256) void brieden_knots_kernel(const double* __restrict__ envelope, const double* __restrict__ pknow,
                                                            const double* __restrict__ ratio_now_fid, const double* __restrict__ k_fid,
                                                            const double* __restrict__ rescale, double kmin, double kmax, double* __restrict__ xk,
                                                            double* __restrict__ yk, long long nb, int n) {
    const long long total = nb * (n + 4);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int row = (int)(i / nb);
        const long long c = i - (long long)row * nb;
        const double r = rescale[c];
        auto logk = [&](int j) { return log10(k_fid[j] / r); };
        auto logp = [&](int j) { return log10(envelope[c * n + j] * pknow[c * n + j] * ratio_now_fid[j]); };
        double x, y;
        if (row >= 2 && row < n + 2) {
            x = logk(row - 2);
            y = logp(row - 2);
        } else if (row < 2) {      // two points on the line through the first two knots, at lmin and 0.1 logk[0] + 0.9 lmin
            const double lmin = log10(fmin(kmin, k_fid[0] / r * (1 - 1e-9)));
            const double x0 = logk(0), x1 = logk(1), y0 = logp(0), y1 = logp(1);
            const double slope = (y1 - y0) / (x1 - x0);
            x = row == 0 ? lmin : x0 * 0.1 + lmin * 0.9;
            y = y0 + slope * (x - x0);
        } else {                   // ... and through the last two, at 0.1 logk[-1] + 0.9 lmax and lmax
            const double lmax = log10(fmax(kmax, k_fid[n - 1] / r * (1 + 1e-9)));
            const double x0 = logk(n - 1), x1 = logk(n - 2), y0 = logp(n - 1), y1 = logp(n - 2);
            const double slope = (y0 - y1) / (x0 - x1);
            x = row == n + 2 ? x0 * 0.1 + lmax * 0.9 : lmax;
            y = y0 + slope * (x - x0);
        }
        xk[i] = x;
        yk[i] = y;
    }
}

// out (nb, nk) = pk, with columns first .. first + n - 1 replaced by 10^resampled[j, c] (resampled is knot-major (n, nb))
__global__ __launch_bounds__(256) void brieden_finish_kernel(const double* __restrict__ pk, const double* __restrict__ resampled, double* __restrict__ out,
                                                             long long nb, int nk, int first, int n) {
    const long long total = nb * nk;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long c = i / nk;
        const int j = (int)(i - c * nk) - first;
        out[i] = (j >= 0 && j < n) ? exp10(resampled[(long long)j * nb + c]) : pk[i];
    }
}

unsigned grid_for(long long total) {
    const long long blocks = (total + 255) / 256;
    return (unsigned)(blocks < 256 * 16 ? (blocks < 1 ? 1 : blocks) : 256 * 16);
}

int finish(const char* what, int status_device_ok) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "%s: launch failed: %s", what, hipGetErrorString(e));
    return CP_OK;
}


// ---- wallish2018: second derivatives at the knots and the box, one kernel ---------------------------------------------------------------------
// The clamped cubic spline through (x = 1 .. n, y) has second derivatives M with  2 M_0 + M_1 = 6 (y_1 - y_0),  M_{i-1} + 4 M_i + M_{i+1} =
// 6 (y_{i+1} - 2 y_i + y_{i-1}),  M_{n-2} + 2 M_{n-1} = -6 (y_{n-1} - y_{n-2}):  what scipy's CubicSpline(bc_type='clamped')(x, nu=2) returns
// (bao_filter.py:377-382).  The operator route applied the inverse of that matrix as a banded GEMM (2048 x 2048 with bands of 64: 0.48 ms per
// 32 768 sequences, then 0.2 ms for the searches over the 537 MB it wrote).  Here a wave takes a sequence into LDS and runs the elimination
// itself: lane l owns the knots [S l, S l + S), S = n / 64.  The modified diagonal c_i = 1 / (4 - c_{i-1}) does not depend on the data and
// converges to 2 - sqrt(3) within 40 knots; the forward recurrence d_i = (rhs_i - d_{i-1}) c_i forgets its start at that rate (0.268 per knot),
// so a lane starts HALO = 32 knots to the left of its own with d = 0 (5e-19 of the starting error is left when it reaches them; the first two
// lanes start at knot 0 and are exact), and the back substitution M_i = d_i - c_i M_{i+1} the same from the right.  64 + 64 dependent steps
// per sequence instead of 4096, nothing but the sequence read from memory; the arg-max searches run on the M in LDS; M itself is written only
// on request (tests).
constexpr int DD_HALO = 32;
constexpr double DD_CINF = 0.26794919243112270647;      // 2 - sqrt(3)
constexpr double DD_CLAST = 1. / (2. - DD_CINF);         // the last row has diagonal 2
constexpr int DD_NTAB = 40;                              // c_i equals its limit to the last bit from knot 30 on
struct DdTable {
    double c[DD_NTAB];
    constexpr DdTable() : c() {
        double v = 0.5;
        c[0] = v;
        for (int i = 1; i < DD_NTAB; ++i) {
            v = 1. / (4. - v);
            c[i] = v;
        }
    }
};
__constant__ DdTable dd_table = DdTable();

__device__ __forceinline__ void dd_argmax_merge(double& v, int& i, double ov, int oi) {      // first index of the maximum, NaN counts as largest (numpy)
    const bool take = (ov > v && !(v != v)) || (ov != ov && !(v != v)) || (((ov == v) || (ov != ov && v != v)) && oi < i);
    if (take) {
        v = ov;
        i = oi;
    }
}

__device__ __forceinline__ int dd_wave_merge(double v, int idx) {
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        dd_argmax_merge(v, idx, ov, oi);
    }
    return idx == 0x7fffffff ? 0 : idx;
}

template <int S>
__device__ __forceinline__ int dd_wave_argmax(const double* buf, int lo, int hi, int lane) {
    double v = -__builtin_inf();
    int idx = 0x7fffffff;
    for (int j = lo + lane; j < hi; j += 64) dd_argmax_merge(v, idx, buf[j + j / S], j);
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        dd_argmax_merge(v, idx, ov, oi);
    }
    return idx == 0x7fffffff ? 0 : idx;
}

// S = n / 64 knots per lane; LDS per wave: n + 64 doubles, index i -> i + i / S (a lane's segment starts on its own bank), the sequence first,
// then d over it (a lane reads ahead of where its left neighbour writes; the one value it needs from its right neighbour's segment it takes
// before the sweep), then M over d (a lane is through its neighbour's segment before the neighbour writes there: the wave runs in lockstep).
template <int S>
__global__ __launch_bounds__(256) void wallish_dd_box_kernel(const double* __restrict__ y, long long nrows, int margin_first, int margin_second, int off0,
                                                             int off1, int* __restrict__ box, double* __restrict__ dd_out) {
    constexpr int N = 64 * S, STRIDE = N + 64;
    extern __shared__ double dd_lds[];
    double* ctab = dd_lds + 4 * STRIDE;           // c_i, i < DD_NTAB (the last entry stands for every later knot)
    if (threadIdx.x < DD_NTAB) ctab[threadIdx.x] = dd_table.c[threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= nrows) return;
    double* buf = dd_lds + wave * STRIDE;
    const double* src = y + row * N;
#pragma unroll 4
    for (int j = lane; j < N; j += 64) buf[j + j / S] = src[j];
    auto at = [&](int i) -> double& { return buf[i + i / S]; };
    auto clamped = [&](int i) { return buf[(i < 0 ? 0 : (i > N - 1 ? N - 1 : i)) + (i < 0 ? 0 : (i > N - 1 ? N - 1 : i)) / S]; };
    auto c_of = [&](int i) { return i >= N - 1 ? DD_CLAST : ctab[i < 0 ? 0 : (i < DD_NTAB ? i : DD_NTAB - 1)]; };
    const int own = S * lane;
    // Both sweeps without a branch: knots read beyond either end repeat the end knot, which makes the right-hand sides of the two clamped rows
    // come out of the general formula (6 ((y_1 - y_0) - (y_0 - y_0)) and 6 ((y_{n-1} - y_{n-1}) - (y_{n-1} - y_{n-2}))) and keeps d = 0 to the
    // left of knot 0.
    {
        const double beyond = clamped(own + S);      // the next segment's first knot, before its owner writes there
        double d = 0., ym = clamped(own - DD_HALO - 1), y0 = clamped(own - DD_HALO);
#pragma unroll 8
        for (int t = 0; t < DD_HALO; ++t) {          // towards the segment: nothing stored
            const int i = own - DD_HALO + t;
            const double yp = clamped(i + 1);
            d = (6. * ((yp - y0) - (y0 - ym)) - d) * c_of(i);
            ym = y0;
            y0 = yp;
        }
#pragma unroll 8
        for (int t = 0; t < S; ++t) {
            const int i = own + t;
            const double yp = t == S - 1 ? beyond : clamped(i + 1);
            d = (6. * ((yp - y0) - (y0 - ym)) - d) * c_of(i);
            at(i) = d;
            ym = y0;
            y0 = yp;
        }
    }
    double best = -__builtin_inf();
    int best_i = 0x7fffffff;
    {
        double m = 0.;
#pragma unroll 8
        for (int t = 0; t < DD_HALO; ++t) {          // towards the segment from the right (beyond the last knot: M_{n-1} = d_{n-1} again)
            const int i = own + S + DD_HALO - 1 - t;
            const double d = clamped(i);
            m = i >= N - 1 ? d : d - c_of(i) * m;
        }
#pragma unroll 8
        for (int t = 0; t < S; ++t) {
            const int i = own + S - 1 - t;
            const double d = at(i);
            m = i >= N - 1 ? d : d - c_of(i) * m;
            at(i) = m;
            // the lane's own maximum inside [margin_first, n - margin_first), first index on ties, NaN as the largest value (numpy's argmax): the
            // sweep runs towards smaller i, so an equal value replaces the one held
            const bool inside = i >= margin_first && i < N - margin_first;
            if (inside && ((m != m) || (!(best != best) && m >= best))) {
                best = m;
                best_i = i;
            }
        }
    }
    if (dd_out) {
        double* dst = dd_out + row * N;
#pragma unroll 4
        for (int j = lane; j < N; j += 64) dst[j] = buf[j + j / S];
    }
    // arg-max over [margin_first, n - margin_first): the lanes' maxima merged; then over [first + margin_second, n - margin_first): the maxima of
    // the lanes whose segments lie inside it, and the segment that straddles its lower end looked at once more, a knot per lane
    const int first = dd_wave_merge(best, best_i);
    const int lower = first + margin_second;
    double v2 = own >= lower ? best : -__builtin_inf();
    int i2 = own >= lower ? best_i : 0x7fffffff;
    {
        const int e = lower - lower % S + lane % S;
        if (e >= lower && e < N - margin_first && lane < S) dd_argmax_merge(v2, i2, at(e), e);
    }
    const int second = dd_wave_merge(v2, i2);
    if (lane == 0) {
        box[2 * row] = first + off0;
        box[2 * row + 1] = second + off1;
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
    hipLaunchKernelGGL(brieden_knots_kernel, dim3(grid_for(nb * (n + 4))), dim3(256), 0, static_cast<hipStream_t>(stream), d_envelope, d_pknow,
                       d_ratio_now_fid, d_k_fid, d_rescale, extrap_kmin, extrap_kmax, d_xk, d_yk, nb, n);
    return finish("cp_brieden_knots", 0);
}

extern "C" int cp_brieden_finish(const double* d_pk, const double* d_resampled, double* d_out, long long nb, int nk, int first, int n, int device,
                                 void* stream) {
    if (nb < 0 || nk < 1 || n < 0 || first < 0 || first + n > nk) return cp::fail(CP_EINVAL, "cp_brieden_finish: bad sizes");
    if (nb == 0) return CP_OK;
    if (!d_pk || !d_resampled || !d_out) return cp::fail(CP_EINVAL, "cp_brieden_finish: null pointer");
    DeviceScope scope(device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_brieden_finish: cannot select device %d", device);
    hipLaunchKernelGGL(brieden_finish_kernel, dim3(grid_for(nb * nk)), dim3(256), 0, static_cast<hipStream_t>(stream), d_pk, d_resampled, d_out, nb, nk,
                       first, n);
    return finish("cp_brieden_finish", 0);
}

extern "C" int cp_wallish_dd_box(const double* d_y, long long nrows, int n, int margin_first, int margin_second, int offset_first, int offset_second,
                                 int* d_box, double* d_dd, int device, void* stream) {
    if (nrows < 0 || margin_first < 0 || 2 * margin_first >= n) return cp::fail(CP_EINVAL, "cp_wallish_dd_box: bad sizes");
    if (n != 2048 && n != 1024) return cp::fail(CP_EUNSUPPORTED, "cp_wallish_dd_box: sequences of %d coefficients (built for 1024 and 2048)", n);
    if (nrows == 0) return CP_OK;
    if (!d_y || !d_box) return cp::fail(CP_EINVAL, "cp_wallish_dd_box: null device pointer");
    if ((nrows + 3) / 4 > 2147483647LL) return cp::fail(CP_EUNSUPPORTED, "cp_wallish_dd_box: too many sequences for one launch");
    DeviceScope scope(device);
    if (!scope.ok) return cp::fail(CP_EDEVICE, "cp_wallish_dd_box: cannot select device %d", device);
    const unsigned grid = (unsigned)((nrows + 3) / 4);
    const size_t lds = ((size_t)4 * (n + 64) + DD_NTAB) * sizeof(double);
    hipStream_t hs = static_cast<hipStream_t>(stream);
#define CP_DD_LAUNCH(S_)                                                                                                                             \
    do {                                                                                                                                             \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&wallish_dd_box_kernel<S_>),                                \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)((4 * (64 * S_ + 64) + DD_NTAB) * sizeof(double))); \
        (void)attr;                                                                                                                                  \
        hipLaunchKernelGGL(wallish_dd_box_kernel<S_>, dim3(grid), dim3(256), lds, hs, d_y, nrows, margin_first, margin_second, offset_first,          \
                           offset_second, d_box, d_dd);                                                                                              \
    } while (0)
    if (n == 1024) CP_DD_LAUNCH(16);
    else CP_DD_LAUNCH(32);
#undef CP_DD_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_wallish_dd_box: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
