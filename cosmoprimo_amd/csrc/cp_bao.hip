// cp_bao.hip -- the elementwise stages of the wallish2018 and brieden2022 filters over batches of spectra, each as ONE pass (they were chains of
// 5-14 torch elementwise / gather / concatenation kernels between the transforms and the spline operators: a quarter of the filters' GPU time).
//   cp_wallish_finish          bao_filter.py:421-431  pknow = spliced spline; wiggles = (pk / pknow - 1) tophat + 1; out = pk / wiggles
//   cp_brieden_ratio           bao_filter.py:493-499  pknow = P_nowiggle x growth x correction; ratio = P / pknow / ratio_fid
//   cp_brieden_knots           bao_filter.py:500-509 + interpolator.py:42-87 (_pad_log): envelope x pknow x ratio_now_fid -> log10, knot-major, with the
//                                                      two log-log extrapolated knots on either side, per cosmology
//   cp_brieden_finish          bao_filter.py:509      out = pk with 10^(re-sampled log10 P) written over the k_fid range
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
