// cp_pairs.hip -- a tensor-product spline evaluated at PAIRS of points: RectBivariateSpline(...)(x, y, grid=False) of the reference's
// Interpolator2D (jax.py:241-287): out[b, q] = sum_i sum_j wx[q, i] f[b, i, j] wy[q, j], with wx (nq, nx) and wy (nq, ny) the rows of the two
// one-dimensional spline operators at the queries' coordinates (built on the host, a handful of non-zeros per row).
// One wave per (table, query): lane j walks column j of the table (coalesced rows), skipping the x knots whose weight is zero (wave-uniform
// test), multiplies by wy[q, j], and the wave adds up.  Small by construction (a few thousand pairs on 500 x 30 tables): latency-bound, no tiling.
#include <hip/hip_runtime.h>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"

namespace {

__global__ __launch_bounds__(256) void bilinear_pairs_kernel(const double* __restrict__ wx, const double* __restrict__ wy, const double* __restrict__ f,
                                                             double* __restrict__ out, long long nbatch, int nq, int nx, int ny) {
    const int lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= nbatch * nq) return;
    const long long b = item / nq;
    const int q = (int)(item - b * nq);
    const double* wq = wx + (long long)q * nx;
    const double* fb = f + b * (long long)nx * ny;
    double total = 0.;
    for (int j0 = 0; j0 < ny; j0 += 64) {
        const int j = j0 + lane;
        double acc = 0.;
        for (int i = 0; i < nx; ++i) {
            const double w = wq[i];      // (the same address for the whole wave: a scalar load)
            if (w != 0. || w != w) acc = fma(w, j < ny ? fb[(long long)i * ny + j] : 0., acc);
        }
        total += j < ny ? acc * wy[(long long)q * ny + j] : 0.;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off);
    if (lane == 0) out[item] = total;
}

}  // namespace

extern "C" int cp_bilinear_pairs(const double* d_wx, const double* d_wy, const double* d_f, double* d_out, long long nbatch, int nq, int nx, int ny, int device,
                                 void* stream) {
    if (nbatch < 0 || nq < 0 || nx < 1 || ny < 1) return cp::fail(CP_EINVAL, "cp_bilinear_pairs: bad sizes");
    if (nbatch == 0 || nq == 0) return CP_OK;
    if (!d_wx || !d_wy || !d_f || !d_out) return cp::fail(CP_EINVAL, "cp_bilinear_pairs: null pointer");
    const long long blocks = (nbatch * nq + 3) / 4;
    if (blocks > 2147483647LL) return cp::fail(CP_EUNSUPPORTED, "cp_bilinear_pairs: %lld x %d pairs exceed one launch", nbatch, nq);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_bilinear_pairs: cannot select device %d", device);
    hipLaunchKernelGGL(bilinear_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), d_wx, d_wy, d_f, d_out, nbatch, nq, nx, ny);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_bilinear_pairs: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
