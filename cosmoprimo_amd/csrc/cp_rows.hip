// Row screening for the transforms that pack two rows into one complex FFT (cp_fftlog_execute, cp_dst_execute): a NaN / Inf in one row
// reaches its pair partner there, which the reference's row-by-row numpy / scipy FFTs do not do.  The Python facades therefore look at
// every input row first; this is that look as ONE pass over the input (a flag per row, optionally the power of two that bounds the
// row's largest magnitude for the ``rescale_rows`` option) instead of five elementwise / reduction passes of torch.
#include <hip/hip_runtime.h>

#include <cmath>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"

namespace {

__global__ __launch_bounds__(256) void rows_screen_kernel(const double* __restrict__ x, long long nrows, long long n, int require_positive,
                                                          unsigned char* __restrict__ ok, double* __restrict__ scale) {
    __shared__ double wmax[4];
    for (long long row = blockIdx.x; row < nrows; row += gridDim.x) {
        const double* xr = x + row * n;
        int good = 1;
        double amax = 0.;
        for (long long i = threadIdx.x; i < n; i += 256) {
            const double v = xr[i];
            good &= (int)(isfinite(v) && (!require_positive || v > 0.));
            amax = fmax(amax, fabs(v));   // fmax ignores NaN; non-finite rows are flagged anyway
        }
        good = __syncthreads_and(good);
        if (scale) {
            for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_down(amax, off));
            if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
            __syncthreads();
            if (threadIdx.x == 0) {
                const double m = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
                int e = 0;
                if (isfinite(m)) (void)frexp(m, &e);   // m = f 2^e with f in [0.5, 1); e = 0 for m = 0
                e = e > 1023 ? 1023 : e;               // 2^1024 is not a double: rows up to the largest finite value are scaled to < 2
                scale[row] = ldexp(1., e);             // 2^e >= max|row| (1 for an all-zero or non-finite row)
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) ok[row] = (unsigned char)good;
    }
}

}  // namespace

extern "C" int cp_rows_screen(const double* d_x, long long nrows, long long n, int require_positive, unsigned char* d_ok, double* d_scale, int device,
                              void* stream) {
    if (nrows < 0 || n < 0) return cp::fail(CP_EINVAL, "cp_rows_screen: negative size");
    if (nrows == 0) return CP_OK;
    if (!d_ok || (n > 0 && !d_x)) return cp::fail(CP_EINVAL, "cp_rows_screen: null pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_rows_screen: cannot select device %d", device);
    const unsigned grid = (unsigned)(nrows < 65536 ? nrows : 65536);
    hipLaunchKernelGGL(rows_screen_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), d_x, nrows, n, require_positive, d_ok, d_scale);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_rows_screen: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
