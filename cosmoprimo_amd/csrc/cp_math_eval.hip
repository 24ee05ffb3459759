// cp_math_eval -- the short forms of the transcendental functions of cp_math.h evaluated on an array: what their stated accuracies are checked on
// (tests/test_math_gpu.py, against 80-bit arithmetic).  Not on any product path.
#include <hip/hip_runtime.h>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_math.h"

namespace {

__global__ __launch_bounds__(256) void math_eval_kernel(int kind, const double* __restrict__ x, double* __restrict__ y, long long n) {
    __shared__ cpmath::MathTables mt;
    cpmath::fill_math_tables(&mt);
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double v = x[i];
        double r;
        switch (kind) {
            case CP_MATH_EXP_MID: r = cpmath::exp_mid(v); break;
            case CP_MATH_EXP_TAB: r = cpmath::exp_tab(v, &mt); break;
            case CP_MATH_LOG_POS: r = cpmath::log_pos(v); break;
            case CP_MATH_LOG_TAB: r = cpmath::log_tab_any(v, &mt); break;
            case CP_MATH_EXP10_MID: r = cpmath::exp10_mid(v); break;
            case CP_MATH_EXP10_TAB: r = cpmath::exp10_tab(v, mt.exp2); break;
            case CP_MATH_SIN_BOUNDED: r = cpmath::sin_bounded(v); break;
            case CP_MATH_RECIP: r = cpmath::recip(v); break;
            default: r = cpmath::rsqrt_pos(v); break;
        }
        y[i] = r;
    }
}

}  // namespace

extern "C" int cp_math_eval(int kind, const double* d_x, double* d_y, long long n, int device, void* stream) {
    if (kind < CP_MATH_EXP_MID || kind > CP_MATH_RSQRT_POS) return cp::fail(CP_EINVAL, "cp_math_eval: unknown function %d", kind);
    if (n < 0) return cp::fail(CP_EINVAL, "cp_math_eval: negative size");
    if (n == 0) return CP_OK;
    if (!d_x || !d_y) return cp::fail(CP_EINVAL, "cp_math_eval: null pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_math_eval: cannot select device %d", device);
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(math_eval_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, static_cast<hipStream_t>(stream), kind, d_x, d_y, n);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_math_eval: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
