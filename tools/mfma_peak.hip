// Sustained rate of v_mfma_f64_16x16x4_f64 on this chip with nothing else going on: 16 independent accumulators per wave, two waves per SIMD,
// every CU.  Prints TFLOP/s and the shader clock it implies (s_memtime ticks of a wave / wall time).  The ceiling against which the operator
// kernels (cp_spline.hip) are to be read: the nominal 78.6 TFLOP/s assume 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void peak(double* out, long long* ticks, int iters, double a0, double b0) {
    v4d acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = v4d{0., 0., 0., 0.};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0.;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int grid = 2 * prop.multiProcessorCount, iters = 20000;
    double* out;
    long long* ticks;
    (void)hipMalloc(&out, (size_t)grid * 256 * 8);
    (void)hipMalloc(&ticks, (size_t)grid * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(peak, dim3(grid), dim3(256), 0, 0, out, ticks, iters, 1.0, 2.0);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(grid);
        (void)hipMemcpy(h.data(), ticks, (size_t)grid * 8, hipMemcpyDeviceToHost);
        double mean = 0.;
        for (long long v : h) mean += (double)v / grid;
        const double flop = (double)grid * 4 * iters * 16 * 2. * 16 * 16 * 4;
        // one wave issues 16 x iters MFMAs; two waves share a SIMD: cycles per MFMA and SIMD = ticks / (2 x 16 x iters) in units of the counter
        std::printf("%d workgroups x 4 waves, %d x 16 MFMAs each: %.3f ms, %.1f TFLOP/s; counter ticks per wave %.3g (%.1f per MFMA of the SIMD), ticks / ms = %.3g\n", grid, iters, ms,
                    flop / ms / 1e9, mean, mean / (2. * 16 * iters), mean / ms);
    }
    return 0;
}
