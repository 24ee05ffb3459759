// tools/fetch_calibration.hip -- how rocprofv3's FETCH_SIZE / WRITE_SIZE count the access widths of the FFTLog kernel's row I/O on gfx950.
// The micro-architecture guide calibrates 16-byte-per-lane streaming reads only (FETCH_SIZE reports half their bytes); since round 2 the
// kernel reads and writes its rows with 8-byte-per-lane accesses, so the same known-size copy is run with both widths under the profiler:
//   hipcc --offload-arch=gfx950 -O3 -o fetch_cal tools/fetch_calibration.hip && rocprofv3 --pmc FETCH_SIZE -- ./fetch_cal
// copy8 / copy16 move exactly NBYTES in and NBYTES out per launch.
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr size_t NBYTES = 1ull << 30;

__global__ __launch_bounds__(256) void copy8(const double* __restrict__ in, double* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

__global__ __launch_bounds__(256) void copy16(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

int main() {
    double *a, *b;
    if (hipMalloc(&a, NBYTES) != hipSuccess || hipMalloc(&b, NBYTES) != hipSuccess) return 1;
    (void)hipMemset(a, 1, NBYTES);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(copy8, dim3(2048), dim3(256), 0, 0, a, b, NBYTES / 8);
        hipLaunchKernelGGL(copy16, dim3(2048), dim3(256), 0, 0, reinterpret_cast<const double2*>(a), reinterpret_cast<double2*>(b), NBYTES / 16);
    }
    (void)hipDeviceSynchronize();
    printf("copied %zu bytes per launch with 8- and 16-byte accesses\n", NBYTES);
    return 0;
}
