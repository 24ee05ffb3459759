// tools/fftlog_microbench.hip -- diagnostic timing harness for the flagship kernel instantiation
// (NP = 4096, zero-padded, cropped).  Built once per -DCP_ABLATE=<mask> (see cp_fft_core.h); prints the
// average kernel time so that differences between masks show what the kernel waits on.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCP_ABLATE=0 -o mb0 tools/fftlog_microbench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../cosmoprimo_amd/csrc/cp_fftlog_kernel.h"

using namespace cpfft;

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

#ifndef MB_NP
#define MB_NP 4096
#endif
#ifndef MB_P
#define MB_P 16
#endif
#ifndef MB_GENERIC  // 1: the generic front / back ends; 2: the HALF front end with 'edge' padding; 0: HALF_ZERO (flagship)
#define MB_GENERIC 0
#endif
#ifndef MB_STREAM_ROWS  // non-temporal row accesses, bit 0 loads, bit 1 stores (3: what cp_fftlog_execute selects for launches of this size)
#define MB_STREAM_ROWS 3
#endif
#ifndef MB_WGS_PER_CU
#define MB_WGS_PER_CU 2
#endif

int main(int argc, char** argv) {
    constexpr int NP = MB_NP, P = MB_P, N = NP / 2;
    const long long nbatch = argc > 1 ? atoll(argv[1]) : 100000;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    std::vector<double> in((size_t)nbatch * N), pre(NP), post(NP), u(2 * (NP / 2 + 1));
    for (size_t i = 0; i < in.size(); ++i) in[i] = 1. + 1e-3 * (double)(i % 977);
    for (int i = 0; i < NP; ++i) pre[i] = 1. + 1e-4 * i, post[i] = 1. - 1e-5 * i;
    for (int i = 0; i <= NP / 2; ++i) u[2 * i] = 0.6, u[2 * i + 1] = (i == 0 || i == NP / 2) ? 0. : 0.8;
    std::vector<cplx> tw, ul(NP);
    build_twiddles<NP, P>(tw);
    build_u_layout<NP, P>(u.data(), ul.data());
    double *d_in, *d_out, *d_pre, *d_post;
    cplx *d_u, *d_tw;
    CHECK(hipMalloc(&d_in, in.size() * 8));
    CHECK(hipMalloc(&d_out, in.size() * 8));
    CHECK(hipMalloc(&d_pre, NP * 8));
    CHECK(hipMalloc(&d_post, NP * 8));
    CHECK(hipMalloc(&d_u, NP * 16));
    CHECK(hipMalloc(&d_tw, tw.size() * 16));
    CHECK(hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_pre, pre.data(), NP * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_post, post.data(), NP * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_u, ul.data(), NP * 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tw, tw.data(), tw.size() * 16, hipMemcpyHostToDevice));
    FftlogArgs A;
    A.in = d_in; A.out = d_out; A.nbatch = nbatch; A.nker = 1; A.n = N; A.in_left = NP / 4; A.out_off = NP / 4; A.n_out = N;
    A.ext_l = A.ext_r = MB_GENERIC == 2 ? 1 : 0; A.val_l = A.val_r = 0.; A.stream_rows = MB_STREAM_ROWS; A.pre = d_pre; A.post = d_post; A.u = d_u; A.tw = d_tw;
#if defined(CP_STAMPS)
    unsigned long long* d_stamp;
    const size_t nstamp = (size_t)2048 * 8 * 16;
    CHECK(hipMalloc(&d_stamp, nstamp * 8));
    CHECK(hipMemset(d_stamp, 0, nstamp * 8));
    A.val_stamp = reinterpret_cast<double*>(d_stamp);
#endif
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = ncu * MB_WGS_PER_CU;
    constexpr int T = Plan<NP, P>::T, lds = Fftlog<NP, P>::LDS_BYTES;
#if MB_GENERIC == 2  // 'edge' padding through the HALF front end
    auto kern = fftlog_kernel<NP, P, IN_HALF, OUT_HALF>;
#elif MB_GENERIC
    auto kern = fftlog_kernel<NP, P, IN_GENERIC, OUT_GENERIC>;
#else
    auto kern = fftlog_kernel<NP, P, IN_HALF_ZERO, OUT_HALF>;
#endif
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // reach the sustained device state first: the first tens of milliseconds of load after an idle period run ~10 % slower (clock ramp)
    for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, 0, A);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, 0, A);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("ablate=%d np=%d grid=%d rows=%lld  %.4f ms  %.2f Mtransforms/s  %.0f GB/s\n", CP_ABLATE, NP, grid, nbatch, ms, nbatch / ms * 1e-3,
           nbatch * 16. * N / ms * 1e-6);
#if defined(CP_STAMPS)
    {
        constexpr int NPH = Fftlog<NP, P>::NPH, W = T / 64, K = 2 * NPH + 1 + 8;
        std::vector<unsigned long long> h((size_t)grid * W * K);
        CHECK(hipMemcpy(h.data(), d_stamp, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> avg(K, 0.);
        for (size_t i = 0; i < (size_t)grid * W; ++i)
            for (int k = 0; k < K; ++k) avg[k] += (double)h[i * K + k] / ((double)grid * W);
        const double npw = (double)((nbatch + 1) / 2) / grid;  // pairs per workgroup
        printf("per pair per wave [s_memtime ticks]: ");
        for (int ph = 0; ph < NPH; ++ph) printf(" work%d=%.0f bar%d=%.0f |", ph, avg[2 * ph] / npw, ph, avg[2 * ph + 1] / npw);
        printf(" total=%.0f (whole kernel %.0f ticks = %.4f ms -> %.1f MHz)\n", avg[2 * NPH] / npw, avg[2 * NPH], ms, avg[2 * NPH] / ms * 1e-3);
        printf("fine: reads landed ph1..4 = %.0f %.0f %.0f %.0f | U applied %.0f | last twiddles applied %.0f | barrier in last phase %.0f\n",
               avg[2 * NPH + 2] / npw, avg[2 * NPH + 3] / npw, avg[2 * NPH + 4] / npw, avg[2 * NPH + 5] / npw, avg[2 * NPH + 6] / npw,
               avg[2 * NPH + 7] / npw, avg[2 * NPH + 8] / npw);
    }
#endif
    return 0;
}
