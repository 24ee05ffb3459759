// tests/abi/consumer.cpp -- a consumer of the C ABI that is neither Python nor torch: plain HIP runtime allocations, plain pointers.
// Builds a small plan from arbitrary tables, runs cp_fftlog_execute and checks the result against a direct O(N^2) evaluation of the
// reference arithmetic irfft(conj(rfft(pad(f) * pre) * u), n=Np) * post, cropped (cosmoprimo/fftlog.py:228-241).
//   hipcc --offload-arch=gfx950 -O2 -I include -o consumer tests/abi/consumer.cpp -L cosmoprimo_amd -lcosmoprimo_amd -Wl,-rpath,$PWD/cosmoprimo_amd
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cosmoprimo_amd.h"

#define CHECK_CP(x)                                                        \
    do {                                                                   \
        int st_ = (x);                                                     \
        if (st_ != CP_OK) {                                                \
            fprintf(stderr, "%s -> %d: %s\n", #x, st_, cp_last_error());  \
            return 1;                                                      \
        }                                                                  \
    } while (0)
#define CHECK_HIP(x)                                                      \
    do {                                                                  \
        hipError_t e_ = (x);                                              \
        if (e_ != hipSuccess) {                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));      \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main() {
    typedef std::complex<double> cd;
    const double pi = 3.14159265358979323846;
    int worst_n = 0;
    double worst = 0.;
    for (int n : {6, 16, 100, 512}) {
        int npad = 1;
        while (npad < 2 * n) npad *= 2;
        const int nker = 2, nbatch = 3, nh = npad / 2 + 1;
        const int in_left = (npad - n) / 2, out_left = (npad - n) - (npad - n) / 2;
        std::vector<double> pre(nker * npad), post(nker * npad), u(nker * nh * 2), in(nbatch * nker * n), out(nbatch * nker * n);
        unsigned s = 12345u + n;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1u << 24); };
        for (auto& v : pre) v = 0.5 + rnd();
        for (auto& v : post) v = 0.5 + rnd();
        for (auto& v : u) v = rnd() - 0.5;
        for (int k = 0; k < nker; ++k) u[(k * nh + 0) * 2 + 1] = u[(k * nh + nh - 1) * 2 + 1] = 0.;  // real DC / Nyquist, as the reference's tables
        for (auto& v : in) v = rnd();
        cp_fftlog_plan* plan = nullptr;
        CHECK_CP(cp_fftlog_plan_create(&plan, n, npad, nker, pre.data(), post.data(), u.data(), 0));
        double *d_in = nullptr, *d_out = nullptr;
        CHECK_HIP(hipMalloc(&d_in, in.size() * sizeof(double)));
        CHECK_HIP(hipMalloc(&d_out, out.size() * sizeof(double)));
        CHECK_HIP(hipMemcpy(d_in, in.data(), in.size() * sizeof(double), hipMemcpyHostToDevice));
        hipStream_t stream;
        CHECK_HIP(hipStreamCreate(&stream));
        CHECK_CP(cp_fftlog_execute(plan, d_in, d_out, nbatch, CP_EXTRAP_EDGE, 0., CP_EXTRAP_CONSTANT, 0.25, 0, stream));
        CHECK_HIP(hipStreamSynchronize(stream));
        CHECK_HIP(hipMemcpy(out.data(), d_out, out.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int r = 0; r < nbatch * nker; ++r) {
            const int k = r % nker;
            std::vector<double> x(npad);
            for (int j = 0; j < npad; ++j) {
                const int idx = j - in_left;
                const double v = idx < 0 ? in[r * n] : (idx >= n ? 0.25 : in[r * n + idx]);  // 'edge' on the left, constant 0.25 on the right
                x[j] = v * pre[k * npad + j];
            }
            std::vector<cd> X(nh);
            for (int m = 0; m < nh; ++m) {
                cd acc = 0.;
                for (int j = 0; j < npad; ++j) acc += x[j] * std::polar(1., -2. * pi * m * j / npad);
                X[m] = std::conj(acc * cd(u[(k * nh + m) * 2], u[(k * nh + m) * 2 + 1]));
            }
            double scale = 0.;
            std::vector<double> g(npad);
            for (int j = 0; j < npad; ++j) {  // irfft: Hermitian extension, imaginary parts of DC / Nyquist ignored
                double acc = X[0].real() + X[nh - 1].real() * ((j & 1) ? -1. : 1.);
                for (int m = 1; m < nh - 1; ++m) acc += 2. * (X[m] * std::polar(1., 2. * pi * m * j / npad)).real();
                g[j] = acc / npad * post[k * npad + j];
                scale = std::fmax(scale, std::fabs(g[j]));
            }
            for (int o = 0; o < n; ++o) {
                const double err = std::fabs(out[r * n + o] - g[o + out_left]) / scale;
                if (err > worst) worst = err, worst_n = n;
            }
        }
        CHECK_CP(cp_fftlog_plan_destroy(plan));
        CHECK_HIP(hipFree(d_in));
        CHECK_HIP(hipFree(d_out));
        CHECK_HIP(hipStreamDestroy(stream));
    }
    printf("worst relative error %.3g (n = %d)\n", worst, worst_n);
    if (!(worst < 1e-12)) return 2;
    // error convention: status + message, no exception, no abort
    cp_fftlog_plan* plan = nullptr;
    double one = 1.;
    if (cp_fftlog_plan_create(&plan, 3, 6, 1, &one, &one, &one, 0) != CP_EINVAL || plan != nullptr) return 3;
    printf("OK (ABI %d): %s\n", cp_abi_version(), cp_last_error());
    return 0;
}
