// tests/abi/consumer.cpp -- a consumer of the C ABI that is neither Python nor torch: plain HIP runtime allocations, plain pointers.
// Builds a small plan from arbitrary tables, runs cp_fftlog_execute and checks the result against a direct O(N^2) evaluation of the
// reference arithmetic irfft(conj(rfft(pad(f) * pre) * u), n=Np) * post, cropped (cosmoprimo/fftlog.py:228-241); then the whole P(k) -> xi(s) -> P(k)
// path with the tables built behind the boundary (cp_fftlog_tables), against an analytic pair.
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
    // ---- the whole path behind the boundary: log grid -> cp_fftlog_tables (P -> xi convention, fftlog.py:318-330) -> plan -> execute, checked
    //      against the analytic pair P(k) = exp(-a^2 k^2) <-> xi(s) = exp(-s^2 / 4 a^2) / (8 pi^1.5 a^3), then back with the xi -> P tables
    //      (fftlog.py:368-377); row 1 of the batch carries a NaN and must stay alone with it
    {
        const int n = 1024, nker = 1, nbatch = 3;
        const double a = 2.;
        std::vector<double> k(n);
        for (int i = 0; i < n; ++i) k[i] = std::pow(10., -5. + 8. * i / (n - 1));
        const int npad = cp_fftlog_padded_size(n, 2);
        if (npad != 2048) return 4;
        const int nh = npad / 2 + 1;
        auto tables = [&](const std::vector<double>& x, double pre_const, std::vector<double>& y, std::vector<double>& pre, std::vector<double>& post,
                          std::vector<double>& u) {
            cp_fftlog_spec spec{CP_KERNEL_SPHERICAL_BESSEL_J, 0., 1.5, 1., 3., pre_const, 1.};
            double delta, lnxy;
            std::vector<double> px(npad), py(npad);
            y.resize(n); pre.resize(npad); post.resize(npad); u.resize(2 * nh);
            return cp_fftlog_tables(n, nker, x.data(), &spec, 2, 1, 1, nullptr, nullptr, &delta, &lnxy, y.data(), px.data(), py.data(), pre.data(),
                                    post.data(), u.data());
        };
        std::vector<double> s, pre, post, u, k2, pre2, post2, u2;
        CHECK_CP(tables(k, std::pow(2. * pi, -1.5), s, pre, post, u));
        CHECK_CP(tables(s, std::pow(2. * pi, 1.5), k2, pre2, post2, u2));
        cp_fftlog_plan *fwd = nullptr, *bwd = nullptr;
        CHECK_CP(cp_fftlog_plan_create(&fwd, n, npad, nker, pre.data(), post.data(), u.data(), 0));
        CHECK_CP(cp_fftlog_plan_create(&bwd, n, npad, nker, pre2.data(), post2.data(), u2.data(), 0));
        std::vector<double> pk(nbatch * n), xi(nbatch * n), back(nbatch * n);
        for (int b = 0; b < nbatch; ++b)
            for (int i = 0; i < n; ++i) pk[b * n + i] = (1. + b) * std::exp(-a * a * k[i] * k[i]);
        pk[1 * n + 17] = std::nan("");
        double *d_a = nullptr, *d_b = nullptr, *d_c = nullptr;
        CHECK_HIP(hipMalloc(&d_a, pk.size() * sizeof(double)));
        CHECK_HIP(hipMalloc(&d_b, pk.size() * sizeof(double)));
        CHECK_HIP(hipMalloc(&d_c, pk.size() * sizeof(double)));
        CHECK_HIP(hipMemcpy(d_a, pk.data(), pk.size() * sizeof(double), hipMemcpyHostToDevice));
        CHECK_CP(cp_fftlog_execute(fwd, d_a, d_b, nbatch, CP_EXTRAP_CONSTANT, 0., CP_EXTRAP_CONSTANT, 0., 0, nullptr));
        CHECK_CP(cp_fftlog_execute(bwd, d_b, d_c, nbatch, CP_EXTRAP_CONSTANT, 0., CP_EXTRAP_CONSTANT, 0., 0, nullptr));
        CHECK_HIP(hipDeviceSynchronize());
        CHECK_HIP(hipMemcpy(xi.data(), d_b, xi.size() * sizeof(double), hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(back.data(), d_c, back.size() * sizeof(double), hipMemcpyDeviceToHost));
        double err_xi = 0., err_back = 0.;
        for (int b = 0; b < nbatch; b += 2)
            for (int i = 0; i < n; ++i) {
                if (std::fabs(k2[i] / k[i] - 1.) > 1e-12) return 5;                       // the inverse transform lands on the input grid
                if (s[i] > 1e-1 && s[i] < 1e1) {
                    const double ref = (1. + b) * std::exp(-s[i] * s[i] / (4. * a * a)) / (8. * std::pow(pi, 1.5) * a * a * a);
                    err_xi = std::fmax(err_xi, std::fabs(xi[b * n + i] / ref - 1.));
                }
                if (k[i] > 1e-2 && k[i] < 1.) err_back = std::fmax(err_back, std::fabs(back[b * n + i] / pk[b * n + i] - 1.));
            }
        for (int i = 0; i < n; ++i)
            if (!std::isnan(xi[1 * n + i]) || !std::isnan(back[1 * n + i])) return 6;     // the NaN row is NaN, alone
        printf("P -> xi against the analytic pair: %.2g; P -> xi -> P: %.2g\n", err_xi, err_back);
        if (!(err_xi < 1e-9) || !(err_back < 1e-9)) return 7;
        CHECK_CP(cp_fftlog_plan_destroy(fwd));
        CHECK_CP(cp_fftlog_plan_destroy(bwd));
        CHECK_HIP(hipFree(d_a));
        CHECK_HIP(hipFree(d_b));
        CHECK_HIP(hipFree(d_c));
    }
    // error convention: status + message, no exception, no abort
    cp_fftlog_plan* plan = nullptr;
    double one = 1.;
    if (cp_fftlog_plan_create(&plan, 3, 6, 1, &one, &one, &one, 0) != CP_EINVAL || plan != nullptr) return 3;
    printf("OK (ABI %d): %s\n", cp_abi_version(), cp_last_error());
    return 0;
}
