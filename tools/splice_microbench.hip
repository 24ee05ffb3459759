// The spliced-spline kernels of cp_bao.hip (scheme 1: cp_splice_uniform.h, scheme 0: elimination in LDS) on 32 768 vectors with the grids of wallish2018, with parts left out (-DCP_SPLICE_ABLATE=1: no sweeps,
// 2: one query per lane instead of 16, 4: no knot loads): where its time goes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCP_SPLICE_ABLATE=0 -o /tmp/smb tools/splice_microbench.hip && /tmp/smb
#include "../cosmoprimo_amd/csrc/cp_bao.hip"

#include <cstdio>

int main() {
    const int nk = 1024, nlin = 4096;
    const long long nrows = 32768;
    std::vector<double> k(nk), klin(nlin), knots;
    for (int i = 0; i < nk; ++i) k[i] = std::pow(10., -7. + 9. * i / (nk - 1));
    for (int i = 0; i < nlin; ++i) klin[i] = 1e-7 + (2. - 1e-7) * i / (nlin - 1);
    int src[3] = {0, 1, 0}, start[3] = {0, 0, 0}, count[3] = {0, 0, 0};
    for (int i = 0; i < nk; ++i) if (k[i] < 5e-4) { knots.push_back(k[i]); ++count[0]; }
    bool first = true;
    for (int i = 0; i < nlin; ++i) if (klin[i] > 1e-2 && klin[i] < 1.5) { if (first) { start[1] = i; first = false; } knots.push_back(klin[i]); ++count[1]; }
    first = true;
    for (int i = 0; i < nk; ++i) if (k[i] > 2.) { if (first) { start[2] = i; first = false; } knots.push_back(k[i]); ++count[2]; }
    cp_splice_plan* plan = nullptr;
    if (cp_splice_plan_create(&plan, (int)knots.size(), knots.data(), 3, src, start, count, nk, k.data(), 0) != CP_OK) { std::printf("plan failed\n"); return 1; }
    std::printf("knots %d, S %d, halo %d, table slots left %d, uniform until %d, LDS %zu\n", plan->T.n, plan->T.S, plan->T.halo, plan->T.ntab_left, plan->T.uniform_end, plan->lds_bytes);
    double *a, *b, *out, *th;
    (void)hipMalloc(&a, nrows * nk * 8);
    (void)hipMalloc(&b, nrows * nlin * 8);
    (void)hipMalloc(&out, nrows * nk * 8);
    (void)hipMalloc(&th, nk * 8);
    std::vector<double> one((size_t)nrows * nlin, 1.);
    (void)hipMemcpy(a, one.data(), nrows * nk * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, one.data(), nrows * nlin * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(th, one.data(), nk * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 6; ++rep) {
        if (cp_splice_plan_set_scheme(plan, rep < 3 ? 1 : 0) != CP_OK) { std::printf("no uniform-stretch scheme for this plan\n"); continue; }
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i)
            if (cp_splice_apply(plan, a, nk, b, nlin, nrows, th, out, nullptr) != CP_OK) { std::printf("apply failed\n"); return 1; }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::printf("scheme %d, ablate %d / %d: %.3f ms per launch\n", cp_splice_plan_scheme(plan), CP_SPLICE_ABLATE, CP_SPLICE_UNIFORM_ABLATE, ms / 20);
    }
    return 0;
}
