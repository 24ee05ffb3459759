// The dense-operator kernel of cp_spline.hip on 16 384 rows x (1024 -> 1024), with parts of its loop left out (-DCP_LINOP_ABLATE=1: the operator
// of the first chunk for all chunks, 2: the rows of the first chunk, 3: both): which of its two operand streams keeps the matrix cores waiting.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DCP_LINOP_ABLATE=0 -o /tmp/lmb tools/linop_microbench.hip && /tmp/lmb
#include "../cosmoprimo_amd/csrc/cp_spline.hip"

#include <cstdio>
#include <random>

int main() {
    const int n = 1024, nq = 1024;
    const long long nrows = 16384;
    std::vector<double> w((size_t)n * nq), y((size_t)nrows * n);
    std::mt19937_64 gen(1);
    std::normal_distribution<double> dist;
    for (auto& v : w) v = dist(gen);
    for (auto& v : y) v = dist(gen);
    cp_spline_plan* plan = nullptr;
    if (cp_linop_plan_create(&plan, n, nq, w.data(), 0) != CP_OK) { std::printf("plan failed\n"); return 1; }
    double *dy, *dout;
    (void)hipMalloc(&dy, y.size() * 8);
    (void)hipMalloc(&dout, (size_t)nrows * nq * 8);
    (void)hipMemcpy(dy, y.data(), y.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i)
            if (cp_spline_apply(plan, dy, dout, nrows, CP_SPLINE_POST_NONE | CP_SPLINE_PATH_MFMA, 1., nullptr) != CP_OK) { std::printf("apply failed\n"); return 1; }
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::printf("ablate %d: %.3f ms per launch, %.1f TFLOP/s\n", CP_LINOP_ABLATE, ms / 20, 2. * n * nq * nrows / (ms / 20) / 1e9);
    }
    return 0;
}
