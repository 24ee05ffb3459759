// tools/lds_pattern_bench.hip -- diagnostic: LDS cycles per 16-read / 16-write burst for the access patterns of the three
// radix-16 passes of the NP = 4096 plan (Pass<4096,16,I>::lds_off), 4 waves per workgroup, 1 or 2 workgroups per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mb/ldsb tools/lds_pattern_bench.hip && /tmp/mb/ldsb
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "../cosmoprimo_amd/csrc/cp_fft_core.h"
using namespace cpfft;

template <int I, bool WRITE>
__global__ __launch_bounds__(256, 2) void k(unsigned long long* out, int iters) {
    extern __shared__ __attribute__((aligned(4096))) char smem[];
    cplx* lds = reinterpret_cast<cplx*>(smem);
    using PS = Pass<4096, 16, I>;
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += 256) lds[i] = cplx{1. * i, 2. * i};
    __syncthreads();
    cplx x[16];
    for (int r = 0; r < 16; ++r) x[r] = cplx{1. * r + t, 2. * r};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (WRITE) {
            PS::store_lds(t, lds, x);
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        } else {
            cplx y[16];
            PS::load_lds(t, lds, y);
            for (int r = 0; r < 16; ++r) x[r].re += y[r].im;
        }
        asm volatile("" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double acc = 0;
    for (int r = 0; r < 16; ++r) acc += x[r].re;
    if (acc == 1.2345e301) out[1] = 1;
    if (t == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int I, bool WRITE>
void run(const char* name, int grid) {
    unsigned long long* d;
    hipMalloc(&d, 64);
    auto kern = k<I, WRITE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 69888);
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 69888, 0, d, iters);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 69888, 0, d, iters);
    hipDeviceSynchronize();
    unsigned long long h = 0;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("%-28s grid=%4d  %.1f ticks per 16-access burst per wave (4 waves/WG -> %.1f B/tick/WG)\n", name, grid, (double)h / iters,
           4. * 16 * 1024 / ((double)h / iters));
    hipFree(d);
}

int main() {
    for (int grid : {256, 512}) {
        run<0, false>("pass0 read  (M=256, contig)", grid);
        run<1, false>("pass1 read  (M=16)", grid);
        run<2, false>("pass2 read  (M=1)", grid);
        run<0, true>("pass0 write (M=256, contig)", grid);
        run<1, true>("pass1 write (M=16)", grid);
        run<2, true>("pass2 write (M=1)", grid);
    }
    return 0;
}
