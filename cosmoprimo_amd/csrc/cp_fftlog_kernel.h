// cp_fftlog_kernel.h -- the __global__ fused FFTLog kernel and its per-size launcher table.
// Included by cp_fftlog_inst.hip (one translation unit per size group, see Makefile) and cp_fftlog.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "cp_fftlog_body.h"
#include "cp_fftlog_dispatch.h"
#include "cp_fftlog_tables.h"

namespace cpfft {

struct Launcher {
    // [variant]: kernel entry (nullptr when the variant does not exist for this size)
    const void* func[VAR_COUNT];
    void (*launch)(int variant, const FftlogArgs&, int grid, hipStream_t stream);
    int np, p, block, lds_bytes;
    void (*build_tw)(std::vector<cplx>&);
    void (*build_u)(const double*, cplx*);
};

// per size group (defined in cp_fftlog_inst.hip compiled with -DCP_INST_GROUP=g)
bool find_launcher_g0(int npad, Launcher* out);
bool find_launcher_g1(int npad, Launcher* out);
bool find_launcher_g2(int npad, Launcher* out);
bool find_launcher_g3(int npad, Launcher* out);
bool find_launcher_g4(int npad, Launcher* out);

#if defined(__HIPCC__)
template <int NP, int P, int IM, int OM, int PH>
__device__ __forceinline__ void run_phases(int t, const FftlogArgs& A, const double* ra, const double* rb, double* oa, double* ob, bool has_b,
                                           int ker, cplx* lds) {
    using F = Fftlog<NP, P, IM, OM>;
    F::template phase<PH>(t, A, ra, rb, oa, ob, has_b, ker, lds);
    if constexpr (PH + 1 < F::NPH) {
        __syncthreads();
        run_phases<NP, P, IM, OM, PH + 1>(t, A, ra, rb, oa, ob, has_b, ker, lds);
    }
}

// One workgroup = T threads = one packed pair of rows per loop iteration (persistent over pairs).
// Occupancy target: the LDS footprint (16 NP bytes per workgroup) allows 2 workgroups per CU at
// NP = 4096, i.e. 2 waves per SIMD, so the register budget is 256 VGPR+AGPR per lane.
template <int NP, int P, int IM, int OM>
__global__ __launch_bounds__(NP / P, 2) void fftlog_kernel(const FftlogArgs A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* lds = reinterpret_cast<cplx*>(smem);
    using F = Fftlog<NP, P, IM, OM>;
    const int t = threadIdx.x;
    const long long nhalf = (A.nbatch + 1) / 2;
    const long long npairs = nhalf * A.nker;
    for (long long p = blockIdx.x; p < npairs; p += gridDim.x) {
        const int ker = (int)(p % A.nker);
        const long long b0 = 2 * (p / A.nker);
        const bool has_b = b0 + 1 < A.nbatch;
        const long long b1 = has_b ? b0 + 1 : b0;  // incomplete pair: alias row b to row a (its results are dropped)
        const double* ra = A.in + (b0 * A.nker + ker) * A.n;
        const double* rb = A.in + (b1 * A.nker + ker) * A.n;
        double* oa = A.out + (b0 * A.nker + ker) * A.n_out;
        double* ob = A.out + (b1 * A.nker + ker) * A.n_out;
        run_phases<NP, P, IM, OM, 0>(t, A, ra, rb, oa, ob, has_b, ker, lds);
        if (F::NPASS > 1) __syncthreads();  // LDS is reused by the next pair
    }
}

template <int NP, int P>
constexpr bool has_half() {
    return P == 16 && NP >= CP_FFTLOG_HALF_MIN_NP;
}

template <int NP, int P>
void launch_impl(int variant, const FftlogArgs& A, int grid, hipStream_t stream) {
    constexpr int T = Plan<NP, P>::T;
    constexpr int lds = Plan<NP, P>::NPASS > 1 ? NP * (int)sizeof(cplx) : 0;
    if constexpr (has_half<NP, P>()) {
        if (variant == VAR_HALF_ZERO) {
            hipLaunchKernelGGL((fftlog_kernel<NP, P, IN_HALF_ZERO, OUT_HALF>), dim3(grid), dim3(T), lds, stream, A);
            return;
        }
        if (variant == VAR_HALF) {
            hipLaunchKernelGGL((fftlog_kernel<NP, P, IN_HALF, OUT_HALF>), dim3(grid), dim3(T), lds, stream, A);
            return;
        }
    }
    if (variant == VAR_LOG)
        hipLaunchKernelGGL((fftlog_kernel<NP, P, IN_LOG, OUT_GENERIC>), dim3(grid), dim3(T), lds, stream, A);
    else
        hipLaunchKernelGGL((fftlog_kernel<NP, P, IN_GENERIC, OUT_GENERIC>), dim3(grid), dim3(T), lds, stream, A);
}

template <int NP, int P>
Launcher make_launcher() {
    Launcher l;
    for (int v = 0; v < VAR_COUNT; ++v) l.func[v] = nullptr;
    l.func[VAR_GENERIC] = reinterpret_cast<const void*>(&fftlog_kernel<NP, P, IN_GENERIC, OUT_GENERIC>);
    l.func[VAR_LOG] = reinterpret_cast<const void*>(&fftlog_kernel<NP, P, IN_LOG, OUT_GENERIC>);
    if constexpr (has_half<NP, P>()) {
        l.func[VAR_HALF] = reinterpret_cast<const void*>(&fftlog_kernel<NP, P, IN_HALF, OUT_HALF>);
        l.func[VAR_HALF_ZERO] = reinterpret_cast<const void*>(&fftlog_kernel<NP, P, IN_HALF_ZERO, OUT_HALF>);
    }
    l.launch = &launch_impl<NP, P>;
    l.np = NP;
    l.p = P;
    l.block = Plan<NP, P>::T;
    l.lds_bytes = Plan<NP, P>::NPASS > 1 ? NP * (int)sizeof(cplx) : 0;
    l.build_tw = &build_twiddles<NP, P>;
    l.build_u = &build_u_layout<NP, P>;
    return l;
}
#endif  // __HIPCC__

}  // namespace cpfft
