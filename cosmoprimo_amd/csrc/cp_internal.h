// cp_internal.h -- what the translation units of libcosmoprimo_amd.so share with each other besides the public C ABI (not installed, not part
// of the ABI): views of plan internals for kernels that fuse several stages (cp_sigma.hip).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include "../../include/cosmoprimo_amd.h"

namespace cpfft {
struct cplx;
}

// device tables of a fused-kernel FFTLog plan (npad <= 8192); false for plans of the general-size path
struct cp_fftlog_tables_view {
    int n, npad, nker, device, in_left, out_left;
    const double* d_pre;
    const double* d_post;
    const cpfft::cplx* d_u;
    const cpfft::cplx* d_tw;
};
bool cp_fftlog_plan_view(const cp_fftlog_plan* plan, cp_fftlog_tables_view* out);

// band form of a spline / operator plan: wb (bw, nq) band-major weights (padded entries are zero), j0 (nq) first knot of each band, -1 for a
// query outside the knots
struct cp_spline_band_view {
    int n, nq, bw, device;
    const double* d_wb;
    const int* d_j0;
};
bool cp_spline_plan_view(const cp_spline_plan* plan, cp_spline_band_view* out);

// the constants of a batch of cosmologies for the evaluation of engine `engine` (cppower::CosmoConsts: fit coefficients of EH98 / no-wiggle, Gamma of
// BBKS, the primordial constants of pk_params -- NULL: transfer functions only) into d_work (cp_power_workspace_bytes(ncosmo) bytes): the first of
// the two kernels cp_power_eval launches
// d_k / d_ln_k (n wavenumbers; optional): extra workgroups of the same launch write the (3, n) table log k, k^1.08, k^1.4 of the kernels that evaluate on
// a shared grid (cp_power_eval.h: powers_of_wavenumber) -- a launch of their own for 1024 values was 3 % of the fused sigma(r, z) call
int cp_power_coefficients(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
                          void* d_work, int device, void* stream, const double* d_k = nullptr, double* d_ln_k = nullptr, int n = 0);

// a cp_spline_rows plan seen from cp_spline.hip: its queries as (interval, four weights) per query -- A y_j + B y_{j+1} + wM0 M_j + wM1 M_{j+1}, the
// second derivatives M from cp_spline_rows_second_derivatives -- for the kernel that evaluates the k direction of (z, k) tables itself
struct cp_spline_rows_view {
    int n, nq, first_knot, nknots, device;
    const int* d_qj;        // (nq) interval relative to first_knot, -1: outside the knots
    const double* d_qw;     // (nq, 4)
};
bool cp_spline_rows_plan_view(const cp_spline_rows_plan* plan, cp_spline_rows_view* out);

// the uniform-stretch scheme of a splice plan (cp_splice_uniform_plan.h: its tables with their device pointers) for the kernel that runs the spline step
// of wallish2018 behind the inverse transform (cp_dst.hip: wallish_tail_kernel); false when the plan does not run that scheme
namespace cpsu {
struct Tables;
}
bool cp_splice_plan_uniform_view(const cp_splice_plan* plan, cpsu::Tables* out, int* device);

// Dynamic LDS above 64 KB is an opt-in per kernel AND per device (hipFuncAttributeMaxDynamicSharedMemorySize).  Raises the limit of KERNEL to the
// whole 160 KB of a CU the first time it is launched on the device that is current (once per (kernel, device): a multi-GPU process configures every
// device it launches on, and a later launch with more LDS than the first finds the limit already at the maximum).
namespace cp {
template <auto KERNEL>
inline hipError_t allow_full_lds() {
    static std::atomic<bool> configured[64];      // (zero-initialised; two threads may both set the same attribute to the same value: harmless, and no data race)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && configured[dev].load(std::memory_order_acquire)) return hipSuccess;
    hipFuncAttributes attr;      // the limit covers static + dynamic LDS: a kernel with __shared__ variables of its own gets what they leave
    e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(KERNEL));
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)attr.sharedSizeBytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) configured[dev].store(true, std::memory_order_release);
    return e;
}

// Kernels that hand data from lane to lane of ONE wave through LDS without a workgroup barrier (the tridiagonal eliminations of cp_bao.hip and
// cp_spline_rows.hip): the hardware executes a wave's LDS instructions in order; this pins the compiler to the same order between the phases
// (staging, forward sweep, backward sweep, evaluation): no LDS access moves across it.  At wavefront scope neither the fences nor the barrier
// emit an instruction.
// the value lane `src` (the same for the whole wave) holds, in every lane
__device__ __forceinline__ double lane_value(double v, int src) {
    const long long bits = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)bits, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(bits >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// a value every lane holds alike, moved to scalar registers (first active lane's)
__device__ __forceinline__ double wave_uniform(double v) {
    const long long bits = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bits), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bits >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ void wave_lds_phase() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
}  // namespace cp
