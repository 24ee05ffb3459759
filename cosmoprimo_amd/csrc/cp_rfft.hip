// cp_rfft.hip -- batched real FFTs of rows (gfx950) + C ABI: the two methods of the reference's FFT engine protocol,
//   forward(fun)  = rfft(fun, axis=-1)                      (cosmoprimo/fftlog.py:536-539)
//   backward(fun) = irfft(conj(fun), n=size, axis=-1)       (cosmoprimo/fftlog.py:541-544)
// as standalone transforms on device rows, for callers that hold an engine object (NumpyFFTEngine / FFTWEngine by name) and call its methods
// themselves.  FFTlog.__call__ does not come through here: it is the fused kernel (cp_fftlog_kernel.h).
//
// A real row of N samples is ONE complex FFT of M = N / 2 points (z_m = x_2m + i x_2m+1) and a pass over its spectrum:
//   E_k = (Z_k + conj Z_{M-k}) / 2,  O_k = (Z_k - conj Z_{M-k}) / 2i,  X_k = E_k + e^{-2 pi i k / N} O_k,  k = 0 .. M  (Z_M = Z_0),
// and back  Z_k = E_k + i O_k  with  E_k = (X_k + conj X_{M-k}) / 2,  O_k = (X_k - conj X_{M-k}) / 2 e^{+2 pi i k / N},  z = IFFT_M(Z) =
// conj(FFT_M(conj Z)) / M.  A workgroup takes a row at a time: rows never share a transform, so a NaN or a huge row cannot reach another
// (numpy transforms row by row).  The complex FFT is the pass machinery of the FFTLog kernel (cp_fft_core.h): radix-16 butterflies in
// registers, swizzled LDS exchanges, the twiddles of the passes behind the first resident in LDS.  The imaginary parts of the DC and Nyquist
// bins are ignored on the way back, as numpy's c2r does.  HBM: 8 N + 16 (M + 1) bytes per row either way.
#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_fft_core.h"
#include "cp_fftlog_tables.h"
#include "cp_internal.h"

namespace {

using namespace cpfft;

struct RArgs {
    const double* in;
    double* out;
    long long nrows;
    const cplx* tw;      // Plan<M, P> twiddles
    const cplx* rot;     // (M) e^{-2 pi i k / N}
    int conj_input;      // backward: transform conj(in) (the reference's engine convention)
};

template <int M, int P>
__device__ __forceinline__ int rpos_of_freq(int k) {
    using PL = Plan<M, P>;
    int pos = 0;
#pragma unroll
    for (int i = 0; i < PL::NPASS; ++i) {
        const int R = PL::radix(i), MM = PL::len(i) / R;
        pos += (k % R) * MM;
        k /= R;
    }
    return pos;
}

template <int M, int P, int I>
__device__ __forceinline__ void rdif_rest(int t, cplx* lds, const cplx* ltw) {
    using PL = Plan<M, P>;
    if constexpr (I < PL::NPASS) {
        cplx x[P];
        __syncthreads();
        asm volatile("" : "+v"(t));
        Pass<M, P, I>::load_lds(t, lds, x);
        Pass<M, P, I>::butterflies(x);
        Pass<M, P, I>::twiddle_apply_lds(t, ltw + (PL::tw_offset(I) - M), x);
        Pass<M, P, I>::store_lds(t, lds, x);
        rdif_rest<M, P, I + 1>(t, lds, ltw);
    }
}

// forward complex FFT of the M points held as x[r] = z[t + T r]; result in LDS, frequency k at swz(rpos_of_freq(k))
template <int M, int P>
__device__ __forceinline__ void rdif_all(int t, const RArgs& A, cplx* x, cplx* lds, const cplx* ltw) {
    using PL = Plan<M, P>;
    Pass<M, P, 0>::butterflies(x);
    cplx w[P];
    Pass<M, P, 0>::twiddle_load(t, A.tw + PL::tw_offset(0), w);
    Pass<M, P, 0>::twiddle_apply(w, x);
    Pass<M, P, 0>::store_lds(t, lds, x);
    rdif_rest<M, P, 1>(t, lds, ltw);
    __syncthreads();
}

template <int M, int P, bool BACKWARD>
__global__ __launch_bounds__(M / P) void rfft_kernel(const RArgs A) {
    using PL = Plan<M, P>;
    constexpr int T = PL::T, N = 2 * M;
    extern __shared__ __attribute__((aligned(4096))) char rsmem[];
    cplx* lds = reinterpret_cast<cplx*>(rsmem);
    cplx* ltw = lds + lds_data_slots(M, P);
    const int t = threadIdx.x;
    for (int i = t; i < PL::TW_TOTAL - M; i += T) ltw[i] = A.tw[M + i];
    for (long long row = blockIdx.x; row < A.nrows; row += gridDim.x) {
        cplx x[P];
        __syncthreads();      // LDS reuse across rows (and the table fill on the first one)
        int tt = t;
        asm volatile("" : "+v"(tt));
        if constexpr (!BACKWARD) {
            const cplx* src = reinterpret_cast<const cplx*>(A.in + row * N);      // (x_2m, x_2m+1) is z_m
            cplx* dst = reinterpret_cast<cplx*>(A.out) + row * (M + 1);
#pragma unroll
            for (int r = 0; r < P; ++r) x[r] = src[tt + T * r];
            rdif_all<M, P>(tt, A, x, lds, ltw);
            asm volatile("" : "+v"(tt));
#pragma unroll 4
            for (int s = 0; s < P; ++s) {
                const int k = tt + T * s;
                const cplx v = lds[swz<M, P>(rpos_of_freq<M, P>(k))];
                const cplx u = lds[swz<M, P>(rpos_of_freq<M, P>((M - k) % M))];
                const cplx w = A.rot[k];
                const cplx e = cplx{0.5 * (v.re + u.re), 0.5 * (v.im - u.im)};
                const cplx o = cplx{0.5 * (v.im + u.im), 0.5 * (u.re - v.re)};
                cplx X = cplx{e.re + (w.re * o.re - w.im * o.im), e.im + (w.re * o.im + w.im * o.re)};
                if (k == 0) {      // real bins: DC here, Nyquist = E_0 - O_0 behind the last bin
                    X.im = 0.;
                    dst[M] = cplx{e.re - o.re, 0.};
                }
                dst[k] = X;
            }
        } else {
            const cplx* src = reinterpret_cast<const cplx*>(A.in) + row * (M + 1);
            cplx* dst = reinterpret_cast<cplx*>(A.out + row * N);
            const double sign = A.conj_input ? -1. : 1.;
#pragma unroll
            for (int r = 0; r < P; ++r) {
                const int k = tt + T * r;
                cplx a = src[k], b = src[M - k];
                a.im *= sign;
                b.im *= sign;
                if (k == 0) a.im = b.im = 0.;      // DC and Nyquist count as real (numpy's c2r)
                const cplx w = A.rot[k];           // e^{-2 pi i k / N}: its conjugate is the factor of O_k
                const cplx e = cplx{0.5 * (a.re + b.re), 0.5 * (a.im - b.im)};
                const cplx d = cplx{0.5 * (a.re - b.re), 0.5 * (a.im + b.im)};
                const cplx o = cplx{d.re * w.re + d.im * w.im, d.im * w.re - d.re * w.im};
                // conj(Z_k) = conj(E_k + i O_k)
                x[r].re = e.re - o.im;
                x[r].im = -(e.im + o.re);
            }
            rdif_all<M, P>(tt, A, x, lds, ltw);
            asm volatile("" : "+v"(tt));
            const double scale = 1. / M;
#pragma unroll 4
            for (int s = 0; s < P; ++s) {
                const int m = tt + T * s;
                const cplx g = lds[swz<M, P>(rpos_of_freq<M, P>(m))];
                dst[m] = cplx{g.re * scale, -g.im * scale};
            }
        }
    }
}

template <int M, int P>
void rlaunch(bool backward, const RArgs& A, int grid, hipStream_t stream) {
    constexpr int T = M / P;
    constexpr int lds = (lds_data_slots(M, P) + Plan<M, P>::TW_TOTAL - M) * (int)sizeof(cplx);
    if (lds > 64 * 1024) {
        (void)cp::allow_full_lds<&rfft_kernel<M, P, true>>();
        (void)cp::allow_full_lds<&rfft_kernel<M, P, false>>();
    }
    if (backward) hipLaunchKernelGGL((rfft_kernel<M, P, true>), dim3(grid), dim3(T), lds, stream, A);
    else hipLaunchKernelGGL((rfft_kernel<M, P, false>), dim3(grid), dim3(T), lds, stream, A);
}

// X(M, P): half sizes and points per thread
#define CP_RFFT_SIZES(X) X(4, 4) X(8, 8) X(16, 16) X(32, 16) X(64, 16) X(128, 16) X(256, 16) X(512, 16) X(1024, 16) X(2048, 16) X(4096, 16) X(8192, 16)

}  // namespace

struct cp_rfft_plan {
    int size, device;
    cplx* d_tw;
    cplx* d_rot;
};

extern "C" int cp_rfft_plan_destroy(cp_rfft_plan* p) {
    if (!p) return CP_OK;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device) (void)hipSetDevice(p->device);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_rot) (void)hipFree(p->d_rot);
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    delete p;
    return CP_OK;
}

extern "C" int cp_rfft_plan_create(cp_rfft_plan** out, int size, int device) {
    if (!out) return cp::fail(CP_EINVAL, "cp_rfft_plan_create: null plan pointer");
    *out = nullptr;
    if (size < 8 || size > 16384 || (size & (size - 1)))
        return cp::fail(CP_EUNSUPPORTED, "cp_rfft_plan_create: size %d (powers of two from 8 to 16384: the padded sizes FFTlog makes)", size);
    const int m = size / 2;
    std::vector<cplx> tw, rot(m);
#define X(M_, P_) \
    if (m == M_) build_twiddles<M_, P_>(tw);
    CP_RFFT_SIZES(X)
#undef X
    for (int k = 0; k < m; ++k) rot[k] = unit_root(k, size);
    cp_rfft_plan* p = new (std::nothrow) cp_rfft_plan();
    if (!p) return cp::fail(CP_ENOMEM, "cp_rfft_plan_create: host allocation failed");
    p->size = size; p->device = device; p->d_tw = nullptr; p->d_rot = nullptr;
    int prev = -1, status = CP_OK;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) status = cp::fail(CP_EDEVICE, "cp_rfft_plan_create: cannot select device %d", device);
    if (status == CP_OK && (hipMalloc(&p->d_tw, tw.size() * sizeof(cplx)) != hipSuccess || hipMalloc(&p->d_rot, m * sizeof(cplx)) != hipSuccess))
        status = cp::fail(CP_ENOMEM, "cp_rfft_plan_create: device allocation failed");
    if (status == CP_OK && (hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemcpy(p->d_rot, rot.data(), m * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess))
        status = cp::fail(CP_EDEVICE, "cp_rfft_plan_create: upload failed");
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (status != CP_OK) {
        cp_rfft_plan_destroy(p);
        return status;
    }
    *out = p;
    return CP_OK;
}

static int rfft_run(const cp_rfft_plan* p, const double* d_in, double* d_out, long long nrows, bool backward, int conj_input, void* stream, const char* what) {
    if (!p) return cp::fail(CP_EINVAL, "%s: null plan", what);
    if (nrows < 0) return cp::fail(CP_EINVAL, "%s: negative row count", what);
    if (nrows == 0) return CP_OK;
    if (!d_in || !d_out) return cp::fail(CP_EINVAL, "%s: null device pointer", what);
    if (d_in == d_out) return cp::fail(CP_EINVAL, "%s: the transform is not in place", what);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "%s: cannot select device %d", what, p->device);
    RArgs A;
    A.in = d_in; A.out = d_out; A.nrows = nrows; A.tw = p->d_tw; A.rot = p->d_rot; A.conj_input = conj_input;
    const int m = p->size / 2;
    const int grid = (int)(nrows < 2048 ? nrows : 2048);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define X(M_, P_) \
    if (m == M_) rlaunch<M_, P_>(backward, A, grid, s);
    CP_RFFT_SIZES(X)
#undef X
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "%s: launch failed: %s", what, hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_rfft_forward(const cp_rfft_plan* p, const double* d_in, double* d_out, long long nrows, void* stream) {
    return rfft_run(p, d_in, d_out, nrows, false, 0, stream, "cp_rfft_forward");
}

extern "C" int cp_rfft_backward(const cp_rfft_plan* p, const double* d_in, double* d_out, long long nrows, int conj_input, void* stream) {
    return rfft_run(p, d_in, d_out, nrows, true, conj_input != 0, stream, "cp_rfft_backward");
}
