// cp_fftlog_large.hip -- FFTLog rows whose padded length does not fit the LDS-resident fused kernel (Np > 8192).
//
// Same arithmetic as the reference (cosmoprimo/fftlog.py:228-241): pad (:436-505) x prefactor -> rfft -> x u -> conj -> irfft ->
// x postfactor -> crop, as three elementwise kernels around a library real FFT (hipFFT D2Z / Z2D, batched over a chunk of rows
// through plan-owned scratch buffers).  This is the general-size path only: 4 <= Np <= 8192 (every size the reference's own callers
// use: nk = 1024 in to_xi / sigma, 4096 in the BAO filters) runs the fused kernel of cp_fftlog.hip.  hipFFT is loaded on first use
// (dlopen), so that the library has no load-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>
#include <new>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_fftlog_large.h"

namespace {

// the part of the hipFFT C API that is used (hipfft/hipfft.h), resolved at run time
typedef struct hipfftHandle_t* fft_handle;
enum { FFT_D2Z = 0x6a, FFT_Z2D = 0x6c, FFT_SUCCESS = 0 };
struct FftApi {
    int (*plan1d)(fft_handle*, int, int, int);
    int (*set_stream)(fft_handle, hipStream_t);
    int (*exec_d2z)(fft_handle, double*, double2*);
    int (*exec_z2d)(fft_handle, double2*, double*);
    int (*destroy)(fft_handle);
};

const FftApi* fft_api() {
    static FftApi api;
    static bool ok = false;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("libhipfft.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libhipfft.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        api.plan1d = reinterpret_cast<int (*)(fft_handle*, int, int, int)>(dlsym(h, "hipfftPlan1d"));
        api.set_stream = reinterpret_cast<int (*)(fft_handle, hipStream_t)>(dlsym(h, "hipfftSetStream"));
        api.exec_d2z = reinterpret_cast<int (*)(fft_handle, double*, double2*)>(dlsym(h, "hipfftExecD2Z"));
        api.exec_z2d = reinterpret_cast<int (*)(fft_handle, double2*, double*)>(dlsym(h, "hipfftExecZ2D"));
        api.destroy = reinterpret_cast<int (*)(fft_handle)>(dlsym(h, "hipfftDestroy"));
        ok = api.plan1d && api.set_stream && api.exec_d2z && api.exec_z2d && api.destroy;
    });
    return ok ? &api : nullptr;
}

struct PadArgs {
    const double* in;   // (nrows, n), row r uses kernel r % nker
    double* work;       // (nrows, npad)
    const double* pre;  // (nker, npad)
    long long nrows;
    int n, npad, nker, in_left;
    int ext_l, ext_r;
    double val_l, val_r;
};

// pad(array, (L, R), extrap) x padded_prefactor (fftlog.py:483-505, 230)
__global__ __launch_bounds__(256) void pad_pre_kernel(const PadArgs A) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.nrows * A.npad) return;
    const long long r = i / A.npad;
    const int j = (int)(i - r * A.npad);
    const double* a = A.in + r * A.n;
    const int idx = j - A.in_left;
    const int cl = idx < 0 ? 0 : (idx >= A.n ? A.n - 1 : idx);
    double v = a[cl];
    if (idx < 0) {
        if (A.ext_l == CP_EXTRAP_CONSTANT) v = A.val_l;
        if (A.ext_l == CP_EXTRAP_LOGLOG) v = v * pow(a[1] / v, (double)idx);
    } else if (idx >= A.n) {
        if (A.ext_r == CP_EXTRAP_CONSTANT) v = A.val_r;
        if (A.ext_r == CP_EXTRAP_LOGLOG) v = v / pow(a[A.n - 2] / v, (double)(idx - A.n + 1));
    }
    A.work[i] = v * A.pre[(r % A.nker) * (long long)A.npad + j];
}

// conj(rfft(.) * u); numpy's irfft ignores the imaginary parts of the DC and Nyquist bins (fftlog.py:231, 542-544)
__global__ __launch_bounds__(256) void mul_u_conj_kernel(double2* c, const double2* u, long long nrows, int nh, int nker) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * nh) return;
    const long long r = i / nh;
    const int m = (int)(i - r * nh);
    const double2 x = c[i], w = u[(r % nker) * (long long)nh + m];
    double2 y;
    y.x = x.x * w.x - x.y * w.y;
    y.y = -(x.x * w.y + x.y * w.x);
    if (m == 0 || m == nh - 1) y.y = 0.;
    c[i] = y;
}

// irfft normalisation 1 / Np, x padded_postfactor, crop (fftlog.py:232-235)
__global__ __launch_bounds__(256) void post_crop_kernel(const double* work, const double* post, double* out, long long nrows, int npad, int nker,
                                                         int out_off, int n_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * n_out) return;
    const long long r = i / n_out;
    const int o = (int)(i - r * n_out) + out_off;
    out[i] = work[r * npad + o] / npad * post[(r % nker) * (long long)npad + o];
}

unsigned blocks(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace

struct cp_fftlog_large {
    int n, npad, nker, device;
    long long chunk;       // rows per FFT batch (a multiple of nker)
    double* d_pre;
    double* d_post;
    double2* d_u;          // (nker, npad / 2 + 1)
    double* d_work;        // (chunk, npad)
    double2* d_cplx;       // (chunk, npad / 2 + 1)
    fft_handle d2z, z2d;
    std::mutex lock;       // hipfftSetStream mutates the plan: one execute is ENQUEUED at a time ...
    hipEvent_t done;       // ... and the scratch (d_work, d_cplx) is shared: an execute waits, on its own stream, for the event the previous
    bool has_done;         // execute recorded behind its last kernel, so that executes of one plan on different streams cannot overlap on the device
};

int cp_fftlog_large_create(cp_fftlog_large** out, int n, int npad, int nker, const double* pre, const double* post, const double* u_re_im, int device) {
    *out = nullptr;
    const FftApi* api = fft_api();
    if (!api) return cp::fail(CP_EUNSUPPORTED, "padded size %d needs the hipFFT path (Np > %d) and libhipfft.so.0 cannot be loaded: %s", npad, 8192, dlerror());
    if (npad > (1 << 24)) return cp::fail(CP_EUNSUPPORTED, "padded size %d is beyond the supported range (2^24)", npad);
    cp_fftlog_large* p = new (std::nothrow) cp_fftlog_large();
    if (!p) return cp::fail(CP_ENOMEM, "cp_fftlog_plan_create: host allocation failed");
    p->n = n; p->npad = npad; p->nker = nker; p->device = device;
    p->d_pre = p->d_post = p->d_work = nullptr;
    p->d_u = p->d_cplx = nullptr;
    p->d2z = p->z2d = nullptr;
    p->done = nullptr;
    p->has_done = false;
    const int nh = npad / 2 + 1;
    long long chunk = (256LL << 20) / ((long long)npad * 24);   // ~256 MB of scratch
    if (chunk < 1) chunk = 1;
    if (chunk > 4096) chunk = 4096;
    chunk = (chunk + nker - 1) / nker * nker;                   // whole batch items: row r uses kernel r % nker
    p->chunk = chunk;
    const size_t tb = (size_t)nker * npad * sizeof(double);
    bool ok = hipMalloc(&p->d_pre, tb) == hipSuccess && hipMalloc(&p->d_post, tb) == hipSuccess &&
              hipMalloc(&p->d_u, (size_t)nker * nh * sizeof(double2)) == hipSuccess &&
              hipMalloc(&p->d_work, (size_t)chunk * npad * sizeof(double)) == hipSuccess &&
              hipMalloc(&p->d_cplx, (size_t)chunk * nh * sizeof(double2)) == hipSuccess;
    ok = ok && hipMemcpy(p->d_pre, pre, tb, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(p->d_post, post, tb, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_u, u_re_im, (size_t)nker * nh * sizeof(double2), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        cp_fftlog_large_destroy(p);
        return cp::fail(CP_ENOMEM, "cp_fftlog_plan_create: cannot allocate the tables / scratch of the large-size path on device %d", device);
    }
    if (hipEventCreateWithFlags(&p->done, hipEventDisableTiming) != hipSuccess) {
        p->done = nullptr;
        cp_fftlog_large_destroy(p);
        return cp::fail(CP_EDEVICE, "cp_fftlog_plan_create: cannot create the stream-ordering event of the large-size path");
    }
    if (api->plan1d(&p->d2z, npad, FFT_D2Z, (int)chunk) != FFT_SUCCESS || api->plan1d(&p->z2d, npad, FFT_Z2D, (int)chunk) != FFT_SUCCESS) {
        cp_fftlog_large_destroy(p);
        return cp::fail(CP_EDEVICE, "cp_fftlog_plan_create: hipFFT cannot plan %lld transforms of size %d", chunk, npad);
    }
    *out = p;
    return CP_OK;
}

void cp_fftlog_large_destroy(cp_fftlog_large* p) {
    if (!p) return;
    const FftApi* api = fft_api();
    if (api && p->d2z) (void)api->destroy(p->d2z);
    if (api && p->z2d) (void)api->destroy(p->z2d);
    if (p->d_pre) (void)hipFree(p->d_pre);
    if (p->d_post) (void)hipFree(p->d_post);
    if (p->d_u) (void)hipFree(p->d_u);
    if (p->d_work) (void)hipFree(p->d_work);
    if (p->d_cplx) (void)hipFree(p->d_cplx);
    if (p->done) (void)hipEventDestroy(p->done);
    delete p;
}

int cp_fftlog_large_execute(cp_fftlog_large* p, const double* d_in, double* d_out, long long nbatch, int ext_l, double val_l, int ext_r, double val_r,
                            int keep_padding, hipStream_t st) {
    const FftApi* api = fft_api();
    if (!api) return cp::fail(CP_EUNSUPPORTED, "cp_fftlog_execute: hipFFT is not available");
    const int nh = p->npad / 2 + 1;
    const int in_left = (p->npad - p->n) / 2, out_left = (p->npad - p->n) - (p->npad - p->n) / 2;   // fftlog.py:152-153
    const int out_off = keep_padding ? 0 : out_left, n_out = keep_padding ? p->npad : p->n;
    const long long nrows_total = nbatch * p->nker;
    std::lock_guard<std::mutex> guard(p->lock);
    if (api->set_stream(p->d2z, st) != FFT_SUCCESS || api->set_stream(p->z2d, st) != FFT_SUCCESS)
        return cp::fail(CP_EDEVICE, "cp_fftlog_execute: hipfftSetStream failed");
    if (p->has_done && hipStreamWaitEvent(st, p->done, 0) != hipSuccess)   // the previous execute (any stream) is done with the scratch
        return cp::fail(CP_EDEVICE, "cp_fftlog_execute: hipStreamWaitEvent failed");
    for (long long r0 = 0; r0 < nrows_total; r0 += p->chunk) {
        const long long nrows = nrows_total - r0 < p->chunk ? nrows_total - r0 : p->chunk;
        PadArgs A;
        A.in = d_in + r0 * p->n; A.work = p->d_work; A.pre = p->d_pre; A.nrows = nrows;
        A.n = p->n; A.npad = p->npad; A.nker = p->nker; A.in_left = in_left;
        A.ext_l = ext_l; A.ext_r = ext_r; A.val_l = val_l; A.val_r = val_r;
        hipLaunchKernelGGL(pad_pre_kernel, dim3(blocks(nrows * p->npad)), dim3(256), 0, st, A);
        if (nrows < p->chunk)   // the FFT plans always transform `chunk` rows: keep the tail of the scratch finite
            (void)hipMemsetAsync(p->d_work + nrows * p->npad, 0, (size_t)(p->chunk - nrows) * p->npad * sizeof(double), st);
        if (api->exec_d2z(p->d2z, p->d_work, p->d_cplx) != FFT_SUCCESS) return cp::fail(CP_EDEVICE, "cp_fftlog_execute: hipfftExecD2Z failed");
        hipLaunchKernelGGL(mul_u_conj_kernel, dim3(blocks(nrows * nh)), dim3(256), 0, st, p->d_cplx, p->d_u, nrows, nh, p->nker);
        if (api->exec_z2d(p->z2d, p->d_cplx, p->d_work) != FFT_SUCCESS) return cp::fail(CP_EDEVICE, "cp_fftlog_execute: hipfftExecZ2D failed");
        hipLaunchKernelGGL(post_crop_kernel, dim3(blocks(nrows * n_out)), dim3(256), 0, st, p->d_work, p->d_post, d_out + r0 * n_out, nrows, p->npad, p->nker,
                           out_off, n_out);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_fftlog_execute: launch failed: %s", hipGetErrorString(e));
    if (hipEventRecord(p->done, st) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_fftlog_execute: hipEventRecord failed");
    p->has_done = true;
    return CP_OK;
}
