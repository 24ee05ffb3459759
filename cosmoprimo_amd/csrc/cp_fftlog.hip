// cp_fftlog.hip -- fused FFTLog kernel for gfx950 + plan management + C ABI.
//
// Replaces the body of FFTlog.__call__ (reference cosmoprimo/fftlog.py:198-241): pad ->
// x prefactor -> rfft -> x u -> irfft(conj) -> x postfactor -> crop, as ONE kernel: one HBM read
// and one HBM write per row (16 N bytes), everything else in registers / LDS.
// Data layout and kernel structure: DESIGN.md "FFTLog kernel"; per-thread phases: cp_fftlog_body.h.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_fftlog_kernel.h"
#include "cp_fftlog_large.h"
#include "cp_internal.h"

using namespace cpfft;

namespace {

bool find_launcher(int npad, Launcher* out) {
    return find_launcher_g0(npad, out) || find_launcher_g1(npad, out) || find_launcher_g2(npad, out) || find_launcher_g3(npad, out) ||
           find_launcher_g4(npad, out);
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct cp_fftlog_plan {
    int n, npad, nker, device;
    int in_left, out_left;
    Launcher l;
    int max_grid[VAR_COUNT];  // resident workgroups on the device (CUs x occupancy) per kernel variant
    double* d_pre;
    double* d_post;
    cplx* d_u;
    cplx* d_tw;
    cp_fftlog_large* large;  // Np > CP_FFTLOG_MAX_NP: the general-size path (cp_fftlog_large.hip), everything above unused
};

#define CP_HIP(call)                                                                               \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            status = cp::fail(e_ == hipErrorOutOfMemory ? CP_ENOMEM : CP_EDEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
            goto done;                                                                             \
        }                                                                                          \
    } while (0)

bool cp_fftlog_plan_view(const cp_fftlog_plan* p, cp_fftlog_tables_view* out) {
    if (!p || p->large || !out) return false;
    *out = cp_fftlog_tables_view{p->n, p->npad, p->nker, p->device, p->in_left, p->out_left, p->d_pre, p->d_post, p->d_u, p->d_tw};
    return true;
}

extern "C" int cp_abi_version(void) { return CP_ABI_VERSION; }

extern "C" const char* cp_last_error(void) { return cp::error_buffer(); }

extern "C" int cp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

extern "C" int cp_fftlog_plan_create(cp_fftlog_plan** out, int n, int npad, int nker, const double* pre, const double* post,
                                     const double* u_re_im, int device) {
    if (!out) return cp::fail(CP_EINVAL, "cp_fftlog_plan_create: null plan pointer");
    *out = nullptr;
    if (!pre || !post || !u_re_im) return cp::fail(CP_EINVAL, "cp_fftlog_plan_create: null table pointer");
    if (n < 2 || nker < 1) return cp::fail(CP_EINVAL, "cp_fftlog_plan_create: need n >= 2 and nker >= 1 (got n=%d, nker=%d)", n, nker);
    if (npad < n || (npad & (npad - 1)) != 0)
        return cp::fail(CP_EINVAL, "cp_fftlog_plan_create: padded size %d must be a power of two >= n=%d (fftlog.py:149-159)", npad, n);
    Launcher l;
    if (npad > (1 << 24)) return cp::fail(CP_EUNSUPPORTED, "cp_fftlog_plan_create: padded size %d is beyond the supported range (2^24)", npad);
    // (the plan stages nker x npad entries of three tables on the host: a request beyond 2^28 entries -- 8 GB -- is refused, not attempted)
    if ((long long)nker * npad > (1LL << 28))
        return cp::fail(CP_EUNSUPPORTED, "cp_fftlog_plan_create: %d transforms of padded size %d: tables of more than 2^28 entries", nker, npad);
    if (npad > CP_FFTLOG_MAX_NP) {  // one packed pair no longer fits the LDS: elementwise kernels around a library FFT
        cp_fftlog_plan* p = new (std::nothrow) cp_fftlog_plan();
        if (!p) return cp::fail(CP_ENOMEM, "cp_fftlog_plan_create: host allocation failed");
        p->n = n; p->npad = npad; p->nker = nker; p->device = device;
        p->in_left = (npad - n) / 2;
        p->out_left = (npad - n) - (npad - n) / 2;
        p->d_pre = p->d_post = nullptr;
        p->d_u = p->d_tw = nullptr;
        p->large = nullptr;
        DeviceGuard guard(device);
        int st = guard.ok ? cp_fftlog_large_create(&p->large, n, npad, nker, pre, post, u_re_im, device)
                          : cp::fail(CP_EDEVICE, "cp_fftlog_plan_create: cannot select device %d", device);
        if (st != CP_OK) {
            delete p;
            return st;
        }
        *out = p;
        return CP_OK;
    }
    if (!find_launcher(npad, &l))
        return cp::fail(CP_EUNSUPPORTED, "cp_fftlog_plan_create: padded size %d outside the LDS-resident kernel range [4, %d]", npad,
                        CP_FFTLOG_MAX_NP);
    int status = CP_OK;
    cp_fftlog_plan* p = new (std::nothrow) cp_fftlog_plan();
    if (!p) return cp::fail(CP_ENOMEM, "cp_fftlog_plan_create: host allocation failed");
    p->large = nullptr;
    p->n = n;
    p->npad = npad;
    p->nker = nker;
    p->device = device;
    p->in_left = (npad - n) / 2;              // fftlog.py:152
    p->out_left = (npad - n) - (npad - n) / 2;  // fftlog.py:153
    p->l = l;
    p->d_pre = p->d_post = nullptr;
    p->d_u = p->d_tw = nullptr;
    {
        DeviceGuard guard(device);
        std::vector<cplx> tw, u((size_t)nker * npad);
        if (!guard.ok) {
            status = cp::fail(CP_EDEVICE, "cp_fftlog_plan_create: cannot select device %d", device);
            goto done;
        }
        l.build_tw(tw);
        for (int k = 0; k < nker; ++k) l.build_u(u_re_im + (size_t)k * 2 * (npad / 2 + 1), u.data() + (size_t)k * npad);
        const size_t tbytes = (size_t)nker * npad * sizeof(double);
        CP_HIP(hipMalloc(&p->d_pre, tbytes));
        CP_HIP(hipMalloc(&p->d_post, tbytes));
        CP_HIP(hipMalloc(&p->d_u, u.size() * sizeof(cplx)));
        CP_HIP(hipMalloc(&p->d_tw, tw.size() * sizeof(cplx)));
        CP_HIP(hipMemcpy(p->d_pre, pre, tbytes, hipMemcpyHostToDevice));
        CP_HIP(hipMemcpy(p->d_post, post, tbytes, hipMemcpyHostToDevice));
        CP_HIP(hipMemcpy(p->d_u, u.data(), u.size() * sizeof(cplx), hipMemcpyHostToDevice));
        CP_HIP(hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice));
        int ncu = 0;
        CP_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
        for (int v = 0; v < VAR_COUNT; ++v) {
            p->max_grid[v] = 0;
            if (!l.func[v]) continue;
            if (l.lds_bytes > 64 * 1024) CP_HIP(hipFuncSetAttribute(l.func[v], hipFuncAttributeMaxDynamicSharedMemorySize, l.lds_bytes));
            int nblk = 0;
            CP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, l.func[v], l.block, l.lds_bytes));
            if (nblk < 1) {
                status = cp::fail(CP_EDEVICE, "cp_fftlog_plan_create: kernel variant %d for npad=%d does not fit on a CU", v, npad);
                goto done;
            }
            p->max_grid[v] = nblk * ncu;
        }
    }
done:
    if (status != CP_OK) {
        cp_fftlog_plan_destroy(p);
        return status;
    }
    *out = p;
    return CP_OK;
}

extern "C" int cp_fftlog_plan_destroy(cp_fftlog_plan* p) {
    if (!p) return CP_OK;
    {
        DeviceGuard guard(p->device);
        if (p->d_pre) (void)hipFree(p->d_pre);
        if (p->d_post) (void)hipFree(p->d_post);
        if (p->d_u) (void)hipFree(p->d_u);
        if (p->d_tw) (void)hipFree(p->d_tw);
        cp_fftlog_large_destroy(p->large);
    }
    delete p;
    return CP_OK;
}

static int grid_for(const cp_fftlog_plan* p, int variant, long long nbatch) {
    const long long npairs = ((nbatch + 1) / 2) * p->nker;
    return (int)(npairs < p->max_grid[variant] ? npairs : p->max_grid[variant]);
}

extern "C" int cp_fftlog_plan_info(const cp_fftlog_plan* p, long long nbatch, int* grid, int* block, int* lds_bytes) {
    if (!p) return cp::fail(CP_EINVAL, "cp_fftlog_plan_info: null plan");
    if (p->large) {  // no persistent kernel on the general-size path
        if (grid) *grid = 0;
        if (block) *block = 0;
        if (lds_bytes) *lds_bytes = 0;
        return CP_OK;
    }
    if (grid) *grid = grid_for(p, select_variant(p->npad, p->l.p, p->n, 0, 0., 0, 0., 0), nbatch);
    if (block) *block = p->l.block;
    if (lds_bytes) *lds_bytes = p->l.lds_bytes;
    return CP_OK;
}

// out_count < 0: whole rows
static int execute_impl(const cp_fftlog_plan* p, const double* d_in, double* d_out, long long nbatch, int extrap_left, double val_left, int extrap_right,
                        double val_right, int keep_padding, int out_first, int out_count, void* stream) {
    if (!p) return cp::fail(CP_EINVAL, "cp_fftlog_execute: null plan");
    if (nbatch < 0) return cp::fail(CP_EINVAL, "cp_fftlog_execute: negative batch");
    if (nbatch == 0) return CP_OK;
    if (!d_in || !d_out) return cp::fail(CP_EINVAL, "cp_fftlog_execute: null device pointer");
    if (extrap_left < CP_EXTRAP_CONSTANT || extrap_left > CP_EXTRAP_LOGLOG || extrap_right < CP_EXTRAP_CONSTANT || extrap_right > CP_EXTRAP_LOGLOG)
        return cp::fail(CP_EINVAL, "cp_fftlog_execute: unknown extrapolation mode (%d, %d)", extrap_left, extrap_right);
    if (p->large) {
        DeviceGuard guard(p->device);
        if (!guard.ok) return cp::fail(CP_EDEVICE, "cp_fftlog_execute: cannot select device %d", p->device);
        return cp_fftlog_large_execute(p->large, d_in, d_out, nbatch, extrap_left, val_left, extrap_right, val_right, keep_padding,
                                       static_cast<hipStream_t>(stream));
    }
    FftlogArgs A;
    A.in = d_in;
    A.out = d_out;
    A.nbatch = nbatch;
    A.nker = p->nker;
    A.n = p->n;
    A.in_left = p->in_left;
    A.out_off = keep_padding ? 0 : p->out_left;
    A.n_out = keep_padding ? p->npad : p->n;
    A.ext_l = extrap_left;
    A.ext_r = extrap_right;
    A.val_l = val_left;
    A.val_r = val_right;
    // once-read / once-written rows bypass the caches when the launch moves more than the Infinity Cache (256 MB) can keep for a consumer
    A.stream_rows = (double)nbatch * p->nker * ((double)p->n + A.n_out) * 8. > 512. * 1024. * 1024. ? 3 : 0;
    A.pre = p->d_pre;
    A.post = p->d_post;
    A.u = p->d_u;
    A.tw = p->d_tw;
    A.out_first = 0;
    A.out_last = A.n_out;
    DeviceGuard guard(p->device);
    if (!guard.ok) return cp::fail(CP_EDEVICE, "cp_fftlog_execute: cannot select device %d", p->device);
    int variant = select_variant(p->npad, p->l.p, p->n, extrap_left, val_left, extrap_right, val_right, keep_padding);
    if (variant == VAR_HALF_ZERO && out_count >= 0 && out_count < A.n_out && p->l.func[VAR_HALF_ZERO_WINDOW]) {      // other kernels store whole rows
        variant = VAR_HALF_ZERO_WINDOW;
        A.out_first = out_first;
        A.out_last = out_first + out_count;
        A.stream_rows = (double)nbatch * p->nker * ((double)p->n + out_count) * 8. > 512. * 1024. * 1024. ? 3 : 0;
    }
    const int grid = grid_for(p, variant, nbatch);
    // the kernel walks the rows with 32-bit element steps (cp_fftlog_kernel.h: PairWalk): nker rows, and 2 grid rows, of
    // the padded length must stay below 2^31 elements (Np <= 8192: nker < 262144)
    if ((long long)p->nker * p->npad >= (1LL << 31) || 2LL * grid * p->npad >= (1LL << 31))
        return cp::fail(CP_EUNSUPPORTED, "cp_fftlog_execute: nker = %d kernels of padded size %d exceed the row-walk range", p->nker, p->npad);
    p->l.launch(variant, A, grid, static_cast<hipStream_t>(stream));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_fftlog_execute: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_fftlog_execute(const cp_fftlog_plan* p, const double* d_in, double* d_out, long long nbatch, int extrap_left,
                                 double val_left, int extrap_right, double val_right, int keep_padding, void* stream) {
    return execute_impl(p, d_in, d_out, nbatch, extrap_left, val_left, extrap_right, val_right, keep_padding, 0, -1, stream);
}

extern "C" int cp_fftlog_execute_window(const cp_fftlog_plan* p, const double* d_in, double* d_out, long long nbatch, int extrap_left, double val_left,
                                        int extrap_right, double val_right, int keep_padding, int out_first, int out_count, void* stream) {
    if (p) {
        const int n_out = keep_padding ? p->npad : p->n;
        if (out_first < 0 || out_count < 0 || out_first > n_out - out_count)
            return cp::fail(CP_EINVAL, "cp_fftlog_execute_window: columns [%d, %d + %d) are not inside the %d of an output row", out_first, out_first, out_count, n_out);
    }
    return execute_impl(p, d_in, d_out, nbatch, extrap_left, val_left, extrap_right, val_right, keep_padding, out_first, out_count, stream);
}
