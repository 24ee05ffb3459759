// cp_pipeline.hip -- sigma(r, z) of a batch of analytic cosmologies as ONE call: P(k) -> TophatVariance FFTLog -> natural spline to r,
// times the growth factor, root taken, (ncosmo, nr, nz) written once (reference interpolator.py:846-875 with :200-292, for interpolators built
// from an analytic engine's callable + growth factor, eisenstein_hu.py:295-329).
//
// The three kernels have different bounds -- the P(k) evaluation and the FFTLog the vector ALUs, the (nr x nz) store HBM writes -- and on one
// stream they run one after the other (0.17 + 0.05 + 0.25 ms for 10 000 cosmologies).  Here the batch is cut in blocks: the caller's stream
// evaluates and transforms block after block, a second stream owned by the library stores block i (behind an event) while the caller's
// stream is already on block i + 1, and the caller's stream finally waits for the last store.  Host side: 4 launches per block from C, no
// Python between them.  Nothing is allocated: rows, variances and fit coefficients live in the caller's workspace.
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_internal.h"

extern "C" int cp_sigma_rz_fused_available(const cp_fftlog_plan* fftlog, const cp_spline_plan* spline);
int cp_sigma_rz_fused(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
                      const double* d_k,
                      const cp_fftlog_plan* fftlog, const cp_spline_plan* spline, const double* d_growth_sq, int nz, double* d_out, double* d_pk_out,
                      void* d_coef, int device, void* stream);

namespace {

constexpr int MAX_BLOCKS = 16;

struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t ready[MAX_BLOCKS] = {nullptr};
    hipEvent_t begin = nullptr, done = nullptr;
    bool ok = false;
};

// one per device, created on first use (like a plan: not inside the steady-state call)
SideStream* side_stream(int device) {
    static SideStream cache[64];
    static std::mutex lock;
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> guard(lock);
    SideStream& s = cache[device];
    if (!s.ok) {
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (int i = 0; i < MAX_BLOCKS; ++i)
            if (hipEventCreateWithFlags(&s.ready[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&s.begin, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return nullptr;
        s.ok = true;
    }
    return &s;
}

void offset_params(const cp_param* in, cp_param* out, int n, long long first) {
    for (int i = 0; i < n; ++i) {
        out[i] = in[i];
        if (in[i].ptr) out[i].ptr = in[i].ptr + first;
    }
}

}  // namespace

extern "C" long long cp_sigma_rz_workspace_bytes(long long ncosmo, int nk) {
    if (ncosmo < 0 || nk < 0) return -1;
    return 2 * ncosmo * (long long)nk * (long long)sizeof(double) + cp_power_workspace_bytes(ncosmo) + 192 + 3 * (long long)nk * (long long)sizeof(double);      // ... + log k, k^1.08, k^1.4
}

extern "C" int cp_sigma_rz_analytic(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm,
                                    const cp_param* pk_params, int nk,
                                    const double* d_k, const cp_fftlog_plan* fftlog, const cp_spline_plan* spline, const double* d_growth_sq, int nz,
                                    double* d_out, double* d_pk_out, void* d_work, int nblocks, int device, void* stream) {
    if (ncosmo < 0 || nk <= 0 || nz <= 0) return cp::fail(CP_EINVAL, "cp_sigma_rz_analytic: bad sizes");
    if (ncosmo == 0) return CP_OK;
    if (!bg_params || !pk_params || !d_k || !fftlog || !spline || !d_growth_sq || !d_out || !d_work)
        return cp::fail(CP_EINVAL, "cp_sigma_rz_analytic: null pointer");
    int n_spline = 0, nq = 0;
    int st = cp_spline_plan_info(spline, &n_spline, &nq, nullptr);
    if (st != CP_OK) return st;
    if (n_spline != nk) return cp::fail(CP_EINVAL, "cp_sigma_rz_analytic: the spline plan has %d knots, the spectra %d samples", n_spline, nk);
    if (nblocks <= 0) {      // the one-kernel route (cp_sigma.hip) where the plans are the ones it is written for, else one block on one stream
        if (cp_sigma_rz_fused_available(fftlog, spline)) {
            char* coef = static_cast<char*>(d_work);
            coef += (64 - (reinterpret_cast<unsigned long long>(coef) & 63u)) & 63u;
            return cp_sigma_rz_fused(engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, d_k, fftlog, spline, d_growth_sq, nz, d_out, d_pk_out, coef, device, stream);
        }
        nblocks = 1;
    }
    if (nblocks > MAX_BLOCKS) nblocks = MAX_BLOCKS;
    if (nblocks > ncosmo) nblocks = (int)ncosmo;
    hipStream_t main = static_cast<hipStream_t>(stream);
    SideStream* side = nblocks > 1 ? side_stream(device) : nullptr;
    if (nblocks > 1 && !side) return cp::fail(CP_EDEVICE, "cp_sigma_rz_analytic: cannot create the second stream on device %d", device);
    double* work = static_cast<double*>(d_work);
    double* rows = d_pk_out ? d_pk_out : work;      // the spectra are kept where the caller wants them: (ncosmo, nk), nothing behind them is the callee's
    double* var = d_pk_out ? work : work + ncosmo * (long long)nk;      // variances and coefficients always live in the workspace
    char* coef = reinterpret_cast<char*>(var + ncosmo * (long long)nk);
    coef += (64 - (reinterpret_cast<unsigned long long>(coef) & 63u)) & 63u;
    if (side) {  // what the caller queued before this call (growth factors, the result buffer) is ready once its stream gets here
        if (hipEventRecord(side->begin, main) != hipSuccess || hipStreamWaitEvent(side->stream, side->begin, 0) != hipSuccess)
            return cp::fail(CP_EDEVICE, "cp_sigma_rz_analytic: cannot order the second stream");
    }
    cp_param bg[CP_BG_NPARAMS], pk[CP_PK_NPARAMS];
    const bool massive = ncdm && ncdm->nspecies > 0 && ncdm->tab;
    for (int i = 0; i < nblocks; ++i) {
        const long long first = ncosmo * i / nblocks, count = ncosmo * (i + 1) / nblocks - first;
        offset_params(bg_params, bg, CP_BG_NPARAMS, first);
        offset_params(pk_params, pk, CP_PK_NPARAMS, first);
        cp_ncdm nu_block{};      // the tables of the block's cosmologies
        if (massive) {
            nu_block = *ncdm;
            nu_block.tab = ncdm->tab + first * (long long)ncdm->nspecies * 4 * CP_NCDM_NKNOTS;
        }
        st = cp_power_eval(engine, CP_PK_MATTER, count, bg, second_is_omega_m, massive ? &nu_block : ncdm, pk, nk, d_k, nullptr, 0, nullptr, rows + first * nk,
                           coef + cp_power_workspace_bytes(first), device, main);
        if (st != CP_OK) return st;
        st = cp_fftlog_execute(fftlog, rows + first * nk, var + first * nk, count, CP_EXTRAP_CONSTANT, 0., CP_EXTRAP_CONSTANT, 0., 0, main);
        if (st != CP_OK) return st;
        hipStream_t store = main;
        if (side) {
            if (hipEventRecord(side->ready[i], main) != hipSuccess || hipStreamWaitEvent(side->stream, side->ready[i], 0) != hipSuccess)
                return cp::fail(CP_EDEVICE, "cp_sigma_rz_analytic: cannot order the second stream");
            store = side->stream;
        }
        st = cp_spline_apply_outer(spline, var + first * nk, d_growth_sq + first * nz, nz, d_out + first * (long long)nq * nz, count, CP_SPLINE_POST_SQRT, 1.,
                                   store);
        if (st != CP_OK) return st;
    }
    if (side) {
        if (hipEventRecord(side->done, side->stream) != hipSuccess || hipStreamWaitEvent(main, side->done, 0) != hipSuccess)
            return cp::fail(CP_EDEVICE, "cp_sigma_rz_analytic: cannot join the second stream");
    }
    return CP_OK;
}
