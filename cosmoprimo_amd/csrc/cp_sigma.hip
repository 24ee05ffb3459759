// cp_sigma.hip -- sigma(r, z) of a batch of analytic cosmologies as ONE kernel (gfx950):
//     P(k) of two cosmologies evaluated into the row registers -> TophatVariance FFTLog of the pair in LDS -> natural spline of the variances
//     to the radii, out of LDS -> x growth factor, root -> the (nr x nz) results of both cosmologies written once.
// Replaces, for interpolators built from an analytic engine's callable + growth factor (eisenstein_hu.py:295-329), the chain
// PowerSpectrumInterpolator2D.sigma_rz -> integrate_sigma_r2(method='fftlog') -> TophatVariance -> Interpolator1D (interpolator.py:846-875,
// 200-292, jax.py:169-175) that the separate kernels power_kernel -> fftlog_kernel -> spline_outer_kernel run one after the other.
//
// Why one kernel: the three stages have different bounds -- evaluating P(k) (~390 fp64 instructions per sample) and the FFTLog the vector ALUs,
// the (nr x nz) store HBM writes (131 KB per cosmology against 8 KB of spectrum) -- and as separate launches they add up: 0.17 + 0.05 + 0.25 ms
// for 10 000 cosmologies.  Two HIP streams co-run them, but the blocks needed for that cost more in launches and small-kernel tails than
// the overlap returns (tools/bench_config3_streams.py).  Inside one persistent kernel every workgroup alternates between the ALU-bound and
// the store-bound stage of ITS pair, different workgroups are in different stages, and the chip sees both kinds of work all the time; the
// spectra and variances never leave the CU.
//
// Mapping: one workgroup of 128 threads (Np = 2048, 16 points per thread) = one pair of cosmologies per iteration of a persistent loop; the
// FFT phases are those of the FFTLog kernel (cp_fftlog_body.h, input mode IN_HALF_ZERO_GEN).  LDS: the 32 KB of the pair + twiddles, reused
// for the 1024 cropped variances of both rows (natural order, 16-byte slots) once the last butterfly has read its inputs, + (2 nq + 2 nz)
// doubles for the roots of the splined variances and of the growth factors.
#include <hip/hip_runtime.h>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_fftlog_body.h"
#include "cp_internal.h"
#include "cp_power_eval.h"

namespace {

using namespace cpfft;
using namespace cppower;

constexpr int NP = 2048, P = 16;
#ifndef CP_SIGMA_ILP
#define CP_SIGMA_ILP 2
#endif
#ifndef CP_SIGMA_ABLATE      // diagnostic builds (tools/sigma_ablate.sh; wrong results): 1 no P(k) evaluation, 2 no spline, 4 no stores, 8 no FFT phases
#define CP_SIGMA_ABLATE 0
#endif

typedef double cp_v2d __attribute__((ext_vector_type(2)));

struct SigmaArgs {
    FftlogArgs fft;               // tables of the TophatVariance plan; in / out unused
    long long ncosmo;
    Param bg[CP_BG_NPARAMS];
    Param pw[CP_PK_NPARAMS];
    int second_is_omega_m;
    const double* k;              // (n) wavenumbers of the transform, h/Mpc
    const double* ln_k;           // (n) their logarithms (log_wavenumbers_kernel)
    const EhScalars* scal;        // (ncosmo) fit coefficients (cp_power_coefficients), unused for BBKS
    const double* wb;             // (bw, nq) band of the spline operator, query fastest
    const int* j0;                // (nq) first knot of each band, -1: outside the knots
    int bw, nq, nz;
    const double* growth_sq;      // (ncosmo, nz)
    double* out;                  // (ncosmo, nq, nz)
    double* pk_out;               // (ncosmo, n) the spectra themselves, or null
};

// phases 0 .. NPH - 2 of the FFTLog of one pair, with the barriers of run_phases (cp_fftlog_kernel.h)
template <class F, int PH>
__device__ __forceinline__ void front_phases(int t, const FftlogArgs& A, bool has_b, cplx* lds, typename F::State& st) {
    F::template phase<PH>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nullptr, nullptr, 0, st);
    if constexpr (F::template barrier_free_after<PH>()) {
        asm volatile("" ::: "memory");
    } else {
        __syncthreads();
    }
    if constexpr (PH + 1 < F::NPH - 1) front_phases<F, PH + 1>(t, A, has_b, lds, st);
}

// P(k) without the growth factor (eisenstein_hu.py:315-324) of cosmologies ia and ib at the thread's H in-range samples j = t0 + T r, the
// arithmetic of power_kernel (cp_power.hip) so that both routes give the same bits.  The loop over the samples is NOT unrolled (16 evaluations
// of ~390 instructions would be 50 KB of code), so nothing in it may be a register array indexed by r: the wavenumbers and their logarithms
// come from memory (L1 hits), and the results go through the thread's own slots of the FFT's data region, which is free at this point.
// The thread's samples are j = t0 + T r of a geometric grid: k_j and log k_j follow from the thread's first sample by one multiplication /
// addition per step (kh0, ln0 in registers; ratio = k[T] / k[0] and its logarithm are uniform).  No memory is read inside the loop: a vector
// load here would make the wave wait for the 256 stores of the previous pair (the vector-memory counter retires in order), which are meant
// to drain under this arithmetic.  (k_j differs from the tabulated one by a few ulp: 1e-15 on P.)
template <int ENGINE, int T, int H>
__device__ __forceinline__ void evaluate_spectrum(const SigmaArgs& S, long long ic, int t0, double kh0, double ln0, double ratio, double ln_ratio, double* slots) {
    const Cosmo c = load_cosmo(S.bg, ic, S.second_is_omega_m);
    double pw[CP_PK_NPARAMS];
#pragma unroll
    for (int i = 0; i < CP_PK_NPARAMS; ++i) pw[i] = S.pw[i].ptr ? S.pw[i].ptr[ic] : S.pw[i].value;
    EhScalars s{};
    if (ENGINE != CP_ENGINE_BBKS) s = S.scal[ic];
    const EhPerCosmology eh = eh_per_cosmology(s, c.h);
    const PkPerCosmology pc = pk_per_cosmology(c, pw);
    double kh = kh0, ln_kh = ln0;
    // CP_SIGMA_ILP samples per iteration: independent chains of logarithms / exponentials / reciprocals for the two waves of a SIMD to interleave
#pragma unroll 1
    for (int r0 = 0; r0 < H; r0 += CP_SIGMA_ILP) {
#pragma unroll
        for (int u = 0; u < CP_SIGMA_ILP; ++u) {
            const int j = t0 + T * (r0 + u);
            double Tk;
            if (ENGINE == CP_ENGINE_BBKS) Tk = transfer_bbks(c.h, c.Omega_cdm, c.Omega_b, kh);
            else Tk = ENGINE == CP_ENGINE_EH ? transfer_eh(eh, kh, ln_kh) : transfer_nowiggle(s, c.h, kh);
            slots[2 * j] = (Tk * Tk) * (kh * pc.pk_unit) * primordial_tilt(pc, ln_kh);
            kh *= ratio;
            ln_kh += ln_ratio;
        }
    }
}

template <int ENGINE, int T, int H>
__device__ __forceinline__ void evaluate_spectra(const SigmaArgs& S, long long ia, long long ib, int t0, double kh0, double ln0, double ratio,
                                                 double ln_ratio, cplx* lds, double* va, double* vb) {
    double* slots = reinterpret_cast<double*>(lds);      // (re, im) of slot j = (row a, row b) at sample j
    evaluate_spectrum<ENGINE, T, H>(S, ia, t0, kh0, ln0, ratio, ln_ratio, slots);
    evaluate_spectrum<ENGINE, T, H>(S, ib, t0, kh0, ln0, ratio, ln_ratio, slots + 1);
#pragma unroll
    for (int r = 0; r < H; ++r) {      // the thread's own slots: no barrier
        va[r] = slots[2 * (t0 + T * r)];
        vb[r] = slots[2 * (t0 + T * r) + 1];
    }
}

// log of the wavenumbers with the kernels' own logarithm (what power_kernel evaluates per sample), once per launch
__global__ void log_wavenumbers_kernel(const double* k, double* ln_k, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ln_k[i] = log_pos(k[i]);
}

// the cropped outputs of the pair, variance j = t0 + T s of (row a, row b), into the (now free) data region in natural order, 16-byte slots
template <int T, int H>
CP_HD void keep_variances(int t0, cplx* lds, const double* ya, const double* yb) {
    const LdsView v(lds);
#pragma unroll
    for (int s = 0; s < H; ++s) v.write(v.l0 + (unsigned)(t0 + T * s) * 16u, cplx{ya[s], yb[s]});
}

// natural spline to the radii: query q = sum_i wb[i, q] var[j0[q] + i] (cp_spline.hip: the same order of additions), root taken.  The band
// weights come from L2 (bw x nq doubles per plan): they are fetched CHUNK at a time ahead of the multiply-adds that use them, for the thread's
// two queries q and q + T at once, so that the loop waits for memory once per chunk instead of once per weight.
template <int T>
CP_HD void spline_to_radii(int t, const cplx* lds, const double* __restrict__ wb, const int* __restrict__ j0s, int bw, int nq, double* roots_r, bool root = true) {
    constexpr int CHUNK = 8;
    const LdsView v(lds);
    for (int q = t; q < nq; q += 2 * T) {
        const int q2 = q + T < nq ? q + T : q;      // the second query of the thread (the first again when there is none: not stored)
        const int j0a = j0s[q], j0b = j0s[q2];
        double acc[4] = {0., 0., 0., 0.};             // (row a, row b) of query q, then of query q2
        for (int i0 = 0; i0 < bw; i0 += CHUNK) {
            double wa[CHUNK], wq[CHUNK];
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) {
                const int i = i0 + u < bw ? i0 + u : bw - 1;
                wa[u] = wb[(long long)i * nq + q];
                wq[u] = wb[(long long)i * nq + q2];
            }
#pragma unroll
            for (int u = 0; u < CHUNK; ++u) {
                if (i0 + u < bw) {
                    int ja = (j0a < 0 ? 0 : j0a) + i0 + u, jb = (j0b < 0 ? 0 : j0b) + i0 + u;
                    ja = ja < NP / 2 ? ja : NP / 2 - 1;      // padded band entries carry w = 0
                    jb = jb < NP / 2 ? jb : NP / 2 - 1;
                    const cplx ya = v.read(v.l0 + (unsigned)ja * 16u), yb = v.read(v.l0 + (unsigned)jb * 16u);
                    acc[0] = fma(wa[u], ya.re, acc[0]);
                    acc[1] = fma(wa[u], ya.im, acc[1]);
                    acc[2] = fma(wq[u], yb.re, acc[2]);
                    acc[3] = fma(wq[u], yb.im, acc[3]);
                }
            }
        }
        const double nan = __builtin_nan("");
        if (j0a < 0) acc[0] = acc[1] = nan;
        if (j0b < 0) acc[2] = acc[3] = nan;
        roots_r[q] = root ? sqrt(acc[0]) : acc[0];
        roots_r[nq + q] = root ? sqrt(acc[1]) : acc[1];
        if (q2 != q) {
            roots_r[q2] = root ? sqrt(acc[2]) : acc[2];
            roots_r[nq + q2] = root ? sqrt(acc[3]) : acc[3];
        }
    }
}

template <int ENGINE>
__global__ __launch_bounds__(NP / P, 2) void sigma_rz_kernel(const SigmaArgs S) {
    using F = Fftlog<NP, P, IN_HALF_ZERO_GEN, OUT_HALF>;
    constexpr int T = F::T, H = F::H, Q = F::Q;
    extern __shared__ __attribute__((aligned(4096))) char smem[];
    cplx* lds = reinterpret_cast<cplx*>(smem);
    double* roots_r = reinterpret_cast<double*>(smem + F::LDS_BYTES);   // (2, nq) sqrt of the splined variances of rows a and b
    double* roots_g = roots_r + 2 * S.nq;                               // (2, nz) sqrt of the growth factors
    const int t = threadIdx.x;
    const FftlogArgs& A = S.fft;
    const long long npairs = (S.ncosmo + 1) / 2;
    long long p = blockIdx.x;
    if (p >= npairs) return;
    typename F::State st;
    F::init_state(t, A, nullptr, nullptr, 0, st);
    F::fill_lds_tables(t, A, lds);
    const int t0 = st.t0;
    const double kh0 = S.k[t0], ln0 = S.ln_k[t0], ratio = S.k[T] / S.k[0], ln_ratio = S.ln_k[T] - S.ln_k[0];
    __syncthreads();
    for (; p < npairs; p += gridDim.x) {
        const long long ia = 2 * p;
        const bool has_b = ia + 1 < S.ncosmo;
        const long long ib = has_b ? ia + 1 : ia;
        if (CP_SIGMA_ABLATE & 1) {
#pragma unroll
            for (int r = 0; r < H; ++r) st.va[r] = 1. + 1e-3 * (t + r), st.vb[r] = 2. - 1e-3 * (t + r);
        } else {
            evaluate_spectra<ENGINE, T, H>(S, ia, ib, t0, kh0, ln0, ratio, ln_ratio, lds, st.va, st.vb);
            if (S.pk_out) {      // the caller keeps the spectra (the sigma8 normalisation: the filters ask for them on these wavenumbers next)
#pragma unroll
                for (int r = 0; r < H; ++r) {
                    S.pk_out[ia * (NP / 2) + t0 + T * r] = st.va[r];
                    if (has_b) S.pk_out[ib * (NP / 2) + t0 + T * r] = st.vb[r];
                }
            }
        }
        if (!(CP_SIGMA_ABLATE & 8)) front_phases<F, 0>(t, A, has_b, lds, st);
        // ---- last phase of the FFTLog, its outputs kept on the CU ----
        cplx x[P];
        Pass<NP, P, 0>::load_lds(t0, lds, x);
        __syncthreads();      // every thread has its inputs: the data region is free for the outputs
        for (int z = t; z < 2 * S.nz; z += T) {      // (written behind the barrier: nobody is still in the previous pair's store loop)
            const long long ic = z < S.nz ? ia : ib;
            roots_g[z] = sqrt(S.growth_sq[ic * S.nz + (z < S.nz ? z : z - S.nz)]);
        }
        Pass<NP, P, 0>::twiddle_apply(st.w, x);
        Pass<NP, P, 0>::butterflies(x);
        {
            double ya[H], yb[H];
#pragma unroll
            for (int s = 0; s < H; ++s) {
                ya[s] = x[s + Q].re * st.fpost[s];
                yb[s] = x[s + Q].im * st.fpost[s];
            }
            F::template fix_output<H>(st.info_cur, ya, yb);
            keep_variances<T, H>(t0, lds, ya, yb);
        }
        __syncthreads();
        // ---- natural spline to the radii: query q = sum_i wb[i, q] var[j0[q] + i] (cp_spline.hip, same order of additions) ----
        if (CP_SIGMA_ABLATE & 2) {
            for (int q = t; q < 2 * S.nq; q += T) roots_r[q] = 1. + q;
        } else {
            spline_to_radii<T>(t, lds, S.wb, S.j0, S.bw, S.nq, roots_r);
        }
        __syncthreads();
        // ---- out[c, q, z] = sqrt(var[q]) sqrt(growth_sq[z]): the (nq x nz) block of a cosmology is contiguous; 16-byte stores when nz is even ----
        {
            const int nz = S.nz, block = S.nq * nz;
            const bool even = (nz & 1) == 0;
            const int stride = even ? 2 * T : T, e0 = even ? 2 * t : t;
            const int dq = stride / nz, dz = stride - dq * nz, q_first = e0 / nz, z_first = e0 - q_first * nz;
            for (int row = 0; row < ((CP_SIGMA_ABLATE & 4) ? 0 : has_b ? 2 : 1); ++row) {
                double* dst = S.out + (ia + row) * (long long)block;
                const double* vr = roots_r + row * S.nq;
                const double* gr = roots_g + row * nz;
                int q = q_first, z = z_first;
                if (even && dz == 0) {      // nz divides the stride (64 redshifts): the thread's two growth factors do not change
                    const double g0 = gr[z], g1 = gr[z + 1];
#pragma unroll 4
                    for (int e = e0; e < block; e += stride) {
                        cp_v2d val;
                        val.x = vr[q] * g0;
                        val.y = vr[q] * g1;
                        __builtin_nontemporal_store(val, reinterpret_cast<cp_v2d*>(dst + e));      // written once, read by nobody on this device soon
                        q += dq;
                    }
                } else if (even) {
                    for (int e = e0; e < block; e += stride) {
                        double2 val;
                        val.x = vr[q] * gr[z];
                        val.y = vr[q] * gr[z + 1];
                        *reinterpret_cast<double2*>(dst + e) = val;
                        q += dq; z += dz;
                        if (z >= nz) { z -= nz; ++q; }
                    }
                } else {
                    for (int e = e0; e < block; e += stride) {
                        dst[e] = vr[q] * gr[z];
                        q += dq; z += dz;
                        if (z >= nz) { z -= nz; ++q; }
                    }
                }
            }
        }
        // the next pair's phase 0 writes only the data region of the FFT, which nobody reads any more; roots_r / roots_g are rewritten behind
        // barriers of the next pair
    }
}

// ---- the same tail behind rows that come from memory: FFTLog of a batch of rows + natural spline of every output row to the radii (+ root), the
// variances never written.  integrate_sigma_r2(method='fftlog') for tabulated spectra (interpolator.py:285-291): 640 000 rows of config 3B went
// through fftlog_kernel (5.2 GB written), then through the spline kernel (5.2 GB read); here 1.3 GB of results leave the kernel.  Front end,
// prefetch and row screening are those of fftlog_kernel<2048, 16, IN_HALF_ZERO, OUT_HALF> (cp_fftlog_kernel.h); out: (nrows, nq).
struct RowsArgs {
    FftlogArgs fft;
    const double* wb;
    const int* j0;
    int bw, nq, post_sqrt;
    double* out;   // (nbatch, nq)
};

__global__ __launch_bounds__(NP / P, 2) void fftlog_spline_kernel(const RowsArgs R) {
    using F = Fftlog<NP, P, IN_HALF_ZERO, OUT_HALF>;
    constexpr int T = F::T, H = F::H, Q = F::Q;
    extern __shared__ __attribute__((aligned(4096))) char smem[];
    cplx* lds = reinterpret_cast<cplx*>(smem);
    double* roots_r = reinterpret_cast<double*>(smem + F::LDS_BYTES);   // (2, nq)
    const int t = threadIdx.x;
    const FftlogArgs& A = R.fft;
    const long long npairs = (A.nbatch + 1) / 2;
    long long p = blockIdx.x;
    if (p >= npairs) return;
    typename F::State st;
    const long long n = A.n;
    {
        const double* ra = A.in + 2 * p * n;
        F::init_state(t, A, ra, ra + (2 * p + 1 < A.nbatch ? n : 0), 0, st);
    }
    F::fill_lds_tables(t, A, lds);
    F::screen_prefetched(t, st.t0, A, 0, lds, st);
    __syncthreads();
    st.info_nxt = F::screen_collect(lds);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): see fftlog_kernel
    const int t0 = st.t0;
    for (; p < npairs; p += gridDim.x) {
        const bool has_b = 2 * p + 1 < A.nbatch;
        const long long pn = p + gridDim.x < npairs ? p + gridDim.x : p;      // the pair prefetched during this one (itself on the last round)
        const double* nra = A.in + 2 * pn * n;
        const double* nrb = nra + (2 * pn + 1 < A.nbatch ? n : 0);
        // phases 0 .. NPH - 2 (phase 0 issues the next pair's row loads; the phase before the last one screens them)
        {
            F::template phase<0>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nra, nrb, 0, st);
            __syncthreads();
            F::template phase<1>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nra, nrb, 0, st);
            if constexpr (!F::template barrier_free_after<1>()) __syncthreads();
            F::template phase<2>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nra, nrb, 0, st);
            if constexpr (!F::template barrier_free_after<2>()) __syncthreads();
            static_assert(F::NPH == 5, "three passes");
            F::template phase<3>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nra, nrb, 0, st);
            __syncthreads();
        }
        cplx x[P];
        Pass<NP, P, 0>::load_lds(t0, lds, x);
        st.info_nxt = F::screen_collect(lds);      // verdict on the next pair's rows, published in front of the barrier above
        __syncthreads();      // every thread has its inputs: the data region is free for the outputs
        Pass<NP, P, 0>::twiddle_apply(st.w, x);
        Pass<NP, P, 0>::butterflies(x);
        {
            double ya[H], yb[H];
#pragma unroll
            for (int s = 0; s < H; ++s) {
                ya[s] = x[s + Q].re * st.fpost[s];
                yb[s] = x[s + Q].im * st.fpost[s];
            }
            F::template fix_output<H>(st.info_cur, ya, yb);
            keep_variances<T, H>(t0, lds, ya, yb);
        }
        __syncthreads();
        spline_to_radii<T>(t, lds, R.wb, R.j0, R.bw, R.nq, roots_r, R.post_sqrt != 0);
        __syncthreads();
        for (int e = t; e < (has_b ? 2 : 1) * R.nq; e += T) R.out[2 * p * R.nq + e] = roots_r[e];      // rows 2 p and 2 p + 1 are adjacent: one contiguous block
    }
}

template <int ENGINE>
hipError_t launch(const SigmaArgs& S, int grid, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024) (void)cp::allow_full_lds<&sigma_rz_kernel<ENGINE>>();
    hipLaunchKernelGGL(sigma_rz_kernel<ENGINE>, dim3(grid), dim3(NP / P), lds, stream, S);
    return hipGetLastError();
}

}  // namespace

// 1 when cp_sigma_rz_analytic takes the fused kernel for these plans (TophatVariance on 1024 samples padded to 2048, one kernel)
extern "C" int cp_sigma_rz_fused_available(const cp_fftlog_plan* fftlog, const cp_spline_plan* spline) {
    cp_fftlog_tables_view f;
    cp_spline_band_view b;
    if (!cp_fftlog_plan_view(fftlog, &f) || !cp_spline_plan_view(spline, &b)) return 0;
    return f.npad == NP && f.n == NP / 2 && f.nker == 1 && f.in_left == NP / 4 && f.out_left == NP / 4 && b.n == f.n && b.device == f.device;
}

int cp_sigma_rz_fused(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_param* pk_params, const double* d_k,
                      const cp_fftlog_plan* fftlog, const cp_spline_plan* spline, const double* d_growth_sq, int nz, double* d_out, double* d_pk_out,
                      void* d_coef, int device, void* stream) {
    cp_fftlog_tables_view f;
    cp_spline_band_view b;
    if (!cp_fftlog_plan_view(fftlog, &f) || !cp_spline_plan_view(spline, &b)) return cp::fail(CP_EINVAL, "cp_sigma_rz_fused: plans without device tables");
    int st = cp_power_coefficients(engine, ncosmo, bg_params, second_is_omega_m, d_coef, device, stream);
    if (st != CP_OK) return st;
    SigmaArgs S{};
    FftlogArgs& A = S.fft;
    A.in = nullptr; A.out = nullptr; A.nbatch = ncosmo; A.nker = 1; A.n = f.n; A.in_left = f.in_left; A.out_off = f.out_left; A.n_out = f.n;
    A.ext_l = A.ext_r = CP_EXTRAP_CONST; A.val_l = A.val_r = 0.; A.stream_rows = 0;
    A.pre = f.d_pre; A.post = f.d_post; A.u = f.d_u; A.tw = f.d_tw;
    S.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) S.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    for (int i = 0; i < CP_PK_NPARAMS; ++i) S.pw[i] = Param{pk_params[i].ptr, pk_params[i].value};
    S.second_is_omega_m = second_is_omega_m;
    S.k = d_k;
    S.scal = static_cast<const EhScalars*>(d_coef);
    double* ln_k = reinterpret_cast<double*>(static_cast<char*>(d_coef) + ((cp_power_workspace_bytes(ncosmo) + 63) / 64) * 64);      // behind the coefficients
    S.ln_k = ln_k;
    S.wb = b.d_wb; S.j0 = b.d_j0; S.bw = b.bw; S.nq = b.nq; S.nz = nz;
    S.growth_sq = d_growth_sq;
    S.out = d_out;
    S.pk_out = d_pk_out;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma_rz_fused: cannot select device %d", device);
    using F = Fftlog<NP, P, IN_HALF_ZERO_GEN, OUT_HALF>;
    const size_t lds = (size_t)F::LDS_BYTES + (size_t)(2 * b.nq + 2 * nz) * sizeof(double);
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
    const long long npairs = (ncosmo + 1) / 2;
    const size_t per_cu = lds ? (160 * 1024) / lds : 4;
    const long long resident = (long long)(ncu > 0 ? ncu : 256) * (long long)(per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    // every workgroup the same number of pairs (5000 pairs on 1024 resident workgroups would leave a fifth of the chip idle in the last round)
    const long long rounds = (npairs + resident - 1) / resident;
    const int grid = (int)((npairs + rounds - 1) / rounds);
    hipError_t e = hipSuccess;
    if (lds > 160 * 1024) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return cp::fail(CP_EUNSUPPORTED, "cp_sigma_rz_fused: %d radii x %d redshifts exceed the LDS staging", b.nq, nz);
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(log_wavenumbers_kernel, dim3((f.n + 255) / 256), dim3(256), 0, hs, d_k, ln_k, f.n);
    if (engine == CP_ENGINE_EH) e = launch<CP_ENGINE_EH>(S, grid, lds, hs);
    else if (engine == CP_ENGINE_EH_NOWIGGLE) e = launch<CP_ENGINE_EH_NOWIGGLE>(S, grid, lds, hs);
    else e = launch<CP_ENGINE_BBKS>(S, grid, lds, hs);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma_rz_fused: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

// ---- few radii: sigma^2(r_q) as a linear functional of the spectrum --------------------------------------------------------------------------
// The transform and the spline are linear in P(k): for fixed wavenumbers and radii sigma^2(r_q) = sum_j F[q, j] P(k_j), with F the rows that
// the pipeline (FFTLog, natural spline to r) returns for unit spectra.  For the sigma8 normalisation of a batch of analytic cosmologies (one radius,
// one redshift: eisenstein_hu.py:94-103, 331-342) that is a dot product per cosmology behind the evaluation of P(k) -- no transform, no LDS, no
// barrier: one wave per cosmology, a lane evaluates nk / 64 samples (the arithmetic of power_kernel on the tabulated log k: the same bits),
// multiplies them into its NQ partial sums, the wave adds them up.  16 384 cosmologies: see DESIGN.md section 4 (config 4).
namespace {

constexpr int FUNCTIONAL_MAX_NQ = 4;

struct FunctionalArgs {
    long long ncosmo;
    Param bg[CP_BG_NPARAMS];
    Param pw[CP_PK_NPARAMS];
    int second_is_omega_m;
    int nk, nq, nz;
    const double* k;              // (nk) wavenumbers, h/Mpc
    const double* ln_k;           // (nk) their logarithms (log_wavenumbers_kernel)
    const EhScalars* scal;        // (ncosmo) fit coefficients, unused for BBKS
    const double* functional;     // (nq, nk)
    const double* growth_sq;      // (ncosmo, nz)
    double* out;                  // (ncosmo, nq, nz)
    double* pk_out;               // (ncosmo, nk) or null
};

template <int ENGINE>
__global__ __launch_bounds__(256) void sigma_functional_kernel(const FunctionalArgs S) {
    const int lane = threadIdx.x & 63;
    const long long ic = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ic >= S.ncosmo) return;
    const Cosmo c = load_cosmo(S.bg, ic, S.second_is_omega_m);
    double pw[CP_PK_NPARAMS];
#pragma unroll
    for (int i = 0; i < CP_PK_NPARAMS; ++i) pw[i] = S.pw[i].ptr ? S.pw[i].ptr[ic] : S.pw[i].value;
    EhScalars s{};
    if (ENGINE != CP_ENGINE_BBKS) s = S.scal[ic];
    const EhPerCosmology eh = eh_per_cosmology(s, c.h);
    const PkPerCosmology pc = pk_per_cosmology(c, pw);
    double acc[FUNCTIONAL_MAX_NQ];
#pragma unroll
    for (int q = 0; q < FUNCTIONAL_MAX_NQ; ++q) acc[q] = 0.;
#pragma unroll 2
    for (int j = lane; j < S.nk; j += 64) {
        const double kh = S.k[j], ln_kh = S.ln_k[j];
        double Tk;
        if (ENGINE == CP_ENGINE_BBKS) Tk = transfer_bbks(c.h, c.Omega_cdm, c.Omega_b, kh);
        else Tk = ENGINE == CP_ENGINE_EH ? transfer_eh(eh, kh, ln_kh) : transfer_nowiggle(s, c.h, kh);
        const double pk = (Tk * Tk) * (kh * pc.pk_unit) * primordial_tilt(pc, ln_kh);
        if (S.pk_out) S.pk_out[ic * S.nk + j] = pk;
#pragma unroll
        for (int q = 0; q < FUNCTIONAL_MAX_NQ; ++q)
            if (q < S.nq) acc[q] = fma(S.functional[q * S.nk + j], pk, acc[q]);
    }
#pragma unroll
    for (int q = 0; q < FUNCTIONAL_MAX_NQ; ++q)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc[q] += __shfl_xor(acc[q], off);
    for (int e = lane; e < S.nq * S.nz; e += 64) {
        const int q = e / S.nz, z = e - q * S.nz;
        double v = acc[0];
#pragma unroll
        for (int qq = 1; qq < FUNCTIONAL_MAX_NQ; ++qq) v = q == qq ? acc[qq] : v;
        S.out[ic * (long long)(S.nq * S.nz) + e] = sqrt(v) * sqrt(S.growth_sq[ic * S.nz + z]);      // the fused kernel's product of roots
    }
}

}  // namespace

extern "C" int cp_sigma_rz_functional(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_param* pk_params, int nk,
                                      const double* d_k, const double* d_functional, int nq, const double* d_growth_sq, int nz, double* d_out,
                                      double* d_pk_out, void* d_work, int device, void* stream) {
    if (ncosmo < 0 || nk <= 0 || nz <= 0 || nq <= 0) return cp::fail(CP_EINVAL, "cp_sigma_rz_functional: bad sizes");
    if (nq > FUNCTIONAL_MAX_NQ) return cp::fail(CP_EUNSUPPORTED, "cp_sigma_rz_functional: %d radii (at most %d: use cp_sigma_rz_analytic)", nq, FUNCTIONAL_MAX_NQ);
    if (engine != CP_ENGINE_EH && engine != CP_ENGINE_EH_NOWIGGLE && engine != CP_ENGINE_BBKS) return cp::fail(CP_EINVAL, "cp_sigma_rz_functional: unknown engine %d", engine);
    if (ncosmo == 0) return CP_OK;
    if (!bg_params || !pk_params || !d_k || !d_functional || !d_growth_sq || !d_out || !d_work) return cp::fail(CP_EINVAL, "cp_sigma_rz_functional: null pointer");
    char* coef = static_cast<char*>(d_work);
    coef += (64 - (reinterpret_cast<unsigned long long>(coef) & 63u)) & 63u;
    int st = cp_power_coefficients(engine, ncosmo, bg_params, second_is_omega_m, coef, device, stream);
    if (st != CP_OK) return st;
    FunctionalArgs S{};
    S.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) S.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    for (int i = 0; i < CP_PK_NPARAMS; ++i) S.pw[i] = Param{pk_params[i].ptr, pk_params[i].value};
    S.second_is_omega_m = second_is_omega_m;
    S.nk = nk; S.nq = nq; S.nz = nz;
    S.k = d_k;
    S.scal = reinterpret_cast<const EhScalars*>(coef);
    double* ln_k = reinterpret_cast<double*>(coef + ((cp_power_workspace_bytes(ncosmo) + 63) / 64) * 64);      // behind the coefficients (cp_sigma_rz_workspace_bytes)
    S.ln_k = ln_k;
    S.functional = d_functional;
    S.growth_sq = d_growth_sq;
    S.out = d_out;
    S.pk_out = d_pk_out;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma_rz_functional: cannot select device %d", device);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(log_wavenumbers_kernel, dim3((nk + 255) / 256), dim3(256), 0, hs, d_k, ln_k, nk);
    const unsigned grid = (unsigned)((ncosmo + 3) / 4);
    if (engine == CP_ENGINE_EH) hipLaunchKernelGGL(sigma_functional_kernel<CP_ENGINE_EH>, dim3(grid), dim3(256), 0, hs, S);
    else if (engine == CP_ENGINE_EH_NOWIGGLE) hipLaunchKernelGGL(sigma_functional_kernel<CP_ENGINE_EH_NOWIGGLE>, dim3(grid), dim3(256), 0, hs, S);
    else hipLaunchKernelGGL(sigma_functional_kernel<CP_ENGINE_BBKS>, dim3(grid), dim3(256), 0, hs, S);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma_rz_functional: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

// FFTLog of (nbatch, n) rows followed by the spline of every output row to the plan's queries (root taken when post_op is CP_SPLINE_POST_SQRT), as one
// kernel; d_out : (nbatch, nq).  Plans as for cp_sigma_rz_fused_available.
extern "C" int cp_fftlog_spline_execute(const cp_fftlog_plan* fftlog, const cp_spline_plan* spline, const double* d_in, double* d_out, long long nbatch,
                                        int post_op, void* stream) {
    if (nbatch < 0) return cp::fail(CP_EINVAL, "cp_fftlog_spline_execute: negative batch");
    if (nbatch == 0) return CP_OK;
    if (!d_in || !d_out) return cp::fail(CP_EINVAL, "cp_fftlog_spline_execute: null device pointer");
    if (post_op != CP_SPLINE_POST_NONE && post_op != CP_SPLINE_POST_SQRT) return cp::fail(CP_EINVAL, "cp_fftlog_spline_execute: unknown post op %d", post_op);
    if (!cp_sigma_rz_fused_available(fftlog, spline)) return cp::fail(CP_EUNSUPPORTED, "cp_fftlog_spline_execute: plans outside the fused kernel's shape (1024 samples padded to 2048, one kernel)");
    cp_fftlog_tables_view f;
    cp_spline_band_view b;
    (void)cp_fftlog_plan_view(fftlog, &f);
    (void)cp_spline_plan_view(spline, &b);
    RowsArgs R{};
    FftlogArgs& A = R.fft;
    A.in = d_in; A.out = nullptr; A.nbatch = nbatch; A.nker = 1; A.n = f.n; A.in_left = f.in_left; A.out_off = f.out_left; A.n_out = f.n;
    A.ext_l = A.ext_r = CP_EXTRAP_CONST; A.val_l = A.val_r = 0.;
    A.stream_rows = (double)nbatch * f.n * 8. > 512. * 1024. * 1024.;
    A.pre = f.d_pre; A.post = f.d_post; A.u = f.d_u; A.tw = f.d_tw;
    R.wb = b.d_wb; R.j0 = b.d_j0; R.bw = b.bw; R.nq = b.nq; R.post_sqrt = post_op == CP_SPLINE_POST_SQRT;
    R.out = d_out;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != f.device && hipSetDevice(f.device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_fftlog_spline_execute: cannot select device %d", f.device);
    using F = Fftlog<NP, P, IN_HALF_ZERO, OUT_HALF>;
    const size_t lds = (size_t)F::LDS_BYTES + (size_t)(2 * b.nq) * sizeof(double);
    if (lds > 160 * 1024) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return cp::fail(CP_EUNSUPPORTED, "cp_fftlog_spline_execute: %d queries exceed the LDS staging", b.nq);
    }
    if (lds > 64 * 1024) (void)cp::allow_full_lds<&fftlog_spline_kernel>();
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, f.device);
    const long long npairs = (nbatch + 1) / 2;
    const size_t per_cu = (160 * 1024) / lds;
    const long long resident = (long long)(ncu > 0 ? ncu : 256) * (long long)(per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    const long long rounds = (npairs + resident - 1) / resident;
    const int grid = (int)((npairs + rounds - 1) / rounds);
    hipLaunchKernelGGL(fftlog_spline_kernel, dim3(grid), dim3(NP / P), lds, static_cast<hipStream_t>(stream), R);
    const hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_fftlog_spline_execute: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
