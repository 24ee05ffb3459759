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

#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include <vector>

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

// the four cubic B-splines of a geometric knot sequence that live on one interval, as polynomials of the position in it (cp_geospline_basis)
struct GeoBasis {
    double k[4][4];             // the B-spline centred on knot j - 1 + i is sum_d k[i][d] x^d on [s_j, s_j+1], x = (r - s_j) / h_j
};

struct SigmaArgs {
    FftlogArgs fft;               // tables of the TophatVariance plan; in / out unused
    long long ncosmo;
    Param bg[CP_BG_NPARAMS];
    Param pw[CP_PK_NPARAMS];
    int second_is_omega_m;
    const double* ncdm_tab;       // massive neutrinos (cp_ncdm.tab; nsp == 0: none): today's densities only -- Omega0_m of pk_callable, Omega_m of BBKS
    int nsp;
    const double* k;              // (n) wavenumbers of the transform, h/Mpc
    const double* ln_k;           // (3, n) log k, k^1.08, k^1.4 (written by the coefficients' launch: cp_power_coefficients)
    const CosmoConsts* consts;    // (ncosmo) the cosmologies' constants (cp_power_coefficients)
    int stagger_div, stagger_mod, stagger_sleeps;      // start offsets between workgroups (see the kernel), 0 sleeps: none
    const double* wb;             // (bw, nq) band of the spline operator, query fastest
    const int* j0;                // (nq) first knot of each band, -1: outside the knots
    int bw, nq, nz;
    // COEF (the transform's u carries the B-spline prefilter: see fftlog_geospline_kernel): the radii's intervals relative to knot ws (-1: outside the
    // knots), their positions x in the interval, the four cubic pieces
    const int* qe;
    const double* qx;
    const GeoBasis* basis;
    int ws;
    const double* growth_sq;      // (ncosmo, nz)
    double* out;                  // (ncosmo, nq, nz)
    double* pk_out;               // (ncosmo, n) the spectra themselves, or null
};

#ifndef CP_SIGMA_DEFER      // 1: the second row of a pair's results stored behind the NEXT pair's wait for U (measurements: tools/sigma_defer.sh)
#define CP_SIGMA_DEFER 0
#endif
#ifndef CP_SIGMA_RZ_TABLES
#define CP_SIGMA_RZ_TABLES 0
#endif

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
__device__ __forceinline__ void evaluate_spectrum(const SigmaArgs& S, long long ic, int t0, double kh0, double ln0, double ratio, double ln_ratio, double2 pw0,
                                                  double2 pw_ratio, double* slots, const MathTables* mt) {
    const CosmoConsts K = load_uniform(S.consts + ic);      // (scalar loads: the cosmology's constants in scalar registers)
    const PkPerCosmology& pc = K.pk;
    double kh = kh0, ln_kh = ln0, kh108 = pw0.x, kh14 = pw0.y;
    // CP_SIGMA_ILP samples per iteration: independent chains of logarithms / exponentials / reciprocals for the two waves of a SIMD to interleave
#pragma unroll 1
    for (int r0 = 0; r0 < H; r0 += CP_SIGMA_ILP) {
#pragma unroll
        for (int u = 0; u < CP_SIGMA_ILP; ++u) {
            const int j = t0 + T * (r0 + u);
            const double Tk = transfer_any<ENGINE>(K, kh, ln_kh, kh108, kh14, mt);
            slots[2 * j] = (Tk * Tk) * (kh * pc.pk_unit) * primordial_tilt(pc, ln_kh, mt);
            kh *= ratio;
            ln_kh += ln_ratio;
            kh108 *= pw_ratio.x;
            kh14 *= pw_ratio.y;
        }
    }
}

template <int ENGINE, int T, int H>
__device__ __forceinline__ void evaluate_spectra(const SigmaArgs& S, long long ia, long long ib, int t0, double kh0, double ln0, double ratio,
                                                 double ln_ratio, double2 pw0, double2 pw_ratio, cplx* lds, double* va, double* vb, const MathTables* mt) {
    double* slots = reinterpret_cast<double*>(lds);      // (re, im) of slot j = (row a, row b) at sample j
    evaluate_spectrum<ENGINE, T, H>(S, ia, t0, kh0, ln0, ratio, ln_ratio, pw0, pw_ratio, slots, mt);
    evaluate_spectrum<ENGINE, T, H>(S, ib, t0, kh0, ln0, ratio, ln_ratio, pw0, pw_ratio, slots + 1, mt);
#pragma unroll
    for (int r = 0; r < H; ++r) {      // the thread's own slots: no barrier
        va[r] = slots[2 * (t0 + T * r)];
        vb[r] = slots[2 * (t0 + T * r) + 1];
    }
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

// The same spline from its B-spline coefficients (the transform carried the prefilter: the data region holds c_j of both rows where the variances would
// be): a radius in [s_j, s_j+1] is c_{j-1} ... c_{j+2} times four cubic weights of its position -- four 16-byte LDS reads (both rows) and 20 multiply-adds
// where the band operator takes ~44 weights from L2 and 88 multiply-adds.
template <int T>
CP_HD void bspline_to_radii(int t, const cplx* lds, const int* __restrict__ qes, const double* __restrict__ qxs, const GeoBasis* basis, int ws, int nq,
                            double* roots_r) {
    const LdsView v(lds);
    const GeoBasis B = load_uniform(basis);      // scalar loads
    for (int q = t; q < nq; q += 2 * T) {
        const int q2 = q + T < nq ? q + T : q;      // the second query of the thread (the first again when there is none: not stored)
        const int ea = (CP_SIGMA_ABLATE & 32) ? 40 + (q >> 1) : qes[q], eb = (CP_SIGMA_ABLATE & 32) ? 40 + (q2 >> 1) : qes[q2];
        const double xa = (CP_SIGMA_ABLATE & 32) ? 0.25 : qxs[q], xb = (CP_SIGMA_ABLATE & 32) ? 0.75 : qxs[q2];
        const int ja = ws + (ea < 0 ? 0 : ea) - 1, jb = ws + (eb < 0 ? 0 : eb) - 1;      // the first of the four coefficients
        cplx ca[4], cb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ca[i] = v.read(v.l0 + (unsigned)(ja + i) * 16u);
            cb[i] = v.read(v.l0 + (unsigned)(jb + i) * 16u);
        }
        double acc[4] = {0., 0., 0., 0.};             // (row a, row b) of query q, then of query q2
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double wa = fma(fma(fma(B.k[i][3], xa, B.k[i][2]), xa, B.k[i][1]), xa, B.k[i][0]);
            const double wq = fma(fma(fma(B.k[i][3], xb, B.k[i][2]), xb, B.k[i][1]), xb, B.k[i][0]);
            acc[0] = fma(wa, ca[i].re, acc[0]);
            acc[1] = fma(wa, ca[i].im, acc[1]);
            acc[2] = fma(wq, cb[i].re, acc[2]);
            acc[3] = fma(wq, cb[i].im, acc[3]);
        }
        const double nan = __builtin_nan("");
        if (ea < 0) acc[0] = acc[1] = nan;
        if (eb < 0) acc[2] = acc[3] = nan;
        roots_r[q] = sqrt(acc[0]);
        roots_r[nq + q] = sqrt(acc[1]);
        if (q2 != q) {
            roots_r[q2] = sqrt(acc[2]);
            roots_r[nq + q2] = sqrt(acc[3]);
        }
    }
}

template <int ENGINE, bool COEF>
__global__ __launch_bounds__(NP / P, 2) void sigma_rz_kernel(const SigmaArgs S) {
    using F = Fftlog<NP, P, IN_HALF_ZERO_GEN, OUT_HALF>;
    constexpr int T = F::T, H = F::H, Q = F::Q;
    extern __shared__ __attribute__((aligned(4096))) char smem[];
    cplx* lds = reinterpret_cast<cplx*>(smem);
    double* roots_r = reinterpret_cast<double*>(smem + F::LDS_BYTES);   // (2, nq) sqrt of the splined variances of rows a and b
    double* roots_g = roots_r + 2 * S.nq;                               // (2, nz) sqrt of the growth factors
    // the table-driven logarithm / exponential of the evaluation (cp_math.h): measured SLOWER here (0.41 against 0.335 ms per 10 000 cosmologies --
    // the gathers sit in dependent chains that this kernel, at two waves per SIMD with the FFT's registers, cannot cover), so off
#if CP_SIGMA_RZ_TABLES
    __shared__ MathTables mt;      // (the barrier behind the FFT tables covers it)
    fill_math_tables(&mt);
    const MathTables* mtp = tables_present(&mt);
#else
    const MathTables* mtp = nullptr;
#endif
    const int t = threadIdx.x;
    const FftlogArgs& A = S.fft;
    const long long npairs = (S.ncosmo + 1) / 2;
    long long p = blockIdx.x;
    if (p >= npairs) return;
    typename F::State st;
    F::init_state(t, A, nullptr, nullptr, 0, st);
    F::fill_lds_tables(t, A, lds);
    const int t0 = st.t0;
    // The thread's first wavenumber, its logarithm, k^1.08 and k^1.4 there and the steps along the geometric grid are the same for every pair.  The four
    // steps are the same in every lane: scalar registers.  The thread's own four values (eight vector registers) do not survive the transform -- it takes
    // what two waves per SIMD leave: held over the loop they were spilled, and a scratch reload is a vector-memory load: it waits for the 256 stores of the
    // previous pair (the counter retires in order), the very thing the evaluation is arranged to avoid; an LDS copy of them, 4 KB, costs the fourth
    // workgroup of the CU (0.33 -> 0.37 ms, measured).  They are FORMED AGAIN at the top of every pair from the thread's index -- log k linear in it, three
    // exponentials: 60 of a pair's ~9 000 instructions -- behind an opaque copy of the index, so that the compiler does not hoist them back out.
    const int nk_tab = NP / 2;
    const double ratio = cp::wave_uniform(S.k[T] / S.k[0]), ln_ratio = cp::wave_uniform(S.ln_k[T] - S.ln_k[0]);
    const double2 pw_ratio = double2{cp::wave_uniform(S.ln_k[nk_tab + T] / S.ln_k[nk_tab]), cp::wave_uniform(S.ln_k[2 * nk_tab + T] / S.ln_k[2 * nk_tab])};
    const double ln_first = cp::wave_uniform(S.ln_k[0]), ln_step = cp::wave_uniform((S.ln_k[nk_tab - 1] - S.ln_k[0]) / (double)(nk_tab - 1));
    __syncthreads();
    // The one-off loads (twiddles, factors) are drained HERE: left in flight into the loop, the compiler's wait counts at its head had to cover this entry path
    // too and became vmcnt(13) in phase 0 and vmcnt(0) behind its barrier of EVERY pair -- waits for the previous pair's stores (the counter retires in
    // order), which are meant to drain under the evaluation and the first two phases (fftlog_kernel does the same: cp_fftlog_kernel.h).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
    // Every workgroup does the same work on its pairs and all start together: left alone they evaluate together and store together -- the chip alternates
    // between its vector ALUs and its memory instead of using both.  The workgroups of a CU start a fraction of a pair apart instead.
    if (S.stagger_sleeps > 0) {
        const int w = (int)((blockIdx.x / S.stagger_div) % S.stagger_mod) * S.stagger_sleeps;
        for (int i = 0; i < w; ++i) __builtin_amdgcn_s_sleep(127);      // 127 x 64 clocks
    }
    // ---- out[c, q, z] = sqrt(var[q]) sqrt(growth_sq[z]): the (nq x nz) block of a cosmology is contiguous; 16-byte stores when nz is even ----
    auto store_rows = [&](long long first, int row_begin, int row_end) {
        int tv = t;
        asm volatile("" : "+v"(tv));      // (the addresses are formed here, not carried through the transform: see the loop)
        const int nz = S.nz, block = S.nq * nz;
        const bool even = (nz & 1) == 0;
        const int stride = even ? 2 * T : T, e0 = even ? 2 * tv : tv;
        const int dq = stride / nz, dz = stride - dq * nz, q_first = e0 / nz, z_first = e0 - q_first * nz;
        for (int row = row_begin; row < ((CP_SIGMA_ABLATE & 4) ? 0 : row_end); ++row) {
            double* dst = S.out + (first + row) * (long long)block;
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
                    if (CP_SIGMA_ABLATE & 64) *reinterpret_cast<cp_v2d*>(dst + e) = val;      // (diagnostic: the ordinary cache policy)
                    else __builtin_nontemporal_store(val, reinterpret_cast<cp_v2d*>(dst + e));      // written once, read by nobody on this device soon
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
    };
    long long pending = -1;      // CP_SIGMA_DEFER: the cosmology whose (second) row of results is still to be stored
    for (; p < npairs; p += gridDim.x) {
        const long long ia = 2 * p;
        const bool has_b = ia + 1 < S.ncosmo;
        const long long ib = has_b ? ia + 1 : ia;
        if (CP_SIGMA_ABLATE & 1) {
#pragma unroll
            for (int r = 0; r < H; ++r) st.va[r] = 1. + 1e-3 * (t + r), st.vb[r] = 2. - 1e-3 * (t + r);
        } else {
            int t0v = t0;
            asm volatile("" : "+v"(t0v));      // (what follows is not loop-invariant for the compiler)
            const double ln0 = fma((double)t0v, ln_step, ln_first), kh0 = exp_mid(ln0);
            const double2 pw0 = double2{exp_mid(1.08 * ln0), exp_mid(1.4 * ln0)};
            evaluate_spectra<ENGINE, T, H>(S, ia, ib, t0, kh0, ln0, ratio, ln_ratio, pw0, pw_ratio, lds, st.va, st.vb, mtp);
            if (S.pk_out) {      // the caller keeps the spectra (the sigma8 normalisation: the filters ask for them on these wavenumbers next)
#pragma unroll
                for (int r = 0; r < H; ++r) {
                    S.pk_out[ia * (NP / 2) + t0 + T * r] = st.va[r];
                    if (has_b) S.pk_out[ib * (NP / 2) + t0 + T * r] = st.vb[r];
                }
            }
        }
        if (CP_SIGMA_DEFER) {
            // the phases written out: the previous pair's second row goes out right behind phase 2 -- behind the wait for U, the pair's only wait for memory
            static_assert(F::NPH == 5, "three passes");
            F::template phase<0>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nullptr, nullptr, 0, st);
            __syncthreads();
            F::template phase<1>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nullptr, nullptr, 0, st);
            if constexpr (!F::template barrier_free_after<1>()) __syncthreads();
            F::template phase<2>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nullptr, nullptr, 0, st);
            if (pending >= 0) store_rows(pending, 1, 2);
            pending = -1;
            if constexpr (!F::template barrier_free_after<2>()) __syncthreads();
            F::template phase<3>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nullptr, nullptr, 0, st);
            __syncthreads();
        } else if (!(CP_SIGMA_ABLATE & 8)) front_phases<F, 0>(t, A, has_b, lds, st);
        // ---- last phase of the FFTLog, its outputs kept on the CU ----
        cplx x[P];
        Pass<NP, P, 0>::load_lds(t0, lds, x);
        __syncthreads();      // every thread has its inputs: the data region is free for the outputs
        // (the addresses the loops below form from the thread's index are formed here, pair by pair, not carried through the transform in registers it
        // does not have -- spilled, their reloads were vector-memory loads at the top of these loops: the copy of the index is opaque to the compiler)
        int tv = t;
        asm volatile("" : "+v"(tv));
        for (int z = tv; z < 2 * S.nz; z += T) {      // (written behind the barrier: nobody is still in the previous pair's store loop)
            const long long ic = z < S.nz ? ia : ib;
            roots_g[z] = (CP_SIGMA_ABLATE & 32) ? sqrt(1. + 1e-3 * (double)(z + ic)) : sqrt(S.growth_sq[ic * S.nz + (z < S.nz ? z : z - S.nz)]);
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
            if constexpr (COEF) bspline_to_radii<T>(tv, lds, S.qe, S.qx, S.basis, S.ws, S.nq, roots_r);
            else spline_to_radii<T>(tv, lds, S.wb, S.j0, S.bw, S.nq, roots_r);
        }
        __syncthreads();
        if (CP_SIGMA_DEFER) {      // the first row now; the second behind the NEXT pair's wait for U (see front of the loop)
            store_rows(ia, 0, 1);
            pending = has_b ? ia : -1;
        } else {
            store_rows(ia, 0, has_b ? 2 : 1);
        }
        // the next pair's phase 0 writes only the data region of the FFT, which nobody reads any more; roots_r / roots_g are rewritten behind
        // barriers of the next pair
    }
    if (CP_SIGMA_DEFER && pending >= 0) store_rows(pending, 1, 2);
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

// ---- the same front end with the spline SOLVED instead of multiplied: natural cubic spline from the (geometric) output grid to the radii, the
// tridiagonal system solved in registers.  fftlog_spline_kernel above fetches a query's band of ~44 weights from L2 per pair of rows (90 KB per
// pair: why it loses beyond a few thousand rows).  On a geometric grid s_i = s_0 rho^i the system has CONSTANT coefficients once the second
// derivatives are scaled by their interval: with N_i = M_i h_{i-1}^2 / 6,
//     rho^2 N_{i-1} + 2 (1 + rho) N_i + N_{i+1} / rho = d_i := (y_{i+1} - y_i) / rho - (y_i - y_{i-1}),
//     s(r) = A y_j + B y_{j+1} + (A^3 - A) rho^2 N_j + (B^3 - B) N_{j+1},   A = (s_{j+1} - r) / h_j,  B = 1 - A,
// and the inverse of a constant tridiagonal matrix is two geometric tails: N_i = kappa (sum_{j <= i} pL^(i-j) d_j + sum_{j > i} pR^(j-i) d_j),
// |pL|, |pR| ~ 0.27: a causal and an anti-causal first-order recursion, F_i = d_i + pL F_{i-1} and B_i = d_i + pR B_{i+1}, N_i = kappa (F_i +
// pR B_{i+1}).  One wave per row; lane l owns the S knots [S l, S l + S) of the stretch the radii see (+ >= 32 knots on either side, 64 S in
// all): it reads its S + 2 values from LDS ONCE, runs both recursions over its own knots from zero, and takes what the knots outside its
// segment contribute from its neighbours' segment totals, handed from lane to lane by DPP shifts of the whole wave (a segment away the weight is
// pL^S: nine lanes reach the 1e-19 that the 32 knots of margin stand for; lanes beyond the stretch count as zero).  No LDS round trip but the
// initial read and the hand-over of N to the evaluation -- a first version that reduced the system cyclically through LDS (five dependent round
// trips, each queueing behind the FFT traffic of the other workgroups of the CU) spent 0.7 ms of 4.1 there (profiles/r4_geospline_ablate.txt).
// Plans whose stretch would leave the interior of the grid are refused and take the operator route.  group > 0: rows come in groups (the
// redshifts of one table) and out is (nbatch / group, nq, group), the transposition PowerSpectrumInterpolator2D.sigma_rz needs, written by the
// kernel itself: the pairs of a group are then handed to workgroups of ONE XCD in the same round (workgroups are dealt round-robin to the 8 XCDs),
// so that the 8-byte pieces of a 512-byte output line meet in that XCD's L2.
//
// COEF (plans of cp_geospline_plan_create_prefiltered): no solve at all.  In the B-spline basis of the geometric knots -- B_j(s) = B_0(s / rho^j):
// the knot sequence is scale invariant -- the spline is s(r) = sum_j c_j B_j(r) and the interpolation conditions are a tridiagonal system with
// CONSTANT coefficients, alpha c_{j-1} + beta c_j + gamma c_{j+1} = y_j.  The FFTLog output is y_j = post_j g_j with post_j = post_0 lambda^j and
// g the inverse DFT of a product, so c_j = post_j ct_j with (alpha / lambda) ct_{j-1} + beta ct_j + gamma lambda ct_{j+1} = g_j, which on the
// periodic padded grid is a division in the frequency domain: the plan's transform carries u_m / conj(alpha / lambda e^{-i w_m} + beta + gamma
// lambda e^{i w_m}) in place of u_m and the kernel's ordinary output IS the coefficient sequence.  Periodic ends on the padded grid instead of
// natural ends on the unpadded one: both are forgotten like 0.27^distance, and the plan refuses radii within GEO_HALO knots of the ends as the
// solve does.  A radius then costs four LDS reads and the four cubic weights (12 + 4 fused multiply-adds, the polynomials in scalar registers).
constexpr int GEO_SMAX = 8;          // knots per lane: stretches of at most 512 knots
constexpr int GEO_SMIN = 4;          // (fewer knots per lane would need more than GEO_REACH lanes of carry)
constexpr int GEO_QMAX = 8;          // queries per lane: at most 512 radii
constexpr int GEO_HALO = 32, GEO_REACH = 9;      // GEO_REACH x GEO_SMIN >= GEO_HALO + 1
#ifndef CP_GEO_EVAL_LAST      // 1: the deferred evaluation of a pair in the NEXT pair's last phase instead of behind its first barrier (measurements)
#define CP_GEO_EVAL_LAST 0
#endif
#ifndef CP_GEO_ABLATE      // diagnostic builds (tools/geospline_ablate.sh; wrong results): 1 no carries between lanes, 2 one query per lane, 4 no root,
#define CP_GEO_ABLATE 0    // 8 no solve at all (the tail is the barriers only), 16 no stores, 128 no reads of the coefficients, 256 the stretch not written
#endif

struct GeoConsts {                  // on the device, read by scalar loads right in front of the solve (kernel arguments would sit in SGPRs through the FFT)
    double inv_rho, rho_sq, kappa, pL, pR;
    double pLk[GEO_SMAX];           // pL^(k + 1): what the carry into the segment weighs at its knot k
    double pRk[GEO_SMAX];           // pR^(S - k)
    double cL[GEO_REACH];           // (pL^S)^m: the segment total of the lane m + 1 to the left
    double cR[GEO_REACH];
    GeoBasis basis;                 // prefiltered plans
};

struct GeoArgs {
    FftlogArgs fft;
    int ws, ne, S;                   // first knot of the stretch, its length (64 S), knots per lane
    int nq, post_sqrt, group;
    int ntables, pt;                 // grouped: tables and pairs per table; plain: pairs and 1
    const GeoConsts* consts;
    const int* qe;                   // (nq) interval of each query relative to ws, -1: outside the knots
    const double* qa;                // (nq) A of each query (prefiltered plans: x = 1 - A)
    double* out;
};

// lane l receives the value of lane l - 1 (lane 0: zero) / of lane l + 1 (lane 63: zero): DPP shifts of the whole wave, no LDS
typedef int geo_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double wave_from_left(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    geo_v2i w = __builtin_bit_cast(geo_v2i, v);
    w.x = __builtin_amdgcn_update_dpp(0, w.x, 0x138, 0xf, 0xf, true);      // wave_shr:1
    w.y = __builtin_amdgcn_update_dpp(0, w.y, 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, w);
#else
    return v;
#endif
}
__device__ __forceinline__ double wave_from_right(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    geo_v2i w = __builtin_bit_cast(geo_v2i, v);
    w.x = __builtin_amdgcn_update_dpp(0, w.x, 0x130, 0xf, 0xf, true);      // wave_shl:1
    w.y = __builtin_amdgcn_update_dpp(0, w.y, 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double, w);
#else
    return v;
#endif
}

// sqrt for the epilogue: positive normal arguments (every variance) take the hardware reciprocal-root estimate with its third-order correction
// (cp_math.h: rsqrt_pos, relative error 1e-16) and one multiplication; zero, subnormal, negative, Inf and NaN take the library's
__device__ __forceinline__ double geo_sqrt(double v) {
    return v >= 2.2250738585072014e-308 && v <= 1.7976931348623157e308 ? v * cpmath::rsqrt_pos(v) : sqrt(v);
}

template <bool COEF, int QS = GEO_QMAX>      // QS: blocks of 64 radii the COEF tail is compiled for (4: plans of at most 256 radii)
__global__ __launch_bounds__(NP / P, 2) void fftlog_geospline_kernel(const GeoArgs R) {
    using F = Fftlog<NP, P, IN_HALF_ZERO, OUT_HALF>;
    constexpr int T = F::T, H = F::H, Q = F::Q;
    extern __shared__ __attribute__((aligned(4096))) char smem[];
    cplx* lds = reinterpret_cast<cplx*>(smem);
    const int t = threadIdx.x;
    const FftlogArgs& A = R.fft;
    // Work items without a division in the loop: an item is (tl, w) = (table of this share, pair inside the table); plain layout: one share,
    // tables of one pair, dealt blockIdx + k gridDim; grouped: a share per XCD (workgroup b runs on XCD b % 8), tables xcd + 8 tl, dealt
    // slot + k slots over the pairs of the share's tables in order.  tl only grows: the first table beyond the batch ends the walk.
    const bool grouped = R.group > 0 && !(CP_GEO_ABLATE & 64);      // (64: the grouped layout written in the plain work order, diagnostic)
    const int xcd = grouped ? (int)(blockIdx.x & 7) : 0, mult = grouped ? 8 : 1;
    const int pt = grouped ? R.pt : 1;
    const int first = grouped ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, step = grouped ? (int)(gridDim.x >> 3) : (int)gridDim.x;
    int tl = first / pt, w = first - tl * pt;
    const int dtl = step / pt, dw = step - dtl * pt;
    const int ntables = grouped ? R.ntables : (int)((A.nbatch + 1) / 2);
    if (tl * mult + xcd >= ntables) return;
    long long p = (long long)(tl * mult + xcd) * pt + w;
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
    const int wave = t >> 6, lane = t & 63;
    // the data region as doubles: the stretch of both rows (one knot of margin on either side), then their scaled second derivatives
    double* base = reinterpret_cast<double*>(lds);
    constexpr int YSTRIDE = 64 * GEO_SMAX + 8;
    static_assert(4 * YSTRIDE <= 2 * NP, "the solve lives in the data region of the FFT");
    // COEF: the coefficients of the stretch live BEHIND the FFT's LDS (two rows of ne + 2), so that the pair's radii are evaluated and stored after
    // the first barrier of the NEXT pair: no barrier of its own (the last pass reads and the next phase 0 writes the same slots of the data region,
    // thread by thread, as in fftlog_kernel), the reads of the coefficients land under phase 1's, and the stores have a whole pair to retire
    // before anything waits on the memory counter behind them (tools/geospline_ablate.sh: the stores at the end of the pair cost 0.3 ms of 3.4).
    double* side = reinterpret_cast<double*>(smem + F::LDS_BYTES);
    const int side_stride = R.ne + 2;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    double* pending_dst = nullptr;      // where the previous pair's first row goes (null: nothing to evaluate)
    bool pending_b = false;
    // A wave takes half of the radii for BOTH rows of the pair: the cubic weights of a radius are formed once for the two, and in the grouped
    // layout -- out[table, q, 2 within + row] -- the two values of a lane are neighbours in memory: one 16-byte store.
    // the radii of this lane, for the whole launch (a load here would sit behind the rows that phase 0 has just requested: the memory counter retires in order)
    constexpr int QW = COEF ? QS / 2 : 1;      // radii per lane and wave
    int qe[QW];
    double qx[QW];
    if constexpr (COEF) {
#pragma unroll
        for (int j = 0; j < QW; ++j) {
            const int q = lane + 64 * (2 * j + wave_s);      // the blocks of 64 radii alternate between the two waves
            qe[j] = q < R.nq ? R.qe[q] : -1;
            qx[j] = q < R.nq ? R.qa[q] : 0.;
        }
        // ... and landed before the loop: in flight at its head they made the compiler wait for vmcnt(0) at their first use in EVERY pair -- behind the
        // rows phase 0 has just requested from HBM (tools/isa_waits.py)
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    auto evaluate = [&](double* dst, bool both) {
        const double* Ca = side + 1;      // Ca[-1 .. ne]: the B-spline coefficients of the stretch, first row (second: side_stride further)
        const GeoConsts* C = R.consts;
        asm volatile("" : "+s"(C));      // (not hoisted out of the loop over pairs: the scalar registers are the FFT's there)
        const GeoBasis B = load_uniform(&C->basis);      // scalar loads
        const int qstride = grouped ? R.group : 1;
        const unsigned voff = (unsigned)(lane * qstride) * 8u;
        const int nqi = (CP_GEO_ABLATE & 2) ? 1 : (((R.nq + 63) >> 6) - wave_s + 1) >> 1;      // (wave-uniform) blocks of 64 radii of this wave
        const long long row_b = grouped ? 1 : R.nq;      // the second row's value, in doubles from the first's
#pragma unroll
        for (int j0 = 0; j0 < QW; j0 += 2) {
            if (j0 >= nqi) break;
            double ca[2][4], cb[2][4];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int e = qe[j0 + jj] < 0 ? 0 : qe[j0 + jj];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ca[jj][i] = (CP_GEO_ABLATE & 128) ? (double)(e + i) : Ca[e - 1 + i];
                    cb[jj][i] = (CP_GEO_ABLATE & 128) ? (double)(e - i) : Ca[side_stride + e - 1 + i];
                }
            }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int j = j0 + jj;
                const double x = qx[j];
                double va = 0., vb = 0.;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double wgt = fma(fma(fma(B.k[i][3], x, B.k[i][2]), x, B.k[i][1]), x, B.k[i][0]);
                    va = fma(wgt, ca[jj][i], va);
                    vb = fma(wgt, cb[jj][i], vb);
                }
                if (R.post_sqrt && !(CP_GEO_ABLATE & 4)) {
                    va = geo_sqrt(va);
                    vb = geo_sqrt(vb);
                }
                if (qe[j] < 0) va = vb = __builtin_nan("");
                char* obase = reinterpret_cast<char*>(dst) + (size_t)(64 * (2 * j + wave_s)) * (size_t)qstride * 8u;      // (wave-uniform)
                if (lane + 64 * (2 * j + wave_s) < R.nq && (!(CP_GEO_ABLATE & 16) || va == 12345.678)) {
                    double* o = reinterpret_cast<double*>(obase + voff);
                    if (grouped) {      // (groups are even: every pair has its second row)
                        *reinterpret_cast<double2*>(o) = make_double2(va, vb);
                    } else {
                        o[0] = va;
                        if (both) o[row_b] = vb;
                    }
                }
            }
        }
    };
    bool more = true;
    while (more) {
        const bool has_b = 2 * p + 1 < A.nbatch;
        const int table = tl * mult + xcd, within = w;
        // the next item (this one again on the last round: its rows are prefetched and screened, never used)
        w += dw;
        tl += dtl;
        if (w >= pt) {
            w -= pt;
            ++tl;
        }
        more = tl * mult + xcd < ntables;
        const long long pn = more ? (long long)(tl * mult + xcd) * pt + w : p;
        const double* nra = A.in + 2 * pn * n;
        const double* nrb = nra + (2 * pn + 1 < A.nbatch ? n : 0);
        {
            F::template phase<0>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nra, nrb, 0, st);
            __syncthreads();
            if constexpr (COEF && !CP_GEO_EVAL_LAST) {
                if (pending_dst && !(CP_GEO_ABLATE & 8)) evaluate(pending_dst, pending_b);
            }
            F::template phase<1>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nra, nrb, 0, st);
            if constexpr (!F::template barrier_free_after<1>()) __syncthreads();
            F::template phase<2>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nra, nrb, 0, st);
            if constexpr (!F::template barrier_free_after<2>()) __syncthreads();
            static_assert(F::NPH == 5, "three passes");
            F::template phase<3>(t, A, nullptr, nullptr, nullptr, nullptr, has_b, 0, lds, nra, nrb, 0, st);
            __syncthreads();
        }
        {
            cplx x[P];
            Pass<NP, P, 0>::load_lds(t0, lds, x);
            st.info_nxt = F::screen_collect(lds);
            if constexpr (COEF && CP_GEO_EVAL_LAST) {      // (measurements: the previous pair evaluated here, a barrier in front of this pair's coefficients)
                if (pending_dst && !(CP_GEO_ABLATE & 8)) evaluate(pending_dst, pending_b);
                __syncthreads();
            }
            if constexpr (!COEF) __syncthreads();      // every thread has its inputs: the data region is free
            Pass<NP, P, 0>::twiddle_apply(st.w, x);
            Pass<NP, P, 0>::butterflies(x);
            double ya[H], yb[H];
#pragma unroll
            for (int s = 0; s < H; ++s) {
                ya[s] = x[s + Q].re * st.fpost[s];
                yb[s] = x[s + Q].im * st.fpost[s];
            }
            F::template fix_output<H>(st.info_cur, ya, yb);
            double* ka = COEF ? side + 1 : base + 1;
            double* kb = COEF ? side + side_stride + 1 : base + YSTRIDE + 1;
#pragma unroll
            for (int s = 0; s < H; ++s) {      // the stretch of both rows (and one knot beyond either end) in natural order
                const int e = t0 + T * s - R.ws;
                if (e >= -1 && e <= R.ne && !(CP_GEO_ABLATE & 256 && ya[s] != 12345.678)) {
                    ka[e] = ya[s];
                    kb[e] = yb[s];
                }
            }
        }
        if constexpr (COEF) {
            // plain: out[row, q]; grouped: out[table, q, 2 within + wave]
            pending_dst = grouped ? R.out + (long long)table * R.nq * R.group + 2 * within : R.out + 2 * p * R.nq;
            pending_b = has_b;
            p = pn;
            continue;
        }
        __syncthreads();
        // ---- one wave per row from here to the stores ----
        if ((wave == 0 || has_b) && !(CP_GEO_ABLATE & 8)) {
            const double* Y = base + wave * YSTRIDE + 1;              // Y[-1 .. ne]
            double* N = base + (2 + wave) * YSTRIDE;                  // N[0 .. ne)
            const int S = R.S;
            // the queries of this lane (12 bytes per radius, L1 / L2 resident): in flight during the solve
            int qe[GEO_QMAX];
            double qa[GEO_QMAX];
#pragma unroll
            for (int j = 0; j < GEO_QMAX; ++j) {
                const int q = lane + 64 * j;
                if (CP_GEO_ABLATE & 32) {      // (queries made up in registers: what their loads cost)
                    qe[j] = q < R.nq ? 32 + q : -1;
                    qa[j] = 0.25;
                } else {
                    qe[j] = q < R.nq ? R.qe[q] : -1;
                    qa[j] = q < R.nq ? R.qa[q] : 0.;
                }
            }
            const GeoConsts* Cp = R.consts;
            asm volatile("" : "+s"(Cp));      // (not hoisted out of the loop over pairs: the scalar registers are the FFT's there)
            const GeoConsts G = load_uniform(Cp);      // scalar loads of the fields the instantiation below uses (through the laundered generic pointer
            const GeoConsts* C = &G;                   // they were flat vector loads, each behind a wait for every earlier store: 0.2 ms of 3.6)
            // The solve, instantiated for the plan's knots per lane (4 ... 8: a switch on a wave-uniform value): loops of exactly S steps and
            // ceil(33 / S) carries, where one body for 8 knots spent a third of its instructions on knots a plan of 5 does not have.
            auto solve = [&](auto sc) {
                constexpr int SS = decltype(sc)::value;
                constexpr int REACH = (GEO_HALO + SS) / SS;      // REACH x SS >= GEO_HALO + 1
                static_assert(REACH <= GEO_REACH, "the plan's carry weights");
                // the lane's knots and their two neighbours, one LDS round trip
                double y[SS + 2];
                const double* mine = Y + SS * lane - 1;
#pragma unroll
                for (int i = 0; i < SS + 2; ++i) y[i] = mine[i];
                const double inv_rho = C->inv_rho, pL = C->pL, pR = C->pR;
                double f[SS], g[SS];
                {      // F_k = d_k + pL F_{k-1} upwards and B_k = d_k + pR B_{k+1} downwards over the segment, both from zero
                    double d[SS];
#pragma unroll
                    for (int k = 0; k < SS; ++k) d[k] = (y[k + 2] - y[k + 1]) * inv_rho - (y[k + 1] - y[k]);
                    double acc = 0.;
#pragma unroll
                    for (int k = 0; k < SS; ++k) {
                        acc = fma(pL, acc, d[k]);
                        f[k] = acc;
                    }
                    acc = 0.;
#pragma unroll
                    for (int k = SS - 1; k >= 0; --k) {
                        acc = fma(pR, acc, d[k]);
                        g[k] = acc;
                    }
                }
                // what the knots left of the segment add to F at its first knot (x pL^(k + 1) further in), and the knots right of it to B
                double fin = 0., gin = 0.;
                if (!(CP_GEO_ABLATE & 1)) {
                    double fl = f[SS - 1], gr = g[0];      // the segment's totals: F at its last knot, B at its first
#pragma unroll
                    for (int m = 0; m < REACH; ++m) {
                        fl = wave_from_left(fl);
                        gr = wave_from_right(gr);
                        fin = fma(C->cL[m], fl, fin);
                        gin = fma(C->cR[m], gr, gin);
                    }
                }
                // N_k = kappa (F_k + pR B_{k+1}), B beyond the segment's last knot being the carry itself
                const double kappa = C->kappa;
                double bnext = gin;      // true B at the first knot of the next segment
#pragma unroll
                for (int k = SS - 1; k >= 0; --k) {
                    const double fk = fma(C->pLk[k], fin, f[k]);
                    N[SS * lane + k] = kappa * fma(pR, bnext, fk);
                    bnext = fma(C->pRk[k], gin, g[k]);
                }
            };
            switch (S) {
                case 4: solve(std::integral_constant<int, 4>{}); break;
                case 5: solve(std::integral_constant<int, 5>{}); break;
                case 6: solve(std::integral_constant<int, 6>{}); break;
                case 7: solve(std::integral_constant<int, 7>{}); break;
                default: solve(std::integral_constant<int, 8>{}); break;
            }
            cp::wave_lds_phase();
            const double rho_sq = C->rho_sq;
            // plain: out[row, q]; grouped: out[table, q, 2 within + wave]
            double* dst = grouped ? R.out + (long long)table * R.nq * R.group + 2 * within + wave : R.out + (2 * p + wave) * R.nq;
            int qstride = grouped ? R.group : 1;
            if ((CP_GEO_ABLATE & 64) && R.group > 0) {
                const long long row = 2 * p + wave;
                dst = R.out + (row / R.group) * R.nq * R.group + row % R.group;
                qstride = R.group;
            }
            // Four radii per lane at a time, without a branch between them: their four LDS reads each are issued together, the stores follow each
            // other (a radius per block of its own -- one predicate, two LDS round trips and a 64-bit address each -- took 0.15 ms of the 3.8 per
            // radius and lane: profiles/r4_geospline_ablate.txt, bit 2).  Lanes past the last radius compute on the last one and do not store.
            const unsigned voff = (unsigned)(lane * qstride) * 8u;
            const int nqi = (CP_GEO_ABLATE & 2) ? 1 : (R.nq + 63) >> 6;      // (wave-uniform)
#pragma unroll
            for (int j0 = 0; j0 < GEO_QMAX; j0 += 4) {
                if (j0 >= nqi) break;
                double ya[4], yb[4], na[4], nb[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int e = qe[j0 + jj] < 0 ? 0 : qe[j0 + jj];
                    ya[jj] = Y[e];
                    yb[jj] = Y[e + 1];
                    na[jj] = N[e];
                    nb[jj] = N[e + 1];
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int j = j0 + jj;
                    const double a = qa[j], b = 1. - a;
                    double v = fma(a, ya[jj], b * yb[jj]) + (fma(a * a, a, -a) * (rho_sq * na[jj]) + fma(b * b, b, -b) * nb[jj]);
                    if (R.post_sqrt && !(CP_GEO_ABLATE & 4)) v = geo_sqrt(v);
                    if (qe[j] < 0) v = __builtin_nan("");
                    char* obase = reinterpret_cast<char*>(dst) + (size_t)(64 * j) * (size_t)qstride * 8u;      // (wave-uniform)
                    if (lane + 64 * j < R.nq && (!(CP_GEO_ABLATE & 16) || v == 12345.678)) *reinterpret_cast<double*>(obase + voff) = v;
                }
            }
        }
        __syncthreads();      // the next pair's phase 0 writes the data region
        p = pn;
    }
    if constexpr (COEF) {
        __syncthreads();
        if (pending_dst && !(CP_GEO_ABLATE & 8)) evaluate(pending_dst, pending_b);
    }
}

template <int ENGINE>
hipError_t launch(const SigmaArgs& S, int grid, size_t lds, hipStream_t stream) {
    if (S.qe) {      // the radii come from B-spline coefficients
        if (lds > 64 * 1024) (void)cp::allow_full_lds<&sigma_rz_kernel<ENGINE, true>>();
        hipLaunchKernelGGL((sigma_rz_kernel<ENGINE, true>), dim3(grid), dim3(NP / P), lds, stream, S);
        return hipGetLastError();
    }
    if (lds > 64 * 1024) (void)cp::allow_full_lds<&sigma_rz_kernel<ENGINE, false>>();
    hipLaunchKernelGGL((sigma_rz_kernel<ENGINE, false>), dim3(grid), dim3(NP / P), lds, stream, S);
    return hipGetLastError();
}

// what the fused kernel needs of a prefiltered spline plan (cp_geospline_plan, below)
struct GeoTail {
    int nq, ws;
    const int* qe;
    const double* qx;
    const GeoBasis* basis;
};

int sigma_rz_fused_launch(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
                          const double* d_k, const cp_fftlog_tables_view& f, const cp_spline_band_view* band, const GeoTail* geo, const double* d_growth_sq,
                          int nz, double* d_out, double* d_pk_out, void* d_coef, int device, void* stream);

}  // namespace

// 1 when cp_sigma_rz_analytic takes the fused kernel for these plans (TophatVariance on 1024 samples padded to 2048, one kernel)
extern "C" int cp_sigma_rz_fused_available(const cp_fftlog_plan* fftlog, const cp_spline_plan* spline) {
    cp_fftlog_tables_view f;
    cp_spline_band_view b;
    if (!cp_fftlog_plan_view(fftlog, &f) || !cp_spline_plan_view(spline, &b)) return 0;
    return f.npad == NP && f.n == NP / 2 && f.nker == 1 && f.in_left == NP / 4 && f.out_left == NP / 4 && b.n == f.n && b.device == f.device;
}

int cp_sigma_rz_fused(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
                      const double* d_k,
                      const cp_fftlog_plan* fftlog, const cp_spline_plan* spline, const double* d_growth_sq, int nz, double* d_out, double* d_pk_out,
                      void* d_coef, int device, void* stream) {
    cp_fftlog_tables_view f;
    cp_spline_band_view b;
    if (!cp_fftlog_plan_view(fftlog, &f) || !cp_spline_plan_view(spline, &b)) return cp::fail(CP_EINVAL, "cp_sigma_rz_fused: plans without device tables");
    return sigma_rz_fused_launch(engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, d_k, f, &b, nullptr, d_growth_sq, nz, d_out, d_pk_out, d_coef, device,
                                 stream);
}

namespace {
int sigma_rz_fused_launch(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
                          const double* d_k, const cp_fftlog_tables_view& f, const cp_spline_band_view* band, const GeoTail* geo, const double* d_growth_sq,
                          int nz, double* d_out, double* d_pk_out, void* d_coef, int device, void* stream) {
    const int nq = geo ? geo->nq : band->nq;
    double* ln_k = reinterpret_cast<double*>(static_cast<char*>(d_coef) + ((cp_power_workspace_bytes(ncosmo) + 63) / 64) * 64);      // behind the coefficients
    // (validates the massive-neutrino tables; the same launch writes the table of the wavenumbers)
    int st = cp_power_coefficients(engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, d_coef, device, stream, d_k, ln_k, f.n);
    if (st != CP_OK) return st;
    SigmaArgs S{};
    S.nsp = ncdm ? ncdm->nspecies : 0;
    if (S.nsp < 0 || (S.nsp > 0 && !ncdm->tab)) return cp::fail(CP_EINVAL, "cp_sigma_rz_fused: bad massive-neutrino tables");
    S.ncdm_tab = S.nsp ? ncdm->tab : nullptr;
    FftlogArgs& A = S.fft;
    A.in = nullptr; A.out = nullptr; A.nbatch = ncosmo; A.nker = 1; A.n = f.n; A.in_left = f.in_left; A.out_off = f.out_left; A.n_out = f.n;
    A.ext_l = A.ext_r = CP_EXTRAP_CONST; A.val_l = A.val_r = 0.; A.stream_rows = 0;
    A.pre = f.d_pre; A.post = f.d_post; A.u = f.d_u; A.tw = f.d_tw;
    S.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) S.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    for (int i = 0; i < CP_PK_NPARAMS; ++i) S.pw[i] = Param{pk_params[i].ptr, pk_params[i].value};
    S.second_is_omega_m = second_is_omega_m;
    S.k = d_k;
    S.consts = static_cast<const CosmoConsts*>(d_coef);
    S.ln_k = ln_k;
    S.nq = nq; S.nz = nz;
    if (geo) {
        S.qe = geo->qe; S.qx = geo->qx; S.basis = geo->basis; S.ws = geo->ws;
    } else {
        S.wb = band->d_wb; S.j0 = band->d_j0; S.bw = band->bw;
    }
    S.growth_sq = d_growth_sq;
    S.out = d_out;
    S.pk_out = d_pk_out;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma_rz_fused: cannot select device %d", device);
    using F = Fftlog<NP, P, IN_HALF_ZERO_GEN, OUT_HALF>;
    const size_t lds = (size_t)F::LDS_BYTES + (size_t)(2 * nq + 2 * nz) * sizeof(double);
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device);
    const long long npairs = (ncosmo + 1) / 2;
    const size_t per_cu = lds ? (160 * 1024) / lds : 4;
    const long long resident = (long long)(ncu > 0 ? ncu : 256) * (long long)(per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    // every workgroup the same number of pairs (5000 pairs on 1024 resident workgroups would leave a fifth of the chip idle in the last round)
    const long long rounds = (npairs + resident - 1) / resident;
    const int grid = (int)((npairs + rounds - 1) / rounds);
    // The workgroups that share a CU (b, b + ncu, b + 2 ncu, ...: the dispatcher deals them out in order) start two s_sleep(127) apart: the first of them
    // has its first results in memory while the others still evaluate, instead of all four storing at once behind a first evaluation at a quarter of the
    // CU each (0.327 -> 0.317 ms per 10 000 cosmologies, profiles/r5_sigma_stagger.txt; larger offsets only delay the last workgroup).  Same results.
    S.stagger_div = ncu > 0 ? ncu : 256; S.stagger_mod = (int)(per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu)); S.stagger_sleeps = rounds >= 2 ? 2 : 0;
    if (const char* env = std::getenv("CP_SIGMA_STAGGER")) {      // "div,mod,sleeps": measurements (tools/ab_sigma_stagger.py)
        int d = 1, m = 1, sl = 0;
        if (std::sscanf(env, "%d,%d,%d", &d, &m, &sl) == 3 && d >= 1 && m >= 1 && sl >= 0 && sl <= 4096) { S.stagger_div = d; S.stagger_mod = m; S.stagger_sleeps = sl; }
    }
    hipError_t e = hipSuccess;
    if (lds > 160 * 1024) {
        if (prev >= 0) (void)hipSetDevice(prev);
        return cp::fail(CP_EUNSUPPORTED, "cp_sigma_rz_fused: %d radii x %d redshifts exceed the LDS staging", nq, nz);
    }
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (engine == CP_ENGINE_EH) e = launch<CP_ENGINE_EH>(S, grid, lds, hs);
    else if (engine == CP_ENGINE_EH_NOWIGGLE) e = launch<CP_ENGINE_EH_NOWIGGLE>(S, grid, lds, hs);
    else e = launch<CP_ENGINE_BBKS>(S, grid, lds, hs);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma_rz_fused: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
}  // namespace

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
    const double* ncdm_tab;       // massive neutrinos (cp_ncdm.tab; nsp == 0: none), their knots for the growth factor at z = 0
    const double* ncdm_knots;
    int nsp;
    int nk, nq, nz;
    const double* k;              // (nk) wavenumbers, h/Mpc
    const double* ln_k;           // (3, nk) log k, k^1.08, k^1.4 (cp_power_coefficients)
    const CosmoConsts* consts;    // (ncosmo) the cosmologies' constants (cp_power_coefficients)
    const double* functional;     // (nq, nk)
    const double* growth_sq;      // (ncosmo, nz)
    double* out;                  // (ncosmo, nq, nz)
    double* pk_out;               // (ncosmo, nk) or null
    // the sigma8 normalisation (sigma8_normalise_kernel): target sigma8, factors sigma8 / sigma8(fiducial amplitude) and normalised amplitudes out
    Param target;
    double* rsigma8_out;          // (ncosmo)
    double* amplitude_out;        // (ncosmo) A_s x rsigma8^2, or null
};

// The sigma8 normalisation of a batch of analytic cosmologies as ONE kernel (eisenstein_hu.py:94-103 with :331-342, cosmology.py:505-510): sigma8 of
// the fiducial amplitude -- P(k) x the functional of r = 8, x the CPT92 growth factor at z = 0 squared (Background.growth_factor(0, znorm=0),
// eisenstein_hu.py:134-139), evaluated here instead of by a launch of the background kernel --, the factor rsigma8 = sigma8 / that, the normalised
// amplitude A_s rsigma8^2, and the spectra at the NORMALISED amplitude (P is linear in A_s): what the engine hands to the filters next, with no
// pass over (ncosmo, 1024) arrays behind the kernel.  1024 wavenumbers: a lane keeps its 16 samples in LDS (its own slots: no barrier) until the factor
// is known -- in registers they put the kernel at 190 VGPRs, two waves per SIMD, where the evaluation runs best at three (power_kernel).
template <int ENGINE>
__global__ __launch_bounds__(256, 3) void sigma8_normalise_kernel(const FunctionalArgs S) {      // three waves per SIMD, as power_kernel (the evaluation is the same)
    constexpr int PER_LANE = 16;      // nk = 1024
    __shared__ MathTables mt;
    fill_math_tables(&mt);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long ic = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (a cosmology per wave: in a scalar register)
    if (ic >= S.ncosmo) return;
    const CosmoConsts K = load_uniform(S.consts + ic);      // (scalar loads: the cosmology's constants in scalar registers)
    const PkPerCosmology& pc = K.pk;
    double acc = 0.;
    __shared__ double kept[4 * PER_LANE * 64];
    double* pks = kept + (threadIdx.x >> 6) * (PER_LANE * 64) + lane;      // sample i of this lane at pks[64 i]
#ifndef CP_SIGMA8_UNROLL
#define CP_SIGMA8_UNROLL 2
#endif
#pragma unroll CP_SIGMA8_UNROLL
    for (int i = 0; i < PER_LANE; ++i) {
        const int j = lane + 64 * i;
        const double kh = S.k[j], ln_kh = S.ln_k[j];
        const double Tk = transfer_any<ENGINE>(K, kh, ln_kh, S.ln_k[1024 + j], S.ln_k[2048 + j], tables_present(&mt));
        const double pk = (Tk * Tk) * (kh * pc.pk_unit) * primordial_tilt(pc, ln_kh, tables_present(&mt));
        pks[64 * i] = pk;
        acc = fma(S.functional[j], pk, acc);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    const double g0 = K.s.growth0;      // (the lane of the coefficient pre-kernel has evaluated it: ~350 instructions per wave here)
    const double sigma8_fid = sqrt(acc) * sqrt(g0 * g0);      // (the product of roots of the fused kernels: sqrt(sigma^2) sqrt(growth^2))
    const double target = S.target.ptr ? S.target.ptr[ic] : S.target.value;
    const double rs = target / sigma8_fid, rs2 = rs * rs;
    if (lane == 0) {
        S.rsigma8_out[ic] = rs;
        if (S.amplitude_out) S.amplitude_out[ic] = K.A_s * rs2;
    }
    if (S.pk_out) {
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) S.pk_out[ic * 1024 + lane + 64 * i] = pks[64 * i] * rs2;
    }
}

template <int ENGINE>
__global__ __launch_bounds__(256) void sigma_functional_kernel(const FunctionalArgs S) {
    __shared__ MathTables mt;
    fill_math_tables(&mt);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const long long ic = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // (a cosmology per wave: in a scalar register)
    if (ic >= S.ncosmo) return;
    const CosmoConsts K = load_uniform(S.consts + ic);      // (scalar loads: the cosmology's constants in scalar registers)
    const PkPerCosmology& pc = K.pk;
    double acc[FUNCTIONAL_MAX_NQ];
#pragma unroll
    for (int q = 0; q < FUNCTIONAL_MAX_NQ; ++q) acc[q] = 0.;
#pragma unroll 2
    for (int j = lane; j < S.nk; j += 64) {
        const double kh = S.k[j], ln_kh = S.ln_k[j];
        const double Tk = transfer_any<ENGINE>(K, kh, ln_kh, S.ln_k[S.nk + j], S.ln_k[2 * S.nk + j], tables_present(&mt));
        const double pk = (Tk * Tk) * (kh * pc.pk_unit) * primordial_tilt(pc, ln_kh, tables_present(&mt));
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

extern "C" int cp_sigma_rz_functional(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
        int nk,
                                      const double* d_k, const double* d_functional, int nq, const double* d_growth_sq, int nz, double* d_out,
                                      double* d_pk_out, void* d_work, int device, void* stream) {
    if (ncosmo < 0 || nk <= 0 || nz <= 0 || nq <= 0) return cp::fail(CP_EINVAL, "cp_sigma_rz_functional: bad sizes");
    if (nq > FUNCTIONAL_MAX_NQ) return cp::fail(CP_EUNSUPPORTED, "cp_sigma_rz_functional: %d radii (at most %d: use cp_sigma_rz_analytic)", nq, FUNCTIONAL_MAX_NQ);
    if (engine != CP_ENGINE_EH && engine != CP_ENGINE_EH_NOWIGGLE && engine != CP_ENGINE_BBKS) return cp::fail(CP_EINVAL, "cp_sigma_rz_functional: unknown engine %d", engine);
    if (ncosmo == 0) return CP_OK;
    if (!bg_params || !pk_params || !d_k || !d_functional || !d_growth_sq || !d_out || !d_work) return cp::fail(CP_EINVAL, "cp_sigma_rz_functional: null pointer");
    char* coef = static_cast<char*>(d_work);
    coef += (64 - (reinterpret_cast<unsigned long long>(coef) & 63u)) & 63u;
    double* ln_k = reinterpret_cast<double*>(coef + ((cp_power_workspace_bytes(ncosmo) + 63) / 64) * 64);      // behind the coefficients (cp_sigma_rz_workspace_bytes)
    int st = cp_power_coefficients(engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, coef, device, stream, d_k, ln_k, nk);      // (and the table of the wavenumbers)
    if (st != CP_OK) return st;
    FunctionalArgs S{};
    {
        int prev_nu = -1;
        if (hipGetDevice(&prev_nu) != hipSuccess) prev_nu = -1;
        if (prev_nu != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cannot select device %d", device);
        NcdmView nu;
        st = ncdm_view(ncdm, device, "cp_sigma_rz_functional / cp_sigma8_normalise", &nu);
        if (prev_nu >= 0 && prev_nu != device) (void)hipSetDevice(prev_nu);
        if (st != CP_OK) return st;
        S.ncdm_tab = nu.tab;
        S.ncdm_knots = nu.knots;
        S.nsp = nu.nsp;
    }
    S.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) S.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    for (int i = 0; i < CP_PK_NPARAMS; ++i) S.pw[i] = Param{pk_params[i].ptr, pk_params[i].value};
    S.second_is_omega_m = second_is_omega_m;
    S.nk = nk; S.nq = nq; S.nz = nz;
    S.k = d_k;
    S.consts = reinterpret_cast<const CosmoConsts*>(coef);
    S.ln_k = ln_k;
    S.functional = d_functional;
    S.growth_sq = d_growth_sq;
    S.out = d_out;
    S.pk_out = d_pk_out;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma_rz_functional: cannot select device %d", device);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((ncosmo + 3) / 4);
    if (engine == CP_ENGINE_EH) hipLaunchKernelGGL(sigma_functional_kernel<CP_ENGINE_EH>, dim3(grid), dim3(256), 0, hs, S);
    else if (engine == CP_ENGINE_EH_NOWIGGLE) hipLaunchKernelGGL(sigma_functional_kernel<CP_ENGINE_EH_NOWIGGLE>, dim3(grid), dim3(256), 0, hs, S);
    else hipLaunchKernelGGL(sigma_functional_kernel<CP_ENGINE_BBKS>, dim3(grid), dim3(256), 0, hs, S);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma_rz_functional: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_sigma8_normalise(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm, const cp_param* pk_params,
        int nk,
                                   const double* d_k, const double* d_functional, cp_param sigma8, double* d_rsigma8, double* d_amplitude, double* d_pk_out,
                                   void* d_work, int device, void* stream) {
    if (ncosmo < 0) return cp::fail(CP_EINVAL, "cp_sigma8_normalise: negative size");
    if (nk != 1024) return cp::fail(CP_EUNSUPPORTED, "cp_sigma8_normalise: %d wavenumbers (built for the 1024 of the default transform)", nk);
    if (engine != CP_ENGINE_EH && engine != CP_ENGINE_EH_NOWIGGLE && engine != CP_ENGINE_BBKS) return cp::fail(CP_EINVAL, "cp_sigma8_normalise: unknown engine %d", engine);
    if (ncosmo == 0) return CP_OK;
    if (!bg_params || !pk_params || !d_k || !d_functional || !d_rsigma8 || !d_work) return cp::fail(CP_EINVAL, "cp_sigma8_normalise: null pointer");
    char* coef = static_cast<char*>(d_work);
    coef += (64 - (reinterpret_cast<unsigned long long>(coef) & 63u)) & 63u;
    double* ln_k = reinterpret_cast<double*>(coef + ((cp_power_workspace_bytes(ncosmo) + 63) / 64) * 64);      // behind the coefficients (cp_sigma_rz_workspace_bytes)
    int st = cp_power_coefficients(engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, coef, device, stream, d_k, ln_k, nk);      // (and the table of the wavenumbers)
    if (st != CP_OK) return st;
    FunctionalArgs S{};
    {
        int prev_nu = -1;
        if (hipGetDevice(&prev_nu) != hipSuccess) prev_nu = -1;
        if (prev_nu != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cannot select device %d", device);
        NcdmView nu;
        st = ncdm_view(ncdm, device, "cp_sigma_rz_functional / cp_sigma8_normalise", &nu);
        if (prev_nu >= 0 && prev_nu != device) (void)hipSetDevice(prev_nu);
        if (st != CP_OK) return st;
        S.ncdm_tab = nu.tab;
        S.ncdm_knots = nu.knots;
        S.nsp = nu.nsp;
    }
    S.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) S.bg[i] = Param{bg_params[i].ptr, bg_params[i].value};
    for (int i = 0; i < CP_PK_NPARAMS; ++i) S.pw[i] = Param{pk_params[i].ptr, pk_params[i].value};
    S.second_is_omega_m = second_is_omega_m;
    S.nk = nk; S.nq = 1; S.nz = 1;
    S.k = d_k;
    S.consts = reinterpret_cast<const CosmoConsts*>(coef);
    S.ln_k = ln_k;
    S.functional = d_functional;
    S.pk_out = d_pk_out;
    S.target = Param{sigma8.ptr, sigma8.value};
    S.rsigma8_out = d_rsigma8;
    S.amplitude_out = d_amplitude;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma8_normalise: cannot select device %d", device);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)((ncosmo + 3) / 4);
    if (engine == CP_ENGINE_EH) hipLaunchKernelGGL(sigma8_normalise_kernel<CP_ENGINE_EH>, dim3(grid), dim3(256), 0, hs, S);
    else if (engine == CP_ENGINE_EH_NOWIGGLE) hipLaunchKernelGGL(sigma8_normalise_kernel<CP_ENGINE_EH_NOWIGGLE>, dim3(grid), dim3(256), 0, hs, S);
    else hipLaunchKernelGGL(sigma8_normalise_kernel<CP_ENGINE_BBKS>, dim3(grid), dim3(256), 0, hs, S);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_sigma8_normalise: launch failed: %s", hipGetErrorString(e));
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
    A.stream_rows = (double)nbatch * f.n * 8. > 512. * 1024. * 1024. ? 3 : 0;
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

// ---- natural spline on a geometric grid, solved inside the FFTLog kernel (fftlog_geospline_kernel above) -----------------------------------------
struct cp_geospline_plan {
    int n, nq, ws, ne, S, device;
    int* d_qe;
    double* d_qa;
    GeoConsts* d_consts;
    cp_fftlog_plan* prefiltered;      // plans of cp_geospline_plan_create_prefiltered: the transform whose output is the B-spline coefficients
};

extern "C" int cp_geospline_plan_destroy(cp_geospline_plan* p) {
    if (!p) return CP_OK;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device) (void)hipSetDevice(p->device);
    if (p->d_qe) (void)hipFree(p->d_qe);
    if (p->d_qa) (void)hipFree(p->d_qa);
    if (p->d_consts) (void)hipFree(p->d_consts);
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    if (p->prefiltered) (void)cp_fftlog_plan_destroy(p->prefiltered);
    delete p;
    return CP_OK;
}

namespace {

// the geometric ratio of the knots, or 0 when they are not an ascending geometric grid (the error is set)
double geometric_ratio(const char* who, const double* knots, int n) {
    if (!(knots[0] > 0.) || !(knots[n - 1] > knots[0])) {
        (void)cp::fail(CP_EUNSUPPORTED, "%s: the knots are not an ascending geometric grid", who);
        return 0.;
    }
    const double rho = pow(knots[n - 1] / knots[0], 1. / (n - 1));
    for (int i = 0; i + 1 < n; ++i)
        if (!(fabs(knots[i + 1] / (knots[i] * rho) - 1.) < 1e-12)) {
            (void)cp::fail(CP_EUNSUPPORTED, "%s: the knots are not a geometric grid (knot %d)", who, i + 1);
            return 0.;
        }
    return rho;
}

// interval and position of every query: qj = -1 outside the knots (or NaN), else s_qj <= r <= s_qj+1 and qa = (s_qj+1 - r) / h_qj
void locate_queries(const double* knots, int n, const double* queries, int nq, std::vector<int>& qj, std::vector<double>& qa, int* jmin, int* jmax) {
    *jmin = n;
    *jmax = -1;
    for (int q = 0; q < nq; ++q) {
        const double r = queries[q];
        if (!(r >= knots[0] && r <= knots[n - 1])) {      // outside the knots (or NaN): NaN, as the spline without extrapolation returns
            qj[q] = -1;
            qa[q] = 0.;
            continue;
        }
        int j = (int)(std::upper_bound(knots, knots + n, r) - knots) - 1;
        if (j > n - 2) j = n - 2;
        qj[q] = j;
        qa[q] = (knots[j + 1] - r) / (knots[j + 1] - knots[j]);
        *jmin = std::min(*jmin, j);
        *jmax = std::max(*jmax, j + 1);
    }
}

// The four cubic B-splines of the knots rho^m that live on [1, rho], as polynomials in x = (r - 1) / (rho - 1): basis[i] is the one centred on the
// knot rho^(i - 1).  Cox - de Boor on the coefficients (the only piece of order 0 on that interval is B_{0,0} = 1).
void geometric_bspline_pieces(long double rho, long double basis[4][4]) {
    auto knot = [&](int m) { return powl(rho, (long double)m); };
    const long double h = rho - 1.0L;
    long double prev[5][4] = {}, cur[5][4];      // slot i + 3 holds B_{i,d}, i = -3 .. 0 (slot 4: B_{1,d} = 0 on the interval)
    prev[3][0] = 1.0L;
    for (int d = 1; d <= 3; ++d) {
        for (int slot = 0; slot < 5; ++slot)
            for (int e = 0; e < 4; ++e) cur[slot][e] = 0.0L;
        for (int i = -d; i <= 0; ++i) {
            // (r - t_i) / (t_{i+d} - t_i) B_{i,d-1} + (t_{i+d+1} - r) / (t_{i+d+1} - t_{i+1}) B_{i+1,d-1},   r = 1 + h x
            const long double wl = 1.0L / (knot(i + d) - knot(i)), wr = 1.0L / (knot(i + d + 1) - knot(i + 1));
            const long double l0 = (1.0L - knot(i)) * wl, l1 = h * wl, r0 = (knot(i + d + 1) - 1.0L) * wr, r1 = -h * wr;
            for (int e = 0; e < 4; ++e) {
                cur[i + 3][e] += l0 * prev[i + 3][e] + r0 * prev[i + 4][e];
                if (e > 0) cur[i + 3][e] += l1 * prev[i + 3][e - 1] + r1 * prev[i + 4][e - 1];
            }
        }
        for (int slot = 0; slot < 5; ++slot)
            for (int e = 0; e < 4; ++e) prev[slot][e] = cur[slot][e];
    }
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) basis[i][e] = prev[i][e];
}

int upload_geospline(cp_geospline_plan* p, const std::vector<int>& qj, const std::vector<double>& qa, const GeoConsts& consts) {
    const int nq = p->nq, device = p->device;
    int prev = -1, status = CP_OK;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) status = cp::fail(CP_EDEVICE, "cp_geospline_plan_create: cannot select device %d", device);
    if (status == CP_OK && (hipMalloc(&p->d_qe, nq * sizeof(int)) != hipSuccess || hipMalloc(&p->d_qa, nq * sizeof(double)) != hipSuccess ||
                            hipMalloc(&p->d_consts, sizeof(GeoConsts)) != hipSuccess))
        status = cp::fail(CP_ENOMEM, "cp_geospline_plan_create: device allocation failed");
    if (status == CP_OK && (hipMemcpy(p->d_qe, qj.data(), nq * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemcpy(p->d_qa, qa.data(), nq * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemcpy(p->d_consts, &consts, sizeof(GeoConsts), hipMemcpyHostToDevice) != hipSuccess))
        status = cp::fail(CP_EDEVICE, "cp_geospline_plan_create: upload failed");
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return status;
}

}  // namespace

extern "C" int cp_geospline_plan_create(cp_geospline_plan** out, const double* knots, int n, const double* queries, int nq, int device) {
    if (!out) return cp::fail(CP_EINVAL, "cp_geospline_plan_create: null plan pointer");
    *out = nullptr;
    if (!knots || !queries || n < 4 || nq < 1) return cp::fail(CP_EINVAL, "cp_geospline_plan_create: bad arguments");
    if (nq > 64 * GEO_QMAX) return cp::fail(CP_EUNSUPPORTED, "cp_geospline_plan_create: %d queries (at most %d)", nq, 64 * GEO_QMAX);
    const double rho = geometric_ratio("cp_geospline_plan_create", knots, n);
    if (!(rho > 0.)) return CP_EUNSUPPORTED;
    std::vector<int> qj(nq);
    std::vector<double> qa(nq);
    int jmin, jmax;
    locate_queries(knots, n, queries, nq, qj, qa, &jmin, &jmax);
    if (jmax < 0) return cp::fail(CP_EUNSUPPORTED, "cp_geospline_plan_create: no query inside the knots");
    const int need = jmax - jmin + 1 + 2 * GEO_HALO;
    const int S = std::max(GEO_SMIN, (need + 63) / 64);
    if (S > GEO_SMAX) return cp::fail(CP_EUNSUPPORTED, "cp_geospline_plan_create: the queries span %d knots (at most %d)", jmax - jmin + 1, 64 * GEO_SMAX - 2 * GEO_HALO);
    const int ne = 64 * S;
    int ws = jmin - GEO_HALO - (ne - need) / 2;
    // the stretch [ws, ws + ne) and one knot on either side must be interior knots (the solve uses the interior equation everywhere)
    if (ws < 2) ws = 2;
    if (ws + ne > n - 2) ws = n - 2 - ne;
    if (ws < 2 || jmin - ws < GEO_HALO || ws + ne - 1 - jmax < GEO_HALO)
        return cp::fail(CP_EUNSUPPORTED, "cp_geospline_plan_create: the queries come within %d knots of the ends of the grid", GEO_HALO);
    cp_geospline_plan* p = new (std::nothrow) cp_geospline_plan();
    if (!p) return cp::fail(CP_ENOMEM, "cp_geospline_plan_create: host allocation failed");
    p->n = n; p->nq = nq; p->ws = ws; p->ne = ne; p->S = S; p->device = device; p->d_qe = nullptr; p->d_qa = nullptr; p->d_consts = nullptr;
    p->prefiltered = nullptr;
    GeoConsts consts{};
    {
        // a N_{i-1} + b N_i + c N_{i+1} = d_i on the infinite grid: N_i = kappa (sum_{j <= i} pL^(i-j) d_j + sum_{j > i} pR^(j-i) d_j), pL the root of
        // c t^2 + b t + a inside the unit circle (a source to the left decays to the right), pR that of a t^2 + b t + c, kappa = 1 / (a pR + b + c pL)
        const long double a = (long double)rho * rho, b = 2.0L * (1.0L + rho), c = 1.0L / rho;
        const long double disc = sqrtl(b * b - 4.0L * a * c);
        const long double pL = -2.0L * a / (b + disc);      // = (-b + disc) / (2 c), without the cancellation
        const long double pR = -2.0L * c / (b + disc);
        const long double kappa = 1.0L / (a * pR + b + c * pL);
        consts.inv_rho = (double)(1.0L / rho); consts.rho_sq = (double)a; consts.kappa = (double)kappa; consts.pL = (double)pL; consts.pR = (double)pR;
        for (int k = 0; k < GEO_SMAX; ++k) {
            consts.pLk[k] = (double)powl(pL, k + 1);
            consts.pRk[k] = k < S ? (double)powl(pR, S - k) : 0.;
        }
        for (int m = 0; m < GEO_REACH; ++m) {
            consts.cL[m] = (double)powl(pL, (long double)S * m);
            consts.cR[m] = (double)powl(pR, (long double)S * m);
        }
    }
    for (int q = 0; q < nq; ++q)
        if (qj[q] >= 0) qj[q] -= ws;
    const int status = upload_geospline(p, qj, qa, consts);
    if (status != CP_OK) {
        cp_geospline_plan_destroy(p);
        return status;
    }
    *out = p;
    return CP_OK;
}

// Host only (no device call): the four cubic B-splines of the geometric knots rho^m on [1, rho] as polynomials in x = (r - 1) / (rho - 1), basis[4 i + d] the
// coefficient of x^d of the one centred on rho^(i - 1) -- what a prefiltered plan evaluates with; their values at x = 0 are the interpolation conditions'
// alpha, beta, gamma (the fourth is 0).
extern "C" int cp_geospline_basis(double rho, double* basis) {
    if (!basis || !(rho > 1.) || !std::isfinite(rho)) return cp::fail(CP_EINVAL, "cp_geospline_basis: need rho > 1 and 16 doubles of room");
    long double b[4][4];
    geometric_bspline_pieces((long double)rho, b);
    for (int i = 0; i < 4; ++i)
        for (int d = 0; d < 4; ++d) basis[4 * i + d] = (double)b[i][d];
    return CP_OK;
}

// The same spline for a transform the plan OWNS: built from the caller's FFTLog tables with the B-spline prefilter folded into u (see
// fftlog_geospline_kernel<true>), so that the kernel's tail is four reads and sixteen multiply-adds per radius.
extern "C" int cp_geospline_plan_create_prefiltered(cp_geospline_plan** out, int n, int npad, const double* pre, const double* post, const double* u_re_im,
                                                    const double* knots, const double* queries, int nq, int device) {
    const char* who = "cp_geospline_plan_create_prefiltered";
    if (!out) return cp::fail(CP_EINVAL, "%s: null plan pointer", who);
    *out = nullptr;
    if (!pre || !post || !u_re_im || !knots || !queries || n < 4 || nq < 1) return cp::fail(CP_EINVAL, "%s: bad arguments", who);
    if (npad != NP || n != NP / 2) return cp::fail(CP_EUNSUPPORTED, "%s: transform outside the fused kernel's shape (1024 samples padded to 2048)", who);
    if (nq > 64 * GEO_QMAX) return cp::fail(CP_EUNSUPPORTED, "%s: %d queries (at most %d)", who, nq, 64 * GEO_QMAX);
    const double rho = geometric_ratio(who, knots, n);
    if (!(rho > 0.)) return CP_EUNSUPPORTED;
    // the postfactor over the padded grid: post_0 lambda^j (a power of the geometric output grid, fftlog.py:175)
    if (!(post[0] != 0.) || !std::isfinite(post[0]) || !std::isfinite(post[npad - 1])) return cp::fail(CP_EUNSUPPORTED, "%s: the postfactor is not a power law", who);
    const long double lambda = powl((long double)post[npad - 1] / (long double)post[0], 1.0L / (npad - 1));
    if (!(lambda > 0.0L)) return cp::fail(CP_EUNSUPPORTED, "%s: the postfactor is not a power law", who);
    for (int j = 0; j + 1 < npad; ++j)
        if (!(fabs(post[j + 1] / (post[j] * (double)lambda) - 1.) < 1e-11)) return cp::fail(CP_EUNSUPPORTED, "%s: the postfactor is not a power law (entry %d)", who, j + 1);
    std::vector<int> qj(nq);
    std::vector<double> qa(nq);
    int jmin, jmax;
    locate_queries(knots, n, queries, nq, qj, qa, &jmin, &jmax);
    if (jmax < 0) return cp::fail(CP_EUNSUPPORTED, "%s: no query inside the knots", who);
    // the kernel keeps the coefficients jmin - 1 ... jmax + 1; periodic ends on the padded grid stand for natural ends on the knots GEO_HALO knots away
    const int ws = jmin, ne = jmax - jmin + 1;
    if (ne + 2 > 64 * GEO_SMAX) return cp::fail(CP_EUNSUPPORTED, "%s: the queries span %d knots (at most %d)", who, ne, 64 * GEO_SMAX - 2);
    if (jmin < GEO_HALO || jmax > n - 1 - GEO_HALO) return cp::fail(CP_EUNSUPPORTED, "%s: the queries come within %d knots of the ends of the grid", who, GEO_HALO);
    long double basis[4][4];
    geometric_bspline_pieces((long double)rho, basis);
    std::vector<double> u((size_t)2 * (npad / 2 + 1));
    {
        // alpha c_{j-1} + beta c_j + gamma c_{j+1} = y_j: the values at s_j (x = 0) of the B-splines centred on s_{j-1}, s_j, s_{j+1}
        const long double alpha = basis[0][0] / lambda, beta = basis[1][0], gamma = basis[2][0] * lambda;
        const long double two_pi = 6.283185307179586476925286766559005768L;
        for (int m = 0; m <= npad / 2; ++m) {
            const long double w = two_pi * m / npad, cw = cosl(w), sw = sinl(w);
            // conj(alpha e^{-i w} + beta + gamma e^{i w}) = (beta + (alpha + gamma) cos w) + i (alpha - gamma) sin w
            const long double dr = beta + (alpha + gamma) * cw, di = (alpha - gamma) * sw, inv = 1.0L / (dr * dr + di * di);
            const long double ur = u_re_im[2 * m], ui = u_re_im[2 * m + 1];
            u[2 * m] = (double)((ur * dr + ui * di) * inv);
            u[2 * m + 1] = (double)((ui * dr - ur * di) * inv);
        }
    }
    cp_geospline_plan* p = new (std::nothrow) cp_geospline_plan();
    if (!p) return cp::fail(CP_ENOMEM, "%s: host allocation failed", who);
    p->n = n; p->nq = nq; p->ws = ws; p->ne = ne; p->S = 0; p->device = device; p->d_qe = nullptr; p->d_qa = nullptr; p->d_consts = nullptr;
    p->prefiltered = nullptr;
    int status = cp_fftlog_plan_create(&p->prefiltered, n, npad, 1, pre, post, u.data(), device);
    if (status == CP_OK) {
        GeoConsts consts{};
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 4; ++e) consts.basis.k[i][e] = (double)basis[i][e];
        for (int q = 0; q < nq; ++q)
            if (qj[q] >= 0) {
                qj[q] -= ws;
                qa[q] = 1. - qa[q];
            }
        status = upload_geospline(p, qj, qa, consts);
    }
    if (status != CP_OK) {
        cp_geospline_plan_destroy(p);
        return status;
    }
    *out = p;
    return CP_OK;
}

// sigma(r, z) of a batch of analytic cosmologies as the fused kernel, its spline evaluated from the B-spline coefficients the plan's transform delivers
extern "C" int cp_sigma_rz_analytic_prefiltered(int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm,
                                                const cp_param* pk_params, int nk, const double* d_k, const cp_geospline_plan* spline,
                                                const double* d_growth_sq, int nz, double* d_out, double* d_pk_out, void* d_work, int device, void* stream) {
    if (ncosmo < 0 || nk <= 0 || nz <= 0) return cp::fail(CP_EINVAL, "cp_sigma_rz_analytic_prefiltered: bad sizes");
    if (ncosmo == 0) return CP_OK;
    if (!bg_params || !pk_params || !d_k || !spline || !d_growth_sq || !d_out || !d_work) return cp::fail(CP_EINVAL, "cp_sigma_rz_analytic_prefiltered: null pointer");
    if (!spline->prefiltered) return cp::fail(CP_EINVAL, "cp_sigma_rz_analytic_prefiltered: the spline plan carries no transform (cp_geospline_plan_create_prefiltered)");
    if (spline->n != nk) return cp::fail(CP_EINVAL, "cp_sigma_rz_analytic_prefiltered: the spline plan has %d knots, the spectra %d samples", spline->n, nk);
    if (spline->device != device) return cp::fail(CP_EINVAL, "cp_sigma_rz_analytic_prefiltered: the spline plan lives on device %d", spline->device);
    cp_fftlog_tables_view f;
    if (!cp_fftlog_plan_view(spline->prefiltered, &f)) return cp::fail(CP_EINVAL, "cp_sigma_rz_analytic_prefiltered: plan without device tables");
    const GeoTail geo{spline->nq, spline->ws, spline->d_qe, spline->d_qa, &spline->d_consts->basis};
    char* coef = static_cast<char*>(d_work);
    coef += (64 - (reinterpret_cast<unsigned long long>(coef) & 63u)) & 63u;
    return sigma_rz_fused_launch(engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, d_k, f, nullptr, &geo, d_growth_sq, nz, d_out, d_pk_out, coef, device,
                                 stream);
}

extern "C" int cp_geospline_plan_info(const cp_geospline_plan* p, int* first_knot, int* nknots, int* nq) {
    if (!p) return cp::fail(CP_EINVAL, "cp_geospline_plan_info: null plan");
    if (first_knot) *first_knot = p->ws;
    if (nknots) *nknots = p->ne;
    if (nq) *nq = p->nq;
    return CP_OK;
}

extern "C" int cp_fftlog_geospline_execute(const cp_fftlog_plan* fftlog, const cp_geospline_plan* spline, const double* d_in, double* d_out,
                                           long long nbatch, int group, int post_op, void* stream) {
    if (nbatch < 0 || group < 0) return cp::fail(CP_EINVAL, "cp_fftlog_geospline_execute: negative size");
    if (nbatch > 2000000000LL) return cp::fail(CP_EUNSUPPORTED, "cp_fftlog_geospline_execute: %lld rows in one call (split the batch)", nbatch);
    if (nbatch == 0) return CP_OK;
    if (!spline || !d_in || !d_out) return cp::fail(CP_EINVAL, "cp_fftlog_geospline_execute: null pointer");
    if ((fftlog != nullptr) == (spline->prefiltered != nullptr))
        return cp::fail(CP_EINVAL, spline->prefiltered ? "cp_fftlog_geospline_execute: a prefiltered spline plan runs its own transform (pass a null fftlog plan)"
                                                       : "cp_fftlog_geospline_execute: null pointer");
    const bool coef = spline->prefiltered != nullptr;
    if (coef) fftlog = spline->prefiltered;
    if (post_op != CP_SPLINE_POST_NONE && post_op != CP_SPLINE_POST_SQRT) return cp::fail(CP_EINVAL, "cp_fftlog_geospline_execute: unknown post op %d", post_op);
    if (group > 0 && ((group & 1) || nbatch % group)) return cp::fail(CP_EINVAL, "cp_fftlog_geospline_execute: %lld rows do not come in groups of an even %d", nbatch, group);
    cp_fftlog_tables_view f;
    if (!cp_fftlog_plan_view(fftlog, &f) || !(f.npad == NP && f.n == NP / 2 && f.nker == 1 && f.in_left == NP / 4 && f.out_left == NP / 4))
        return cp::fail(CP_EUNSUPPORTED, "cp_fftlog_geospline_execute: transform outside the fused kernel's shape (1024 samples padded to 2048, one kernel)");
    if (spline->n != f.n || spline->device != f.device) return cp::fail(CP_EINVAL, "cp_fftlog_geospline_execute: the spline plan is for %d knots on device %d", spline->n, spline->device);
    GeoArgs R{};
    FftlogArgs& A = R.fft;
    A.in = d_in; A.out = nullptr; A.nbatch = nbatch; A.nker = 1; A.n = f.n; A.in_left = f.in_left; A.out_off = f.out_left; A.n_out = f.n;
    A.ext_l = A.ext_r = CP_EXTRAP_CONST; A.val_l = A.val_r = 0.;
    A.stream_rows = (double)nbatch * f.n * 8. > 512. * 1024. * 1024. ? 3 : 0;
    A.pre = f.d_pre; A.post = f.d_post; A.u = f.d_u; A.tw = f.d_tw;
    R.ws = spline->ws; R.ne = spline->ne; R.S = spline->S; R.nq = spline->nq; R.post_sqrt = post_op == CP_SPLINE_POST_SQRT; R.group = group;
    R.ntables = group > 0 ? (int)(nbatch / group) : (int)((nbatch + 1) / 2);
    R.pt = group > 0 ? group / 2 : 1;
    R.consts = spline->d_consts;
    R.qe = spline->d_qe; R.qa = spline->d_qa; R.out = d_out;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != f.device && hipSetDevice(f.device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_fftlog_geospline_execute: cannot select device %d", f.device);
    using F = Fftlog<NP, P, IN_HALF_ZERO, OUT_HALF>;
    // (prefiltered: the coefficients of the stretch behind the FFT's tables; up to 373 knots keep four workgroups on a CU)
    const size_t lds = (size_t)F::LDS_BYTES + (coef ? (((size_t)2 * (spline->ne + 2) * sizeof(double) + 15) / 16) * 16 : 0);
    const bool few = spline->nq <= 256;
    if (lds > 64 * 1024)
        (void)(!coef ? cp::allow_full_lds<&fftlog_geospline_kernel<false>>()
                     : few ? cp::allow_full_lds<&fftlog_geospline_kernel<true, 4>>() : cp::allow_full_lds<&fftlog_geospline_kernel<true, GEO_QMAX>>());
    int ncu = 0;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, f.device);
    const long long npairs = (nbatch + 1) / 2;
    const size_t per_cu = (160 * 1024) / lds;
    const long long resident = (long long)(ncu > 0 ? ncu : 256) * (long long)(per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu));
    long long grid;
    if (group > 0) {      // a multiple of 8 (one share per XCD), every workgroup of an XCD the same number of rounds
        const long long per_xcd = ((nbatch / group + 7) / 8) * (group / 2), slots = std::max(1LL, resident / 8);
        const long long rounds = (per_xcd + slots - 1) / slots;
        grid = 8 * ((per_xcd + rounds - 1) / rounds);
    } else {
        const long long rounds = (npairs + resident - 1) / resident;
        grid = (npairs + rounds - 1) / rounds;
    }
    if (coef && few) hipLaunchKernelGGL((fftlog_geospline_kernel<true, 4>), dim3((unsigned)grid), dim3(NP / P), lds, static_cast<hipStream_t>(stream), R);
    else if (coef) hipLaunchKernelGGL((fftlog_geospline_kernel<true, GEO_QMAX>), dim3((unsigned)grid), dim3(NP / P), lds, static_cast<hipStream_t>(stream), R);
    else hipLaunchKernelGGL(fftlog_geospline_kernel<false>, dim3((unsigned)grid), dim3(NP / P), lds, static_cast<hipStream_t>(stream), R);
    const hipError_t e = hipGetLastError();
    if (prev >= 0) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_fftlog_geospline_execute: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
