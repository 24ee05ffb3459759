// cp_dst.hip -- batched orthonormal DST-II / DST-III of rows (gfx950) + C ABI.
//
// Replaces scipy.fftpack.dst / idst (type=2, norm='ortho') of the wallish2018 BAO filter
// (reference cosmoprimo/bao_filter.py:371-372, 412):
//   Y_k = f_k sum_n x_n sin(pi (k+1) (2n+1) / (2N)),  f_k = sqrt(2/N) (k < N-1), sqrt(1/N) (k = N-1);  inverse = transpose.
// With x'_n = (-1)^n x_n the DST-II is the DCT-II of x' read backwards (Y_k = f_k C'_{N-1-k}), and the DCT-II comes from ONE
// N-point complex FFT of the reordered sequence v_m = x'_{2m}, v_{N-1-m} = x'_{2m+1} (Makhoul 1980):
// C'_k = Re(e^{-i pi k / 2N} V_k).  Two real rows are packed as v_a + i v_b and separated after the FFT through
// V_a[k] = (V[k] + conj V[N-k]) / 2, V_b[k] = (V[k] - conj V[N-k]) / 2i.  The inverse runs the same network backwards:
// Hermitian-symmetrised spectra H = H_a + i H_b, v = conj(FFT(conj H)).
// The FFT reuses the pass machinery of the FFTLog kernel (cp_fft_core.h): radix-16 butterflies in registers, swizzled
// LDS exchanges, middle-pass twiddles resident in LDS.  Optional fused elementwise maps for the filter:
// forward input log(k_n x_n), inverse output exp(y_n) / k_n.  HBM traffic: 16 N bytes per row (read + write), HBM-bound.
#include <hip/hip_runtime.h>

#include <cmath>
#include <new>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_internal.h"
#include "cp_fft_core.h"
#include "cp_fftlog_tables.h"
#include "cp_math.h"
#include "cp_power_eval.h"
#include "cp_wallish_dd.h"
#include "cp_splice_uniform_plan.h"

namespace {

using namespace cpfft;

struct Args {
    const double* in;   // (nrows, N)
    double* out;        // (nrows, N)
    long long nrows;
    const cplx* tw;     // Plan twiddles
    const cplx* rot;    // (N) e^{-i pi k / (2N)} = (cos, -sin)
    const double* kx;   // (N) optional abscissa for the fused maps, or nullptr
    const double* ikx;  // (N) its reciprocals (the plan's table), or nullptr
    int fused;          // forward: x_n := log(kx_n * in_n); inverse: out_n := exp(y_n) / kx_n
    int split;          // coefficients stored de-interleaved: Y_0, Y_2, ... in the first half of the row, Y_1, Y_3, ... in the second
};

// LDS position of frequency k after the full DIF network (digit reversal of the mixed-radix plan)
template <int N, int P>
__device__ __forceinline__ int pos_of_freq(int k) {
    using PL = Plan<N, P>;
    int pos = 0;
#pragma unroll
    for (int i = 0; i < PL::NPASS; ++i) {
        const int R = PL::radix(i), M = PL::len(i) / R;
        pos += (k % R) * M;
        k /= R;
    }
    return pos;
}

template <int N, int P>
__device__ __forceinline__ cplx lds_at(const cplx* lds, int pos) {
    return lds[swz<N>(pos)];
}

// all DIF passes; pass 0 takes x from registers, the result is left in LDS in digit-reversed order
template <int N, int P, int I>
__device__ __forceinline__ void dif_rest(int t, const Args& A, cplx* lds, const cplx* ltw) {
    using PL = Plan<N, P>;
    if constexpr (I < PL::NPASS) {
        cplx x[P];
        __syncthreads();
        asm volatile("" : "+v"(t));  // no address sharing across passes (register pressure, see cp_fftlog_body.h)
        Pass<N, P, I>::load_lds(t, lds, x);
        Pass<N, P, I>::butterflies(x);
        Pass<N, P, I>::twiddle_apply_lds(t, ltw + (PL::tw_offset(I) - N), x);
        Pass<N, P, I>::store_lds(t, lds, x);
        dif_rest<N, P, I + 1>(t, A, lds, ltw);
    }
}

template <int N, int P>
__device__ __forceinline__ void dif_all(int t, const Args& A, cplx* x, cplx* lds, const cplx* ltw) {
    using PL = Plan<N, P>;
    Pass<N, P, 0>::butterflies(x);
    cplx w[P];
    Pass<N, P, 0>::twiddle_load(t, A.tw + PL::tw_offset(0), w);
    Pass<N, P, 0>::twiddle_apply(w, x);
    Pass<N, P, 0>::store_lds(t, lds, x);
    dif_rest<N, P, 1>(t, A, lds, ltw);
    __syncthreads();
}

#ifndef CP_DST_INVERSE_VIA_LDS      // inverse transform: 1 the rows are read once, in memory order, and handed to the threads through LDS; 0 every coefficient is read by
#define CP_DST_INVERSE_VIA_LDS 1    // two threads (one an iteration ahead, one in place)
#endif
#ifndef CP_DST_ABLATE      // diagnostic builds of the inverse transform (wrong results): 1 no fused map, 2 no in-place loads of the second half, 4 no stores, 8 no transform
#define CP_DST_ABLATE 0
#endif

template <int N, int P, bool INVERSE>
__global__ __launch_bounds__(N / P, 2) void dst_kernel(const Args A) {
    using PL = Plan<N, P>;
    constexpr int T = PL::T;
    extern __shared__ __attribute__((aligned(4096))) char smem[];  // 4096: see LdsView (cp_fft_core.h)
    cplx* lds = reinterpret_cast<cplx*>(smem);
    cplx* ltw = lds + lds_data_slots(N, P);  // data slots of one packed pair (padded layout for N = 4096), then the tables
    const int t = threadIdx.x;
    for (int i = t; i < PL::TW_TOTAL - N; i += T) ltw[i] = A.tw[N + i];
    const long long npairs = (A.nrows + 1) / 2;
    const double fn = sqrt(2. / N), fl = sqrt(1. / N);
    // Rows stay independent although two of them share one complex FFT (scipy transforms row by row): a row holding a sample that is
    // not finite -- or, for the fused log map, not positive -- goes into the transform as a harmless constant and is stored as NaN; its
    // partner is untouched.  The flags of the pair are raised by whichever thread meets such a sample and read behind one barrier.
    __shared__ int bad_row[2];
    __shared__ int pair_flag;   // inverse transform: some sample of the pair is not finite
    __shared__ cpmath::MathTables mt;      // the table-driven exponential of the fused inverse map (the barrier at the top of the first pair covers the fill)
    if (INVERSE) cpmath::fill_math_tables(&mt);
    const double nan = __builtin_nan("");
    // forward transform: the rows of the NEXT pair are fetched (in Makhoul order) while this pair is transformed -- the loads of a pair used to sit
    // at the top of its iteration, in front of everything that needs them, with two waves per SIMD to cover their latency
    double na[P], nb[P];
    auto fetch = [&](long long p) {
        const double* ra = A.in + 2 * p * N;
        const double* rb = 2 * p + 1 < A.nrows ? ra + N : ra;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            const int m = t + T * r;
            const int n = m < N / 2 ? 2 * m : 2 * (N - 1 - m) + 1;
            na[r] = ra[n];
            nb[r] = rb[n];
        }
    };
    // inverse transform: half of a pair's samples (the coefficients N - 1 - k of both rows) the same way; the other half (k - 1) are read in place
    auto at_ = [&](int j) { return A.split ? ((j & 1) * (N / 2) + (j >> 1)) : j; };
    auto fetch_inverse = [&](long long p) {
        const double* ra = A.in + 2 * p * N;
        const double* rb = 2 * p + 1 < A.nrows ? ra + N : ra;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            // CP_DST_INVERSE_VIA_LDS: the rows as they lie in memory (every coefficient read ONCE; the threads pick what they need out of LDS)
            const int ia = CP_DST_INVERSE_VIA_LDS ? t + T * r : at_(N - 1 - (t + T * r));
            na[r] = ra[ia];
            nb[r] = rb[ia];
        }
    };
    if ((long long)blockIdx.x < npairs) {
        if (INVERSE) fetch_inverse(blockIdx.x);
        else fetch(blockIdx.x);
    }
    for (long long p = blockIdx.x; p < npairs; p += gridDim.x) {
        const bool has_b = 2 * p + 1 < A.nrows;
        const double* ra = A.in + 2 * p * N;
        const double* rb = has_b ? ra + N : ra;
        double* oa = A.out + 2 * p * N;
        double* ob = has_b ? oa + N : oa;
        cplx x[P];
        // position of coefficient j in its row: the wallish2018 filter treats even- and odd-indexed coefficients as two sequences
        // (bao_filter.py:373), so they can be written / read as two half rows instead of being gathered by separate copy kernels
        auto at = [&](int j) { return A.split ? ((j & 1) * (N / 2) + (j >> 1)) : j; };
        if (t == 0) bad_row[0] = bad_row[1] = pair_flag = 0;
        __syncthreads();  // LDS reuse across pairs (and the table fill on the first one)
        int tt = t;
        asm volatile("" : "+v"(tt));
        if constexpr (!INVERSE) {
            // Makhoul reordering with the (-1)^n sign folded in
            bool bad_a = false, bad_b = false;
#pragma unroll
            for (int r = 0; r < P; ++r) {
                const int m = tt + T * r;
                const bool lower = m < N / 2;
                const int n = lower ? 2 * m : 2 * (N - 1 - m) + 1;
                double a = na[r], b = nb[r];
                bad_a |= !(fabs(a) <= 1.7976931348623157e308) || (A.fused && !(a > 0.));
                bad_b |= !(fabs(b) <= 1.7976931348623157e308) || (A.fused && !(b > 0.));
                if (A.fused) {
                    const double kk = A.kx[n];
                    a = cpmath::log_pos(kk * a);      // rows with a sample that is not positive are flagged above and left out
                    b = cpmath::log_pos(kk * b);
                }
                x[r].re = lower ? a : -a;
                x[r].im = has_b ? (lower ? b : -b) : 0.;
            }
            if (p + gridDim.x < npairs) fetch(p + gridDim.x);
            if (bad_a) bad_row[0] = 1;
            if (bad_b) bad_row[1] = 1;
            __syncthreads();
            const bool skip_a = bad_row[0] != 0, skip_b = bad_row[1] != 0;
            if (skip_a | skip_b) {
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    if (skip_a) x[r].re = 0.;
                    if (skip_b) x[r].im = 0.;
                }
            }
            dif_all<N, P>(tt, A, x, lds, ltw);
            asm volatile("" : "+v"(tt));
            // separate the two rows, rotate, scale; frequency k -> output index N - 1 - k
#pragma unroll 4
            for (int s = 0; s < P; ++s) {
                const int k = tt + T * s;
                const cplx v = lds_at<N, P>(lds, pos_of_freq<N, P>(k));
                const cplx u = lds_at<N, P>(lds, pos_of_freq<N, P>((N - k) % N));
                const cplx rot = A.rot[k];  // (cos, -sin)
                const double f = k == 0 ? fl : fn;
                // V_a = ((p + r)/2, (q - s)/2), V_b = ((q + s)/2, (r - p)/2);  C' = cos * re + sin * im
                const double ya = 0.5 * (rot.re * (v.re + u.re) - rot.im * (v.im - u.im));
                const double yb = 0.5 * (rot.re * (v.im + u.im) - rot.im * (u.re - v.re));
                oa[at(N - 1 - k)] = skip_a ? nan : f * ya;
                if (has_b) ob[at(N - 1 - k)] = skip_b ? nan : f * yb;
            }
        } else {
            // Hermitian-symmetrised, conjugated spectrum of the pair; a row that is skipped does not take part
            const double2* staged = reinterpret_cast<const double2*>(lds);      // (a, b) of the pair at their positions in memory (CP_DST_INVERSE_VIA_LDS)
            if (CP_DST_INVERSE_VIA_LDS) {
                double2* w = reinterpret_cast<double2*>(lds);
#pragma unroll
                for (int r = 0; r < P; ++r) w[tt + T * r] = double2{na[r], nb[r]};
                __syncthreads();
            }
            auto spectrum = [&](bool keep_a, bool keep_b, bool prefetched) {
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    const int k = tt + T * r;
                    const double fa = k == 0 ? fl : fn;                 // f_{N-1-k}
                    const double fb = (k == 0 || k == 1) ? (k == 0 ? fl : fn) : fn;   // f_{k-1} (k = 0 handled below)
                    const int ia = N - 1 - k, ib = k == 0 ? N - 1 : k - 1;
                    double Aa, Ab, Ba, Bb;
                    if (CP_DST_INVERSE_VIA_LDS && prefetched) {
                        const double2 va = staged[at(ia)], vb = staged[at(ib)];
                        Aa = fa * va.x; Ab = fa * va.y;
                        Ba = (k == 0 ? fa : fb) * vb.x; Bb = (k == 0 ? fa : fb) * vb.y;
                    } else {
                        Aa = fa * (prefetched ? na[r] : ra[at(ia)]); Ab = fa * (prefetched ? nb[r] : rb[at(ia)]);
                        Ba = (k == 0 ? fa : fb) * ((CP_DST_ABLATE & 2) ? na[r] : ra[at(ib)]); Bb = (k == 0 ? fa : fb) * ((CP_DST_ABLATE & 2) ? nb[r] : rb[at(ib)]);
                    }
                    const cplx rot = A.rot[k];
                    const double cs = rot.re, sn = -rot.im;
                    cplx Ha, Hb;
                    if (k == 0) {
                        Ha = cplx{Aa, 0.};
                        Hb = cplx{Ab, 0.};
                    } else {
                        Ha = cplx{0.5 * (Aa * cs + Ba * sn), 0.5 * (Aa * sn - Ba * cs)};
                        Hb = cplx{0.5 * (Ab * cs + Bb * sn), 0.5 * (Ab * sn - Bb * cs)};
                    }
                    if (!keep_a) Ha = cplx{0., 0.};
                    if (!keep_b) Hb = cplx{0., 0.};
                    // conj(H_a + i H_b)
                    x[r].re = Ha.re - Hb.im;
                    x[r].im = -(Ha.im + Hb.re);
                }
            };
            spectrum(true, has_b, true);
            if (p + gridDim.x < npairs) fetch_inverse(p + gridDim.x);
            // a sample that is not finite shows in the packed spectrum, though not which row it came from: the pair is flagged from the registers
            // (no second pass over the rows), and only a flagged pair -- rare -- reads its rows again to tell the two apart and leave the bad one out
            bool suspect = false;
#pragma unroll
            for (int r = 0; r < P; ++r) suspect |= !(fabs(x[r].re) <= 1.7976931348623157e308) || !(fabs(x[r].im) <= 1.7976931348623157e308);
            if (suspect) pair_flag = 1;
            __syncthreads();
            bool skip_a = false, skip_b = false;
            if (pair_flag) {     // uniform over the workgroup
                bool bad_a = false, bad_b = false;
                asm volatile("" : "+v"(tt));
#pragma unroll 1
                for (int r = 0; r < P; ++r) {
                    const int k = tt + T * r;
                    const int ia = N - 1 - k, ib = k == 0 ? N - 1 : k - 1;
                    bad_a |= !(fabs(ra[at(ia)]) <= 1.7976931348623157e308) || !(fabs(ra[at(ib)]) <= 1.7976931348623157e308);
                    bad_b |= !(fabs(rb[at(ia)]) <= 1.7976931348623157e308) || !(fabs(rb[at(ib)]) <= 1.7976931348623157e308);
                }
                if (bad_a) bad_row[0] = 1;
                if (bad_b) bad_row[1] = 1;
                __syncthreads();
                skip_a = bad_row[0] != 0;
                skip_b = bad_row[1] != 0;
                // finite rows whose sum overflowed in the packing stay as they are (skip_a = skip_b = false): nothing to separate
                if (skip_a | skip_b) spectrum(!skip_a, has_b && !skip_b, false);
            }
            if (!(CP_DST_ABLATE & 8)) dif_all<N, P>(tt, A, x, lds, ltw);
            else {
                Pass<N, P, 0>::store_lds(tt, lds, x);
                __syncthreads();
            }
            asm volatile("" : "+v"(tt));
            // LDS holds (v_a[m], -v_b[m]) at pos(m); undo the reordering and the (-1)^n sign
#pragma unroll 4
            for (int s = 0; s < P; ++s) {
                const int n = tt + T * s;
                const bool even = (n & 1) == 0;
                const int m = even ? n / 2 : N - 1 - (n - 1) / 2;
                const cplx g = lds_at<N, P>(lds, pos_of_freq<N, P>(m));
                double ya = even ? g.re : -g.re;
                double yb = even ? -g.im : g.im;
                if (A.fused && !(CP_DST_ABLATE & 1)) {
                    const double ik = A.ikx ? A.ikx[n] : cpmath::recip(A.kx[n]);
                    ya = cpmath::exp_tab(ya, &mt) * ik;
                    yb = cpmath::exp_tab(yb, &mt) * ik;
                }
                if ((CP_DST_ABLATE & 4) && ya != 12345.678) continue;
                oa[n] = skip_a ? nan : ya;
                if (has_b) ob[n] = skip_b ? nan : yb;
            }
        }
    }
}

// ---- forward transform of log(k P_c(k)) with the spectra of an analytic engine EVALUATED in the kernel (wallish2018 on a batch of cosmologies,
// bao_filter.py:371 behind eisenstein_hu.py:315-324): the evaluation (~400 fp64 instructions per sample, vector-ALU bound) and the transform
// (16 N bytes of LDS / HBM traffic per row, memory bound) were two launches with 4096 doubles per vector written and read between them; here a
// workgroup evaluates the 2 x 4096 samples of a pair of cosmologies straight into the transform's input (Makhoul order, (-1)^n folded in) and the
// chip sees both kinds of work at once, as in the fused sigma(r, z) kernel (cp_sigma.hip).  The arithmetic of a sample is power_kernel's
// CP_PK_LOG_K_MATTER (cp_power.hip), term by term, with k^1.08 and k^1.4 from tables next to log k (transfer_eh_powers) where power_kernel takes
// exponentials of the tabulated logarithm: the two agree to rounding (3e-16), not bit for bit.
struct GenArgs {
    Args dst;                        // out, tw, rot, split; in / kx / fused unused
    long long ncosmo;
    cpcosmo::Param bg[CP_BG_NPARAMS];
    cpcosmo::Param pw[CP_PK_NPARAMS];
    int second_is_omega_m;
    const double* ncdm_tab;          // massive neutrinos (cp_ncdm.tab; nsp == 0: none): today's densities only -- Omega0_m of pk_callable, Omega_m of BBKS
    int nsp;
    const double* k;                 // (N) wavenumbers, h/Mpc
    const double* ln_k;              // (N) their logarithms (log_pos, as power_kernel takes them)
    const cppower::CosmoConsts* consts;  // (ncosmo) the cosmologies' constants (cp_power_coefficients)
    // the next step of wallish2018 in the epilogue (bao_filter.py:373-405): second derivatives of the clamped splines through the even- and the
    // odd-indexed coefficients, the box between their maxima, the box rewritten -- box != null (split layout only)
    int* box;                        // (2 ncosmo, 2) or null
    int margin_first, margin_second, off0, off1;
};

#ifndef CP_DST_GEN_ILP
#define CP_DST_GEN_ILP 2
#endif
#ifndef CP_DST_GEN_PREFETCH
#define CP_DST_GEN_PREFETCH 1
#endif

// ln_k: (4, n) log k, k^1.08, k^1.4 (cp_power_eval.h), 1 / k; then (n, 4): (k, log k, k^1.08, k^1.4) of the sample the generating transform puts at
// position m of its reordered sequence (Makhoul's order: m < n / 2: sample 2 m, else 2 (n - 1 - m) + 1) -- a thread of generate_row walks m = t + T r:
// its four numbers are 32 contiguous bytes, a wave's 2 KB, where four tables indexed by the sample number took four scattered 8-byte loads and their addresses
__global__ void dst_log_kernel(const double* k, double* ln_k, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        cppower::powers_of_wavenumber(k[i], ln_k, i, n);
        ln_k[3 * n + i] = 1. / k[i];
        const int m = (i & 1) ? n - 1 - (i - 1) / 2 : i / 2;
        double* g = ln_k + 4 * (size_t)n + 4 * (size_t)m;
        g[0] = k[i];
        g[1] = ln_k[i];
        g[2] = ln_k[n + i];
        g[3] = ln_k[2 * n + i];
    }
}

// log(k P(k)) of cosmology ic at the thread's P samples m = t + T r of the reordered sequence, into the thread's own slots of the data region
// (double index 2 (t + T r) + row).  NOT unrolled: sixteen copies of a 400-instruction evaluation would not fit the instruction cache.
// GP: a pointer to the arguments -- const GenArgs* (a by-value kernel parameter: dst_generate_kernel), or GenArgsK: the kernel-argument segment itself, read
// through the constant address space where a value is used (wallish_full_kernel: its arguments held in scalar registers over the whole loop over pairs were
// ~800 v_readlane / v_writelane among the 9 300 vector instructions of a pair).  Everything is taken into locals that live as long as this function.
typedef const GenArgs __attribute__((address_space(4))) * GenArgsK;

template <int N, int P, int ENGINE, typename GP>
__device__ __forceinline__ void generate_row(GP G, long long ic, int t, double* slots, const cpmath::MathTables* mt) {
    using namespace cppower;
    constexpr int T = N / P;
    const CosmoConsts K = load_uniform(G->consts + ic);      // (scalar loads: the cosmology's constants in scalar registers)
    const PkPerCosmology& pc = K.pk;
    const double2* gen = reinterpret_cast<const double2*>(G->ln_k + 4 * (size_t)N);      // (k, log k), (k^1.08, k^1.4) of position m of the reordered sequence (dst_log_kernel)
    const double ln_pk_unit = K.ln_pk_unit;
    // CP_DST_GEN_ILP samples per iteration: independent chains of logarithms / exponentials / reciprocals for the two waves of a SIMD to interleave
    // ... whose table entries are requested an iteration ahead (CP_DST_GEN_PREFETCH: the loads of iteration r0 + ILP are in flight under the ~500
    // instructions of iteration r0; at the top of the iteration that uses them they sat in front of everything, with one other wave on the SIMD to cover them)
    double2 kl[CP_DST_GEN_ILP], pw[CP_DST_GEN_ILP];
#pragma unroll
    for (int u = 0; u < CP_DST_GEN_ILP; ++u) {
        kl[u] = gen[2 * (t + T * u)];
        pw[u] = gen[2 * (t + T * u) + 1];
    }
#pragma unroll 1
    for (int r0 = 0; r0 < P; r0 += CP_DST_GEN_ILP) {
        double2 kl_now[CP_DST_GEN_ILP], pw_now[CP_DST_GEN_ILP];
#pragma unroll
        for (int u = 0; u < CP_DST_GEN_ILP; ++u) {
            kl_now[u] = kl[u];
            pw_now[u] = pw[u];
            if (CP_DST_GEN_PREFETCH) {
                const int mn = t + T * ((r0 + CP_DST_GEN_ILP + u) & (P - 1));      // (the last iteration asks for the first entries again: never used)
                kl[u] = gen[2 * mn];
                pw[u] = gen[2 * mn + 1];
            } else if (r0 + CP_DST_GEN_ILP < P) {
                const int mn = t + T * (r0 + CP_DST_GEN_ILP + u);
                kl[u] = gen[2 * mn];
                pw[u] = gen[2 * mn + 1];
            }
        }
        if (!CP_DST_GEN_PREFETCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (measurements: the entries waited for where they are requested)
#pragma unroll
        for (int u = 0; u < CP_DST_GEN_ILP; ++u) {
            const int m = t + T * (r0 + u);
            const double kh = kl_now[u].x, ln_kh = kl_now[u].y;
            const double Tk = transfer_any<ENGINE>(K, kh, ln_kh, pw_now[u].x, pw_now[u].y, mt);
            slots[2 * m] = 2. * (ln_kh + (CP_MATH_TABLES_OFF ? log_pos(fabs(Tk)) : log_tab_any(fabs(Tk), mt))) + ln_pk_unit + primordial_tilt_exponent(pc, ln_kh);
        }
    }
}

template <int N, int P, int ENGINE>
__global__ __launch_bounds__(N / P, 2) void dst_generate_kernel(const GenArgs G) {
    using PL = Plan<N, P>;
    constexpr int T = PL::T;
    const Args& A = G.dst;
    extern __shared__ __attribute__((aligned(4096))) char smem[];
    cplx* lds = reinterpret_cast<cplx*>(smem);
    cplx* ltw = lds + lds_data_slots(N, P);
    const int t = threadIdx.x;
    for (int i = t; i < PL::TW_TOTAL - N; i += T) ltw[i] = A.tw[N + i];
    double* dd_tabs = reinterpret_cast<double*>(ltw + (PL::TW_TOTAL - N));      // 2 x DD_NTAB elimination factors of the box step (G.box only)
    if (G.box) cpdd::fill_tables(dd_tabs);
    const long long npairs = (G.ncosmo + 1) / 2;
    const double fn = sqrt(2. / N), fl = sqrt(1. / N);
    __shared__ int bad_row[2];
    __shared__ cpmath::MathTables mt;      // (the barrier at the top of the first pair covers it)
    cpmath::fill_math_tables(&mt);
    const double nan = __builtin_nan("");
    static_assert(!padded_lds(N, P), "the generated samples go through natural-order slots of the data region");
    for (long long p = blockIdx.x; p < npairs; p += gridDim.x) {
        const bool has_b = 2 * p + 1 < G.ncosmo;
        double* oa = A.out + 2 * p * N;
        double* ob = has_b ? oa + N : oa;
        auto at = [&](int j) { return A.split ? ((j & 1) * (N / 2) + (j >> 1)) : j; };
        if (t == 0) bad_row[0] = bad_row[1] = 0;
        __syncthreads();  // LDS reuse across pairs (and the table fill on the first one)
        double* slots = reinterpret_cast<double*>(lds);
        int tt = t;
        asm volatile("" : "+v"(tt));      // (nothing derived from the thread's number is kept in registers from one pair to the next: what the compiler hoists
                                           // out of this loop lives through the transform, in registers the kernel does not have -- spilled and reloaded)
        generate_row<N, P, ENGINE>(&G, 2 * p, tt, slots, cpmath::tables_present(&mt));
        if (has_b) generate_row<N, P, ENGINE>(&G, 2 * p + 1, tt, slots + 1, cpmath::tables_present(&mt));
        cplx x[P];
        bool bad_a = false, bad_b = false;
#pragma unroll
        for (int r = 0; r < P; ++r) {      // the thread's own slots: no barrier
            const int m = tt + T * r;
            const bool lower = m < N / 2;
            const double a = slots[2 * m], b = has_b ? slots[2 * m + 1] : 0.;
            bad_a |= !(fabs(a) <= 1.7976931348623157e308);
            bad_b |= !(fabs(b) <= 1.7976931348623157e308);
            x[r].re = lower ? a : -a;
            x[r].im = lower ? b : -b;
        }
        if (bad_a) bad_row[0] = 1;
        if (bad_b) bad_row[1] = 1;
        __syncthreads();      // flags published; every thread has read its slots before the first pass overwrites the region
        const bool skip_a = bad_row[0] != 0, skip_b = bad_row[1] != 0;
        if (skip_a | skip_b) {
#pragma unroll
            for (int r = 0; r < P; ++r) {
                if (skip_a) x[r].re = 0.;
                if (skip_b) x[r].im = 0.;
            }
        }
        asm volatile("" : "+v"(tt));
        dif_all<N, P>(tt, A, x, lds, ltw);
        asm volatile("" : "+v"(tt));
        if (!G.box) {
#pragma unroll 4
            for (int s = 0; s < P; ++s) {
                const int k = tt + T * s;
                const cplx v = lds_at<N, P>(lds, pos_of_freq<N, P>(k));
                const cplx u = lds_at<N, P>(lds, pos_of_freq<N, P>((N - k) % N));
                const cplx rot = A.rot[k];
                const double f = k == 0 ? fl : fn;
                const double ya = 0.5 * (rot.re * (v.re + u.re) - rot.im * (v.im - u.im));
                const double yb = 0.5 * (rot.re * (v.im + u.im) - rot.im * (u.re - v.re));
                oa[at(N - 1 - k)] = skip_a ? nan : f * ya;
                if (has_b) ob[at(N - 1 - k)] = skip_b ? nan : f * yb;
            }
            continue;
        }
        // ---- the coefficients stay on the CU for the next step of the filter: the four sequences of the pair (even / odd coefficients of rows a, b),
        // one per wave, in the data region of the transform (4 x 2048 doubles, XOR layout) ----
        if constexpr (N == 4096) {
            using namespace cpdd;
            constexpr int S = 32, NS = N / 2;
            double va[P], vb[P];
#pragma unroll
            for (int s = 0; s < P; ++s) {
                const int k = tt + T * s;
                const cplx v = lds_at<N, P>(lds, pos_of_freq<N, P>(k));
                const cplx u = lds_at<N, P>(lds, pos_of_freq<N, P>((N - k) % N));
                const cplx rot = A.rot[k];
                const double f = k == 0 ? fl : fn;
                va[s] = skip_a ? nan : f * (0.5 * (rot.re * (v.re + u.re) - rot.im * (v.im - u.im)));
                vb[s] = skip_b ? nan : f * (0.5 * (rot.re * (v.im + u.im) - rot.im * (u.re - v.re)));
            }
            __syncthreads();      // every thread has its coefficients: the data region is free
            double* seqs = reinterpret_cast<double*>(lds);
#pragma unroll
            for (int s = 0; s < P; ++s) {
                const int j = N - 1 - (tt + T * s);      // coefficient j of its row: sequence (j & 1), knot j >> 1
                const int slot = Xor32Layout::at(j >> 1);
                seqs[(j & 1) * NS + slot] = va[s];
                seqs[(2 + (j & 1)) * NS + slot] = vb[s];
            }
            __syncthreads();
            int lane = tt;
            asm volatile("" : "+v"(lane));      // (nothing of the tail is hoisted out of the loop over pairs: 32 squared abscissae and 32 slots would live through the transform)
            const int wave = __builtin_amdgcn_readfirstlane(lane >> 6);
            lane &= 63;
            if (wave < 2 || has_b) {
                double* buf = seqs + wave * NS;
                const long long row = 2 * p + (wave >> 1);
                double* seq = A.out + row * N + (wave & 1) * NS;      // split layout: [even-indexed | odd-indexed] coefficients of the row
                double own[S];      // the wave's sequence, knot lane + 64 k in register k: the solve overwrites the buffer
#pragma unroll
                for (int k = 0; k < S; ++k) own[k] = buf[Xor32Layout::at(lane + 64 * k)];
                int first, second;
                second_derivatives_and_box<S, Xor32Layout>(buf, dd_tabs, lane, G.margin_first, G.margin_second, first, second);
                const long long srow = 2 * row + (wave & 1);
                if (lane == 0) {
                    G.box[2 * srow] = first + G.off0;
                    G.box[2 * srow + 1] = second + G.off1;
                }
                wave_lds_phase();      // the last reads of M are done: the buffer is free
                const int a = first + G.off0, b = second + G.off1;
                const bool removed = remove_box<S>(buf, dd_tabs + DD_NTAB, lane, a, b,
                                                   [&](int lo, int L, int R, int hi, double* zl, double* zr) {
#pragma unroll
                                                       for (int k = 0; k < S; ++k) {
                                                           const int i = lane + 64 * k;
                                                           const double x = (double)(i + 1);
                                                           if (i >= lo && i <= L) zl[i - lo] = own[k] * (x * x);
                                                           if (i >= R && i <= hi) zr[i - R] = own[k] * (x * x);
                                                       }
                                                   },
                                                   seq);
                // the knots that stay: every address of the sequence is written exactly once (the box by remove_box, the rest here)
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const int i = lane + 64 * k;
                    if (!(removed && i >= a && i <= b)) seq[i] = own[k];
                }
            }
        }
    }
}


#ifndef CP_TAIL_HEADER      // (measurements: another version of the kernel, tools/variants/, built beside the shipped library by tools/variant_lib.sh)
#define CP_TAIL_HEADER "cp_wallish_tail.h"
#endif
#include CP_TAIL_HEADER      // wallish_tail_kernel: everything of the filter behind its forward transform (cp_wallish_tail below)

template <int N>
void launch(bool inverse, const Args& A, int grid, hipStream_t stream) {
    constexpr int P = 16, T = N / P;
    constexpr int lds = (lds_data_slots(N, P) + Plan<N, P>::TW_TOTAL - N) * (int)sizeof(cplx);
    if (lds > 64 * 1024) {  // opt in to more than 64 KiB of dynamic LDS (once per process would do; the call is cheap)
        (void)cp::allow_full_lds<&dst_kernel<N, P, true>>();
        (void)cp::allow_full_lds<&dst_kernel<N, P, false>>();
    }
    if (inverse) hipLaunchKernelGGL((dst_kernel<N, P, true>), dim3(grid), dim3(T), lds, stream, A);
    else hipLaunchKernelGGL((dst_kernel<N, P, false>), dim3(grid), dim3(T), lds, stream, A);
}

}  // namespace

struct cp_dst_plan {
    int n, device;
    cplx* d_tw;
    cplx* d_rot;
    double* d_kx;
    double* d_ln_kx;      // log of the abscissa, filled at the first cp_dst_forward_analytic (device kernel: the bits of power_kernel's own logarithm)
};

extern "C" int cp_dst_plan_destroy(cp_dst_plan* p) {
    if (!p) return CP_OK;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device) (void)hipSetDevice(p->device);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_rot) (void)hipFree(p->d_rot);
    if (p->d_kx) (void)hipFree(p->d_kx);
    if (p->d_ln_kx) (void)hipFree(p->d_ln_kx);
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    delete p;
    return CP_OK;
}

extern "C" int cp_dst_plan_create(cp_dst_plan** out, int n, const double* kx, int device) {
    if (!out) return cp::fail(CP_EINVAL, "cp_dst_plan_create: null plan pointer");
    *out = nullptr;
    if (n != 256 && n != 1024 && n != 4096) return cp::fail(CP_EUNSUPPORTED, "cp_dst_plan_create: length %d not instantiated (256, 1024, 4096)", n);
    std::vector<cplx> tw, rot(n);
    if (n == 256) build_twiddles<256, 16>(tw);
    else if (n == 1024) build_twiddles<1024, 16>(tw);
    else build_twiddles<4096, 16>(tw);
    for (int k = 0; k < n; ++k) rot[k] = unit_root(k, 4LL * n);  // e^{-2 pi i k / 4N} = e^{-i pi k / 2N}
    cp_dst_plan* p = new (std::nothrow) cp_dst_plan();
    if (!p) return cp::fail(CP_ENOMEM, "cp_dst_plan_create: host allocation failed");
    p->n = n; p->device = device; p->d_tw = nullptr; p->d_rot = nullptr; p->d_kx = nullptr; p->d_ln_kx = nullptr;
    int prev = -1, status = CP_OK;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) status = cp::fail(CP_EDEVICE, "cp_dst_plan_create: cannot select device %d", device);
    if (status == CP_OK && (hipMalloc(&p->d_tw, tw.size() * sizeof(cplx)) != hipSuccess || hipMalloc(&p->d_rot, n * sizeof(cplx)) != hipSuccess ||
                            (kx && (hipMalloc(&p->d_kx, n * sizeof(double)) != hipSuccess || hipMalloc(&p->d_ln_kx, 8 * (size_t)n * sizeof(double)) != hipSuccess))))
        status = cp::fail(CP_ENOMEM, "cp_dst_plan_create: device allocation failed");
    if (status == CP_OK && (hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemcpy(p->d_rot, rot.data(), n * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess ||
                            (kx && hipMemcpy(p->d_kx, kx, n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)))
        status = cp::fail(CP_EDEVICE, "cp_dst_plan_create: upload failed");
    if (status == CP_OK && kx) {      // log of the abscissa with the kernels' own logarithm (default stream, then waited for: plan creation is synchronous)
        hipLaunchKernelGGL(dst_log_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, p->d_kx, p->d_ln_kx, n);
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) status = cp::fail(CP_EDEVICE, "cp_dst_plan_create: cannot tabulate log k");
    }
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (status != CP_OK) {
        cp_dst_plan_destroy(p);
        return status;
    }
    *out = p;
    return CP_OK;
}

extern "C" int cp_dst_execute(const cp_dst_plan* p, const double* d_in, double* d_out, long long nrows, int inverse, int flags, void* stream) {
    if (flags & ~(CP_DST_FUSED | CP_DST_SPLIT)) return cp::fail(CP_EINVAL, "cp_dst_execute: unknown flags %d", flags);
    const int fused = flags & CP_DST_FUSED;
    if (!p) return cp::fail(CP_EINVAL, "cp_dst_execute: null plan");
    if (nrows < 0) return cp::fail(CP_EINVAL, "cp_dst_execute: negative row count");
    if (nrows == 0) return CP_OK;
    if (!d_in || !d_out) return cp::fail(CP_EINVAL, "cp_dst_execute: null device pointer");
    if (fused && !p->d_kx) return cp::fail(CP_EINVAL, "cp_dst_execute: the fused log / exp maps need the abscissa given at plan creation");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_dst_execute: cannot select device %d", p->device);
    Args A;
    A.in = d_in; A.out = d_out; A.nrows = nrows; A.tw = p->d_tw; A.rot = p->d_rot; A.kx = p->d_kx; A.ikx = p->d_ln_kx ? p->d_ln_kx + 3 * (size_t)p->n : nullptr; A.fused = fused; A.split = (flags & CP_DST_SPLIT) != 0;
    const long long npairs = (nrows + 1) / 2;
    const int grid = (int)(npairs < 512 ? npairs : 512);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (p->n == 256) launch<256>(inverse != 0, A, grid, s);
    else if (p->n == 1024) launch<1024>(inverse != 0, A, grid, s);
    else launch<4096>(inverse != 0, A, grid, s);
    hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_dst_execute: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

namespace {

template <int ENGINE>
void launch_generate(const GenArgs& G, int grid, hipStream_t stream) {
    constexpr int N = 4096, P = 16, T = N / P;
    constexpr int lds = (lds_data_slots(N, P) + Plan<N, P>::TW_TOTAL - N) * (int)sizeof(cplx) + 2 * cpdd::DD_NTAB * (int)sizeof(double);
    (void)cp::allow_full_lds<&dst_generate_kernel<N, P, ENGINE>>();
    hipLaunchKernelGGL((dst_generate_kernel<N, P, ENGINE>), dim3(grid), dim3(T), lds, stream, G);
}

}  // namespace

extern "C" long long cp_dst_forward_analytic_workspace_bytes(long long ncosmo) { return ncosmo < 0 ? -1 : cp_power_workspace_bytes(ncosmo) + 64; }

static int dst_forward_analytic(const cp_dst_plan* p, int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm,
                                const cp_param* pk_params, double* d_out, void* d_work, int flags, int* d_box, int margin_first, int margin_second, int offset_first, int offset_second,
                                void* stream);

extern "C" int cp_dst_forward_analytic(const cp_dst_plan* p, int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m,
                                       const cp_ncdm* ncdm, const cp_param* pk_params, double* d_out, void* d_work, int flags, void* stream) {
    return dst_forward_analytic(p, engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, d_out, d_work, flags, nullptr, 0, 0, 0, 0, stream);
}

extern "C" int cp_dst_forward_analytic_box(const cp_dst_plan* p, int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m,
                                           const cp_ncdm* ncdm, const cp_param* pk_params, double* d_out, void* d_work, int* d_box, int margin_first, int margin_second,
                                           int offset_first, int offset_second, void* stream) {
    if (!d_box) return cp::fail(CP_EINVAL, "cp_dst_forward_analytic_box: null box pointer");
    if (margin_first < 0 || 2 * margin_first >= 2048 || margin_second < 0 || margin_second >= 2048)
        return cp::fail(CP_EINVAL, "cp_dst_forward_analytic_box: bad margins");
    return dst_forward_analytic(p, engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, d_out, d_work, CP_DST_SPLIT, d_box, margin_first, margin_second,
                                offset_first, offset_second, stream);
}

static int dst_forward_analytic(const cp_dst_plan* p, int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m, const cp_ncdm* ncdm,
                                const cp_param* pk_params, double* d_out, void* d_work, int flags, int* d_box, int margin_first, int margin_second, int offset_first, int offset_second,
                                void* stream) {
    if (!p) return cp::fail(CP_EINVAL, "cp_dst_forward_analytic: null plan");
    if (flags & ~CP_DST_SPLIT) return cp::fail(CP_EINVAL, "cp_dst_forward_analytic: unknown flags %d", flags);
    if (p->n != 4096 || !p->d_kx) return cp::fail(CP_EUNSUPPORTED, "cp_dst_forward_analytic: needs a plan of length 4096 made with its abscissa (wallish2018's linear grid)");
    if (engine != CP_ENGINE_EH && engine != CP_ENGINE_EH_NOWIGGLE && engine != CP_ENGINE_BBKS) return cp::fail(CP_EINVAL, "cp_dst_forward_analytic: unknown engine %d", engine);
    if (ncosmo < 0) return cp::fail(CP_EINVAL, "cp_dst_forward_analytic: negative batch");
    if (ncosmo == 0) return CP_OK;
    if (!bg_params || !pk_params || !d_out || !d_work) return cp::fail(CP_EINVAL, "cp_dst_forward_analytic: null pointer");
    char* coef = static_cast<char*>(d_work);
    coef += (64 - (reinterpret_cast<unsigned long long>(coef) & 63u)) & 63u;
    int st = cp_power_coefficients(engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, coef, p->device, stream);      // (validates the massive-neutrino tables)
    if (st != CP_OK) return st;
    const int nsp = ncdm ? ncdm->nspecies : 0;
    if (nsp < 0 || (nsp > 0 && !ncdm->tab)) return cp::fail(CP_EINVAL, "cp_dst_forward_analytic: bad massive-neutrino tables");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_dst_forward_analytic: cannot select device %d", p->device);
    GenArgs G{};
    G.dst.in = nullptr; G.dst.out = d_out; G.dst.nrows = ncosmo; G.dst.tw = p->d_tw; G.dst.rot = p->d_rot; G.dst.kx = nullptr; G.dst.ikx = nullptr; G.dst.fused = 0;
    G.dst.split = (flags & CP_DST_SPLIT) != 0;
    G.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) G.bg[i] = cpcosmo::Param{bg_params[i].ptr, bg_params[i].value};
    for (int i = 0; i < CP_PK_NPARAMS; ++i) G.pw[i] = cpcosmo::Param{pk_params[i].ptr, pk_params[i].value};
    G.second_is_omega_m = second_is_omega_m;
    G.ncdm_tab = nsp ? ncdm->tab : nullptr;
    G.nsp = nsp;
    G.k = p->d_kx;
    G.ln_k = p->d_ln_kx;
    G.consts = reinterpret_cast<const cppower::CosmoConsts*>(coef);
    G.box = d_box; G.margin_first = margin_first; G.margin_second = margin_second; G.off0 = offset_first; G.off1 = offset_second;
    const long long npairs = (ncosmo + 1) / 2;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || ncu <= 0) ncu = 256;
    // two workgroups per CU are resident (LDS): every workgroup the same number of pairs
    const long long resident = 2LL * ncu, rounds = (npairs + resident - 1) / resident;
    const int grid = (int)((npairs + rounds - 1) / rounds);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (engine == CP_ENGINE_EH) launch_generate<CP_ENGINE_EH>(G, grid, s);
    else if (engine == CP_ENGINE_EH_NOWIGGLE) launch_generate<CP_ENGINE_EH_NOWIGGLE>(G, grid, s);
    else launch_generate<CP_ENGINE_BBKS>(G, grid, s);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_dst_forward_analytic: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}


// ---- cp_wallish_tail: everything of wallish2018 behind its forward transform, one kernel (wallish_tail_kernel above) ----
extern "C" int cp_wallish_tail(const cp_dst_plan* p, const cp_splice_plan* splice, double* d_coef, const double* d_pk, int npk, long long nrows, int margin_first,
                               int margin_second, int offset_first, int offset_second, const double* d_tophat, int* d_box, double* d_out, void* stream) {
    if (!p || !splice) return cp::fail(CP_EINVAL, "cp_wallish_tail: null plan");
    if (nrows < 0) return cp::fail(CP_EINVAL, "cp_wallish_tail: negative batch");
    if (margin_first < 0 || 2 * margin_first >= 2048 || margin_second < 0 || margin_second >= 2048) return cp::fail(CP_EINVAL, "cp_wallish_tail: bad margins");
    if (p->n != 4096 || !p->d_kx || !p->d_ln_kx)
        return cp::fail(CP_EUNSUPPORTED, "cp_wallish_tail: needs a transform plan of length 4096 made with its abscissa (wallish2018's linear grid)");
    cpsu::Tables U;
    int device = -1;
    if (!cp_splice_plan_uniform_view(splice, &U, &device) || device != p->device)
        return cp::fail(CP_EUNSUPPORTED, "cp_wallish_tail: the splice plan does not run the uniform-stretch scheme on the transform's device");
    // the stretch comes from the transformed rows, everything else from the rows of P; the stretch, the slot in front of it and the 64 S knots the
    // lanes own lie inside a row of the transform; one query per column of P
    if (U.src_u != 1 || (U.wl > 0 && U.src_l != 0) || (U.wr > 0 && U.src_r != 0) || U.col_u < 1 || U.col_u + 64 * U.S + 1 > 4096 || U.nq != npk || U.ngb > cpsu::NGB ||
        U.S != 49 || U.col_l + U.wl > npk || U.col_r + U.wr > npk)
        return cp::fail(CP_EUNSUPPORTED, "cp_wallish_tail: the splice plan is not the filter's (stretch of %d knots at column %d of array %d, %d queries)", U.nm, U.col_u,
                        U.src_u, U.nq);
    if (nrows == 0) return CP_OK;
    if (!d_coef || !d_pk || !d_box || !d_out) return cp::fail(CP_EINVAL, "cp_wallish_tail: null device pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_wallish_tail: cannot select device %d", p->device);
    TailArgs G{};
    G.nrows = nrows; G.coef = d_coef; G.tw = p->d_tw; G.rot = p->d_rot; G.ikx = p->d_ln_kx + 3 * (size_t)p->n;
    G.U = U;
    G.pk = d_pk; G.tophat = d_tophat; G.out = d_out; G.box = d_box;
    G.margin_first = margin_first; G.margin_second = margin_second; G.off0 = offset_first; G.off1 = offset_second;
    const long long npairs = (nrows + 1) / 2;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || ncu <= 0) ncu = 256;
    const long long resident = 2LL * ncu, rounds = (npairs + resident - 1) / resident;      // two workgroups per CU are resident (LDS): every workgroup the same number of pairs
    const int grid = (int)((npairs + rounds - 1) / rounds);
    constexpr int lds = (4096 + Plan<4096, 16>::TW_TOTAL - 4096) * (int)sizeof(cplx) + 2 * cpdd::DD_NTAB * (int)sizeof(double);
    (void)cp::allow_full_lds<&wallish_tail_kernel<49>>();
    hipLaunchKernelGGL(wallish_tail_kernel<49>, dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), G);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_wallish_tail: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}


// ---- cp_wallish_full: wallish2018 of a batch of analytic cosmologies, one kernel (wallish_full_kernel, cp_wallish_tail.h) ----
namespace {
template <int ENGINE>
void launch_full(const FullArgs& F, int grid, int lds, hipStream_t stream) {
    (void)cp::allow_full_lds<&wallish_full_kernel<49, ENGINE>>();
    hipLaunchKernelGGL((wallish_full_kernel<49, ENGINE>), dim3(grid), dim3(256), lds, stream, F);
}
}  // namespace

extern "C" int cp_wallish_full(const cp_dst_plan* p, const cp_splice_plan* splice, int engine, long long ncosmo, const cp_param* bg_params, int second_is_omega_m,
                               const cp_ncdm* ncdm, const cp_param* pk_params, const double* d_pk, int npk, int margin_first, int margin_second, int offset_first,
                               int offset_second, const double* d_tophat, int* d_box, double* d_coef, double* d_out, void* d_work, void* stream) {
    if (!p || !splice) return cp::fail(CP_EINVAL, "cp_wallish_full: null plan");
    if (ncosmo < 0) return cp::fail(CP_EINVAL, "cp_wallish_full: negative batch");
    if (engine != CP_ENGINE_EH && engine != CP_ENGINE_EH_NOWIGGLE && engine != CP_ENGINE_BBKS) return cp::fail(CP_EINVAL, "cp_wallish_full: unknown engine %d", engine);
    if (margin_first < 0 || 2 * margin_first >= 2048 || margin_second < 0 || margin_second >= 2048) return cp::fail(CP_EINVAL, "cp_wallish_full: bad margins");
    if (p->n != 4096 || !p->d_kx || !p->d_ln_kx)
        return cp::fail(CP_EUNSUPPORTED, "cp_wallish_full: needs a transform plan of length 4096 made with its abscissa (wallish2018's linear grid)");
    cpsu::Tables U;
    int device = -1;
    if (!cp_splice_plan_uniform_view(splice, &U, &device) || device != p->device)
        return cp::fail(CP_EUNSUPPORTED, "cp_wallish_full: the splice plan does not run the uniform-stretch scheme on the transform's device");
    if (U.src_u != 1 || (U.wl > 0 && U.src_l != 0) || (U.wr > 0 && U.src_r != 0) || U.col_u < 1 || U.col_u + 64 * U.S + 1 > 4096 || U.nq != npk || U.ngb > cpsu::NGB ||
        U.S != 49 || U.col_l + U.wl > npk || U.col_r + U.wr > npk)
        return cp::fail(CP_EUNSUPPORTED, "cp_wallish_full: the splice plan is not the filter's (stretch of %d knots at column %d of array %d, %d queries)", U.nm, U.col_u,
                        U.src_u, U.nq);
    if (ncosmo == 0) return CP_OK;
    if (!bg_params || !pk_params || !d_pk || !d_box || !d_out || !d_work) return cp::fail(CP_EINVAL, "cp_wallish_full: null pointer");
    char* coef = static_cast<char*>(d_work);
    coef += (64 - (reinterpret_cast<unsigned long long>(coef) & 63u)) & 63u;
    int st = cp_power_coefficients(engine, ncosmo, bg_params, second_is_omega_m, ncdm, pk_params, coef, p->device, stream);      // (validates the massive-neutrino tables)
    if (st != CP_OK) return st;
    const int nsp = ncdm ? ncdm->nspecies : 0;
    if (nsp < 0 || (nsp > 0 && !ncdm->tab)) return cp::fail(CP_EINVAL, "cp_wallish_full: bad massive-neutrino tables");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_wallish_full: cannot select device %d", p->device);
    FullArgs F{};
    TailArgs& G = F.tail;
    G.nrows = ncosmo; G.coef = d_coef; G.tw = p->d_tw; G.rot = p->d_rot; G.ikx = p->d_ln_kx + 3 * (size_t)p->n;
    G.U = U;
    G.pk = d_pk; G.tophat = d_tophat; G.out = d_out; G.box = d_box;
    G.margin_first = margin_first; G.margin_second = margin_second; G.off0 = offset_first; G.off1 = offset_second;
    GenArgs& E = F.gen;
    E.dst.in = nullptr; E.dst.out = nullptr; E.dst.nrows = ncosmo; E.dst.tw = p->d_tw; E.dst.rot = p->d_rot; E.dst.kx = nullptr; E.dst.ikx = nullptr; E.dst.fused = 0; E.dst.split = 1;
    E.ncosmo = ncosmo;
    for (int i = 0; i < CP_BG_NPARAMS; ++i) E.bg[i] = cpcosmo::Param{bg_params[i].ptr, bg_params[i].value};
    for (int i = 0; i < CP_PK_NPARAMS; ++i) E.pw[i] = cpcosmo::Param{pk_params[i].ptr, pk_params[i].value};
    E.second_is_omega_m = second_is_omega_m;
    E.ncdm_tab = nsp ? ncdm->tab : nullptr;
    E.nsp = nsp;
    E.k = p->d_kx;
    E.ln_k = p->d_ln_kx;
    E.consts = reinterpret_cast<const cppower::CosmoConsts*>(coef);
    E.box = nullptr;
    const long long npairs = (ncosmo + 1) / 2;
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || ncu <= 0) ncu = 256;
    const long long resident = 2LL * ncu, rounds = (npairs + resident - 1) / resident;
    const int grid = (int)((npairs + rounds - 1) / rounds);
    constexpr int lds = (4096 + Plan<4096, 16>::TW_TOTAL - 4096) * (int)sizeof(cplx) + 2 * cpdd::DD_NTAB * (int)sizeof(double);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (engine == CP_ENGINE_EH) launch_full<CP_ENGINE_EH>(F, grid, lds, s);
    else if (engine == CP_ENGINE_EH_NOWIGGLE) launch_full<CP_ENGINE_EH_NOWIGGLE>(F, grid, lds, s);
    else launch_full<CP_ENGINE_BBKS>(F, grid, lds, s);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_wallish_full: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
