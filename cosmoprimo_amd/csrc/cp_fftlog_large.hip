// cp_fftlog_large.hip -- FFTLog rows whose padded length does not fit the LDS-resident fused kernel (8192 < Np <= 2^24).
//
// Same arithmetic as the reference (cosmoprimo/fftlog.py:228-241) in the form the fused kernel uses (cp_fftlog_body.h): two rows packed as
// z = (a + i b) x prefactor, g = FFT(FFT(z) x U) / Np with U the Hermitian extension of u, Re -> row a, Im -> row b.  A transform of Np =
// N1 x 4096 points is split the four-step way, with the order of the intermediate spectrum left permuted (U is stored in that order, the
// second transform runs the transposed network back), so that no transposition pass exists:
//
//   screen   per row: the largest |sample x prefactor| (its exponent) -- rows are scaled to [1, 2) and back by exact powers of two, a row that
//            is not finite is transformed as zeros and stored as NaN: the two rows of a pair stay independent, as numpy's row-by-row FFTs are
//   columns  A[k1][n2] = sum_n1 z[n1 4096 + n2] w_N1^(n1 k1): pad + prefactor + pack on the way in, N1-point FFTs down the columns of a
//            (N1 x C) tile in LDS, written to the scratch at row k1
//   rows     for each k1: x w_Np^(k1 n2), 4096-point FFT (radix 16 x 3 in registers, exchanges through swizzled LDS), x U[k1 + N1 k2],
//            4096-point FFT back (transposed network), x w_Np^(k1 m2): one kernel, one read and one write of the scratch row
//   columns  g[m2 + 4096 m1] = sum_k1 B[k1][m2] w_N1^(k1 m1), x postfactor, crop, unpack on the way out
//
// The scratch of a chunk of pairs (16 Np bytes per pair, <= 128 MB per chunk) is written and read by consecutive kernels: it stays in the
// Infinity Cache.  Every size the reference's own callers use (nk = 1024 in to_xi / sigma, 4096 in the BAO filters: Np <= 8192) runs the fused
// kernel of cp_fftlog.hip; this path completes FFTlog's size range.
#include <hip/hip_runtime.h>

#include <mutex>
#include <new>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_fft_core.h"
#include "cp_math.h"
#include "cp_fftlog_large.h"
#include "cp_fftlog_tables.h"

namespace {

using cpfft::cplx;
using cpfft::cmul;
using cpfft::Dft;

constexpr int ROW = 4096;        // N2: the length of the in-LDS transforms of the row kernel
constexpr int TILE = 4096;       // complex points of a column tile (N1 x C)
constexpr int MAXPASS = 3;       // N1 <= 4096 = 16^3

struct PadSpec {
    int n, npad, nker, in_left;
    int ext_l, ext_r;
    double val_l, val_r;
};

// ratio^e of the log-log continuation (see Fftlog::ratio_pow, cp_fftlog_body.h): exp(e log ratio) for a positive, normal ratio, the library's pow otherwise
__device__ __attribute__((noinline)) double ratio_pow(double ratio, int e) {
    if (ratio > 2.2250738585072014e-308 && ratio < 1.7976931348623157e308) {
        const double x = (double)e * cpmath::log_pos(ratio);
        if (fabs(x) < 700.) return cpmath::exp_mid(x);
    }
    return pow(ratio, (double)e);
}

// pad(array, (L, R), extrap)[j] (fftlog.py:483-505)
__device__ __forceinline__ double padded_sample(const double* a, int j, const PadSpec& S) {
    const int idx = j - S.in_left;
    const int cl = idx < 0 ? 0 : (idx >= S.n ? S.n - 1 : idx);
    double v = a[cl];
    if (idx < 0) {
        if (S.ext_l == CP_EXTRAP_CONSTANT) v = S.val_l;
        if (S.ext_l == CP_EXTRAP_LOGLOG) v = v * ratio_pow(a[1] / v, idx);
    } else if (idx >= S.n) {
        if (S.ext_r == CP_EXTRAP_CONSTANT) v = S.val_r;
        if (S.ext_r == CP_EXTRAP_LOGLOG) v = v / ratio_pow(a[S.n - 2] / v, idx - S.n + 1);
    }
    return v;
}

// pair P of the batch -> its two rows (the second one may not exist): rows of one kernel index, consecutive batch items (as the fused kernel)
struct PairRows {
    long long ra, rb;
    int ker;
    bool has_b;
};
__device__ __forceinline__ PairRows pair_rows(long long pair, int nker, long long nbatch) {
    PairRows r;
    r.ker = (int)(pair % nker);
    const long long b0 = 2 * (pair / nker);
    r.ra = b0 * nker + r.ker;
    r.has_b = b0 + 1 < nbatch;
    r.rb = r.has_b ? (b0 + 1) * nker + r.ker : r.ra;
    return r;
}

struct ScreenArgs {
    const double* in;
    const double* pre;
    unsigned* mag;      // (2 x pairs of the chunk): high dword of the largest |sample x prefactor| of each row
    long long pair0, nbatch;
    PadSpec S;
    int segs;           // workgroups per row
};

__global__ __launch_bounds__(256) void screen_kernel(const ScreenArgs A) {
    __shared__ unsigned top[4];
    const long long slot = blockIdx.x / A.segs;
    const int seg = blockIdx.x % A.segs;
    const PairRows pr = pair_rows(A.pair0 + slot / 2, A.S.nker, A.nbatch);
    unsigned m = 0u;
    if (!(slot & 1) || pr.has_b) {
        const double* a = A.in + ((slot & 1) ? pr.rb : pr.ra) * A.S.n;
        const double* pre = A.pre + (long long)pr.ker * A.S.npad;
#pragma unroll
        for (int i = 0; i < 16; ++i) {       // (the padded length is a multiple of the 4096 samples of a workgroup)
            const int j = seg * 4096 + i * 256 + threadIdx.x;
            const unsigned h = cpfft::hi_abs(padded_sample(a, j, A.S) * pre[j]);
            m = h > m ? h : m;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)m, d);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) top[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) m = top[i] > m ? top[i] : m;
        if (m) atomicMax(A.mag + slot, m);
    }
}

// what a row's magnitude says: the exact power of two that brings its largest sample to [1, 2), its inverse, and whether the row is finite
struct RowScale {
    double down, up;
    bool finite;
};
__device__ __forceinline__ RowScale row_scale(unsigned mag) {
    RowScale r;
    r.finite = mag < 0x7ff00000u;
    int sh = 1023 - (int)(mag >> 20);
    sh = sh > 1022 ? 1022 : (sh < -1022 ? -1022 : sh);
    if (!r.finite || mag == 0u) sh = 0;
    r.down = __builtin_bit_cast(double, (unsigned long long)(1023 + sh) << 52);
    r.up = __builtin_bit_cast(double, (unsigned long long)(1023 - sh) << 52);
    return r;
}

// One decimation-in-frequency pass of radix R over the N-point columns of an (N x C) tile in LDS (column index fastest): the butterflies of
// span L, twiddles w_L^(j q) from the table of w_4096^i.  In place; after the passes of radices r_0, r_1, ... position p holds the frequency
// sum_i d_i prod_{l<i} r_l, d_i the mixed-radix digits of p (freq_of_pos).
template <int R>
__device__ __forceinline__ void column_pass(cplx* lds, int L, int N, int log2c, const cplx* tw, int tid) {
    const int M = L / R, C = 1 << log2c;
    const int nbf = (N / R) << log2c;
    const int tws = ROW / L;
    for (int w = tid; w < nbf; w += 256) {
        const int c = w & (C - 1), bf = w >> log2c;
        const int j = bf % M, b = bf / M;
        cplx* at = lds + (((long long)b * L + j) << log2c) + c;
        const int stride = M << log2c;
        cplx x[R];
#pragma unroll
        for (int s = 0; s < R; ++s) x[s] = at[s * stride];
        Dft<R>::run(x);
        if (M > 1) {
#pragma unroll
            for (int q = 1; q < R; ++q) x[q] = cmul(x[q], tw[(j * q) * tws]);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) at[q * stride] = x[q];
    }
}

struct Radices {
    int npass;
    int r[MAXPASS];
};

__device__ __forceinline__ int freq_of_pos(int p, int N, const Radices& rd) {
    int k = 0, mult = 1, rem = p, L = N;
    for (int i = 0; i < rd.npass; ++i) {
        const int M = L / rd.r[i];
        k += (rem / M) * mult;
        rem %= M;
        mult *= rd.r[i];
        L = M;
    }
    return k;
}

struct ColumnArgs {
    const double* in;     // first column kernel: the rows of the batch
    double* out;          // second column kernel: the results
    const double* pre;
    const double* post;
    cplx* work;           // (pairs of the chunk, N1, 4096)
    const unsigned* mag;
    const cplx* tw;       // w_4096^i
    long long pair0, nbatch;
    PadSpec S;
    int n1, log2c;
    Radices rd;
    int out_off, n_out;
};

// FIRST: rows -> scratch (row k1 of the scratch holds A[k1][.]); otherwise scratch -> rows
template <bool FIRST>
__global__ __launch_bounds__(256) void column_kernel(const ColumnArgs A) {
    __shared__ cplx lds[TILE];
    const int C = 1 << A.log2c, ntiles = ROW >> A.log2c;
    const long long slot = blockIdx.x / ntiles;
    const int c0 = (int)(blockIdx.x % ntiles) << A.log2c;
    const PairRows pr = pair_rows(A.pair0 + slot, A.S.nker, A.nbatch);
    const RowScale sa = row_scale(A.mag[2 * slot]), sb = row_scale(A.mag[2 * slot + 1]);
    cplx* work = A.work + slot * ((long long)A.n1 * ROW);
    const int tid = threadIdx.x;
    const int npts = A.n1 << A.log2c;
    if (FIRST) {
        const double* a = A.in + pr.ra * A.S.n;
        const double* b = A.in + pr.rb * A.S.n;
        const double* pre = A.pre + (long long)pr.ker * A.S.npad;
        for (int w = tid; w < npts; w += 256) {
            const int c = w & (C - 1), r = w >> A.log2c;
            const int j = r * ROW + c0 + c;
            const double f = pre[j];
            cplx z;
            z.re = sa.finite ? padded_sample(a, j, A.S) * f * sa.down : 0.;
            z.im = (pr.has_b && sb.finite) ? padded_sample(b, j, A.S) * f * sb.down : 0.;
            lds[w] = z;
        }
    } else {
        for (int w = tid; w < npts; w += 256) {
            const int c = w & (C - 1), r = w >> A.log2c;
            lds[w] = work[(long long)r * ROW + c0 + c];
        }
    }
    int L = A.n1;
    for (int i = 0; i < A.rd.npass; ++i) {
        __syncthreads();
        switch (A.rd.r[i]) {
            case 16: column_pass<16>(lds, L, A.n1, A.log2c, A.tw, tid); break;
            case 8: column_pass<8>(lds, L, A.n1, A.log2c, A.tw, tid); break;
            case 4: column_pass<4>(lds, L, A.n1, A.log2c, A.tw, tid); break;
            default: column_pass<2>(lds, L, A.n1, A.log2c, A.tw, tid); break;
        }
        L /= A.rd.r[i];
    }
    __syncthreads();
    if (FIRST) {
        for (int w = tid; w < npts; w += 256) {
            const int c = w & (C - 1), p = w >> A.log2c;
            work[(long long)freq_of_pos(p, A.n1, A.rd) * ROW + c0 + c] = lds[w];
        }
    } else {
        double* oa = A.out + pr.ra * A.n_out;
        double* ob = A.out + pr.rb * A.n_out;
        const double* post = A.post + (long long)pr.ker * A.S.npad;
        const double nan = __builtin_nan("");
        for (int w = tid; w < npts; w += 256) {
            const int c = w & (C - 1), p = w >> A.log2c;
            const int m = c0 + c + ROW * freq_of_pos(p, A.n1, A.rd);
            const int o = m - A.out_off;
            if (o < 0 || o >= A.n_out) continue;
            const cplx g = lds[w];
            const double f = post[m];
            oa[o] = sa.finite ? g.re * f * sa.up : nan;
            if (pr.has_b) ob[o] = sb.finite ? g.im * f * sb.up : nan;
        }
    }
}

// N1 <= 16: a column is one butterfly -- a thread takes a column, no LDS
template <bool FIRST, int N1>
__global__ __launch_bounds__(256) void column_direct_kernel(const ColumnArgs A) {
    constexpr int PER = ROW / 256;
    const long long slot = blockIdx.x / PER;
    const int c = (int)(blockIdx.x % PER) * 256 + threadIdx.x;
    const PairRows pr = pair_rows(A.pair0 + slot, A.S.nker, A.nbatch);
    const RowScale sa = row_scale(A.mag[2 * slot]), sb = row_scale(A.mag[2 * slot + 1]);
    cplx* work = A.work + slot * ((long long)N1 * ROW) + c;
    cplx x[N1];
    if (FIRST) {
        const double* a = A.in + pr.ra * A.S.n;
        const double* b = A.in + pr.rb * A.S.n;
        const double* pre = A.pre + (long long)pr.ker * A.S.npad;
#pragma unroll
        for (int r = 0; r < N1; ++r) {
            const int j = r * ROW + c;
            const double f = pre[j];
            x[r].re = sa.finite ? padded_sample(a, j, A.S) * f * sa.down : 0.;
            x[r].im = (pr.has_b && sb.finite) ? padded_sample(b, j, A.S) * f * sb.down : 0.;
        }
        Dft<N1>::run(x);
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) work[k1 * ROW] = x[k1];
    } else {
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) x[k1] = work[k1 * ROW];
        Dft<N1>::run(x);
        double* oa = A.out + pr.ra * A.n_out;
        double* ob = A.out + pr.rb * A.n_out;
        const double* post = A.post + (long long)pr.ker * A.S.npad;
        const double nan = __builtin_nan("");
        double fpost[N1];      // (requested together: see fftlog_body.h, store_output)
#pragma unroll
        for (int m1 = 0; m1 < N1; ++m1) fpost[m1] = post[c + ROW * m1];
#pragma unroll
        for (int m1 = 0; m1 < N1; ++m1) {
            const int m = c + ROW * m1;
            const int o = m - A.out_off;
            if (o < 0 || o >= A.n_out) continue;
            const double f = fpost[m1];
            oa[o] = sa.finite ? x[m1].re * f * sa.up : nan;
            if (pr.has_b) ob[o] = sb.finite ? x[m1].im * f * sb.up : nan;
        }
    }
}

template <bool FIRST>
void launch_columns(const ColumnArgs& C, long long np, hipStream_t st) {
    const unsigned direct = (unsigned)(np * (ROW / 256));
    if (C.n1 == 4) hipLaunchKernelGGL((column_direct_kernel<FIRST, 4>), dim3(direct), dim3(256), 0, st, C);
    else if (C.n1 == 8) hipLaunchKernelGGL((column_direct_kernel<FIRST, 8>), dim3(direct), dim3(256), 0, st, C);
    else if (C.n1 == 16) hipLaunchKernelGGL((column_direct_kernel<FIRST, 16>), dim3(direct), dim3(256), 0, st, C);
    else hipLaunchKernelGGL(column_kernel<FIRST>, dim3((unsigned)(np * (ROW >> C.log2c))), dim3(256), 0, st, C);
}

struct RowArgs {
    cplx* work;           // (pairs of the chunk, N1, 4096), transformed in place
    const cplx* u;        // (nker, N1, 16, 256): U[k1 + N1 k2] / Np at [ker][k1][q][t], k2 the frequency position 16 t + q holds after the forward passes
    const cplx* tw0;      // (16, 256): w_4096^(t q) at [q][t]
    const cplx* tw;       // w_4096^i
    const cplx* tw_a;     // w_Np^i, i < 256 N1
    const cplx* tw_b;     // w_(16 N1)^i, i < 16 N1
    long long pair0, nrows;
    int n1, nker;
};

__device__ __forceinline__ int swz(int p) { return p ^ ((p >> 4) & 15); }

// Rows k1 of the pairs of a chunk, one per iteration of a persistent workgroup: x w_Np^(k1 n2), FFT_4096, x U, FFT_4096 (transposed network),
// x w_Np^(k1 m2).  A thread owns the 16 points t + 256 s on the way in and out (radix-16 butterflies in registers, three passes per transform,
// exchanges through LDS under the XOR swizzle of the fused kernel); the twiddles of the middle pass come from a 4 KB table in LDS, those of
// the outer pass and U from tables laid out so that a wave reads 1 KB segments.  The slots a thread reads last are the slots it writes first
// in the next row: no barrier between rows.  (Fetching the next row into registers during the transform costs 46 spilled registers at two
// waves per SIMD and gains 3 %: not kept.)
__global__ __launch_bounds__(256, 2) void row_kernel(const RowArgs A) {
    __shared__ cplx lds[ROW + 256];
    cplx* w1 = lds + ROW;      // w_256^i
    const int t = threadIdx.x, b = t >> 4, j = t & 15;
    w1[t] = A.tw[16 * t];
    const cplx* w0 = A.tw0 + t;     // w_4096^(t q) at [q][t]: L2 hits in 1 KB segments (no register is left to keep them)
    const int mask_b = 16 * A.n1 - 1;
    long long row = blockIdx.x;
    cplx x[16];
    __syncthreads();
    for (; row < A.nrows; row += gridDim.x) {
        const long long slot = row / A.n1;
        const int k1 = (int)(row - slot * A.n1);
        const int ker = (int)((A.pair0 + slot) % A.nker);
        cplx* y = A.work + row * ROW;
        const cplx* u = A.u + ((long long)ker * A.n1 + k1) * ROW + t;
        const cplx fa = A.tw_a[k1 * t];       // w_Np^(k1 (t + 256 s)) = w_Np^(k1 t) w_(16 N1)^(k1 s)
#pragma unroll
        for (int s = 0; s < 16; ++s) x[s] = cmul(y[t + 256 * s], cmul(fa, A.tw_b[(k1 * s) & mask_b]));
        // forward passes of spans 4096, 256, 16
        Dft<16>::run(x);
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[swz(t + 256 * q)] = q ? cmul(x[q], w0[256 * q]) : x[q];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s) x[s] = lds[swz(b * 256 + j + 16 * s)];
        Dft<16>::run(x);
#pragma unroll
        for (int q = 0; q < 16; ++q) lds[swz(b * 256 + j + 16 * q)] = q ? cmul(x[q], w1[j * q]) : x[q];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 16; ++s) x[s] = lds[swz(16 * t + s)];
        Dft<16>::run(x);
        // position 16 t + q holds k2 = (p >> 8) + 16 ((p >> 4) & 15) + 256 (p & 15); the passes of the second transform in the opposite order
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = cmul(x[q], u[256 * q]);
        Dft<16>::run(x);
#pragma unroll
        for (int s = 0; s < 16; ++s) lds[swz(16 * t + s)] = x[s];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            x[q] = lds[swz(b * 256 + j + 16 * q)];
            if (q) x[q] = cmul(x[q], w1[j * q]);
        }
        Dft<16>::run(x);
#pragma unroll
        for (int s = 0; s < 16; ++s) lds[swz(b * 256 + j + 16 * s)] = x[s];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            x[q] = lds[swz(t + 256 * q)];
            if (q) x[q] = cmul(x[q], w0[256 * q]);
        }
        Dft<16>::run(x);
        {      // (the sixteen twiddles requested together: between the stores each one waited for the store before it -- the memory counter retires in order)
            cplx tb[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) tb[s] = A.tw_b[(k1 * s) & mask_b];
#pragma unroll
            for (int s = 0; s < 16; ++s) y[t + 256 * s] = cmul(x[s], cmul(fa, tb[s]));
        }
    }
}

}  // namespace

struct cp_fftlog_large {
    int n, npad, nker, device;
    int n1, log2c;
    Radices rd;
    long long chunk;       // pairs per pass through the scratch
    double* d_pre;
    double* d_post;
    cplx* d_u;             // (nker, n1, 4096)
    cplx* d_tw;            // w_4096^i | w_Np^i (i < 256 n1) | w_(16 n1)^i (i < 16 n1) | w_4096^(t q) at [q][t]
    int row_grid;          // workgroups of the persistent row kernel: two per CU (68 KB of LDS each)
    cplx* d_work;          // (chunk, npad)
    unsigned* d_mag;       // (2 chunk)
    std::mutex lock;       // the scratch is shared: one execute is ENQUEUED at a time ...
    hipEvent_t done;       // ... and an execute waits, on its own stream, for the event the previous execute recorded behind its last kernel,
    bool has_done;         // so that executes of one plan on different streams cannot overlap on the device
};

int cp_fftlog_large_create(cp_fftlog_large** out, int n, int npad, int nker, const double* pre, const double* post, const double* u_re_im, int device) {
    *out = nullptr;
    if (npad > (1 << 24) || npad < 4 * ROW) return cp::fail(CP_EUNSUPPORTED, "padded size %d is outside the range of the large-size path (2^14 .. 2^24)", npad);
    cp_fftlog_large* p = new (std::nothrow) cp_fftlog_large();
    if (!p) return cp::fail(CP_ENOMEM, "cp_fftlog_plan_create: host allocation failed");
    p->n = n; p->npad = npad; p->nker = nker; p->device = device;
    p->d_pre = p->d_post = nullptr;
    p->d_u = p->d_tw = p->d_work = nullptr;
    p->d_mag = nullptr;
    p->done = nullptr;
    p->has_done = false;
    const int n1 = npad / ROW;
    p->n1 = n1;
    int log2c = 0;
    while ((n1 << (log2c + 1)) <= TILE) ++log2c;
    p->log2c = log2c;
    p->rd.npass = 0;
    for (int rem = n1; rem > 1;) {
        const int r = rem > 16 ? 16 : rem;
        p->rd.r[p->rd.npass++] = r;
        rem /= r;
    }
    long long chunk = (128LL << 20) / ((long long)npad * (long long)sizeof(cplx));   // the scratch of a chunk stays in the Infinity Cache
    if (chunk < 1) chunk = 1;
    p->chunk = chunk;
    std::vector<cplx> tw, u;
    try {
        tw.resize((size_t)2 * ROW + 256 * (size_t)n1 + 16 * (size_t)n1);
        for (int q = 0; q < 16; ++q)
            for (int t = 0; t < 256; ++t) tw[tw.size() - ROW + q * 256 + t] = cpfft::unit_root((long long)t * q, ROW);
        for (int i = 0; i < ROW; ++i) tw[i] = cpfft::unit_root(i, ROW);
        for (long long i = 0; i < 256LL * n1; ++i) tw[ROW + i] = cpfft::unit_root(i, npad);
        for (long long i = 0; i < 16LL * n1; ++i) tw[ROW + 256 * (size_t)n1 + i] = cpfft::unit_root(i, 16LL * n1);
        u.resize((size_t)nker * npad);
    } catch (const std::bad_alloc&) {
        delete p;
        return cp::fail(CP_ENOMEM, "cp_fftlog_plan_create: host allocation failed");
    }
    const int nh = npad / 2 + 1;
    const double inv = 1. / (double)npad;
    for (int ker = 0; ker < nker; ++ker) {
        const double* uk = u_re_im + (size_t)ker * 2 * nh;
        for (int k1 = 0; k1 < n1; ++k1)
            for (int pos = 0; pos < ROW; ++pos) {
                const long long k2 = (pos >> 8) + 16 * ((pos >> 4) & 15) + 256 * (pos & 15);
                const long long k = k1 + (long long)n1 * k2;
                cplx v;   // Hermitian extension with real DC and Nyquist bins: what numpy's irfft assumes (fftlog.py:544)
                if (k == 0) v = cplx{uk[0], 0.};
                else if (k == npad / 2) v = cplx{uk[2 * (size_t)(npad / 2)], 0.};
                else if (k < npad / 2) v = cplx{uk[2 * k], uk[2 * k + 1]};
                else v = cplx{uk[2 * (npad - k)], -uk[2 * (npad - k) + 1]};
                v.re *= inv;
                v.im *= inv;
                u[((size_t)ker * n1 + k1) * ROW + (pos & 15) * 256 + (pos >> 4)] = v;      // position 16 t + q at [q][t]: a wave reads 1 KB segments
            }
    }
    const size_t tb = (size_t)nker * npad * sizeof(double);
    bool ok = hipMalloc(&p->d_pre, tb) == hipSuccess && hipMalloc(&p->d_post, tb) == hipSuccess &&
              hipMalloc(&p->d_u, u.size() * sizeof(cplx)) == hipSuccess && hipMalloc(&p->d_tw, tw.size() * sizeof(cplx)) == hipSuccess &&
              hipMalloc(&p->d_work, (size_t)chunk * npad * sizeof(cplx)) == hipSuccess &&
              hipMalloc(&p->d_mag, (size_t)chunk * 2 * sizeof(unsigned)) == hipSuccess;
    ok = ok && hipMemcpy(p->d_pre, pre, tb, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(p->d_post, post, tb, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_u, u.data(), u.size() * sizeof(cplx), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_tw, tw.data(), tw.size() * sizeof(cplx), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        cp_fftlog_large_destroy(p);
        return cp::fail(CP_ENOMEM, "cp_fftlog_plan_create: cannot allocate the tables / scratch of the large-size path on device %d", device);
    }
    if (hipEventCreateWithFlags(&p->done, hipEventDisableTiming) != hipSuccess) {
        p->done = nullptr;
        cp_fftlog_large_destroy(p);
        return cp::fail(CP_EDEVICE, "cp_fftlog_plan_create: cannot create the stream-ordering event of the large-size path");
    }
    hipDeviceProp_t prop;
    p->row_grid = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0 ? 2 * prop.multiProcessorCount : 512;
    *out = p;
    return CP_OK;
}

void cp_fftlog_large_destroy(cp_fftlog_large* p) {
    if (!p) return;
    if (p->d_pre) (void)hipFree(p->d_pre);
    if (p->d_post) (void)hipFree(p->d_post);
    if (p->d_u) (void)hipFree(p->d_u);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_work) (void)hipFree(p->d_work);
    if (p->d_mag) (void)hipFree(p->d_mag);
    if (p->done) (void)hipEventDestroy(p->done);
    delete p;
}

int cp_fftlog_large_execute(cp_fftlog_large* p, const double* d_in, double* d_out, long long nbatch, int ext_l, double val_l, int ext_r, double val_r,
                            int keep_padding, hipStream_t st) {
    const int out_left = (p->npad - p->n) - (p->npad - p->n) / 2;   // fftlog.py:152-153
    PadSpec S;
    S.n = p->n; S.npad = p->npad; S.nker = p->nker; S.in_left = (p->npad - p->n) / 2;
    S.ext_l = ext_l; S.ext_r = ext_r; S.val_l = val_l; S.val_r = val_r;
    const long long npairs = (nbatch + 1) / 2 * p->nker;
    std::lock_guard<std::mutex> guard(p->lock);
    if (p->has_done && hipStreamWaitEvent(st, p->done, 0) != hipSuccess)   // the previous execute (any stream) is done with the scratch
        return cp::fail(CP_EDEVICE, "cp_fftlog_execute: hipStreamWaitEvent failed");
    const int segs = p->npad / 4096;           // 16 independent samples per thread, one atomic per workgroup
    for (long long p0 = 0; p0 < npairs; p0 += p->chunk) {
        const long long np = npairs - p0 < p->chunk ? npairs - p0 : p->chunk;
        if (hipMemsetAsync(p->d_mag, 0, (size_t)np * 2 * sizeof(unsigned), st) != hipSuccess)
            return cp::fail(CP_EDEVICE, "cp_fftlog_execute: hipMemsetAsync failed");
        ScreenArgs SA;
        SA.in = d_in; SA.pre = p->d_pre; SA.mag = p->d_mag; SA.pair0 = p0; SA.nbatch = nbatch; SA.S = S; SA.segs = segs;
        hipLaunchKernelGGL(screen_kernel, dim3((unsigned)(2 * np * segs)), dim3(256), 0, st, SA);
        ColumnArgs C;
        C.in = d_in; C.out = d_out; C.pre = p->d_pre; C.post = p->d_post; C.work = p->d_work; C.mag = p->d_mag; C.tw = p->d_tw;
        C.pair0 = p0; C.nbatch = nbatch; C.S = S; C.n1 = p->n1; C.log2c = p->log2c; C.rd = p->rd;
        C.out_off = keep_padding ? 0 : out_left;
        C.n_out = keep_padding ? p->npad : p->n;
        launch_columns<true>(C, np, st);
        RowArgs R;
        R.work = p->d_work; R.u = p->d_u; R.tw = p->d_tw; R.tw_a = p->d_tw + ROW; R.tw_b = p->d_tw + ROW + 256 * (size_t)p->n1;
        R.tw0 = R.tw_b + 16 * (size_t)p->n1;
        R.pair0 = p0; R.nrows = np * p->n1; R.n1 = p->n1; R.nker = p->nker;
        hipLaunchKernelGGL(row_kernel, dim3((unsigned)(R.nrows < p->row_grid ? R.nrows : p->row_grid)), dim3(256), 0, st, R);
        launch_columns<false>(C, np, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_fftlog_execute: launch failed: %s", hipGetErrorString(e));
    if (hipEventRecord(p->done, st) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_fftlog_execute: hipEventRecord failed");
    p->has_done = true;
    return CP_OK;
}
