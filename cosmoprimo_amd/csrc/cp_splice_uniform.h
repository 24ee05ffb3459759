// cp_splice_uniform.h -- the spliced clamped spline of wallish2018 (cp_splice_*, cp_bao.hip; reference bao_filter.py:415-431) when nearly all of
// its knots lie on a UNIFORM grid and every query that needs the spline falls inside that stretch or in the interval on either side of it.
//
// On a uniform grid the system for the second derivatives has constant coefficients, M_{i-1} + 4 M_i + M_{i+1} = (6 / h^2) r_i with r_i = y_{i+1} -
// 2 y_i + y_{i-1}, and the inverse of a constant tridiagonal matrix is two geometric tails: M_i = (6 kappa / h^2) sum_j p^|i-j| r_j, p = sqrt 3 - 2,
// kappa = 1 / (2 sqrt 3) -- a causal and an anti-causal first-order recursion (as in fftlog_geospline_kernel, cp_sigma.hip), no elimination and no
// table of factors.  What the knots OUTSIDE the stretch (and the clamped ends) change is a solution of the homogeneous recurrence on the stretch,
// A p^i + B p^(n-1-i): two numbers per row, each a weighted sum of the ~80 knot values next to the junction it decays from (beyond, the weights
// fall below 1e-20) -- the weights are rows of the exact inverse computed on the host when the plan is made (splice_uniform_build).  The second
// derivatives at the one knot outside either end of the stretch (for the two long intervals that border it) are two more such sums.
//
// One wave per row, lane l owns S consecutive knots (S odd: conflict-free LDS strides): it reads its S + 2 values from LDS once, runs the
// anti-causal recursion g_t = r_t + p g_{t+1} over them in registers, then the causal one in the same registers (r_t = g_t - p g_{t+1}:
// e_t = g_t + p f_{t-1} is the second derivative, f_t = e_t - p g_{t+1}), both from zero; what the knots outside its segment add comes from the
// neighbours' segment totals by one DPP shift each way (a segment away the weight is p^S < 1e-18).  The elimination kernel it replaces
// (splice_kernel) walks S + halo knots twice through LDS, three LDS accesses and a table entry per knot and sweep.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "cp_internal.h"
#include "cp_math.h"
#include "cp_splice_uniform_plan.h"

namespace cpsu {

constexpr double ipow(double x, int k) {
    double v = 1.;
    for (int i = 0; i < k; ++i) v *= x;
    return v;
}

typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double from_left(double v) {      // lane l receives lane l - 1's value, lane 0 zero (wave_shr:1)
#if defined(__HIP_DEVICE_COMPILE__)
    v2i w = __builtin_bit_cast(v2i, v);
    w.x = __builtin_amdgcn_update_dpp(0, w.x, 0x138, 0xf, 0xf, true);
    w.y = __builtin_amdgcn_update_dpp(0, w.y, 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, w);
#else
    return v;
#endif
}
__device__ __forceinline__ double from_right(double v) {     // lane l receives lane l + 1's value, lane 63 zero (wave_shl:1)
#if defined(__HIP_DEVICE_COMPILE__)
    v2i w = __builtin_bit_cast(v2i, v);
    w.x = __builtin_amdgcn_update_dpp(0, w.x, 0x130, 0xf, 0xf, true);
    w.y = __builtin_amdgcn_update_dpp(0, w.y, 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double, w);
#else
    return v;
#endif
}

template <int S>
constexpr int lds_doubles_per_wave() { return 64 * S + 2 + 128; }

#ifndef CP_SPLICE_UNIFORM_ABLATE      // diagnostic builds (wrong results): 1 no recursions, 2 no evaluation, 4 no window sums
#define CP_SPLICE_UNIFORM_ABLATE 0
#endif

// LDS: per wave yu[-1 .. 64 S] (the stretch; its first value repeated in front, its last value behind: the second differences vanish beyond the
// stretch, and the one-sided differences at its two end knots are part of what the host's weights account for), gl[64], gr[64] (the knots
// outside); per workgroup the query table.  The second derivatives overwrite yu in place once every lane has read its values.
// One wave per SIMD (the buffers of four rows fill most of the LDS): 512 registers, of which the next row's knot values take ~100 while this row
// is solved; the row of the first array (every query's own column) is fetched behind the evaluation that consumed the previous one.
template <int S>
__global__ __launch_bounds__(256) void splice_uniform_kernel(const Args A) {
    extern __shared__ __attribute__((aligned(32))) double su_lds[];
    const Tables& T = A.T;
    constexpr int STRIDE = lds_doubles_per_wave<S>();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* yu = su_lds + wave * STRIDE + 1;
    double* gl = su_lds + wave * STRIDE + 64 * S + 2;
    double* gr = gl + 64;
    double4* qw = reinterpret_cast<double4*>(su_lds + 4 * STRIDE);      // (64 ngb) weights of y_j, y_{j+1}, M'_j, M'_{j+1}
    double* qt = reinterpret_cast<double*>(qw + 64 * T.ngb);             // (64 ngb) damping factor of the query
    int* qe = reinterpret_cast<int*>(qt + 64 * T.ngb);                   // (64 ngb) interval
    for (int e = threadIdx.x; e < 64 * T.ngb; e += 256) {
        const int q = 64 * T.gb0 + e;
        qw[e] = reinterpret_cast<const double4*>(T.qw)[e];
        qt[e] = (A.tophat && q < T.nq) ? A.tophat[q] : 0.;
        qe[e] = T.qe[e];
    }
    __syncthreads();
    double ww[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) ww[o] = T.win[64 * o + lane];
    const int nm = T.nm;
    const double m_a = lane == 0 ? 1. : 0.;
    const double m_b = lane == T.lane_b ? T.mb0 : (lane == T.lane_b - 1 ? T.mb1 : 0.);
    const int nblocks = (T.nq + 63) >> 6, gb0 = T.gb0, gb1 = T.gb0 + T.ngb;
    double u[S], own[NPB], gvl, gvr;
    auto fetch_knots = [&](long long r) {      // the stretch (lane + 64 k: contiguous in its source row; beyond its end the last value again) and the knots outside it
        const double* a = A.src0 + r * A.n0;
        const double* b = A.src1 + r * A.n1;
        const double* pu = (T.src_u ? b : a) + T.col_u;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            const int i = lane + 64 * k;
            u[k] = pu[i < nm ? i : nm - 1];
        }
        gvl = lane < T.wl ? (T.src_l ? b : a)[T.col_l + lane] : 0.;
        gvr = lane < T.wr ? (T.src_r ? b : a)[T.col_r + lane] : 0.;
    };
    auto fetch_own = [&](long long r) {        // the row of the first array
        const double* a = A.src0 + r * A.n0;
#pragma unroll
        for (int blk = 0; blk < NPB; ++blk) {
            const int q = 64 * blk + lane;
            own[blk] = (blk < nblocks && q < T.nq) ? a[q] : 0.;
        }
    };
    long long row = (long long)blockIdx.x * 4 + wave;
    const long long step = (long long)gridDim.x * 4;
    if (row < A.nrows) {
        fetch_knots(row);
        fetch_own(row);
    }
    for (; row < A.nrows; row += step) {
        double* out = A.out + row * T.nq;
        // ---- staging ----
#pragma unroll
        for (int k = 0; k < S; ++k) yu[lane + 64 * k] = u[k];
        if (lane == 0) yu[-1] = u[0];
        if (lane == 63) yu[64 * S] = u[S - 1];
        gl[lane] = gvl;
        gr[lane] = gvr;
        cp::wave_lds_phase();
        if (row + step < A.nrows) fetch_knots(row + step);
        // ---- the lane's knots: second differences ----
        double g[S];
        {
            const double* mine = yu + S * lane - 1;
            double y1 = mine[1];
            double dprev = y1 - mine[0];
#pragma unroll
            for (int t = 0; t < S; ++t) {
                const double y2 = mine[t + 2];
                const double dn = y2 - y1;
                g[t] = dn - dprev;
                dprev = dn;
                y1 = y2;
            }
        }
        // ---- A / p, M_left, B, M_right: weighted sums over the knots next to the two junctions.  Each set of weights sums to zero (a constant has
        // no second derivative): the sums run over the DIFFERENCES to the knot they belong to -- exact, and as small as the second differences the
        // recursions work on, where the values themselves would leave 1e-16 of y in sums that are 1e-3 ... 1e-7 of it (and the long interval
        // behind the stretch multiplies its second derivatives by (its length / h)^2 ~ 1e6) ----
        const double gl_last = gl[T.wl > 0 ? T.wl - 1 : 0], gr_first = gr[0];
        double sums[4] = {0., 0., 0., 0.};
        if (!(CP_SPLICE_UNIFORM_ABLATE & 4)) {
            const double vl = gl[lane], ul = yu[lane < WIN_U ? lane : 0], ur = yu[lane < WIN_U ? nm - 1 - lane : 0], vr = gr[lane];
            const double yfirst = yu[0], ylast = yu[nm - 1];
            sums[0] = fma(ww[0], vl - yfirst, ww[1] * (ul - yfirst));
            sums[1] = fma(ww[2], vl - gl_last, ww[3] * (ul - gl_last));
            sums[2] = fma(ww[4], ur - ylast, ww[5] * (vr - ylast));
            sums[3] = fma(ww[6], ur - gr_first, ww[7] * (vr - gr_first));
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
                for (int o = 0; o < 4; ++o) sums[o] += __shfl_xor(sums[o], off);
        }
        // ---- the queries' knot values, before the second derivatives take their place ----
        double part[NPB];
        {
#pragma unroll
            for (int blk = 0; blk < NPB; ++blk) {
                part[blk] = 0.;
                if (blk >= gb0 && blk < gb1) {
                    const int slot = 64 * (blk - gb0) + lane;
                    const int e = qe[slot];
                    const double4 w = qw[slot];
                    const double ya = yu[e], yb = yu[e + 1];      // (e = -1 and e + 1 = nm read the repeated end values: replaced)
                    part[blk] = fma(w.x, e < 0 ? gl_last : ya, w.y * (e + 1 >= nm ? gr_first : yb));
                }
            }
        }
        cp::wave_lds_phase();      // every read of the knot values is done
        // ---- the two recursions over the lane's knots, from zero; the neighbours' totals; the second derivatives (unscaled) into LDS ----
        if (!(CP_SPLICE_UNIFORM_ABLATE & 1)) {
#pragma unroll
            for (int t = S - 2; t >= 0; --t) g[t] = fma(P, g[t + 1], g[t]);
            double c = fma(m_b, sums[2], from_right(g[0]));      // G at the first knot of the next segment (+ what B adds there)
            double f = 0.;
#pragma unroll
            for (int t = 0; t < S; ++t) {
                const double e = fma(P, f, g[t]);
                f = t + 1 < S ? fma(-P, g[t + 1], e) : e;
                g[t] = e;
            }
#pragma unroll
            for (int t = S - 1; t >= 0; --t) {      // (all of the segment: in the lane of the stretch's last knot the carry stands for B p^(distance to that knot))
                c *= P;
                g[t] += c;
            }
            c = fma(m_a, sums[0], from_left(f));                 // F at the last knot of the previous segment (+ A / p in lane 0)
#pragma unroll
            for (int t = 0; t < (S < 36 ? S : 36); ++t) {
                c *= P;
                g[t] += c;
            }
            double* mine = yu + S * lane;
#pragma unroll
            for (int t = 0; t < S; ++t) mine[t] = g[t];
        }
        cp::wave_lds_phase();      // (the slots beyond the stretch hold no second derivative anyone reads)
        if (lane == 0) yu[-1] = sums[1];
        if (lane == 1) yu[nm] = sums[3];
        cp::wave_lds_phase();
        // ---- evaluation; the other queries return their own column ----
#pragma unroll
        for (int blk = 0; blk < NPB; ++blk) {
            const int q = 64 * blk + lane;
            if (blk >= gb0 && blk < gb1 && !(CP_SPLICE_UNIFORM_ABLATE & 2)) {
                const int slot = 64 * (blk - gb0) + lane;
                const int e = qe[slot];
                const double4 w = qw[slot];
                double v = part[blk] + fma(w.z, yu[e], w.w * yu[e + 1]);
                const double p = own[blk];
                // pk / ((pk / pknow - 1) tophat + 1), bao_filter.py:421-431, as pk pknow / ((pk - pknow) tophat + pknow) with one reciprocal: the two IEEE
                // divisions per output were four fifths of the vector instructions of this kernel
                if (A.tophat) v = p * (v * cpmath::recip(fma(p - v, qt[slot], v)));      // (pknow / denominator is O(1): no product of two spectra)
                if (q < T.nq) out[q] = (q >= T.gfirst && q < T.gend) ? v : p;
            } else if (blk < nblocks && q < T.nq) {
                out[q] = own[blk];
            }
        }
        if (row + step < A.nrows) fetch_own(row + step);
        cp::wave_lds_phase();      // last reads of the second derivatives before the next row is staged
    }
}

template <int S>
inline hipError_t launch_s(const Args& A, unsigned grid, size_t lds, hipStream_t stream) {
    (void)cp::allow_full_lds<&splice_uniform_kernel<S>>();
    hipLaunchKernelGGL(splice_uniform_kernel<S>, dim3(grid), dim3(256), lds, stream, A);
    return hipGetLastError();
}

inline hipError_t launch(const Args& A, unsigned grid, size_t lds, hipStream_t stream) {
    switch (A.T.S) {
        case 33: return launch_s<33>(A, grid, lds, stream);
        case 37: return launch_s<37>(A, grid, lds, stream);
        case 41: return launch_s<41>(A, grid, lds, stream);
        case 45: return launch_s<45>(A, grid, lds, stream);
        case 49: return launch_s<49>(A, grid, lds, stream);
        case 53: return launch_s<53>(A, grid, lds, stream);
        case 57: return launch_s<57>(A, grid, lds, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace cpsu
