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

namespace cpsu {

constexpr int WIN_G = 44;      // knots outside the stretch that A, B and the outer second derivatives see on either side (all of them if fewer)
constexpr int WIN_U = 40;      // knots of the stretch they see
constexpr int NGB = 8;         // blocks of 64 queries evaluated through the spline
constexpr int NPB = 16;        // blocks of 64 queries in all (the others return their own column of the first array)
constexpr double P = -0.26794919243112270647;      // sqrt 3 - 2

struct Tables {
    int S, nm;                   // knots per lane; knots of the uniform stretch
    int wl, wr;                  // knots in front of / behind the stretch that are staged
    int src_u, col_u;            // the stretch: array (0 / 1) and first column
    int src_l, col_l;            // the wl knots in front of it
    int src_r, col_r;            // the wr knots behind it
    int nq, gb0, ngb;            // queries; first block of 64 queries that holds a spline query, number of such blocks
    int gfirst, gend;            // queries [gfirst, gend) go through the spline
    int lane_b;                  // the lane that owns the last knot of the stretch
    double mb0, mb1;             // what B weighs in the incoming carry of lane_b and of lane_b - 1
    const double* win;           // (8, 64) weights of A / p, M_left, B, M_right (units of the unscaled recursion) on the window entries of a lane
    const int* qe;               // (64 ngb) interval of query 64 gb0 + i relative to the first knot of the stretch (-1: the interval in front of it)
    const double* qw;            // (64 ngb, 4) weights of y_j, y_{j+1}, M'_j, M'_{j+1}
};

struct Args {
    Tables T;
    const double* src0;
    const double* src1;
    int n0, n1;
    long long nrows;
    const double* tophat;        // (nq) or null
    double* out;                 // (nrows, nq)
};

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
                if (A.tophat) v = p / ((p / v - 1.) * qt[slot] + 1.);      // pk / ((pk / pknow - 1) tophat + 1), bao_filter.py:421-431
                if (q < T.nq) out[q] = (q >= T.gfirst && q < T.gend) ? v : p;
            } else if (blk < nblocks && q < T.nq) {
                out[q] = own[blk];
            }
        }
        if (row + step < A.nrows) fetch_own(row + step);
        cp::wave_lds_phase();      // last reads of the second derivatives before the next row is staged
    }
}

// ---- host: does the scheme fit these knots and queries, and if so its tables ----
struct Built {
    bool ok = false;
    Tables T;                       // (device pointers left null)
    std::vector<double> win, qw;
    std::vector<int> qe;
    size_t lds_bytes = 0;
};

inline int pick_S(int nm) {
    for (int S = 33; S <= 57; S += 4)
        if (64 * S >= nm) return S;
    return 0;
}

// x: the n knots; pieces as in cp_splice_plan_create (first knot, source, start column of each of the 3 pieces); qj / xq: the queries and their
// intervals (-1: outside the knots).  The uniform stretch is knots [u0, u1].
inline Built build(int n, const double* x, const int* piece_first, const int* piece_src, const int* piece_start, int npieces, int nq, const double* xq,
                   const int* qj, int u0, int u1, int generic_first, int generic_end) {
    Built B;
    const int nm = u1 - u0 + 1;
    if (nm < 256 || nq > 64 * NPB || generic_end <= generic_first) return B;
    const int S = pick_S(nm);
    if (!S) return B;
    auto piece_of = [&](int i) {
        int k = npieces - 1;
        while (k > 0 && i < piece_first[k]) --k;
        return k;
    };
    auto piece_end = [&](int k) { return k + 1 < npieces ? piece_first[k + 1] : n; };
    const int pu = piece_of(u0);
    if (piece_of(u1) != pu) return B;
    // the windows: knots of ONE piece each, contiguous in their source rows
    int wl = 0, wr = 0, pl = pu, pr = pu;
    if (u0 > 0) {
        pl = piece_of(u0 - 1);
        wl = std::min(WIN_G, u0 - piece_first[pl]);
    }
    if (u1 < n - 1) {
        pr = piece_of(u1 + 1);
        wr = std::min(WIN_G, piece_end(pr) - (u1 + 1));
    }
    // every spline query inside [x_{u0 - 1}, x_{u1 + 1}]
    const int gb0 = generic_first / 64, gb1 = (generic_end + 63) / 64;
    if (gb1 - gb0 > NGB) return B;
    for (int q = generic_first; q < generic_end; ++q) {
        const int j = qj[q];
        if (j < 0 || j < u0 - 1 || j > u1 || (j == u0 - 1 && wl < 1) || (j == u1 && wr < 1)) return B;
    }
    std::vector<double> h(n);
    for (int i = 0; i + 1 < n; ++i) h[i] = x[i + 1] - x[i];
    h[n - 1] = h[n - 2];
    const double h0 = (x[u1] - x[u0]) / (nm - 1);
    const double kappa = 1. / (2. * std::sqrt(3.)), kscale = 6. * kappa / (h0 * h0);
    // rows of the exact inverse: M = T^-1 R y, T symmetric tridiagonal (diagonal 2 (h_{i-1} + h_i), clamped ends 2 h_0 and 2 h_{n-2}; off-diagonal
    // h_i), R the slopes' differences times 6
    auto inverse_row = [&](int i) {
        std::vector<double> diag(n), xs(n, 0.), c(n), row(n, 0.);
        for (int k = 0; k < n; ++k) diag[k] = k == 0 ? 2. * h[0] : (k == n - 1 ? 2. * h[n - 2] : 2. * (h[k - 1] + h[k]));
        xs[i] = 1.;
        // Thomas
        c[0] = h[0] / diag[0];
        xs[0] /= diag[0];
        for (int k = 1; k < n; ++k) {
            const double m = diag[k] - h[k - 1] * c[k - 1];
            c[k] = k < n - 1 ? h[k] / m : 0.;
            xs[k] = (xs[k] - h[k - 1] * xs[k - 1]) / m;
        }
        for (int k = n - 2; k >= 0; --k) xs[k] -= c[k] * xs[k + 1];
        // row_j = sum_i xs_i R_ij
        for (int k = 0; k < n; ++k) {
            const double v = xs[k];
            if (v == 0.) continue;
            if (k == 0) {
                row[0] += -6. / h[0] * v;
                row[1] += 6. / h[0] * v;
            } else if (k == n - 1) {
                row[n - 2] += 6. / h[n - 2] * v;
                row[n - 1] += -6. / h[n - 2] * v;
            } else {
                row[k - 1] += 6. / h[k - 1] * v;
                row[k] += -6. * (1. / h[k - 1] + 1. / h[k]) * v;
                row[k + 1] += 6. / h[k] * v;
            }
        }
        for (double& v : row) v /= kscale;      // units of the unscaled recursion
        return row;
    };
    // what the device's recursion gives at knot i of the stretch, as weights on the stretch's values: sum_j p^|i-j| D'[j, :], D' the second
    // differences with the end values repeated beyond the stretch
    auto model_row = [&](int i, std::vector<double>& row) {      // subtracts it from row (global knot numbering)
        for (int j = 0; j < nm; ++j) {
            const double w = std::pow(P, std::abs(i - j));
            if (std::fabs(w) < 1e-40) continue;
            const double lo = j > 0 ? 1. : 0., hi = j < nm - 1 ? 1. : 0.;      // y_{j-1} and y_{j+1} exist; else the knot's own value stands in
            row[u0 + j] -= w * (-2. + (1. - lo) + (1. - hi));
            if (j > 0) row[u0 + j - 1] -= w;
            if (j < nm - 1) row[u0 + j + 1] -= w;
        }
    };
    std::vector<double> aL = inverse_row(u0), aR = inverse_row(u1), bL, bR;
    model_row(0, aL);
    model_row(nm - 1, aR);
    if (u0 > 0) bL = inverse_row(u0 - 1);
    if (u1 < n - 1) bR = inverse_row(u1 + 1);
    // all that is left outside the windows must be negligible
    auto outside = [&](const std::vector<double>& row, int lo, int hi) {      // largest |weight| outside [lo, hi) over the largest inside
        if (row.empty()) return 0.;
        double in = 0., out = 0.;
        for (int k = 0; k < n; ++k) {
            const double v = std::fabs(row[k]);
            if (k >= lo && k < hi) in = std::max(in, v);
            else out = std::max(out, v);
        }
        return in > 0. ? out / in : 1.;
    };
    const int llo = u0 - wl, lhi = u0 + WIN_U, rlo = u1 + 1 - WIN_U, rhi = u1 + 1 + wr;
    const double worst = std::max(std::max(outside(aL, llo, lhi), outside(bL, llo, lhi)), std::max(outside(aR, rlo, rhi), outside(bR, rlo, rhi)));
    if (!(worst < 1e-18)) return B;
    B.win.assign(8 * 64, 0.);
    for (int l = 0; l < 64; ++l) {
        if (l < wl) {
            B.win[0 * 64 + l] = aL[u0 - wl + l] / P;
            if (!bL.empty()) B.win[2 * 64 + l] = bL[u0 - wl + l];
        }
        if (l < WIN_U) {
            B.win[1 * 64 + l] = aL[u0 + l] / P;
            if (!bL.empty()) B.win[3 * 64 + l] = bL[u0 + l];
            B.win[4 * 64 + l] = aR[u1 - l];
            if (!bR.empty()) B.win[6 * 64 + l] = bR[u1 - l];
        }
        if (l < wr) {
            B.win[5 * 64 + l] = aR[u1 + 1 + l];
            if (!bR.empty()) B.win[7 * 64 + l] = bR[u1 + 1 + l];
        }
    }
    const int ngb = gb1 - gb0;
    B.qe.assign((size_t)64 * ngb, 0);
    B.qw.assign((size_t)256 * ngb, 0.);
    for (int e = 0; e < 64 * ngb; ++e) {
        const int q = 64 * gb0 + e;
        if (q < generic_first || q >= generic_end) continue;      // (weights zero, interval 0: evaluated and not stored)
        const int j = qj[q];
        const double a = (x[j + 1] - xq[q]) / h[j], b = (xq[q] - x[j]) / h[j];
        B.qe[e] = j - u0;
        B.qw[4 * e] = a;
        B.qw[4 * e + 1] = b;
        B.qw[4 * e + 2] = (a * a * a - a) * (h[j] * h[j]) / 6. * kscale;
        B.qw[4 * e + 3] = (b * b * b - b) * (h[j] * h[j]) / 6. * kscale;
    }
    Tables& T = B.T;
    T.S = S; T.nm = nm; T.wl = wl; T.wr = wr;
    T.src_u = piece_src[pu]; T.col_u = piece_start[pu] + (u0 - piece_first[pu]);
    T.src_l = piece_src[pl]; T.col_l = piece_start[pl] + (u0 - wl - piece_first[pl]);
    T.src_r = piece_src[pr]; T.col_r = piece_start[pr] + (u1 + 1 - piece_first[pr]);
    T.nq = nq; T.gb0 = gb0; T.ngb = ngb; T.gfirst = generic_first; T.gend = generic_end;
    T.lane_b = (nm - 1) / S;
    const int tb = (nm - 1) - S * T.lane_b;
    T.mb0 = std::pow(P, tb - S);
    T.mb1 = std::pow(P, tb);
    T.win = nullptr; T.qe = nullptr; T.qw = nullptr;
    B.lds_bytes = ((size_t)4 * (64 * S + 2 + 128) + (size_t)(256 + 64) * ngb) * sizeof(double) + (size_t)64 * ngb * sizeof(int);
    B.ok = B.lds_bytes <= 160 * 1024;
    return B;
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
