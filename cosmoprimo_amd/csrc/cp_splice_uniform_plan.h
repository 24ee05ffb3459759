// cp_splice_uniform_plan.h -- the host side of cp_splice_uniform.h: its constants, the tables a launch takes, and the function that decides whether
// the scheme fits a set of knots and queries and builds those tables (plain C++: compiled by hipcc into the library, and by g++ for the CPU tests,
// tests/host_emu/emu_splice.cpp).  The scheme itself is described in cp_splice_uniform.h.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

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

}  // namespace cpsu
