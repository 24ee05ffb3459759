// cp_spline_rows.hip -- a natural or clamped cubic spline through fixed knots, applied to very many rows and evaluated at fixed queries, by
// solving its tridiagonal system per row in LDS (scipy.interpolate.CubicSpline / the reference's Interpolator1D, jax.py:169-175; the step of
// sigma_r / sigma_rz that takes the FFTLog output to the radii, interpolator.py:285-291, 846-876).
//
// As an operator (cp_spline_apply) the spline costs bandwidth x queries multiply-adds per row -- 64 per query, because the inverse of its
// tridiagonal matrix decays by 0.27 per knot and has to be kept to 1e-17 -- and the matrix-core route multiplies whole windows of knots on top
// (33 000 multiply-adds per row for 1024 knots -> 256 radii).  The elimination itself is ~10 operations per knot and 8 per query:
//   * only the knots the queries touch, plus `halo` knots on either side, are read and solved (a sweep forgets where it started at the rate
//     its factors decay; the halo is chosen at plan creation so that 1e-18 of the start is left);
//   * R = 1, 2 or 4 rows per wave: 64 / R lanes per row, a lane owns S knots (S odd: its neighbours' segments start on other banks) and runs
//     both sweeps over halo + S knots -- values, eliminated right-hand sides and second derivatives share one LDS buffer with the end values
//     repeated beyond either end, exactly as in splice_kernel (cp_bao.hip), whose elimination this is;
//   * the factors (6 / h_i, 1 / pivot_i, h_{i-1} / pivot_i, h_i / pivot_i) depend on the knots only: a table in LDS, built on the host;
//   * queries: A y_j + B y_{j+1} + ((A^3 - A) M_j + (B^3 - B) M_{j+1}) h_j^2 / 6, weights from the plan; scale, root, and the store with the
//     rows of a group as the fastest axis (cp_spline_apply_grouped's layout) in the same pass.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_internal.h"

namespace {

struct RowsTables {
    int n_src;                // knots of a full row (its length in memory)
    int w0, nw;               // the knots [w0, w0 + nw) are read and solved
    int nq, S, halo, R;       // R rows per wave, S knots per lane
    int nslots;               // table entries: window position i -> slot min(max(i, 0), nslots - 1)
    const double* tab;        // (nslots, 4): 6 / h_i, 1 / pivot_i, h_{i-1} / pivot_i, h_i / pivot_i
    const int* qj;            // (nq) interval of each query relative to w0, -1: outside the knots
    const double* qw;         // (nq, 4): A, B, (A^3 - A) h^2 / 6, (B^3 - B) h^2 / 6
    // not-a-knot: the outermost second derivatives follow from their neighbours once the sweeps are done, M_0 = fix[0] M_1 + fix[1] M_2 and
    // M_{n-1} = fix[2] M_{n-2} + fix[3] M_{n-3} (continuity of the third derivative at the second and the last-but-one knot)
    int fix_first, fix_last;
    double fix[4];
};

struct RowsArgs {
    RowsTables T;
    const double* y;          // (nrows, n_src)
    long long nrows;
    int post_op, group;
    double scale;
    double* out;
    double* out_m;            // (nrows, n_src) or null: the second derivatives at the knots of the window instead of the queries
    int pairs;                // out_m is (nrows, n_src, 2): the knot values with their second derivatives, (y_j, M_j)
};

__device__ __forceinline__ long long rows_out_index(long long row, int q, int nq, int group) {      // cp_spline.hip: out_index
    if (group <= 0) return row * nq + q;
    const long long b = row / group;
    return (b * nq + q) * group + (row - b * group);
}

template <int R>
__global__ __launch_bounds__(256) void spline_rows_kernel(const RowsArgs A) {
    constexpr int LPR = 64 / R;      // lanes per row
    extern __shared__ __attribute__((aligned(32))) double sr_lds[];
    const RowsTables& T = A.T;
    const int S = T.S, nw = T.nw, halo = T.halo, pad = halo + 1;
    const int rstride = LPR * S + 2 * pad + 1;      // doubles per row buffer
    double* tab = sr_lds + ((4 * R * rstride + 3) & ~3);
    for (int e = threadIdx.x; e < 4 * T.nslots; e += 256) tab[e] = T.tab[e];
    __syncthreads();
    const double4* coef = reinterpret_cast<const double4*>(tab);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rr = lane / LPR, l = lane % LPR;
    double* buf = sr_lds + (wave * R + rr) * rstride + pad;      // buf[i], -pad <= i < LPR S + pad
    const int own = S * l, last_slot = T.nslots - 1;
    auto slot_of = [&](int i) { return i < 0 ? 0 : (i > last_slot ? last_slot : i); };
    const long long ngroups = (A.nrows + R - 1) / R;
    for (long long g = (long long)blockIdx.x * 4 + wave; g < ngroups; g += (long long)gridDim.x * 4) {
        const long long row = g * R + rr;
        const bool live = row < A.nrows;
        const double* src = A.y + (live ? row : A.nrows - 1) * T.n_src + T.w0;
        cp::wave_lds_phase();      // the previous group's last reads are done before its buffers are staged over
        // the knots of the window, the end values repeated beyond either end (slopes there vanish: the clamped condition; for the natural one and
        // for an end that is not an end of the row the table makes the outermost equations harmless)
#pragma unroll 8
        for (int i = l; i < nw; i += LPR) buf[i] = src[i];
        {
            const double first = src[0], last = src[nw - 1];
            for (int e = l; e < pad; e += LPR) buf[-1 - e] = first;
            for (int e = nw + l; e < LPR * S + pad; e += LPR) buf[e] = last;
        }
        cp::wave_lds_phase();      // knot values staged
#ifndef CP_ROWS_ABLATE
#define CP_ROWS_ABLATE 0      // tools/spline_rows_variants.sh, what the parts cost: 1 no forward sweep, 2 no back substitution, 4 one query per lane,
#endif                        // 8 no values gathered from memory, 16 no root, 32 no stores
        if (!(CP_ROWS_ABLATE & 1)) {
            const double beyond = buf[own + S];
            const int start = own - halo;
            double y0 = buf[start], d = 0.;
            double sigma_m = (y0 - buf[start - 1]) * coef[slot_of(start - 1)].x;
#pragma unroll 8
            for (int t = 0; t < halo; ++t) {
                const int i = start + t;
                const double4 c = coef[slot_of(i)];
                const double yp = buf[i + 1];
                const double sigma = (yp - y0) * c.x;
                d = fma(-c.z, d, c.y * (sigma - sigma_m));
                sigma_m = sigma;
                y0 = yp;
            }
            cp::wave_lds_phase();      // run-ins done before the neighbours' knot values become d
#pragma unroll 8
            for (int t = 0; t < S - 1; ++t) {
                const int i = own + t;
                const double4 c = coef[slot_of(i)];
                const double yp = buf[i + 1];
                const double sigma = (yp - y0) * c.x;
                d = fma(-c.z, d, c.y * (sigma - sigma_m));
                buf[i] = d;
                sigma_m = sigma;
                y0 = yp;
            }
            {
                const int i = own + S - 1;
                const double4 c = coef[slot_of(i)];
                const double sigma = (beyond - y0) * c.x;
                buf[i] = fma(-c.z, d, c.y * (sigma - sigma_m));
            }
        }
        cp::wave_lds_phase();      // d complete
        if (!(CP_ROWS_ABLATE & 2)) {
            double m = 0.;
#pragma unroll 8
            for (int t = 0; t < halo; ++t) {
                const int i = own + S + halo - 1 - t;
                const double d = buf[i];
                m = i >= nw - 1 ? d : fma(-coef[slot_of(i)].w, m, d);
            }
            cp::wave_lds_phase();
#pragma unroll 8
            for (int t = 0; t < S; ++t) {
                const int i = own + S - 1 - t;
                const double d = buf[i];
                m = i >= nw - 1 ? d : fma(-coef[slot_of(i)].w, m, d);
                buf[i] = m;
            }
        }
        cp::wave_lds_phase();      // M complete
        if (T.fix_first && l == 0) buf[0] = T.fix[0] * buf[1] + T.fix[1] * buf[2];
        if (T.fix_last && l == 0) buf[nw - 1] = T.fix[2] * buf[nw - 2] + T.fix[3] * buf[nw - 3];
        cp::wave_lds_phase();      // end fixes in place: the evaluation gathers across segments
        if (A.out_m) {
            if (A.pairs) {
                double2* dst = reinterpret_cast<double2*>(A.out_m) + row * T.n_src + T.w0;
                if (live) {      // (the row's values four at a time ahead of their stores: one by one each load waited behind the store of the pair before it)
                    for (int i0 = l; i0 < nw; i0 += 4 * LPR) {
                        double y[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int i = i0 + u * LPR;
                            y[u] = src[i < nw ? i : nw - 1];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int i = i0 + u * LPR;
                            if (i < nw) dst[i] = double2{y[u], buf[i]};
                        }
                    }
                }
                continue;
            }
            double* dst = A.out_m + row * T.n_src + T.w0;
            if (live)
                for (int i = l; i < nw; i += LPR) dst[i] = buf[i];
            continue;
        }
#pragma unroll 16
        for (int q = l; q < ((CP_ROWS_ABLATE & 4) ? LPR : T.nq); q += LPR) {
            const int j = T.qj[q];
            const int jj = j < 0 ? 0 : j;
            const double4 w = reinterpret_cast<const double4*>(T.qw)[q];
            double v = ((CP_ROWS_ABLATE & 8) ? w.x + w.y : w.x * src[jj] + w.y * src[jj + 1]) + (w.z * buf[jj] + w.w * buf[jj + 1]);
            v = j < 0 ? __builtin_nan("") : v * A.scale;
            if (A.post_op == CP_SPLINE_POST_SQRT && !(CP_ROWS_ABLATE & 16)) v = sqrt(v);
            if (live && (!(CP_ROWS_ABLATE & 32) || v == 12345.678)) A.out[rows_out_index(row, q, T.nq, A.group)] = v;
        }
    }
}

}  // namespace

struct cp_spline_rows_plan {
    RowsTables T;
    int device;
    double* d_tab;
    int* d_qj;
    double* d_qw;
    size_t lds_bytes;
};

extern "C" int cp_spline_rows_plan_destroy(cp_spline_rows_plan* p) {
    if (!p) return CP_OK;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device) (void)hipSetDevice(p->device);
    if (p->d_tab) (void)hipFree(p->d_tab);
    if (p->d_qj) (void)hipFree(p->d_qj);
    if (p->d_qw) (void)hipFree(p->d_qw);
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    delete p;
    return CP_OK;
}

extern "C" int cp_spline_rows_plan_create(cp_spline_rows_plan** out, int n, const double* x, int bc, int extrapolate, int nq, const double* xq, int device) {
    if (!out) return cp::fail(CP_EINVAL, "cp_spline_rows_plan_create: null plan pointer");
    *out = nullptr;
    if (n < 4 || !x || nq < 1 || !xq) return cp::fail(CP_EINVAL, "cp_spline_rows_plan_create: bad arguments");
    if (nq > (1 << 26) || n > (1 << 26))      // (a query is staged as an interval and four weights on the host: 2^26 queries are 2.4 GB; a catalogue comes in pieces)
        return cp::fail(CP_EUNSUPPORTED, "cp_spline_rows_plan_create: %d knots, %d queries (at most 2^26 each): evaluate in pieces", n, nq);
    if (bc != CP_SPLINE_NATURAL && bc != CP_SPLINE_CLAMPED && bc != CP_SPLINE_NOT_A_KNOT) return cp::fail(CP_EINVAL, "cp_spline_rows_plan_create: unknown boundary condition %d", bc);
    for (int i = 0; i + 1 < n; ++i)
        if (!(x[i + 1] > x[i])) return cp::fail(CP_EINVAL, "cp_spline_rows_plan_create: knots must increase");
    // the system for the second derivatives, sub_i M_{i-1} + diag_i M_i + sup_i M_{i+1} = 6 (s_i - s_{i-1}), and its elimination factors.
    // clamped: 2 h_0 M_0 + h_0 M_1 = 6 s_0 and its mirror image; natural: M_0 = M_{n-1} = 0 (factor 0 in the outermost rows); not-a-knot: the
    // outermost unknowns eliminated through M_0 = (1 + h_0 / h_1) M_1 - (h_0 / h_1) M_2 and its mirror image (factor 0 again, filled in afterwards)
    std::vector<double> h(n), inv(n), c(n), q(n), sub(n), diag(n), sup(n);
    for (int i = 0; i + 1 < n; ++i) h[i] = x[i + 1] - x[i];
    h[n - 1] = h[n - 2];
    for (int i = 1; i < n - 1; ++i) {
        sub[i] = h[i - 1];
        diag[i] = 2. * (h[i - 1] + h[i]);
        sup[i] = h[i];
    }
    sub[0] = 0.; diag[0] = 2. * h[0]; sup[0] = h[0];
    sub[n - 1] = h[n - 2]; diag[n - 1] = 2. * h[n - 2]; sup[n - 1] = 0.;
    const bool open_ends = bc != CP_SPLINE_CLAMPED;      // rows 0 and n - 1 are not equations of the system
    double fix[4] = {0., 0., 0., 0.};
    if (bc == CP_SPLINE_NOT_A_KNOT) {
        if (n < 5) return cp::fail(CP_EUNSUPPORTED, "cp_spline_rows_plan_create: a not-a-knot spline through %d knots", n);
        fix[0] = 1. + h[0] / h[1]; fix[1] = -h[0] / h[1];
        fix[2] = 1. + h[n - 2] / h[n - 3]; fix[3] = -h[n - 2] / h[n - 3];
        diag[1] += h[0] * fix[0]; sup[1] += h[0] * fix[1]; sub[1] = 0.;
        diag[n - 2] += h[n - 2] * fix[2]; sub[n - 2] += h[n - 2] * fix[3]; sup[n - 2] = 0.;
    }
    inv[0] = open_ends ? 0. : 1. / diag[0];
    c[0] = sup[0] * inv[0];
    q[0] = 0.;
    for (int i = 1; i < n; ++i) {
        inv[i] = (open_ends && i == n - 1) ? 0. : 1. / (diag[i] - sub[i] * c[i - 1]);
        c[i] = sup[i] * inv[i];
        q[i] = sub[i] * inv[i];
    }
    // the queries' intervals
    std::vector<int> qj(nq);
    int jmin = n, jmax = -1;
    for (int k = 0; k < nq; ++k) {
        const double v = xq[k];
        if (!(v >= x[0] && v <= x[n - 1]) && !(extrapolate && v == v)) {      // outside the knots: NaN, or the cubic of the end interval continued
            qj[k] = -1;
            continue;
        }
        int j = (int)(std::upper_bound(x, x + n, v) - x) - 1;
        j = j > n - 2 ? n - 2 : (j < 0 ? 0 : j);
        qj[k] = j;
        jmin = j < jmin ? j : jmin;
        jmax = j > jmax ? j : jmax;
    }
    if (jmax < 0) jmin = jmax = 0;      // every query outside the knots: NaN everywhere, any window
    // how far a sweep remembers its start
    int halo = 8;
    for (;; halo += 8) {
        if (halo > 128) return cp::fail(CP_EUNSUPPORTED, "cp_spline_rows_plan_create: the elimination does not forget its start within 128 knots");
        double worst = 0.;
        for (int i = halo; i < n; ++i) {
            double f = 1., b = 1.;
            for (int t = 0; t < halo; ++t) {
                f *= std::fabs(q[i - t]);
                b *= std::fabs(c[i - t - 1]);
            }
            worst = std::max(worst, std::max(f, b));
        }
        if (worst < 1e-18 || halo >= n) break;
    }
    const int w0 = std::max(0, jmin - halo), w1 = std::min(n, jmax + 2 + halo), nw = w1 - w0;
    int R = nw <= 16 * 27 ? 4 : (nw <= 32 * 59 ? 2 : 1);
    const int lpr = 64 / R;
    const int S = ((nw + lpr - 1) / lpr) | 1;
    const int pad = halo + 1;
    const int nslots = lpr * S + pad;
    const int rstride = lpr * S + 2 * pad + 1;
    const size_t lds = ((size_t)((4 * R * rstride + 3) & ~3) + 4 * (size_t)nslots) * sizeof(double);
    if (lds > 160 * 1024) return cp::fail(CP_EUNSUPPORTED, "cp_spline_rows_plan_create: a window of %d knots exceeds the LDS budget of the kernel", nw);
    std::vector<double> tab((size_t)4 * nslots), qw((size_t)4 * nq, 0.);
    for (int s = 0; s < nslots; ++s) {
        const int i = std::min(w0 + s, n - 1);
        tab[4 * s] = 6. / h[i];
        tab[4 * s + 1] = inv[i];
        tab[4 * s + 2] = q[i];
        tab[4 * s + 3] = c[i];
    }
    for (int k = 0; k < nq; ++k) {
        if (qj[k] < 0) continue;
        const int j = qj[k];
        const double a = (x[j + 1] - xq[k]) / h[j], b = (xq[k] - x[j]) / h[j];
        qw[4 * k] = a;
        qw[4 * k + 1] = b;
        qw[4 * k + 2] = (a * a * a - a) * (h[j] * h[j]) / 6.;
        qw[4 * k + 3] = (b * b * b - b) * (h[j] * h[j]) / 6.;
        qj[k] = j - w0;
    }
    cp_spline_rows_plan* p = new (std::nothrow) cp_spline_rows_plan();
    if (!p) return cp::fail(CP_ENOMEM, "cp_spline_rows_plan_create: host allocation failed");
    p->device = device;
    p->d_tab = nullptr; p->d_qj = nullptr; p->d_qw = nullptr;
    p->lds_bytes = lds;
    p->T.fix_first = bc == CP_SPLINE_NOT_A_KNOT && w0 == 0;
    p->T.fix_last = bc == CP_SPLINE_NOT_A_KNOT && w1 == n;
    for (int i = 0; i < 4; ++i) p->T.fix[i] = fix[i];
    p->T.n_src = n; p->T.w0 = w0; p->T.nw = nw; p->T.nq = nq; p->T.S = S; p->T.halo = halo; p->T.R = R; p->T.nslots = nslots;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    bool ok = prev == device || hipSetDevice(device) == hipSuccess;
    ok = ok && hipMalloc(&p->d_tab, tab.size() * sizeof(double)) == hipSuccess && hipMalloc(&p->d_qj, qj.size() * sizeof(int)) == hipSuccess &&
         hipMalloc(&p->d_qw, qw.size() * sizeof(double)) == hipSuccess &&
         hipMemcpy(p->d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_qj, qj.data(), qj.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(p->d_qw, qw.data(), qw.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (!ok) {
        cp_spline_rows_plan_destroy(p);
        return cp::fail(CP_ENOMEM, "cp_spline_rows_plan_create: cannot place the tables on device %d", device);
    }
    p->T.tab = p->d_tab; p->T.qj = p->d_qj; p->T.qw = p->d_qw;
    *out = p;
    return CP_OK;
}

extern "C" int cp_spline_rows_plan_info(const cp_spline_rows_plan* p, int* first_knot, int* nknots, int* rows_per_wave, int* halo) {
    if (!p) return cp::fail(CP_EINVAL, "cp_spline_rows_plan_info: null plan");
    if (first_knot) *first_knot = p->T.w0;
    if (nknots) *nknots = p->T.nw;
    if (rows_per_wave) *rows_per_wave = p->T.R;
    if (halo) *halo = p->T.halo;
    return CP_OK;
}

static int launch_rows(const cp_spline_rows_plan* p, const RowsArgs& A, long long nrows, void* stream) {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || ncu <= 0) ncu = 256;
    const long long blocks = ((nrows + p->T.R - 1) / p->T.R + 3) / 4;
    long long per_cu = (160 * 1024) / (long long)p->lds_bytes;
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    const unsigned grid = (unsigned)std::min(blocks, (long long)ncu * per_cu);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (p->T.R == 4) (void)cp::allow_full_lds<&spline_rows_kernel<4>>();
    else if (p->T.R == 2) (void)cp::allow_full_lds<&spline_rows_kernel<2>>();
    else (void)cp::allow_full_lds<&spline_rows_kernel<1>>();
    if (p->T.R == 4) hipLaunchKernelGGL(spline_rows_kernel<4>, dim3(grid), dim3(256), p->lds_bytes, hs, A);
    else if (p->T.R == 2) hipLaunchKernelGGL(spline_rows_kernel<2>, dim3(grid), dim3(256), p->lds_bytes, hs, A);
    else hipLaunchKernelGGL(spline_rows_kernel<1>, dim3(grid), dim3(256), p->lds_bytes, hs, A);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_rows: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_spline_rows_apply(const cp_spline_rows_plan* p, const double* d_y, long long nrows, int post_op, double scale, int group, double* d_out,
                                    void* stream) {
    if (!p) return cp::fail(CP_EINVAL, "cp_spline_rows_apply: null plan");
    if (nrows < 0) return cp::fail(CP_EINVAL, "cp_spline_rows_apply: negative batch");
    if (nrows == 0) return CP_OK;
    if (!d_y || !d_out) return cp::fail(CP_EINVAL, "cp_spline_rows_apply: null device pointer");
    if (post_op != CP_SPLINE_POST_NONE && post_op != CP_SPLINE_POST_SQRT) return cp::fail(CP_EINVAL, "cp_spline_rows_apply: post op %d (none or sqrt)", post_op);
    if (group < 0 || (group > 0 && nrows % group != 0)) return cp::fail(CP_EINVAL, "cp_spline_rows_apply: %lld rows are not whole groups of %d", nrows, group);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_rows_apply: cannot select device %d", p->device);
    RowsArgs A;
    A.T = p->T;
    A.y = d_y; A.nrows = nrows; A.post_op = post_op; A.group = group; A.scale = scale; A.out = d_out; A.out_m = nullptr; A.pairs = 0;
    const int st = launch_rows(p, A, nrows, stream);
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    return st;
}

static int second_derivatives(const cp_spline_rows_plan* p, const double* d_y, long long nrows, double* d_m, int pairs, void* stream);

extern "C" int cp_spline_rows_second_derivatives(const cp_spline_rows_plan* p, const double* d_y, long long nrows, double* d_m, void* stream) {
    return second_derivatives(p, d_y, nrows, d_m, 0, stream);
}

// the same, written with the knot values beside them: d_ym (nrows, n, 2) = (y_j, M_j) -- the layout cp_tables_rows_direct reads with d_m null
extern "C" int cp_spline_rows_pairs(const cp_spline_rows_plan* p, const double* d_y, long long nrows, double* d_ym, void* stream) {
    return second_derivatives(p, d_y, nrows, d_ym, 1, stream);
}

static int second_derivatives(const cp_spline_rows_plan* p, const double* d_y, long long nrows, double* d_m, int pairs, void* stream) {
    if (!p) return cp::fail(CP_EINVAL, "cp_spline_rows_second_derivatives: null plan");
    if (nrows < 0) return cp::fail(CP_EINVAL, "cp_spline_rows_second_derivatives: negative batch");
    if (nrows == 0) return CP_OK;
    if (!d_y || !d_m) return cp::fail(CP_EINVAL, "cp_spline_rows_second_derivatives: null device pointer");
    if (p->T.w0 != 0 || p->T.nw != p->T.n_src)
        return cp::fail(CP_EUNSUPPORTED, "cp_spline_rows_second_derivatives: the plan solves the knots %d .. %d only (its queries do not span the knots)", p->T.w0, p->T.w0 + p->T.nw - 1);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_rows_second_derivatives: cannot select device %d", p->device);
    RowsArgs A;
    A.T = p->T;
    A.y = d_y; A.nrows = nrows; A.post_op = CP_SPLINE_POST_NONE; A.group = 0; A.scale = 1.; A.out = nullptr; A.out_m = d_m; A.pairs = pairs;
    const int st = launch_rows(p, A, nrows, stream);
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    return st;
}

// what the two-direction kernel of cp_spline.hip needs of a plan whose queries are the output wavenumbers of (z, k) tables (cp_internal.h)
bool cp_spline_rows_plan_view(const cp_spline_rows_plan* p, cp_spline_rows_view* out) {
    if (!p || !out) return false;
    *out = cp_spline_rows_view{p->T.n_src, p->T.nq, p->T.w0, p->T.nw, p->device, p->d_qj, p->d_qw};
    return true;
}
