// cp_spline.hip -- cubic-spline interpolation from fixed knots to fixed query points as a banded linear operator
// applied to batches of rows (gfx950) + C ABI.
//
// Replaces the scipy spline calls of the reference wherever knots and queries are shared by a batch:
//   Interpolator1D: CubicSpline(x, fun, bc_type='natural') + evaluation          jax.py:169-175, 196
//   (used by integrate_sigma_r2 method 'fftlog': 1024 FFTLog output knots -> r)   interpolator.py:285-289
//   Interpolator2D: RectBivariateSpline(kx=ky=3, s=0) == separable not-a-knot cubic splines, jax.py:241-271
//   clamped CubicSpline second derivatives of the wallish2018 filter              bao_filter.py:377-382
// A cubic spline is linear in its data: out = W y with W (nq x n) fixed by (knots, queries, boundary condition,
// derivative order).  W is built once per plan on the host with n tridiagonal solves (scipy CubicSpline's system for
// the knot first derivatives + Hermite evaluation) and is exponentially banded (each query couples to ~30-40 knots
// either side at 1e-17), so only the band is stored and applied: bandwidth x nq multiply-adds per row instead of a
// tridiagonal solve per row.  The kernel stages, for up to 16 rows, the knots under a tile of consecutive queries in LDS; every lane owns one query.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "../../include/cosmoprimo_amd.h"
#include "cp_error.h"
#include "cp_internal.h"
#include "cp_math.h"

namespace {

using cpmath::exp10_mid;

constexpr int TILE_Q = 256;      // at most one query per lane
constexpr int SPAN_CAP = 480;    // knots a tile may cover when it holds more than one query: 16 rows x 480 knots = 60 KB of LDS

// where the value of (row, query) is stored: (nrows, nq) row-major, or -- group > 0 -- (nrows / group, nq, group): the rows of a group (the
// redshifts of one table) become the fastest axis, i.e. the (nz, nr) -> (nr, nz) transposition of sigma_rz is part of the store
__device__ __forceinline__ long long out_index(long long row, int q, int nq, int group) {
    if (group <= 0) return row * nq + q;
    const long long b = row / group;
    return (b * nq + q) * group + (row - b * group);
}

struct Args {
    const double* y;    // (nrows, n)
    double* out;        // (nrows, nq)
    long long nrows;
    int n, nq, bw;
    const double* wb;   // (bw, nq) band, query fastest
    const int* j0;      // (nq) first knot of each query's band; -1: query outside the knots -> NaN
    const int* tile;    // (ntiles, 4): first query, number of queries (<= TILE_Q), first knot and number of knots their bands cover
    int ntiles, span_max;
    int post_op;
    double scale;
    int group;          // > 0: output (nrows / group, nq, group) -- row r lands at [r / group, q, r % group] (a transposition fused into the store)
};

// One work item = R rows x one tile of consecutive queries.  The knots the tile's bands cover are staged in LDS for the R rows; a lane owns
// one query and streams its band weights once for the R rows (weights come from L2: bw x nq doubles per plan, shared by all rows).  With
// R = 16 the weight traffic per output is a quarter of the first version's (4 rows, whole rows in LDS), which ran at the L2 rate.  Tiles are
// cut where their knots would exceed SPAN_CAP, so operators that thin the knots out unevenly (log-spaced queries on 3666 linear knots: 83
// knots per query at the top) get many narrow tiles there instead of one tile that forces whole rows into LDS.
template <int R>
__global__ __launch_bounds__(256) void spline_apply_kernel(const Args A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ylds = reinterpret_cast<double*>(smem);   // (R, span_max)
    const int tid = threadIdx.x;
    const long long ngroups = (A.nrows + R - 1) / R;
    const long long nitems = ngroups * A.ntiles;
    for (long long it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int t = (int)(it % A.ntiles);   // tiles of one row group run together: the group's rows are read from HBM once, the rest from L2
        const long long r0 = (it / A.ntiles) * R;
        const int nr = (int)((A.nrows - r0) < R ? (A.nrows - r0) : R);
        const int q0 = A.tile[4 * t], nqt = A.tile[4 * t + 1], tj0 = A.tile[4 * t + 2], span = A.tile[4 * t + 3];
        __syncthreads();
        // (the R rows' entries of a column requested together, then stored: row by row each load was waited for before the next was issued -- R memory round
        // trips in a row per tile; the band weights eight at a time ahead of their multiply-adds: one by one they were a round trip per weight.  Same arithmetic,
        // same order: tools/isa_waits.py)
        for (int i0 = 0; i0 < span; i0 += 256) {
            const int i = i0 + tid, ic = i < span ? i : span - 1;
            double v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = A.y[(r0 + (r < nr ? r : nr - 1)) * A.n + tj0 + ic];   // rows past the end repeat the last one (never stored)
            if (i < span) {
#pragma unroll
                for (int r = 0; r < R; ++r) ylds[r * A.span_max + i] = v[r];
            }
        }
        __syncthreads();
        if (tid >= nqt) continue;
        const int q = q0 + tid;
        const int j0 = A.j0[q];
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.;
        if (j0 >= 0) {
            const int base = j0 - tj0;
            constexpr int CHUNK = 8;
            for (int jj0 = 0; jj0 < A.bw; jj0 += CHUNK) {
                double w[CHUNK];
#pragma unroll
                for (int u = 0; u < CHUNK; ++u) w[u] = A.wb[(long long)(jj0 + u < A.bw ? jj0 + u : A.bw - 1) * A.nq + q];
#pragma unroll
                for (int u = 0; u < CHUNK; ++u) {
                    if (jj0 + u < A.bw) {
                        int j = base + jj0 + u;
                        j = j < span ? j : span - 1;  // padded band entries carry w = 0
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r] = fma(w[u], ylds[r * A.span_max + j], acc[r]);
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (r < nr) {
                double v = j0 >= 0 ? acc[r] * A.scale : __builtin_nan("");
                if (A.post_op == CP_SPLINE_POST_SQRT) v = sqrt(v);
                else if (A.post_op == CP_SPLINE_POST_EXP10) v = exp10_mid(v);
                A.out[out_index(r0 + r, q, A.nq, A.group)] = v;
            }
        }
    }
}

// The same with an outer-product epilogue: out[row, q, z] = f(scale x spline(y)[row, q] x g[row, z]), f = sqrt or identity, written once as
// (nrows, nq, nz).  sigma(r, z) of separable spectra P(k, z) = P(k) D^2(z) is of that form (sigma^2(r) x D^2(z), rooted: reference
// interpolator.py:846-875 evaluates one integral per redshift).  The interpolated values of a tile are staged in LDS and the workgroup
// then streams the rows' (nqt x nz) blocks -- contiguous in the output -- with 16-byte stores: the result, by far the largest array of the
// chain (nq nz values per input row), is produced by the kernel that computes it instead of by elementwise passes behind it.
struct OuterArgs {
    Args a;
    const double* g;   // (nrows, nz)
    int nz;
};

template <int R>
__global__ __launch_bounds__(256) void spline_outer_kernel(const OuterArgs O) {
    const Args& A = O.a;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* ylds = reinterpret_cast<double*>(smem);   // (R, span_max) knots, then (R, TILE_Q) interpolated values, then (R, nz) factors
    double* vlds = ylds + R * A.span_max;
    double* glds = vlds + R * TILE_Q;
    const int tid = threadIdx.x;
    const int nz = O.nz;
    const long long ngroups = (A.nrows + R - 1) / R;
    const long long nitems = ngroups * A.ntiles;
    for (long long it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int t = (int)(it % A.ntiles);
        const long long r0 = (it / A.ntiles) * R;
        const int nr = (int)((A.nrows - r0) < R ? (A.nrows - r0) : R);
        const int q0 = A.tile[4 * t], nqt = A.tile[4 * t + 1], tj0 = A.tile[4 * t + 2], span = A.tile[4 * t + 3];
        __syncthreads();   // the previous item's stores have read vlds / glds
        // (loads in batches, as in spline_apply_kernel: the R rows' entries of a column together, the band weights eight at a time)
        for (int i0 = 0; i0 < span; i0 += 256) {
            const int i = i0 + tid, ic = i < span ? i : span - 1;
            double v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = A.y[(r0 + (r < nr ? r : nr - 1)) * A.n + tj0 + ic];
            if (i < span) {
#pragma unroll
                for (int r = 0; r < R; ++r) ylds[r * A.span_max + i] = v[r];
            }
        }
        for (int i0 = 0; i0 < nz; i0 += 256) {
            const int i = i0 + tid, ic = i < nz ? i : nz - 1;
            double v[R];
#pragma unroll
            for (int r = 0; r < R; ++r) v[r] = O.g[(r0 + (r < nr ? r : nr - 1)) * nz + ic];
            if (i < nz) {
                // f = sqrt: sqrt(v g) is written as sqrt(v) sqrt(g) -- nq + nz roots per row instead of nq nz (both factors are >= 0 or the result is NaN either way)
#pragma unroll
                for (int r = 0; r < R; ++r) glds[r * nz + i] = A.post_op == CP_SPLINE_POST_SQRT ? sqrt(v[r]) : v[r];
            }
        }
        __syncthreads();
        if (tid < nqt) {
            const int q = q0 + tid;
            const int j0 = A.j0[q];
            double acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0.;
            if (j0 >= 0) {
                const int base = j0 - tj0;
                constexpr int CHUNK = 8;
                for (int jj0 = 0; jj0 < A.bw; jj0 += CHUNK) {
                    double w[CHUNK];
#pragma unroll
                    for (int u = 0; u < CHUNK; ++u) w[u] = A.wb[(long long)(jj0 + u < A.bw ? jj0 + u : A.bw - 1) * A.nq + q];
#pragma unroll
                    for (int u = 0; u < CHUNK; ++u) {
                        if (jj0 + u < A.bw) {
                            int j = base + jj0 + u;
                            j = j < span ? j : span - 1;
#pragma unroll
                            for (int r = 0; r < R; ++r) acc[r] = fma(w[u], ylds[r * A.span_max + j], acc[r]);
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double v = j0 >= 0 ? acc[r] * A.scale : __builtin_nan("");
                vlds[r * TILE_Q + tid] = A.post_op == CP_SPLINE_POST_SQRT ? sqrt(v) : v;
            }
        }
        __syncthreads();
        const int block = nqt * nz;   // values of one row of this tile, contiguous in the output (nqt <= 256: well below 2^31 for any sensible nz)
        // element e = q nz + z of the block; a thread walks e = e0 + stride i without dividing: (q, z) advance by (stride / nz, stride % nz)
        const int stride = (nz & 1) == 0 ? 512 : 256, e0 = (nz & 1) == 0 ? 2 * tid : tid;
        const int dq = stride / nz, dz = stride - dq * nz, q_first = e0 / nz, z_first = e0 - q_first * nz;
        for (int r = 0; r < nr; ++r) {
            double* dst = A.out + ((r0 + r) * A.nq + q0) * nz;
            const double* vr = vlds + r * TILE_Q;
            const double* gr = glds + r * nz;
            int q = q_first, z = z_first;
            if ((nz & 1) == 0) {
                for (int e = e0; e < block; e += 512) {
                    double2 v;
                    v.x = vr[q] * gr[z];
                    v.y = vr[q] * gr[z + 1];
                    *reinterpret_cast<double2*>(dst + e) = v;
                    q += dq; z += dz;
                    if (z >= nz) { z -= nz; ++q; }
                }
            } else {
                for (int e = e0; e < block; e += 256) {
                    dst[e] = vr[q] * gr[z];
                    q += dq; z += dz;
                    if (z >= nz) { z -= nz; ++q; }
                }
            }
        }
    }
}

template <int R>
hipError_t launch_outer(const OuterArgs& O, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024)
        (void)cp::allow_full_lds<&spline_outer_kernel<R>>();
    const long long nitems = ((O.a.nrows + R - 1) / R) * O.a.ntiles;
    const int grid = (int)(nitems < 256 * 8 ? nitems : 256 * 8);
    hipLaunchKernelGGL(spline_outer_kernel<R>, dim3(grid), dim3(256), lds, stream, O);
    return hipGetLastError();
}

// ---- dense operators on the matrix cores -------------------------------------------------------------------------------------------
// out (nrows, nq) = Y (nrows, n) W^T for an operator W (nq, n) that is genuinely dense (quadrature weights, least-squares projectors, products
// of spline operators: cp_linop_plan_create): a GEMM in float64, v_mfma_f64_16x16x4_f64.  Lane l of a wave supplies A[l & 15][k = l >> 4] =
// Y[row][k] and B[k = l >> 4][l & 15] = W[q][k]; both matrices are row-major with k contiguous, so a lane loads FOUR consecutive k of its row
// (32 bytes) and spends them on four MFMAs -- lane group g = l >> 4 then carries k = kb + 4 g + m in step m for A and for B alike.  The
// results sit at D[row = (l >> 4) + 4 reg][col = l & 15] (cdna_hip_programming.md, fragment layout of the f64 form).
// Wave tile 32 rows x 64 queries (2 x 4 accumulator tiles, 64 VGPRs); the four waves of a workgroup sit side by side in q on the same rows.
// No LDS: W is small and lives in L2, a Y row segment is re-read by the four waves from L1 and nq / 256 times from L2 / HBM.
typedef double cp_v4d __attribute__((ext_vector_type(4)));

struct DenseArgs {
    const double* y;
    double* out;
    long long nrows;
    int n, nq, n_pad, nq_pad;
    const double* w;   // (nq_pad, n_pad), zero padded
    const int* j0;     // (nq): < 0 marks a query that evaluates to NaN
    const int* kwin;   // (nq_pad / 64, 2): the knots [lo, hi), multiples of 16, that the bands of each tile of 64 queries cover; then (nq_pad / 16, 2): per 16 queries
    int post_op;
    double scale;
    int group;         // as Args::group
};

// Round 3: wave tile 64 rows x 64 queries (4 x 4 accumulator tiles, 128 registers) instead of 32 x 64.  The kernel reads its operands straight from
// L1 / L2 in the fragment layout, and what bounded the 32 x 64 tile was that path: per 16-knot chunk a wave fetched (32 + 64) x 16 doubles = 12 KB
// for 32 MFMAs (512 matrix-core cycles), 24 B per cycle and wave, 96 B per cycle for the four SIMDs of a CU against the 64 B per cycle the L1
// delivers -- 44-46 TFLOP/s = 58 % of the matrix peak, the ratio of the two.  A 64 x 64 tile fetches 16 KB for 64 MFMAs: 16 B per cycle and wave.
// Two waves per SIMD (launch bound: 256 registers) alternate between fetching a chunk and multiplying one.
constexpr int LINOP_MT = 4;   // 16-row tiles per wave
#ifndef CP_LINOP_ABLATE
#define CP_LINOP_ABLATE 0
#endif

// SUB: the four tiles of 16 queries of a wave each skip the chunks of the wave's window that their own bands do not reach (plans whose tiles of 16
// need at most two thirds of the chunks: cp_spline_plan::sub_windows; elsewhere the tests and the split loop cost more than the MFMAs they save)
template <bool SUB>
__global__ __launch_bounds__(256, 2) void linop_mfma_kernel(const DenseArgs A) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;      // (in a scalar register: what follows from it is wave-uniform for the compiler too)
    const int l15 = lane & 15, g = lane >> 4;
    constexpr int MT = LINOP_MT, ROWS = 16 * MT;
    const long long nrt = (A.nrows + ROWS - 1) / ROWS;
    const int nqt = (A.nq_pad + 255) / 256;
    for (long long item = blockIdx.x; item < nrt * nqt; item += gridDim.x) {
        const long long row0 = (item / nqt) * ROWS;   // consecutive items share their rows: Y comes from L2 for all but the first
        const int q0 = (int)(item % nqt) * 256 + wave * 64;
        if (q0 >= A.nq_pad) continue;
        cp_v4d acc[MT][4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = cp_v4d{0., 0., 0., 0.};
        const double* yr[MT];
        const double* wr[4];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            long long row = row0 + 16 * i + l15;
            row = row < A.nrows ? row : A.nrows - 1;   // rows past the end repeat the last one (never stored)
            yr[i] = A.y + row * A.n;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[j] = A.w + (long long)(q0 + 16 * j + l15) * A.n_pad;
        // a banded operator (spline) is a block-banded GEMM: the wave's 64 queries only couple to the knots of their window
        const int klo = A.kwin[2 * (q0 >> 6)], khi = A.kwin[2 * (q0 >> 6) + 1];
        // ... and each of its four tiles of 16 queries to a part of that window (64 + 16 / density knots of 64 + 64 / density): chunks of the
        // window outside it are skipped for that tile -- a third of the MFMAs of a spline between grids of similar density (wave-uniform tests)
        int slo[4], shi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            slo[j] = SUB ? A.kwin[2 * (A.nq_pad >> 6) + 2 * ((q0 >> 4) + j)] : klo;
            shi[j] = SUB ? A.kwin[2 * (A.nq_pad >> 6) + 2 * ((q0 >> 4) + j) + 1] : khi;
        }
#pragma unroll 1
        for (int kb = klo; kb < khi; kb += 16) {
            const int k = kb + 4 * g;
            // (CP_LINOP_ABLATE, diagnostic builds of tools/linop_microbench.hip only: bit 0 re-reads the operator, bit 1 the rows, of the first chunk)
            double a[MT][4];
            cp_v4d b[4];
            bool act[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                act[j] = !SUB || (kb >= slo[j] && kb < shi[j]);
                if (act[j]) b[j] = *reinterpret_cast<const cp_v4d*>(wr[j] + ((CP_LINOP_ABLATE & 1) ? klo + 4 * g : k));
            }
            if (CP_LINOP_ABLATE & 2) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[i][m] = yr[i][klo + 4 * g + m];
            } else if (kb + 16 <= A.n) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[i][m] = yr[i][k + m];
            } else {   // last, partial chunk: W is zero there, but Y must not bring in the next row's values (0 x NaN)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[i][m] = k + m < A.n ? yr[i][k + m] : 0.;
            }
            if (SUB) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (act[j]) {
#pragma unroll
                        for (int m = 0; m < 4; ++m)
#pragma unroll
                            for (int i = 0; i < MT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][m], b[j][m], acc[i][j], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][m], b[j][m], acc[i][j], 0, 0, 0);
            }
        }
        // where element (row, q) goes is out_index(row, q): with the rows of the tile inside one group (or no groups) that is a base of the tile plus
        // row x rstride + q x qstride -- one 64-bit division per tile instead of one per stored element (64 of them per lane)
        const bool plain = A.group <= 0, one_group = A.group > 0 && A.group % ROWS == 0;
        long long base = row0 * A.nq;
        long long rstride = A.nq, qstride = 1;
        if (one_group) {
            const long long b = row0 / A.group;
            base = b * A.nq * A.group + (row0 - b * A.group);
            rstride = 1;
            qstride = A.group;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + 16 * j + l15;
            if (q >= A.nq) continue;
            const bool nanq = A.j0[q] < 0;
            double* outq = A.out + base + q * qstride;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rowin = 16 * i + g + 4 * r;
                    if (row0 + rowin >= A.nrows) continue;
                    double v = nanq ? __builtin_nan("") : acc[i][j][r] * A.scale;
                    if (A.post_op == CP_SPLINE_POST_SQRT) v = sqrt(v);
                    else if (A.post_op == CP_SPLINE_POST_EXP10) v = exp10_mid(v);
                    if (plain || one_group) outq[rowin * rstride] = v;
                    else A.out[out_index(row0 + rowin, q, A.nq, A.group)] = v;
                }
        }
    }
}

// The same contraction along the MIDDLE axis of (nbatch, n, ninner) arrays: out[b, q, c] = f(scale x sum_j W[q, j] y[b, j, c]), c contiguous --
// the redshift interpolation of batches of (z, k) tables whose rows along k feed the FFTLog next (P(k, z) tables are splined in log10 P: f =
// 10^x in the epilogue writes the spectra once, z-major, instead of interpolating, transposing and exponentiating in three passes).
// Here the operator is the A matrix, A[m = l & 15][k = l >> 4] = W[q0 + m][j], and y the B matrix, B[k = l >> 4][n = l & 15] =
// y[b, j, c0 + n]: per k step a lane group reads 16 consecutive doubles of one row j (128-byte segments, four adjacent ones for the four
// column tiles of a wave).  Wave tile 64 queries x 32 columns (4 x 2 accumulator tiles: with 4 x 4 the kernel needs 434 registers, one wave per SIMD, and runs at 2.2 TB/s); the four waves of a workgroup take adjacent columns.
struct MidArgs {
    const double* y;
    double* out;
    long long nbatch, ninner;
    int n, nq, n_pad, nq_pad;
    const double* w;   // (nq_pad, n_pad), zero padded
    const int* j0;     // (nq): < 0 marks a query that evaluates to NaN
    int post_op;
    double scale;
};

constexpr int MID_NT = 2;   // 16-column tiles per wave: 64 queries x 32 columns (4 x 2 accumulator tiles)
constexpr int MID_WSTRIDE = 80;   // LDS row stride (doubles) of the resident operator, 64 queries + padding: lane groups fall on different banks

// The kernel moves 3 bytes for every multiply-add: what it needs is memory operations in flight, i.e. waves -- four per SIMD (128 registers: the
// accumulators of a 64 x 32 tile, the operator in LDS rather than in registers).
__global__ __launch_bounds__(256, 4) void linop_mid_mfma_kernel(const MidArgs A) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;      // (in a scalar register: what follows from it is wave-uniform for the compiler too)
    const int l15 = lane & 15, g = lane >> 4;
    constexpr int NT = MID_NT, WCOLS = 16 * NT, GCOLS = 4 * WCOLS;   // columns per wave and per workgroup
    const long long nct = (A.ninner + GCOLS - 1) / GCOLS;
    const int nqt = A.nq_pad / 64;
    const long long nitems = A.nbatch * nct * nqt;
    // a small operator (one tile of 64 queries, at most 32 knots: the redshift grids of P(k, z) tables) stays in LDS for all items, knot-major
    const bool resident = nqt == 1 && A.n_pad <= 32;
    __shared__ double wl[32 * MID_WSTRIDE];
    if (resident) {
        for (int e = threadIdx.x; e < 64 * A.n_pad; e += 256) {
            const int q = e / A.n_pad, k = e - q * A.n_pad;
            wl[k * MID_WSTRIDE + q] = A.w[e];
        }
        __syncthreads();
    }
    for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int qt = (int)(item % nqt);
        const long long ct = (item / nqt) % nct, b = item / (nqt * nct);
        const long long c0 = ct * GCOLS + wave * WCOLS;
        if (c0 >= A.ninner) continue;
        const int q0 = qt * 64;
        const double* yb = A.y + b * A.n * A.ninner;
        cp_v4d acc[4][NT];   // [query tile][column tile]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = cp_v4d{0., 0., 0., 0.};
        long long col[NT];
        bool colok[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            col[j] = c0 + 16 * j + l15;
            colok[j] = col[j] < A.ninner;
            col[j] = colok[j] ? col[j] : A.ninner - 1;
        }
        auto step = [&](int kb, const double* a) {
            const int jrow = kb + g;
            double bv[NT];
            const bool rowok = jrow < A.n;     // W is zero in its padding, but y must not bring in another batch entry's values (0 x NaN)
            const double* yr = yb + (long long)(rowok ? jrow : 0) * A.ninner;
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = rowok ? yr[col[j]] : 0.;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bv[j], acc[i][j], 0, 0, 0);
        };
        if (resident) {
            for (int kb = 0; kb < A.n_pad; kb += 4) {
                double a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = wl[(kb + g) * MID_WSTRIDE + 16 * i + l15];
                step(kb, a);
            }
        } else {
            for (int kb = 0; kb < A.n_pad; kb += 4) {
                double a[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = A.w[(long long)(q0 + 16 * i + l15) * A.n_pad + kb + g];
                step(kb, a);
            }
        }
        double* ob = A.out + b * A.nq * A.ninner;
        int jq[4][4];      // (the queries' band starts requested together, not one by one between the stores)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + 16 * i + g + 4 * r;
                jq[i][r] = A.j0[q < A.nq ? q : A.nq - 1];
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q0 + 16 * i + g + 4 * r;
                if (q >= A.nq) continue;
                const bool nanq = jq[i][r] < 0;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if (!colok[j]) continue;
                    double v = nanq ? __builtin_nan("") : acc[i][j][r] * A.scale;
                    if (A.post_op == CP_SPLINE_POST_SQRT) v = sqrt(v);
                    else if (A.post_op == CP_SPLINE_POST_EXP10) v = exp10_mid(v);
                    ob[(long long)q * A.ninner + col[j]] = v;
                }
            }
    }
}

template <int R>
hipError_t launch_apply(const Args& A, size_t lds, hipStream_t stream) {
    if (lds > 64 * 1024)
        (void)cp::allow_full_lds<&spline_apply_kernel<R>>();
    const long long nitems = ((A.nrows + R - 1) / R) * A.ntiles;
    const int grid = (int)(nitems < 256 * 8 ? nitems : 256 * 8);
    hipLaunchKernelGGL(spline_apply_kernel<R>, dim3(grid), dim3(256), lds, stream, A);
    return hipGetLastError();
}

// scipy.interpolate.CubicSpline: tridiagonal system for the knot first derivatives s (lower, diag, upper) and the
// right-hand side as a sparse linear map of y (entries rhs[i] = sum_k c[i][k] y[col[i][k]], at most 4 per row)
struct RhsRow {
    int col[4];
    double c[4];
    int nnz;
};

void add(RhsRow& r, int col, double c) {
    for (int k = 0; k < r.nnz; ++k)
        if (r.col[k] == col) {
            r.c[k] += c;
            return;
        }
    r.col[r.nnz] = col;
    r.c[r.nnz] = c;
    ++r.nnz;
}

// slope_i = (y[i+1] - y[i]) / dx[i] scaled by f, added to the row
void add_slope(RhsRow& r, int i, double f, const std::vector<double>& dx) {
    add(r, i + 1, f / dx[i]);
    add(r, i, -f / dx[i]);
}

int build_system(int n, const double* x, int bc, std::vector<double>& lo, std::vector<double>& di, std::vector<double>& up, std::vector<RhsRow>& rhs,
                 std::vector<double>& dx) {
    dx.resize(n - 1);
    for (int i = 0; i < n - 1; ++i) {
        dx[i] = x[i + 1] - x[i];
        if (!(dx[i] > 0.)) return cp::fail(CP_EINVAL, "cp_spline_plan_create: knots must be strictly increasing");
    }
    lo.assign(n, 0.);
    di.assign(n, 0.);
    up.assign(n, 0.);
    rhs.assign(n, RhsRow{{0, 0, 0, 0}, {0., 0., 0., 0.}, 0});
    for (int i = 1; i < n - 1; ++i) {  // interior rows
        lo[i] = dx[i];
        di[i] = 2. * (dx[i - 1] + dx[i]);
        up[i] = dx[i - 1];
        add_slope(rhs[i], i - 1, 3. * dx[i], dx);
        add_slope(rhs[i], i, 3. * dx[i - 1], dx);
    }
    if (bc == CP_SPLINE_NATURAL) {
        di[0] = 2. * dx[0]; up[0] = dx[0];
        add(rhs[0], 1, 3.); add(rhs[0], 0, -3.);
        di[n - 1] = 2. * dx[n - 2]; lo[n - 1] = dx[n - 2];
        add(rhs[n - 1], n - 1, 3.); add(rhs[n - 1], n - 2, -3.);
    } else if (bc == CP_SPLINE_CLAMPED) {
        di[0] = 1.; up[0] = 0.;
        di[n - 1] = 1.; lo[n - 1] = 0.;
    } else if (bc == CP_SPLINE_NOT_A_KNOT) {
        if (n < 4) return cp::fail(CP_EINVAL, "cp_spline_plan_create: not-a-knot needs at least 4 knots");
        double d = x[2] - x[0];
        di[0] = dx[1]; up[0] = d;
        add_slope(rhs[0], 0, (dx[0] + 2. * d) * dx[1] / d, dx);
        add_slope(rhs[0], 1, dx[0] * dx[0] / d, dx);
        d = x[n - 1] - x[n - 3];
        di[n - 1] = dx[n - 3]; lo[n - 1] = d;
        add_slope(rhs[n - 1], n - 3, dx[n - 2] * dx[n - 2] / d, dx);
        add_slope(rhs[n - 1], n - 2, (2. * d + dx[n - 2]) * dx[n - 3] / d, dx);
    } else {
        return cp::fail(CP_EINVAL, "cp_spline_plan_create: unknown boundary condition %d", bc);
    }
    return CP_OK;
}

}  // namespace

struct cp_spline_plan {
    int n, nq, bw, device;
    double* d_wb;
    int* d_j0;
    int* d_tile;
    int ntiles, span_max;
    double* d_wdense;    // operators that are dense (band wider than half the knots): (nq_pad, n_pad) row-major, zero padded, for the MFMA kernel
    int* d_kwin;         // windows of knots per tile of 64 queries, for the matrix-core kernel
    int n_pad, nq_pad;
    bool prefer_dense;
    bool sub_windows;          // linop_mfma_kernel<true>: the tiles of 16 queries need at most two thirds of the chunks of their tiles of 64
    int col_first, col_last;   // cp_spline_plan_columns: the entries of a row that cp_spline_apply(_grouped) may read, on either route
};

bool cp_spline_plan_view(const cp_spline_plan* p, cp_spline_band_view* out) {
    if (!p || !out || !p->d_wb || !p->d_j0) return false;
    *out = cp_spline_band_view{p->n, p->nq, p->bw, p->device, p->d_wb, p->d_j0};
    return true;
}

extern "C" int cp_spline_plan_destroy(cp_spline_plan* p) {
    if (!p) return CP_OK;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device) (void)hipSetDevice(p->device);
    if (p->d_wb) (void)hipFree(p->d_wb);
    if (p->d_j0) (void)hipFree(p->d_j0);
    if (p->d_tile) (void)hipFree(p->d_tile);
    if (p->d_wdense) (void)hipFree(p->d_wdense);
    if (p->d_kwin) (void)hipFree(p->d_kwin);
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    delete p;
    return CP_OK;
}

// Dense operator on the host (row-major nq x n); exposed so that tests can pin it against scipy without a GPU.
extern "C" int cp_spline_operator(int n, const double* x, int nq, const double* xq, int bc, int nu, int extrapolate, double* w_out, int* inside_out) {
    if (n < 2 || nq < 0 || !x || (nq > 0 && !xq) || !w_out) return cp::fail(CP_EINVAL, "cp_spline_operator: bad arguments");
    if (nu < 0 || nu > 2) return cp::fail(CP_EINVAL, "cp_spline_operator: derivative order %d not in [0, 2]", nu);
    std::vector<double> lo, di, up, dx;
    std::vector<RhsRow> rhs;
    int st = build_system(n, x, bc, lo, di, up, rhs, dx);
    if (st != CP_OK) return st;
    // Thomas factorisation (matrix only)
    std::vector<double> cp(n), inv(n);
    inv[0] = 1. / di[0];
    cp[0] = up[0] * inv[0];
    for (int i = 1; i < n; ++i) {
        inv[i] = 1. / (di[i] - lo[i] * cp[i - 1]);
        cp[i] = up[i] * inv[i];
    }
    // interval of each query
    std::vector<int> kq(nq);
    for (int q = 0; q < nq; ++q) {
        const double v = xq[q];
        const bool in = v >= x[0] && v <= x[n - 1];
        if (inside_out) inside_out[q] = in ? 1 : 0;
        if (!in && !extrapolate) {
            kq[q] = -1;
            continue;
        }
        int a = 0, b = n - 1;
        while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (v >= x[mid]) a = mid;
            else b = mid;
        }
        kq[q] = a;
    }
    std::memset(w_out, 0, sizeof(double) * (size_t)nq * n);
    std::vector<double> s(n), d(n);
    for (int j = 0; j < n; ++j) {  // column j: data = e_j
        for (int i = 0; i < n; ++i) {
            double v = 0.;
            for (int k = 0; k < rhs[i].nnz; ++k)
                if (rhs[i].col[k] == j) v += rhs[i].c[k];
            d[i] = v;
        }
        d[0] = d[0] * inv[0];
        for (int i = 1; i < n; ++i) d[i] = (d[i] - lo[i] * d[i - 1]) * inv[i];
        s[n - 1] = d[n - 1];
        for (int i = n - 2; i >= 0; --i) s[i] = d[i] - cp[i] * s[i + 1];
        for (int q = 0; q < nq; ++q) {
            const int k = kq[q];
            if (k < 0) continue;
            // PPoly coefficients of interval k (scipy CubicSpline): t = (s_k + s_{k+1} - 2 slope) / dx
            const double h = dx[k];
            const double yk = (j == k) ? 1. : 0., yk1 = (j == k + 1) ? 1. : 0.;
            const double slope = (yk1 - yk) / h;
            const double t = (s[k] + s[k + 1] - 2. * slope) / h;
            const double c3 = t / h, c2 = (slope - s[k]) / h - t, c1 = s[k], c0 = yk;
            const double u = xq[q] - x[k];
            double v;
            if (nu == 0) v = c0 + u * (c1 + u * (c2 + u * c3));
            else if (nu == 1) v = c1 + u * (2. * c2 + u * 3. * c3);
            else v = 2. * c2 + 6. * c3 * u;
            w_out[(size_t)q * n + j] = v;
        }
    }
    for (int q = 0; q < nq; ++q)
        if (kq[q] < 0)
            for (int j = 0; j < n; ++j) w_out[(size_t)q * n + j] = std::numeric_limits<double>::quiet_NaN();
    return CP_OK;
}

// plan from a dense (nq x n) operator; rows whose first entry is NaN mark queries that evaluate to NaN
static int plan_from_dense(cp_spline_plan** out, int n, int nq, const double* w, int device, bool keep_dense) {
    std::vector<int> j0(nq), j1(nq);
    int bw = 1;
    for (int q = 0; q < nq; ++q) {
        if (std::isnan(w[(size_t)q * n])) {
            j0[q] = -1;
            j1[q] = -1;
            continue;
        }
        double mx = 0.;
        for (int j = 0; j < n; ++j) mx = std::fmax(mx, std::fabs(w[(size_t)q * n + j]));
        if (mx == 0.) {  // a row of zeros (an operator that acts on part of its queries only): placed below
            j0[q] = j1[q] = -2;
            continue;
        }
        const double thr = mx * 1e-18;  // band: entries above 1e-18 of the row maximum
        int a = 0, b = n - 1;
        while (a < b && std::fabs(w[(size_t)q * n + a]) <= thr) ++a;
        while (b > a && std::fabs(w[(size_t)q * n + b]) <= thr) --b;
        j0[q] = a;
        j1[q] = b;
        if (b - a + 1 > bw) bw = b - a + 1;
    }
    // rows of zeros get a one-entry band (weight 0) next to the band of the nearest row that has one, so that they widen neither the tiles of the
    // vector kernel nor the windows of the matrix-core kernel
    {
        int last = -1;
        for (int q = 0; q < nq; ++q) {
            if (j0[q] >= 0) last = j0[q];
            else if (j0[q] == -2 && last >= 0) j0[q] = j1[q] = last;
        }
        last = -1;
        for (int q = nq - 1; q >= 0; --q) {
            if (j0[q] >= 0) last = j0[q];
            else if (j0[q] == -2) j0[q] = j1[q] = last >= 0 ? last : 0;
        }
    }
    std::vector<double> wb((size_t)bw * nq, 0.);
    for (int q = 0; q < nq; ++q) {
        if (j0[q] < 0) continue;
        for (int jj = 0; jj < bw; ++jj) {
            const int j = j0[q] + jj;
            if (j <= j1[q]) wb[(size_t)jj * nq + q] = w[(size_t)q * n + j];
        }
    }
    // tiles of consecutive queries: as many as TILE_Q, cut earlier where the knots under their bands would exceed SPAN_CAP
    std::vector<int> tile;
    int span_max = 1;
    for (int q0 = 0; q0 < nq;) {
        int lo = n, hi = 0, q = q0;
        for (; q < nq && q - q0 < TILE_Q; ++q) {
            int nlo = lo, nhi = hi;
            if (j0[q] >= 0) {
                nlo = j0[q] < lo ? j0[q] : lo;
                const int end = j0[q] + bw < n ? j0[q] + bw : n;
                nhi = end > hi ? end : hi;
            }
            if (q > q0 && nhi > nlo && nhi - nlo > SPAN_CAP) break;
            lo = nlo;
            hi = nhi;
        }
        if (hi <= lo) lo = 0, hi = 1;   // every query of the tile is outside the knots: nothing is read
        const int entry[4] = {q0, q - q0, lo, hi - lo};
        tile.insert(tile.end(), entry, entry + 4);
        span_max = hi - lo > span_max ? hi - lo : span_max;
        q0 = q;
    }
    const int ntiles = (int)(tile.size() / 4);
    cp_spline_plan* p = new (std::nothrow) cp_spline_plan();
    if (!p) return cp::fail(CP_ENOMEM, "cp_spline_plan_create: host allocation failed");
    p->n = n; p->nq = nq; p->bw = bw; p->device = device; p->d_wb = nullptr; p->d_j0 = nullptr; p->d_tile = nullptr;
    p->ntiles = ntiles; p->span_max = span_max;
    p->d_wdense = nullptr;
    p->d_kwin = nullptr;
    p->n_pad = (n + 15) / 16 * 16;
    p->nq_pad = (nq + 63) / 64 * 64;
    // A dense copy (zero outside the bands) serves the matrix-core kernel, which treats the operator as a block-banded GEMM: a tile of 64
    // queries times the window of knots its bands cover.  That does (window / bandwidth) times the multiply-adds of the banded vector kernel
    // but at about five times its rate (the vector kernel reads one LDS word per multiply-add): the matrix cores are the default up to a
    // factor 3.5 (5 until round 6) -- dense operators (quadrature weights, projectors: factor 1), splines between grids of similar density (factor 1.5 for 504 ->
    // 1024 knots of a P(k) table), the wallish2018 splice from 3666 linear knots to 1024 log-spaced ones was one of them at factor 4.x (0.98 against 1.18 ms) until the vector kernel caught up --
    // and a measurement option otherwise (CP_SPLINE_PATH_MFMA).
    const size_t dense_bytes = (size_t)p->n_pad * p->nq_pad * sizeof(double);
    bool dense = keep_dense && dense_bytes <= ((size_t)256 << 20);
    std::vector<double> wd;
    std::vector<int> kwin((size_t)2 * (p->nq_pad / 64), 0);
    if (dense) {
        double work = 0.;
        for (int t = 0; t < p->nq_pad / 64; ++t) {
            int lo = n, hi = 0;
            for (int q = 64 * t; q < 64 * (t + 1) && q < nq; ++q)
                if (j0[q] >= 0) {
                    lo = j0[q] < lo ? j0[q] : lo;
                    hi = j1[q] + 1 > hi ? j1[q] + 1 : hi;
                }
            if (hi <= lo) lo = hi = 0;
            kwin[2 * t] = lo / 16 * 16;
            kwin[2 * t + 1] = (hi + 15) / 16 * 16;
            work += 64. * (kwin[2 * t + 1] - kwin[2 * t]);
        }
        // behind them, the same per tile of 16 queries (linop_mfma_kernel)
        const size_t sub = kwin.size();
        kwin.resize(sub + (size_t)2 * (p->nq_pad / 16), 0);
        double work_sub = 0.;
        for (int t = 0; t < p->nq_pad / 16; ++t) {
            int lo = n, hi = 0;
            for (int q = 16 * t; q < 16 * (t + 1) && q < nq; ++q)
                if (j0[q] >= 0) {
                    lo = j0[q] < lo ? j0[q] : lo;
                    hi = j1[q] + 1 > hi ? j1[q] + 1 : hi;
                }
            if (hi <= lo) lo = hi = 0;
            kwin[sub + 2 * t] = lo / 16 * 16;
            kwin[sub + 2 * t + 1] = (hi + 15) / 16 * 16;
            work_sub += 16. * (kwin[sub + 2 * t + 1] - kwin[sub + 2 * t]);
        }
#ifndef CP_LINOP_SUB_FRACTION
#define CP_LINOP_SUB_FRACTION 0.67      // (tools/ab_linop_sub.sh measures 0: never)
#endif
        p->sub_windows = work_sub <= CP_LINOP_SUB_FRACTION * work;
        // (round 6: the vector kernel requests its rows and weights in batches and runs 1.9-2.7 times faster -- tools/bench_linop.py: 8.9 -> 4.8, 2.6 -> 1.4,
        // 1.06 -> 0.64, 1.19 -> 0.44 ms --, the matrix cores still win at factors 1 - 1.5 (3.9, 1.13, 0.45 ms) and lose at the splice's 4.x (0.50 ms): 3.5)
        p->prefer_dense = n >= 16 && work <= 3.5 * (double)nq * bw;
        // a plan that will run on the vector route keeps no large dense copy (1024 knots -> 16 384 queries: 134 MB of device memory and a
        // blocking upload per plan); small ones stay, for the measurement option and for cp_tables_rows
        const bool valu_fits = (size_t)4 * span_max * sizeof(double) <= 160 * 1024;
        const bool near_the_choice = n >= 16 && work <= 5. * (double)nq * bw;      // (the dense copy stays where the matrix cores were the choice until round 6)
        if (!p->prefer_dense && !near_the_choice && valu_fits && dense_bytes > ((size_t)16 << 20)) dense = false;
    }
    if (dense) {
        wd.assign((size_t)p->n_pad * p->nq_pad, 0.);
        for (int q = 0; q < nq; ++q)
            if (j0[q] >= 0)
                for (int j = j0[q]; j <= j1[q]; ++j) wd[(size_t)q * p->n_pad + j] = w[(size_t)q * n + j];
    } else {
        p->prefer_dense = false;
        p->sub_windows = false;
    }
    // the entries of a row the kernels read: the tiles of the vector route, the windows (multiples of 16 knots) of the matrix-core route
    p->col_first = n;
    p->col_last = 0;
    for (int t = 0; t < ntiles; ++t) {
        p->col_first = std::min(p->col_first, tile[4 * t + 2]);
        p->col_last = std::max(p->col_last, tile[4 * t + 2] + tile[4 * t + 3]);
    }
    if (dense)
        for (size_t t = 0; t < (size_t)(p->nq_pad / 64); ++t)
            if (kwin[2 * t + 1] > kwin[2 * t]) {
                p->col_first = std::min(p->col_first, kwin[2 * t]);
                p->col_last = std::max(p->col_last, kwin[2 * t + 1]);
            }
    p->col_last = std::min(p->col_last, n);
    p->col_first = std::min(p->col_first, p->col_last);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    int status = CP_OK;
    if (prev != device && hipSetDevice(device) != hipSuccess) status = cp::fail(CP_EDEVICE, "cp_spline_plan_create: cannot select device %d", device);
    if (status == CP_OK && (hipMalloc(&p->d_wb, wb.size() * sizeof(double)) != hipSuccess || hipMalloc(&p->d_j0, nq * sizeof(int)) != hipSuccess ||
                            hipMalloc(&p->d_tile, tile.size() * sizeof(int)) != hipSuccess))
        status = cp::fail(CP_ENOMEM, "cp_spline_plan_create: device allocation failed");
    if (status == CP_OK && (hipMemcpy(p->d_wb, wb.data(), wb.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemcpy(p->d_j0, j0.data(), nq * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
                            hipMemcpy(p->d_tile, tile.data(), tile.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess))
        status = cp::fail(CP_EDEVICE, "cp_spline_plan_create: upload failed");
    if (status == CP_OK && dense &&
        (hipMalloc(&p->d_wdense, wd.size() * sizeof(double)) != hipSuccess ||
         hipMemcpy(p->d_wdense, wd.data(), wd.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
         hipMalloc(&p->d_kwin, kwin.size() * sizeof(int)) != hipSuccess ||
         hipMemcpy(p->d_kwin, kwin.data(), kwin.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess))
        status = cp::fail(CP_ENOMEM, "cp_spline_plan_create: cannot upload the dense operator");
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (status != CP_OK) {
        cp_spline_plan_destroy(p);
        return status;
    }
    *out = p;
    return CP_OK;
}

extern "C" int cp_spline_plan_create(cp_spline_plan** out, int n, const double* x, int nq, const double* xq, int bc, int nu, int extrapolate,
                                     int device) {
    if (!out) return cp::fail(CP_EINVAL, "cp_spline_plan_create: null plan pointer");
    *out = nullptr;
    if (nq < 1) return cp::fail(CP_EINVAL, "cp_spline_plan_create: need at least one query point");
    // the operator is staged as a dense (nq, n) matrix on the host: a catalogue or a mesh of queries (1e7 x 600 = 48 GB) must not get here -- a host
    // that runs out of memory is killed, it does not throw (cp_spline_points evaluates splines point by point, any number of points)
    if (n >= 1 && (size_t)nq * (size_t)n > ((size_t)1 << 29))
        return cp::fail(CP_EUNSUPPORTED, "cp_spline_plan_create: %d queries x %d knots: the operator form is for up to 2^29 weights (4 GB on the host); "
                                         "evaluate point by point (cp_spline_points) or in pieces", nq, n);
    std::vector<double> w((size_t)nq * n);
    int st = cp_spline_operator(n, x, nq, xq, bc, nu, extrapolate, w.data(), nullptr);
    if (st != CP_OK) return st;
    return plan_from_dense(out, n, nq, w.data(), device, true);
}

extern "C" int cp_linop_plan_create(cp_spline_plan** out, int n, int nq, const double* w_dense, int device) {
    if (!out) return cp::fail(CP_EINVAL, "cp_linop_plan_create: null plan pointer");
    *out = nullptr;
    if (n < 1 || nq < 1 || !w_dense) return cp::fail(CP_EINVAL, "cp_linop_plan_create: bad arguments");
    if ((long long)n * nq > (1LL << 29))      // (the limit of cp_spline_plan_create: the plan stages bands and padded copies of the operator)
        return cp::fail(CP_EUNSUPPORTED, "cp_linop_plan_create: an operator of %d x %d weights (at most 2^29): apply it in pieces of queries", nq, n);
    return plan_from_dense(out, n, nq, w_dense, device, true);
}

extern "C" int cp_spline_plan_info(const cp_spline_plan* p, int* n, int* nq, int* bandwidth) {
    if (!p) return cp::fail(CP_EINVAL, "cp_spline_plan_info: null plan");
    if (n) *n = p->n;
    if (nq) *nq = p->nq;
    if (bandwidth) *bandwidth = p->bw;
    return CP_OK;
}

extern "C" int cp_spline_plan_columns(const cp_spline_plan* p, int* first, int* count) {
    if (!p) return cp::fail(CP_EINVAL, "cp_spline_plan_columns: null plan");
    if (first) *first = p->col_first;
    if (count) *count = p->col_last - p->col_first;
    return CP_OK;
}

extern "C" int cp_spline_apply_grouped(const cp_spline_plan* p, const double* d_y, double* d_out, long long nrows, int group, int post_op, double scale,
                                       void* stream);

extern "C" int cp_spline_apply(const cp_spline_plan* p, const double* d_y, double* d_out, long long nrows, int post_op, double scale, void* stream) {
    return cp_spline_apply_grouped(p, d_y, d_out, nrows, 0, post_op, scale, stream);
}

extern "C" int cp_spline_apply_grouped(const cp_spline_plan* p, const double* d_y, double* d_out, long long nrows, int group, int post_op, double scale,
                                       void* stream) {
    if (!p) return cp::fail(CP_EINVAL, "cp_spline_apply: null plan");
    if (group < 0 || (group > 0 && nrows % group != 0)) return cp::fail(CP_EINVAL, "cp_spline_apply_grouped: %lld rows are not a whole number of groups of %d", nrows, group);
    if (nrows < 0) return cp::fail(CP_EINVAL, "cp_spline_apply: negative row count");
    if (nrows == 0) return CP_OK;
    if (!d_y || !d_out) return cp::fail(CP_EINVAL, "cp_spline_apply: null device pointer");
    const int path = post_op & (CP_SPLINE_PATH_VALU | CP_SPLINE_PATH_MFMA);
    post_op &= ~(CP_SPLINE_PATH_VALU | CP_SPLINE_PATH_MFMA);
    if (post_op != CP_SPLINE_POST_NONE && post_op != CP_SPLINE_POST_SQRT && post_op != CP_SPLINE_POST_EXP10)
        return cp::fail(CP_EINVAL, "cp_spline_apply: unknown post op %d", post_op);
    if (path == CP_SPLINE_PATH_MFMA && !p->d_wdense) return cp::fail(CP_EINVAL, "cp_spline_apply: the plan holds no dense copy of the operator (more than 256 MB, or more than 16 MB for an operator that runs on the vector route), it has no matrix-core path");
    // the vector kernel stages the knots under a tile of queries in LDS, 4 rows at least: operators wider than that only have the dense route
    const bool valu_fits = (size_t)4 * p->span_max * sizeof(double) <= 160 * 1024;
    if (p->d_wdense && path != CP_SPLINE_PATH_VALU &&
        ((p->prefer_dense && nrows >= 16) || path == CP_SPLINE_PATH_MFMA || !valu_fits)) {   // dense operator: GEMM on the matrix cores
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_apply: cannot select device %d", p->device);
        DenseArgs D;
        D.y = d_y; D.out = d_out; D.nrows = nrows; D.n = p->n; D.nq = p->nq; D.n_pad = p->n_pad; D.nq_pad = p->nq_pad; D.w = p->d_wdense; D.j0 = p->d_j0;
        D.kwin = p->d_kwin; D.post_op = post_op; D.scale = scale; D.group = group;
        const long long items = ((nrows + 16 * LINOP_MT - 1) / (16 * LINOP_MT)) * ((p->nq_pad + 255) / 256);
        const int grid = (int)(items < 256 * 2 ? items : 256 * 2);
        if (p->sub_windows) hipLaunchKernelGGL(linop_mfma_kernel<true>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), D);
        else hipLaunchKernelGGL(linop_mfma_kernel<false>, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), D);
        const hipError_t e = hipGetLastError();
        if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
        if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_apply: launch failed: %s", hipGetErrorString(e));
        return CP_OK;
    }
    // rows per work item: as many as keep the staged knots within 64 KB of LDS (two workgroups per CU), and no more than there are rows
    int rows = 16;
    while (rows > 4 && ((size_t)rows * p->span_max * sizeof(double) > 64 * 1024 || rows / 2 >= nrows)) rows /= 2;
    const size_t lds = (size_t)rows * p->span_max * sizeof(double);
    if (lds > 160 * 1024)
        return cp::fail(CP_EUNSUPPORTED, "cp_spline_apply: one query couples to %d knots, more than the LDS staging buffer holds (max %d)", p->span_max,
                        160 * 1024 / 8 / 4);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_apply: cannot select device %d", p->device);
    Args A;
    A.y = d_y; A.out = d_out; A.nrows = nrows; A.n = p->n; A.nq = p->nq; A.bw = p->bw; A.wb = p->d_wb; A.j0 = p->d_j0;
    A.tile = p->d_tile; A.ntiles = p->ntiles; A.span_max = p->span_max;
    A.post_op = post_op; A.scale = scale; A.group = group;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    const hipError_t e = rows == 16 ? launch_apply<16>(A, lds, hs) : rows == 8 ? launch_apply<8>(A, lds, hs) : launch_apply<4>(A, lds, hs);
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_apply: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}


extern "C" int cp_spline_apply_outer(const cp_spline_plan* p, const double* d_y, const double* d_g, int nz, double* d_out, long long nrows, int post_op,
                                     double scale, void* stream) {
    if (!p) return cp::fail(CP_EINVAL, "cp_spline_apply_outer: null plan");
    if (nrows < 0 || nz < 1) return cp::fail(CP_EINVAL, "cp_spline_apply_outer: bad sizes (nrows=%lld, nz=%d)", nrows, nz);
    if (nrows == 0) return CP_OK;
    if (!d_y || !d_g || !d_out) return cp::fail(CP_EINVAL, "cp_spline_apply_outer: null device pointer");
    if (post_op != CP_SPLINE_POST_NONE && post_op != CP_SPLINE_POST_SQRT) return cp::fail(CP_EINVAL, "cp_spline_apply_outer: unknown post op %d", post_op);
    // rows per work item: the staged knots, interpolated values and factors of the group within 64 KB of LDS (two to three workgroups per CU)
    auto bytes = [&](int rows) { return (size_t)rows * ((size_t)p->span_max + TILE_Q + (size_t)nz) * sizeof(double); };
    int rows = 8;
    while (rows > 2 && (bytes(rows) > 64 * 1024 || rows / 2 >= nrows)) rows /= 2;
    const size_t lds = bytes(rows);
    if (lds > 160 * 1024) return cp::fail(CP_EUNSUPPORTED, "cp_spline_apply_outer: %d knots per query and %d factors per row exceed the LDS staging buffer", p->span_max, nz);
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_apply_outer: cannot select device %d", p->device);
    OuterArgs O;
    Args& A = O.a;
    A.y = d_y; A.out = d_out; A.nrows = nrows; A.n = p->n; A.nq = p->nq; A.bw = p->bw; A.wb = p->d_wb; A.j0 = p->d_j0;
    A.tile = p->d_tile; A.ntiles = p->ntiles; A.span_max = p->span_max;
    A.post_op = post_op; A.scale = scale; A.group = 0;
    O.g = d_g; O.nz = nz;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    const hipError_t e = rows == 8 ? launch_outer<8>(O, lds, hs) : rows == 4 ? launch_outer<4>(O, lds, hs) : launch_outer<2>(O, lds, hs);
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_apply_outer: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}


extern "C" int cp_linop_apply_mid(const cp_spline_plan* p, const double* d_y, double* d_out, long long nbatch, long long ninner, int post_op, double scale,
                                  void* stream) {
    if (!p) return cp::fail(CP_EINVAL, "cp_linop_apply_mid: null plan");
    if (nbatch < 0 || ninner < 0 || ninner > 2147483647LL - 1024) return cp::fail(CP_EINVAL, "cp_linop_apply_mid: bad sizes");
    if (nbatch == 0 || ninner == 0) return CP_OK;
    if (!d_y || !d_out) return cp::fail(CP_EINVAL, "cp_linop_apply_mid: null device pointer");
    if (post_op != CP_SPLINE_POST_NONE && post_op != CP_SPLINE_POST_SQRT && post_op != CP_SPLINE_POST_EXP10)
        return cp::fail(CP_EINVAL, "cp_linop_apply_mid: unknown post op %d", post_op);
    if (!p->d_wdense) return cp::fail(CP_EUNSUPPORTED, "cp_linop_apply_mid: the plan holds no dense operator (create it with cp_linop_plan_create)");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != p->device && hipSetDevice(p->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_linop_apply_mid: cannot select device %d", p->device);
    MidArgs M;
    M.y = d_y; M.out = d_out; M.nbatch = nbatch; M.ninner = ninner; M.n = p->n; M.nq = p->nq; M.n_pad = p->n_pad; M.nq_pad = p->nq_pad;
    M.w = p->d_wdense; M.j0 = p->d_j0; M.post_op = post_op; M.scale = scale;
    const long long items = nbatch * ((ninner + 64 * MID_NT - 1) / (64 * MID_NT)) * (p->nq_pad / 64);
    const int grid = (int)(items < 256 * 8 ? items : 256 * 8);
    hipLaunchKernelGGL(linop_mid_mfma_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), M);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != p->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_linop_apply_mid: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}


// ---- batches of (z, k) tables -> rows of P(k, z): both spline operators of a table in ONE kernel ---------------------------------------------
// PowerSpectrumInterpolator2D(k, z, pk)(k_out, z_out) for a batch of tables on shared grids (reference interpolator.py:667-672, jax.py:241-271:
// RectBivariateSpline = spline along k, then spline along z): out[b, zq, q] = f(sum_zi Wz[zq, zi] sum_j Wk[q, j] T[b, zi, j]), f = 10^x for
// tables splined in log10 P.  As two launches (k-spline, then the middle-axis GEMM) the k-splined tables (B, nz, nq) make a round trip through
// HBM -- 2.5 GB written and read for 10 000 tables of 30 x 504 -> 1024, next to 1.2 GB of input and 5.2 GB of result.  Here they never exist:
// a wave computes L = T[b] Wk^T for all (up to 32) input redshifts and 64 output wavenumbers as a block-banded GEMM, and its accumulators ARE the B
// fragments of the second GEMM R = Wz L: in v_mfma_f64_16x16x4_f64 a lane holds D[row = (l >> 4) + 4 r][col = l & 15] in register r, and as a B
// operand supplies B[k = l >> 4][n = l & 15] -- with rows = input redshifts, register r of row tile i is exactly the B fragment of the knots
// z_in = 16 i + 4 r + (l >> 4), the chunk kk = 4 i + r of the second contraction.  No LDS, no shuffles, no intermediate.
namespace {

struct TablesArgs {
    const double* t;     // (nbatch, nzin, n) tables, k fastest
    double* out;         // (nbatch, nzq, nq)
    long long nbatch;
    int n, nq, n_pad, nq_pad, nzin, nzq;
    const double* wk;    // (nq_pad, n_pad) k operator, zero padded
    const int* kwin;     // windows of knots per 64 queries
    const int* j0k;      // (nq) < 0: NaN query
    const double* wz;    // (64, nz_pad) z operator, zero padded
    int nz_pad;          // 16 or 32
    const int* j0z;      // (nzq)
    int post_op;
    double scale;
};

constexpr int TABLES_WSTRIDE = 80;

template <int POST>
__global__ __launch_bounds__(256, 2) void tables_rows_kernel(const TablesArgs A) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;      // (in a scalar register: what follows from it is wave-uniform for the compiler too)
    const int l15 = lane & 15, g = lane >> 4;
    const int nqt = (A.nq_pad + 255) / 256;
    // the z operator in LDS for the launch, knot-major with a skewed stride (the lanes of an A fragment, 16 queries x 4 knots, on different banks):
    // A fragment of chunk kk and row tile mi = wl[(4 kk + (l >> 4)) * TABLES_WSTRIDE + 16 mi + (l & 15)]
    __shared__ double wl[32 * TABLES_WSTRIDE];
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
        const int q = e >> 5, kz = e & 31;
        wl[kz * TABLES_WSTRIDE + q] = kz < A.nz_pad ? A.wz[q * A.nz_pad + kz] : 0.;      // (the plan pads its knots to 16 or 32)
    }
    __syncthreads();
    // output redshifts whose operator row is NaN, among the 16 this lane stores (zq = 16 mi + (l >> 4) + 4 r): one bit each, for the launch
    unsigned nan_z = 0u;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int zq = 16 * mi + g + 4 * r;
            if (zq < A.nzq && A.j0z[zq] < 0) nan_z |= 1u << (4 * mi + r);
        }
    for (long long item = blockIdx.x; item < A.nbatch * nqt; item += gridDim.x) {
        const long long b = item / nqt;
        const int q0 = (int)(item % nqt) * 256 + wave * 64;
        if (q0 >= A.nq_pad) continue;
        // ---- L = T[b] Wk^T: rows = input redshifts (2 tiles of 16), columns = the wave's 64 output wavenumbers ----
        cp_v4d acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = cp_v4d{0., 0., 0., 0.};
        const double* yr[2];
        const double* wr[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = 16 * i + l15;
            row = row < A.nzin ? row : A.nzin - 1;      // rows past the table repeat its last one: Wz is zero there
            yr[i] = A.t + (b * A.nzin + row) * A.n;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) wr[j] = A.wk + (long long)(q0 + 16 * j + l15) * A.n_pad;
        const int klo = A.kwin[2 * (q0 >> 6)], khi = A.kwin[2 * (q0 >> 6) + 1];
        for (int kb = klo; kb < khi; kb += 16) {
            const int k = kb + 4 * g;
            double a[2][4];
            cp_v4d bw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bw[j] = *reinterpret_cast<const cp_v4d*>(wr[j] + k);
            if (kb + 16 <= A.n) {      // (wave-uniform)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[i][m] = yr[i][k + m];
            } else {   // last, partial chunk: Wk is zero there, but the table must not bring in the next row's values (0 x NaN)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int m = 0; m < 4; ++m) a[i][m] = k + m < A.n ? yr[i][k + m] : 0.;
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][m], bw[j][m], acc[i][j], 0, 0, 0);
        }
        // ---- R = Wz L, one column tile at a time (accumulators: 4 tiles of 16 output redshifts), f applied, rows of P(k, z) stored ----
        double* ob = A.out + b * (long long)A.nzq * A.nq;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            cp_v4d r2[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) r2[mi] = cp_v4d{0., 0., 0., 0.};
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
                    r2[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(wl[(4 * kk + g) * TABLES_WSTRIDE + 16 * mi + l15], acc[kk >> 2][jj][kk & 3], r2[mi], 0, 0, 0);
            // the epilogue without divergent control flow: NaN queries by selects, one wave-uniform test for tiles that reach past the arrays
            const bool full = q0 + 16 * (jj + 1) <= A.nq && A.nzq == 64;
            const int q = q0 + 16 * jj + l15;
            const int qc = q < A.nq ? q : A.nq - 1;
            const bool nanq = A.j0k[qc] < 0;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int zq = 16 * mi + g + 4 * r;
                    double v = r2[mi][r] * A.scale;
                    if (POST == CP_SPLINE_POST_SQRT) v = sqrt(v);
                    else if (POST == CP_SPLINE_POST_EXP10) v = exp10_mid(v);
                    v = (nanq || ((nan_z >> (4 * mi + r)) & 1u)) ? __builtin_nan("") : v;
                    if (full || (q < A.nq && zq < A.nzq)) ob[(long long)zq * A.nq + q] = v;
                }
        }
    }
}

// The same two directions with the k direction evaluated instead of multiplied: a cubic spline at a query is A y_j + B y_{j+1} + wM0 M_j + wM1 M_{j+1}
// with M the second derivatives of the spline at its knots -- which belong to the table, not to the queries (cp_spline_rows_second_derivatives:
// one tridiagonal solve per table row, kept by the caller as long as the table lives, like the coefficients scipy's RectBivariateSpline computes
// when it is built).  The lane that holds element (row z = (l >> 4) + 4 r + 16 i, column q = q0 + 16 j + (l & 15)) of the B fragments of the z
// contraction evaluates it from four table entries: 128 multiply-adds per lane and tile of 64 wavenumbers, where the k operator cost 224
// MFMAs over a window of 112 knots -- the matrix cores are left with the z contraction (128 MFMAs per tile).
struct TablesDirectArgs {
    const double* t;     // (nbatch, nzin, n) tables, k fastest
    const double* m;     // (nbatch, nzin, n) their second derivatives along k
    double* out;         // (nbatch, nzq, nq)
    long long nbatch;
    int n, nq, nzin, nzq;
    const int* qj;       // (nq) interval of each output wavenumber, -1: outside the knots -> NaN
    const double* qw;    // (nq, 4)
    const double* wz;    // (64, nz_pad) z operator, zero padded
    int nz_pad;
    const int* j0z;      // (nzq)
    double scale;
};

#ifndef CP_TABLES_WAVES      // waves per SIMD the direct kernel is built for (register budget 512 / CP_TABLES_WAVES)
#define CP_TABLES_WAVES 3
#endif
#ifndef CP_TABLES_PREFETCH   // 1: the table entries of a step are fetched during the step before (64 more registers in flight)
#define CP_TABLES_PREFETCH 0
#endif
#ifndef CP_TABLES_ABLATE      // diagnostic builds (tools/tables_ablate.sh; wrong results): 1 no stores, 2 no exponential, 4 no z contraction, 8 no table loads
#define CP_TABLES_ABLATE 0
#endif

// The grid is a multiple of the number of 256-column tiles of a row, so a workgroup keeps ITS tile for every table it takes: the intervals and weights
// of its 256 wavenumbers sit in LDS for the whole launch, and a wave's work is one flat sequence of steps (table, jj) whose table entries are
// fetched ONE STEP AHEAD -- issued as soon as the splines of the current step are evaluated, in flight during its contraction, exponentials and
// stores (fetched at the top of the step they stalled every wave for a memory round trip: with loads, stores, exponentials or the contraction
// taken out one at a time the kernel lost 0.5, 0.4, 0.2 and 0.45 of its 2.0 ms -- parts that add up do not overlap).
// PAIRS: tables and second derivatives interleaved, (nbatch, nzin, n, 2) = (y_j, M_j): the four entries of a query are 32 contiguous bytes of ONE array, a
// step of the workgroup walks 576 contiguous bytes of a row instead of 288 in each of two (a partly used cache line at either end of every piece:
// 3.25 lines touched for 2.25 needed against 5.5 for 4.5).
template <int POST, bool PAIRS>
__global__ __launch_bounds__(256, CP_TABLES_WAVES) void tables_rows_direct_kernel(const TablesDirectArgs A) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int l15 = lane & 15, g = lane >> 4;
    const int nqt = (A.nq + 255) / 256;
    __shared__ double wl[32 * TABLES_WSTRIDE];
    __shared__ double4 tile_w[256];
    __shared__ int tile_j[256];
    __shared__ double exp_tab[64];
    if (threadIdx.x < 64) exp_tab[threadIdx.x] = cpmath::exp2_table[threadIdx.x];
    const int tile = (int)(blockIdx.x % nqt);
    // (the plan entries of the workgroup requested together, then stored / tested: entry by entry behind a test they were 8 + 16 memory round trips in a row in
    // front of the first step of each of the 2048 workgroups -- tools/isa_waits.py)
    {
        double wv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = (int)threadIdx.x + 256 * i, q = e >> 5, kz = e & 31;
            wv[i] = A.wz[q * A.nz_pad + (kz < A.nz_pad ? kz : 0)];
        }
        const int q = tile * 256 + threadIdx.x, qc = q < A.nq ? q : A.nq - 1;
        tile_w[threadIdx.x] = reinterpret_cast<const double4*>(A.qw)[qc];
        tile_j[threadIdx.x] = A.qj[qc];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = (int)threadIdx.x + 256 * i, qq = e >> 5, kz = e & 31;
            wl[kz * TABLES_WSTRIDE + qq] = kz < A.nz_pad ? wv[i] : 0.;
        }
    }
    __syncthreads();
    unsigned nan_z = 0u;
    {
        int jz[16];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int zq = 16 * mi + g + 4 * r;
                jz[4 * mi + r] = A.j0z[zq < A.nzq ? zq : 0];
            }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int zq = 16 * mi + g + 4 * r;
                if (zq < A.nzq && jz[4 * mi + r] < 0) nan_z |= 1u << (4 * mi + r);
            }
    }
    typedef double v2u __attribute__((ext_vector_type(2), aligned(8)));
    // The four waves of the workgroup take ADJACENT column tiles at every step (wave w: wavenumbers q0 + 64 jj + 16 w ...): together they walk the
    // table rows 256 bytes at a time, two whole cache lines that all four touch within the same few hundred cycles.  (Each wave walking its own
    // 64 wavenumbers used half of a line per step and came back for the other half thousands of cycles later, when it had often left the L2:
    // FETCH_SIZE 4.7 GB for 2.4 GB of tables.)
    const int q0 = tile * 256 + wave * 16;
    if (q0 >= A.nq) return;
    int rowoff[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = 16 * i + g + 4 * r;
            row = row < A.nzin ? row : A.nzin - 1;      // rows past the table repeat its last one: Wz is zero there
            rowoff[i][r] = row * A.n;
        }
    const long long first = blockIdx.x / nqt, stride = gridDim.x / nqt;
    if (first >= A.nbatch) return;
    const long long nsteps = 4 * ((A.nbatch - first + stride - 1) / stride);
    v2u ty[2][4], tm[2][4];
    auto fetch = [&](long long s) {      // the table entries of step s: the two knots around the lane's wavenumber in each of its 8 rows, and their second derivatives
        const long long b = first + (s >> 2) * stride;
        const int slot = 64 * (int)(s & 3) + 16 * wave + l15;
        const int jraw = tile_j[slot];
        const int jq = jraw < 0 ? 0 : jraw;
        if (PAIRS) {
            const double* tb = A.t + 2 * (b * (long long)A.nzin * A.n + jq);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ty[i][r] = *reinterpret_cast<const v2u*>(tb + 2 * rowoff[i][r]);           // (y_j, M_j)
                    tm[i][r] = *reinterpret_cast<const v2u*>(tb + 2 * rowoff[i][r] + 2);       // (y_j+1, M_j+1)
                }
            return;
        }
        const double* tb = A.t + b * (long long)A.nzin * A.n + jq;
        const double* mb = A.m + b * (long long)A.nzin * A.n + jq;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ty[i][r] = *reinterpret_cast<const v2u*>(tb + rowoff[i][r]);
                tm[i][r] = *reinterpret_cast<const v2u*>(mb + rowoff[i][r]);
            }
    };
    if (!(CP_TABLES_ABLATE & 8) && CP_TABLES_PREFETCH) fetch(0);
#pragma unroll 1
    for (long long s = 0; s < nsteps; ++s) {
        if (!(CP_TABLES_ABLATE & 8) && !CP_TABLES_PREFETCH) fetch(s);
        const long long b = first + (s >> 2) * stride;
        const int jj = (int)(s & 3);
        // One column tile (16 wavenumbers: the lane's is q0 + 64 jj + (l & 15), its interval and weights from the plan) at a time: the k splines
        // of the table's rows there, in the layout of the B fragments of the z contraction (2 x 4 registers); R = Wz L on the matrix cores (4 tiles
        // of 16 output redshifts); f applied, rows of P(k, z) stored.
        const int slot = 64 * jj + 16 * wave + l15;
        const int q = q0 + 64 * jj + l15;
        const bool nanq = tile_j[slot] < 0;
        const double4 w = tile_w[slot];
        cp_v4d lt[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (CP_TABLES_ABLATE & 8) {
                    lt[i][r] = w.x * (double)rowoff[i][r] + w.y;
                    continue;
                }
                lt[i][r] = PAIRS ? w.x * ty[i][r].x + w.y * tm[i][r].x + (w.z * ty[i][r].y + w.w * tm[i][r].y)
                                 : w.x * ty[i][r].x + w.y * ty[i][r].y + (w.z * tm[i][r].x + w.w * tm[i][r].y);
            }
        if (s + 1 < nsteps && !(CP_TABLES_ABLATE & 8) && CP_TABLES_PREFETCH) fetch(s + 1);
        cp_v4d r2[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) r2[mi] = cp_v4d{0., 0., 0., 0.};
#pragma unroll
        for (int kk = 0; kk < ((CP_TABLES_ABLATE & 4) ? 1 : 8); ++kk)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
                r2[mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(wl[(4 * kk + g) * TABLES_WSTRIDE + 16 * mi + l15], lt[kk >> 2][kk & 3], r2[mi], 0, 0, 0);
        if (CP_TABLES_ABLATE & 4)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) r2[mi][r] += lt[mi & 1][r];
        const bool full = q0 + 64 * jj + 16 <= A.nq && A.nzq == 64;
        double* ob = A.out + b * (long long)A.nzq * A.nq;
        if (POST == CP_SPLINE_POST_EXP10 && full && A.nq <= (1 << 20) && !(CP_TABLES_ABLATE & 2)) {
            // The tile lies inside the arrays: when, besides, no lane holds a NaN query / redshift or an exponent beyond +-300 (a wave-uniform test), the
            // sixteen exponentials take the table-driven form and the stores neither a select nor a predicate.  The epilogue was 54 instructions per
            // element (860 of the 1 200 a step issues; the kernel is bound by instruction issue: 625 steps per SIMD x 1 200 x 4 cycles + the
            // contraction = the 2.0 ms it took).
            double amax = 0., sum = 0.;      // (fmax drops a NaN operand: the sum catches them)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    amax = fmax(amax, fabs(r2[mi][r]));
                    sum += r2[mi][r];
                }
            const bool plain = !nanq && nan_z == 0u && amax * fabs(A.scale) < 300. && sum == sum;
            if (__builtin_amdgcn_ballot_w64(!plain) == 0ull) {
                char* base = reinterpret_cast<char*>(ob);                                   // (wave-uniform)
                const unsigned voff = (unsigned)((g * A.nq + q) * 8);
                const unsigned rstride = (unsigned)A.nq * 8u;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double v = cpmath::exp10_tab(r2[mi][r] * A.scale, exp_tab);
                        if (!(CP_TABLES_ABLATE & 1) || v == 12345.678)
                            __builtin_nontemporal_store(v, reinterpret_cast<double*>(base + (size_t)(16 * mi + 4 * r) * rstride + voff));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int zq = 16 * mi + g + 4 * r;
                double v = r2[mi][r] * A.scale;
                if (POST == CP_SPLINE_POST_SQRT) v = sqrt(v);
                else if (POST == CP_SPLINE_POST_EXP10 && !(CP_TABLES_ABLATE & 2)) v = exp10_mid(v);
                v = (nanq || ((nan_z >> (4 * mi + r)) & 1u)) ? __builtin_nan("") : v;
                if ((CP_TABLES_ABLATE & 1) && v != 12345.678) continue;
                // written once, read by the next kernel from memory: the non-temporal policy keeps these 5.2 GB (config 3B) from evicting the table
                // lines the wave comes back to at its next column tile (FETCH_SIZE of this kernel: 5.8 GB with plain stores for 2.4 GB of tables)
                if (full || (q < A.nq && zq < A.nzq)) __builtin_nontemporal_store(v, ob + zq * A.nq + q);      // (at most 64 rows of nq: 32-bit)
            }
            __builtin_amdgcn_sched_barrier(0);      // (four exponentials in flight, not sixteen: their temporaries set the register count)
        }
    }
}

}  // namespace

extern "C" int cp_tables_rows_available(const cp_spline_plan* kplan, const cp_spline_plan* zplan) {
    return kplan && zplan && kplan->d_wdense && kplan->d_kwin && zplan->d_wdense && zplan->n <= 32 && zplan->nq <= 64 && zplan->n_pad <= 32 &&
           zplan->nq_pad == 64 && kplan->device == zplan->device;
}

extern "C" int cp_tables_rows(const cp_spline_plan* kplan, const cp_spline_plan* zplan, const double* d_tables, double* d_out, long long nbatch, int post_op,
                              double scale, void* stream) {
    if (!kplan || !zplan) return cp::fail(CP_EINVAL, "cp_tables_rows: null plan");
    if (nbatch < 0) return cp::fail(CP_EINVAL, "cp_tables_rows: negative batch");
    if (nbatch == 0) return CP_OK;
    if (!d_tables || !d_out) return cp::fail(CP_EINVAL, "cp_tables_rows: null device pointer");
    if (post_op < CP_SPLINE_POST_NONE || post_op > CP_SPLINE_POST_EXP10) return cp::fail(CP_EINVAL, "cp_tables_rows: unknown post op %d", post_op);
    if (!cp_tables_rows_available(kplan, zplan))
        return cp::fail(CP_EUNSUPPORTED, "cp_tables_rows: needs a k operator with a dense copy and a z operator of at most 32 knots and 64 queries");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != kplan->device && hipSetDevice(kplan->device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_tables_rows: cannot select device %d", kplan->device);
    TablesArgs T;
    T.t = d_tables; T.out = d_out; T.nbatch = nbatch;
    T.n = kplan->n; T.nq = kplan->nq; T.n_pad = kplan->n_pad; T.nq_pad = kplan->nq_pad; T.nzin = zplan->n; T.nzq = zplan->nq;
    T.wk = kplan->d_wdense; T.kwin = kplan->d_kwin; T.j0k = kplan->d_j0; T.wz = zplan->d_wdense; T.nz_pad = zplan->n_pad; T.j0z = zplan->d_j0;
    T.post_op = post_op; T.scale = scale;
    const long long items = nbatch * ((kplan->nq_pad + 255) / 256);
    const int grid = (int)(items < 256 * 2 ? items : 256 * 2);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    if (post_op == CP_SPLINE_POST_EXP10) hipLaunchKernelGGL(tables_rows_kernel<CP_SPLINE_POST_EXP10>, dim3(grid), dim3(256), 0, hs, T);
    else if (post_op == CP_SPLINE_POST_SQRT) hipLaunchKernelGGL(tables_rows_kernel<CP_SPLINE_POST_SQRT>, dim3(grid), dim3(256), 0, hs, T);
    else hipLaunchKernelGGL(tables_rows_kernel<CP_SPLINE_POST_NONE>, dim3(grid), dim3(256), 0, hs, T);
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != kplan->device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_tables_rows: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

extern "C" int cp_tables_rows_direct(const cp_spline_rows_plan* kplan, const cp_spline_plan* zplan, const double* d_tables, const double* d_m, double* d_out,
                                     long long nbatch, int post_op, double scale, void* stream) {
    if (!kplan || !zplan) return cp::fail(CP_EINVAL, "cp_tables_rows_direct: null plan");
    if (nbatch < 0) return cp::fail(CP_EINVAL, "cp_tables_rows_direct: negative batch");
    if (nbatch == 0) return CP_OK;
    if (!d_tables || !d_out) return cp::fail(CP_EINVAL, "cp_tables_rows_direct: null device pointer");      // (d_m null: d_tables holds (y, M) pairs)
    if (post_op < CP_SPLINE_POST_NONE || post_op > CP_SPLINE_POST_EXP10) return cp::fail(CP_EINVAL, "cp_tables_rows_direct: unknown post op %d", post_op);
    cp_spline_rows_view kv;
    if (!cp_spline_rows_plan_view(kplan, &kv)) return cp::fail(CP_EINVAL, "cp_tables_rows_direct: bad k plan");
    if (kv.first_knot != 0 || kv.nknots != kv.n) return cp::fail(CP_EUNSUPPORTED, "cp_tables_rows_direct: the output wavenumbers must span the knots of the tables");
    if ((long long)kv.nq * 64 >= (1LL << 31) || (long long)kv.n * 64 >= (1LL << 31))
        return cp::fail(CP_EUNSUPPORTED, "cp_tables_rows_direct: %d wavenumbers (%d knots) per row exceed the 32-bit offsets inside a table", kv.nq, kv.n);
    if (!(zplan->d_wdense && zplan->n <= 32 && zplan->nq <= 64 && zplan->n_pad <= 32 && zplan->nq_pad == 64 && zplan->device == kv.device))
        return cp::fail(CP_EUNSUPPORTED, "cp_tables_rows_direct: needs a z operator of at most 32 knots and 64 queries on the device of the k plan");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != kv.device && hipSetDevice(kv.device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_tables_rows_direct: cannot select device %d", kv.device);
    TablesDirectArgs T;
    T.t = d_tables; T.m = d_m; T.out = d_out; T.nbatch = nbatch;
    T.n = kv.n; T.nq = kv.nq; T.nzin = zplan->n; T.nzq = zplan->nq;
    T.qj = kv.d_qj; T.qw = kv.d_qw; T.wz = zplan->d_wdense; T.nz_pad = zplan->n_pad; T.j0z = zplan->d_j0; T.scale = scale;
    const int nqt = (kv.nq + 255) / 256;
    const long long items = nbatch * nqt;
    const int grid = (int)(items < 256 * 8 ? items : (256 * 8 / nqt > 0 ? (256 * 8 / nqt) * nqt : nqt));      // a multiple of the tiles of a row: a workgroup keeps its tile
    hipStream_t hs = static_cast<hipStream_t>(stream);
    auto launch = [&](auto pairs) {
        constexpr bool PAIRS = decltype(pairs)::value;
        if (post_op == CP_SPLINE_POST_EXP10) hipLaunchKernelGGL((tables_rows_direct_kernel<CP_SPLINE_POST_EXP10, PAIRS>), dim3(grid), dim3(256), 0, hs, T);
        else if (post_op == CP_SPLINE_POST_SQRT) hipLaunchKernelGGL((tables_rows_direct_kernel<CP_SPLINE_POST_SQRT, PAIRS>), dim3(grid), dim3(256), 0, hs, T);
        else hipLaunchKernelGGL((tables_rows_direct_kernel<CP_SPLINE_POST_NONE, PAIRS>), dim3(grid), dim3(256), 0, hs, T);
    };
    if (d_m) launch(std::false_type{});
    else launch(std::true_type{});
    const hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != kv.device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_tables_rows_direct: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}

// ---- clamped cubic spline through uniformly spaced knots with one run of knots removed -------------------------------------
// wallish2018 (reference bao_filter.py:387-405): per column the DST coefficients y_i (positions x_i = i + 1) are multiplied by
// x^2, the knots of a data-dependent box [a, b] are dropped, a clamped CubicSpline through the remaining knots is evaluated at ALL
// positions and divided by x^2.  At a kept knot the spline returns the datum, so only the box interior is interpolated, and that
// needs just the knot derivatives at the two knots bounding the gap.  The tridiagonal system is eliminated from both sides
// towards the gap (two-sided elimination, no storage).  Its off-diagonal coupling decays as (2 - sqrt 3)^distance = 0.268^d, so
// the sweeps start WINDOW = 64 knots away from the gap (true clamped end conditions if the array end is closer, a centred
// difference otherwise; the error of that start value is damped by 0.268^64 ~ 1e-37).
namespace {

constexpr int GAP_WINDOW = 64;

__global__ __launch_bounds__(64) void gap_spline_kernel(const double* y, const int* __restrict__ box, double* out, long long ncol, int n) {
    const long long col = blockIdx.x;
    if (col >= ncol) return;
    const double* yc = y + col * n;
    double* oc = out + col * n;
    // out may be y itself (in place): only the box is rewritten, and nothing inside the box is read -- the copy of the kept knots, two thirds of
    // this kernel's time for the 2048-knot sequences of the filter, is then not made at all
    if (out != y)
        for (int i = threadIdx.x; i < n; i += 64) oc[i] = yc[i];
    const int a = box[2 * col], b = box[2 * col + 1];
    if (a < 1 || b > n - 2 || b < a) return;  // nothing removed (or an invalid box): identity
    const int L = a - 1, R = b + 1;
    const double g = (double)(R - L);
    auto z = [&](int i) { const double x = (double)(i + 1); return yc[i] * (x * x); };
    // forward sweep up to L: s_L + cpL s_R = dpL
    double cp, dp;
    const int i0 = L - GAP_WINDOW > 0 ? L - GAP_WINDOW : 0;
    if (i0 == 0) {
        cp = 0.; dp = 0.;  // clamped: s_0 = 0
    } else {
        cp = 0.; dp = 0.5 * (z(i0 + 1) - z(i0 - 1));
    }
    double cpL, dpL;
    if (L == i0) {
        cpL = cp; dpL = dp;
    } else {
        for (int i = i0 + 1; i < L; ++i) {
            const double d = 3. * (z(i + 1) - z(i - 1));
            const double den = 4. - cp;
            cp = 1. / den;
            dp = (d - dp) / den;
        }
        const double d = 3. * (g * (z(L) - z(L - 1)) + (z(R) - z(L)) / g);
        const double den = 2. * (1. + g) - g * cp;
        cpL = 1. / den;
        dpL = (d - g * dp) / den;
    }
    // backward sweep down to R: s_R + bqR s_L = dqR
    double bq, dq;
    const int i1 = R + GAP_WINDOW < n - 1 ? R + GAP_WINDOW : n - 1;
    if (i1 == n - 1) {
        bq = 0.; dq = 0.;  // clamped: s_{n-1} = 0
    } else {
        bq = 0.; dq = 0.5 * (z(i1 + 1) - z(i1 - 1));
    }
    double bqR, dqR;
    if (R == i1) {
        bqR = bq; dqR = dq;
    } else {
        for (int i = i1 - 1; i > R; --i) {
            const double d = 3. * (z(i + 1) - z(i - 1));
            const double den = 4. - bq;
            bq = 1. / den;
            dq = (d - dq) / den;
        }
        const double d = 3. * ((z(R) - z(L)) / g + g * (z(R + 1) - z(R)));
        const double den = 2. * (g + 1.) - g * bq;
        bqR = 1. / den;
        dqR = (d - g * dq) / den;
    }
    const double sL = (dpL - cpL * dqR) / (1. - cpL * bqR);
    const double sR = dqR - bqR * sL;
    const double zL = z(L), zR = z(R);
    const double slope = (zR - zL) / g;
    const double tt = (sL + sR - 2. * slope) / g;
    const double c3 = tt / g, c2 = (slope - sL) / g - tt;
    for (int i = a + threadIdx.x; i <= b; i += 64) {
        const double u = (double)(i - L), x = (double)(i + 1);
        oc[i] = (zL + u * (sL + u * (c2 + u * c3))) / (x * x);
    }
}

}  // namespace

extern "C" int cp_gap_spline(const double* d_y, const int* d_box, double* d_out, long long ncol, int n, int device, void* stream) {
    if (ncol < 0 || n < 4) return cp::fail(CP_EINVAL, "cp_gap_spline: bad sizes");
    if (ncol == 0) return CP_OK;
    if (!d_y || !d_box || !d_out) return cp::fail(CP_EINVAL, "cp_gap_spline: null device pointer");
    if (ncol > 2147483647LL) return cp::fail(CP_EUNSUPPORTED, "cp_gap_spline: too many columns for one launch");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_gap_spline: cannot select device %d", device);
    hipLaunchKernelGGL(gap_spline_kernel, dim3((unsigned)ncol), dim3(64), 0, static_cast<hipStream_t>(stream), d_y, d_box, d_out, ncol, n);
    hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_gap_spline: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}


// ---- the box of knots wallish2018 removes: between the two maxima of the second derivative (reference bao_filter.py:390-394) ----
// argmax over [margin_first, n - margin_first), then the argmax over [argmax + margin_second, n - margin_first): first index of the maximum,
// NaN counting as the largest value, as numpy / torch argmax do; an empty second range gives index 0 (the arg-max of an all -inf row).
// One wave per column, one pass over its second derivatives (the torch version filled and masked a copy of the whole array).
namespace {

__device__ __forceinline__ void argmax_merge(double& v, int& i, double ov, int oi) {
    const bool take = (ov > v && !(v != v)) || (ov != ov && !(v != v)) || (((ov == v) || (ov != ov && v != v)) && oi < i);
    if (take) {
        v = ov;
        i = oi;
    }
}

__device__ __forceinline__ int wave_argmax(const double* __restrict__ row, int lo, int hi, int lane) {
    double v = -__builtin_inf();
    int idx = 0x7fffffff;
    for (int j = lo + lane; j < hi; j += 64) argmax_merge(v, idx, row[j], j);   // ascending j: ties keep the first
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        argmax_merge(v, idx, ov, oi);
    }
    return idx == 0x7fffffff ? 0 : idx;
}

__global__ __launch_bounds__(256) void wallish_box_kernel(const double* __restrict__ dd, long long ncol, int n, int margin_first, int margin_second,
                                                          int off0, int off1, int* __restrict__ box) {
    const long long col = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= ncol) return;
    const int lane = threadIdx.x & 63;
    const double* row = dd + col * n;
    const int first = wave_argmax(row, margin_first, n - margin_first, lane);
    const int second = wave_argmax(row, first + margin_second, n - margin_first, lane);
    if (lane == 0) {
        box[2 * col] = first + off0;
        box[2 * col + 1] = second + off1;
    }
}

}  // namespace

extern "C" int cp_wallish_box(const double* d_dd, long long ncol, int n, int margin_first, int margin_second, int offset_first, int offset_second,
                              int* d_box, int device, void* stream) {
    if (ncol < 0 || n < 1 || margin_first < 0 || 2 * margin_first >= n) return cp::fail(CP_EINVAL, "cp_wallish_box: bad sizes");
    if (ncol == 0) return CP_OK;
    if (!d_dd || !d_box) return cp::fail(CP_EINVAL, "cp_wallish_box: null device pointer");
    if ((ncol + 3) / 4 > 2147483647LL) return cp::fail(CP_EUNSUPPORTED, "cp_wallish_box: too many columns for one launch");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_wallish_box: cannot select device %d", device);
    hipLaunchKernelGGL(wallish_box_kernel, dim3((unsigned)((ncol + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), d_dd, ncol, n, margin_first,
                       margin_second, offset_first, offset_second, d_box);
    hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_wallish_box: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}


// ---- natural cubic spline per column with per-column knots --------------------------------------------------------------------
// brieden2022 re-samples its smooth spectrum through the input interpolator cloned on the knots k_fid / rescale (reference
// bao_filter.py:503-509): with one rescale per column (batches of cosmologies) the knots differ per column and no fixed operator
// exists.  One thread per column, arrays stored knot-major (n, ncol) so that lanes (columns) are coalesced: a Thomas sweep whose
// modified coefficients go to a global scratch, then the back substitution with the (shared, ascending) queries evaluated on the
// fly from the top interval down.  Queries outside a column's knots give NaN (Interpolator1D, jax.py:200).
namespace {

// Each column is cut in `parts` runs of intervals, a thread per (part, column): the system of a natural spline is diagonally dominant, its
// elimination forgets where it started by a factor <= 0.27 per knot, so a part eliminates from COL_HALO knots below its run to COL_HALO knots
// above it -- treating those two knots as if the spline ended there -- and has its own slopes to 1e-18.  Four times the threads (the kernel
// waits on memory: a wave per CU could not hide it) on chains a third as long.
constexpr int COL_HALO = 32;

__host__ __device__ inline int column_parts(int n) {
    const int p = (n - 1) / 80;
    return p < 1 ? 1 : (p > 8 ? 8 : p);
}
__host__ __device__ inline int column_run(int n) { return (n - 1 + column_parts(n) - 1) / column_parts(n); }      // intervals per part
__host__ __device__ inline int column_span(int n) { return column_run(n) + 2 * COL_HALO + 2; }                       // knots a part eliminates, at most

__global__ __launch_bounds__(64) void column_spline_kernel(const double* __restrict__ xk, const double* __restrict__ yk, long long ncol, int n,
                                                          const double* __restrict__ xq, int nq, double* __restrict__ out,
                                                          double* __restrict__ scratch) {
    const long long ncol_r = (ncol + 63) / 64 * 64;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int part = (int)(g / ncol_r);
    const long long c = g - part * ncol_r;
    if (c >= ncol) return;
    const int run = column_run(n), span = column_span(n);
    const int a = part * run, b = a + run < n - 1 ? a + run : n - 1;      // the part evaluates the intervals [a, b)
    if (a >= b) return;
    const int f0 = a - COL_HALO > 0 ? a - COL_HALO : 0, f1 = b + COL_HALO < n - 1 ? b + COL_HALO : n - 1;      // ... and eliminates the knots f0 .. f1
    double* cp = scratch + (long long)part * 2 * span * ncol + c;      // (span, ncol) of this part, local index = knot - f0
    double* dp = cp + (long long)span * ncol;
    auto X = [&](int i) { return xk[(long long)i * ncol + c]; };
    auto Y = [&](int i) { return yk[(long long)i * ncol + c]; };
    // forward elimination of the system for the knot derivatives (scipy CubicSpline, bc_type='natural'), the first knot as a natural end
    double x0 = X(f0), x1 = X(f0 + 1), y0 = Y(f0), y1 = Y(f0 + 1);
    double dxm = x1 - x0, slm = (y1 - y0) / dxm;   // left interval of knot f0 + 1
    double cprev = 0.5, dprev = 3. * slm / 2.;     // row 0: 2 dx0 s0 + dx0 s1 = 3 dx0 slope0
    cp[0] = cprev;
    dp[0] = dprev;
    // the sweeps are chains of dependent divisions fed by one knot per step: the knots of the next CHUNK steps are fetched together, ahead of
    // the arithmetic
    constexpr int CHUNK = 8;
    for (int i0 = f0 + 1; i0 < f1; i0 += CHUNK) {
        double xs[CHUNK], ys[CHUNK];
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            const int j = i0 + u + 1 < f1 ? i0 + u + 1 : f1;
            xs[u] = X(j);
            ys[u] = Y(j);
        }
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            const int i = i0 + u;
            if (i < f1) {
                const double x2 = xs[u], y2 = ys[u];
                const double dxp = x2 - x1, slp = (y2 - y1) / dxp;
                // dxp s_{i-1} + 2 (dxm + dxp) s_i + dxm s_{i+1} = 3 (dxp slm + dxm slp)
                const double den = 2. * (dxm + dxp) - dxp * cprev;
                cprev = dxm / den;
                dprev = (3. * (dxp * slm + dxm * slp) - dxp * dprev) / den;
                cp[(long long)(i - f0) * ncol] = cprev;
                dp[(long long)(i - f0) * ncol] = dprev;
                x1 = x2; y1 = y2; dxm = dxp; slm = slp;
            }
        }
    }
    // last row (knot f1, the last knot or a natural end in its place): dx s_{f1-1} + 2 dx s_{f1} = 3 dx slope
    double s_hi = (3. * slm - dprev) / (2. - cprev);
    const double nan = __builtin_nan("");
    int iq = nq - 1;
    const bool top = b == n - 1, bottom = a == 0;
    if (top) {
        const double xtop = X(n - 1);
        while (iq >= 0 && xq[iq] > xtop) out[(long long)iq-- * ncol + c] = nan;
    } else {      // queries at or above knot b belong to the parts above
        const double xb = X(b);
        while (iq >= 0 && xq[iq] >= xb) --iq;
    }
    double xh = x1, yh = y1;   // upper knot of the current interval (knot f1)
    for (int i0 = f1 - 1; i0 >= a; i0 -= CHUNK) {
        double xs[CHUNK], ys[CHUNK], cs[CHUNK], ds[CHUNK];
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            const int j = i0 - u >= a ? i0 - u : a;
            xs[u] = X(j);
            ys[u] = Y(j);
            cs[u] = cp[(long long)(j - f0) * ncol];
            ds[u] = dp[(long long)(j - f0) * ncol];
        }
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            if (i0 - u < a) continue;
            const double xl = xs[u], yl = ys[u];
            const double s_lo = ds[u] - cs[u] * s_hi;
            if (i0 - u < b) {      // an interval of this part: its queries
                const double h = xh - xl, slope = (yh - yl) / h;
                const double tt = (s_lo + s_hi - 2. * slope) / h;
                const double c3 = tt / h, c2 = (slope - s_lo) / h - tt;
                while (iq >= 0 && xq[iq] >= xl) {
                    const double v = xq[iq] - xl;
                    out[(long long)iq * ncol + c] = yl + v * (s_lo + v * (c2 + v * c3));
                    --iq;
                }
            }
            xh = xl; yh = yl; s_hi = s_lo;
        }
    }
    if (bottom)
        while (iq >= 0) out[(long long)iq-- * ncol + c] = nan;
}

}  // namespace

extern "C" long long cp_spline_columns_scratch_doubles(long long ncol, int n) {
    if (ncol < 0 || n < 3) return -1;
    return 2LL * column_parts(n) * column_span(n) * ncol;
}

extern "C" int cp_spline_columns(const double* d_xk, const double* d_yk, long long ncol, int n, const double* d_xq, int nq, double* d_out,
                                 double* d_scratch, int device, void* stream) {
    if (ncol < 0 || n < 3 || nq < 0) return cp::fail(CP_EINVAL, "cp_spline_columns: bad sizes");
    if (ncol == 0 || nq == 0) return CP_OK;
    if (!d_xk || !d_yk || !d_xq || !d_out || !d_scratch) return cp::fail(CP_EINVAL, "cp_spline_columns: null device pointer");
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device && hipSetDevice(device) != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_columns: cannot select device %d", device);
    hipLaunchKernelGGL(column_spline_kernel, dim3((unsigned)((ncol + 63) / 64 * column_parts(n))), dim3(64), 0, static_cast<hipStream_t>(stream), d_xk, d_yk,
                       ncol, n, d_xq, nq, d_out, d_scratch);
    hipError_t e = hipGetLastError();
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    if (e != hipSuccess) return cp::fail(CP_EDEVICE, "cp_spline_columns: launch failed: %s", hipGetErrorString(e));
    return CP_OK;
}
