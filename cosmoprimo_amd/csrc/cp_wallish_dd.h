// cp_wallish_dd.h -- wallish2018: second derivatives of the clamped spline through a sequence of sine-transform coefficients, the box between
// their two maxima, and the removal of that box, as wave-level device functions (reference bao_filter.py:377-405).  Used by the stand-alone kernel
// (cp_bao.hip: wallish_dd_box_kernel, sequences read from memory) and by the forward transform that produces the sequences itself (cp_dst.hip:
// dst_generate_kernel, the four sequences of a pair of vectors one per wave of its workgroup).
//
// The clamped cubic spline through (x = 1 .. n, y) has second derivatives M with  2 M_0 + M_1 = 6 (y_1 - y_0),  M_{i-1} + 4 M_i + M_{i+1} =
// 6 (y_{i+1} - 2 y_i + y_{i-1}),  M_{n-2} + 2 M_{n-1} = -6 (y_{n-1} - y_{n-2}):  what scipy's CubicSpline(bc_type='clamped')(x, nu=2) returns
// (bao_filter.py:377-382).  A wave holds a sequence in LDS and runs the elimination itself: lane l owns the knots [S l, S l + S), S = n / 64.
// The modified diagonal c_i = 1 / (4 - c_{i-1}) does not depend on the data and converges to 2 - sqrt(3) within 40 knots; the forward recurrence
// d_i = (rhs_i - d_{i-1}) c_i forgets its start at that rate (0.268 per knot), so a lane starts DD_HALO = 32 knots to the left of its own with
// d = 0 (5e-19 of the starting error is left when it reaches them; the first two lanes start at knot 0 and are exact), and the back
// substitution M_i = d_i - c_i M_{i+1} the same from the right.  64 + 64 dependent steps per sequence instead of 4096.
#pragma once
#include <hip/hip_runtime.h>

#include "cp_internal.h"
#include "cp_math.h"

namespace cpdd {

using cp::wave_lds_phase;

constexpr int DD_HALO = 32;
constexpr int DD_GAP_WINDOW = 64;      // knots on either side of the box from which its end slopes are eliminated (gap_spline_kernel: GAP_WINDOW)
constexpr double DD_CINF = 0.26794919243112270647;      // 2 - sqrt(3)
constexpr double DD_CLAST = 1. / (2. - DD_CINF);         // the last row has diagonal 2
constexpr int DD_NTAB = 40;                              // c_i equals its limit to the last bit from knot 30 on
struct DdTable {
    double c[DD_NTAB];
    constexpr DdTable() : c() {
        double v = 0.5;
        c[0] = v;
        for (int i = 1; i < DD_NTAB; ++i) {
            v = 1. / (4. - v);
            c[i] = v;
        }
    }
};
struct GapTable {      // the same recurrence started from 0 (gap_spline_kernel starts its eliminations inside the sequence with c = 0)
    double c[DD_NTAB];
    constexpr GapTable() : c() {
        double v = 0.;
        c[0] = v;
        for (int i = 1; i < DD_NTAB; ++i) {
            v = 1. / (4. - v);
            c[i] = v;
        }
    }
};

// the two tables into LDS (2 DD_NTAB doubles at `tabs`), by threads 0 .. 39 and 64 .. 103 of a workgroup (the recurrences run in IEEE arithmetic: the
// values of the constexpr tables above, which the host uses); a barrier follows at the caller's
__device__ __forceinline__ void fill_tables(double* tabs) {
    const int t = threadIdx.x;
    const bool gap = t >= 64;
    const int n = gap ? t - 64 : t;
    if (n < DD_NTAB && t < 64 + DD_NTAB) {
        double v = gap ? 0. : 0.5;
        for (int i = 0; i < n; ++i) v = 1. / (4. - v);
        tabs[(gap ? DD_NTAB : 0) + n] = v;
    }
}

__device__ __forceinline__ void argmax_merge(double& v, int& i, double ov, int oi) {      // first index of the maximum, NaN counts as largest (numpy)
    const bool take = (ov > v && !(v != v)) || (ov != ov && !(v != v)) || (((ov == v) || (ov != ov && v != v)) && oi < i);
    if (take) {
        v = ov;
        i = oi;
    }
}

__device__ __forceinline__ int wave_merge(double v, int idx) {
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(idx, off);
        argmax_merge(v, idx, ov, oi);
    }
    return idx == 0x7fffffff ? 0 : idx;
}

// Where knot i of a sequence sits in its LDS buffer.  A lane walks S consecutive knots, so consecutive lanes are S doubles apart: without a
// permutation all of them would sit on one bank.  Padded: i + i / S (n + 64 doubles per sequence); Xor32 (S = 32 only): the low five bits XOR-ed
// with the segment number -- a permutation inside blocks of 32, exactly n doubles per sequence (four sequences of 2048 then fill the 64 KB data
// region of a 4096-point transform).
template <int S>
struct PaddedLayout {
    static __device__ __forceinline__ int at(int i) { return i + i / S; }
};
struct Xor32Layout {
    static __device__ __forceinline__ int at(int i) { return i ^ ((i >> 5) & 31); }
};

// S = n / 64 knots per lane; buf: the sequence in LDS under the layout LAY (every lane's LDS writes fenced by the caller), its second derivatives M
// on exit.  first / second: the arg-max of M over [margin_first, n - margin_first) and over [first + margin_second, n - margin_first)
// (bao_filter.py:390-394, before the offsets).
template <int S, class LAY = PaddedLayout<S>>
__device__ __forceinline__ void second_derivatives_and_box(double* buf, const double* ctab, int lane, int margin_first, int margin_second, int& first,
                                                           int& second) {
    constexpr int N = 64 * S;
    auto at = [&](int i) -> double& { return buf[LAY::at(i)]; };
    auto clamped = [&](int i) { return buf[LAY::at(i < 0 ? 0 : (i > N - 1 ? N - 1 : i))]; };
    auto c_of = [&](int i) { return i >= N - 1 ? DD_CLAST : ctab[i < 0 ? 0 : (i < DD_NTAB ? i : DD_NTAB - 1)]; };
    const int own = S * lane;
    // Both sweeps without a branch: knots read beyond either end repeat the end knot, which makes the right-hand sides of the two clamped rows
    // come out of the general formula (6 ((y_1 - y_0) - (y_0 - y_0)) and 6 ((y_{n-1} - y_{n-1}) - (y_{n-1} - y_{n-2}))) and keeps d = 0 to the
    // left of knot 0.
    {
        const double beyond = clamped(own + S);      // the next segment's first knot, before its owner writes there
        double d = 0., ym = clamped(own - DD_HALO - 1), y0 = clamped(own - DD_HALO);
#pragma unroll 8
        for (int t = 0; t < DD_HALO; ++t) {          // towards the segment: nothing stored
            const int i = own - DD_HALO + t;
            const double yp = clamped(i + 1);
            d = (6. * ((yp - y0) - (y0 - ym)) - d) * c_of(i);
            ym = y0;
            y0 = yp;
        }
        wave_lds_phase();      // every lane is through its run-in (knots of its left neighbours) before those are overwritten by d
#pragma unroll 8
        for (int t = 0; t < S; ++t) {
            const int i = own + t;
            const double yp = t == S - 1 ? beyond : clamped(i + 1);
            d = (6. * ((yp - y0) - (y0 - ym)) - d) * c_of(i);
            at(i) = d;
            ym = y0;
            y0 = yp;
        }
    }
    wave_lds_phase();      // d complete
    double best = -__builtin_inf();
    int best_i = 0x7fffffff;
    {
        double m = 0.;
#pragma unroll 8
        for (int t = 0; t < DD_HALO; ++t) {          // towards the segment from the right (beyond the last knot: M_{n-1} = d_{n-1} again)
            const int i = own + S + DD_HALO - 1 - t;
            const double d = clamped(i);
            m = i >= N - 1 ? d : d - c_of(i) * m;
        }
        wave_lds_phase();      // run-in from the right done before the neighbours' d become M
#pragma unroll 8
        for (int t = 0; t < S; ++t) {
            const int i = own + S - 1 - t;
            const double d = at(i);
            m = i >= N - 1 ? d : d - c_of(i) * m;
            at(i) = m;
            // the lane's own maximum inside [margin_first, n - margin_first), first index on ties, NaN as the largest value (numpy's argmax): the
            // sweep runs towards smaller i, so an equal value replaces the one held
            const bool inside = i >= margin_first && i < N - margin_first;
            if (inside && ((m != m) || (!(best != best) && m >= best))) {
                best = m;
                best_i = i;
            }
        }
    }
    wave_lds_phase();      // M complete
    // arg-max over [margin_first, n - margin_first): the lanes' maxima merged; then over [first + margin_second, n - margin_first): the maxima of
    // the lanes whose segments lie inside it, and the segment that straddles its lower end looked at once more, a knot per lane
    first = wave_merge(best, best_i);
    const int lower = first + margin_second;
    double v2 = own >= lower ? best : -__builtin_inf();
    int i2 = own >= lower ? best_i : 0x7fffffff;
    {
        const int e = lower - lower % S + lane % S;
        if (e >= lower && e < N - margin_first && lane < S) argmax_merge(v2, i2, at(e), e);
    }
    second = wave_merge(v2, i2);
}

// The same for sequences of 2048 knots (S = 32) without an elimination: the knots are uniformly spaced, so the system has constant coefficients and
// its inverse is two geometric tails, M_i = sqrt 3 sum_j p^|i-j| r_j with p = sqrt 3 - 2 and r_j = y_{j+1} - 2 y_j + y_{j-1} -- over the sequence
// continued by REFLECTION about its two end knots, which is what the clamped ends amount to (the spline of the even continuation has no slope at
// the mirror, and its rows there are the clamped rows: 2 M_0 + M_1 = 6 (y_1 - y_0)).  A lane reads its 32 knots (and one on either side) from LDS
// ONCE and runs the anti-causal recursion g_t = r_t + p g_{t+1}, then the causal one in the same registers (r_t = g_t - p g_{t+1}: e_t = g_t +
// p f_{t-1} is M_t / sqrt 3, f_t = e_t - p g_{t+1}), both from zero; what the knots outside its segment add comes from the neighbours' segment
// totals by one DPP shift each way (a segment away the weight is p^32 = 5e-19), and at the two ends from the lane's own sums (the mirror image of
// the causal sum in front of knot 0 is the anti-causal sum at knot 1).  64 register steps and 66 LDS accesses per lane where the elimination walks
// 128 knots through LDS with three accesses each (cp_splice_uniform.h: the same recursions on the uniform stretch of the spliced spline).
typedef int dd_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double dd_from_left(double v) {      // lane l receives lane l - 1's value, lane 0 zero (wave_shr:1)
#if defined(__HIP_DEVICE_COMPILE__)
    dd_v2i w = __builtin_bit_cast(dd_v2i, v);
    w.x = __builtin_amdgcn_update_dpp(0, w.x, 0x138, 0xf, 0xf, true);
    w.y = __builtin_amdgcn_update_dpp(0, w.y, 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, w);
#else
    return v;
#endif
}
__device__ __forceinline__ double dd_from_right(double v) {     // lane l receives lane l + 1's value, lane 63 zero (wave_shl:1)
#if defined(__HIP_DEVICE_COMPILE__)
    dd_v2i w = __builtin_bit_cast(dd_v2i, v);
    w.x = __builtin_amdgcn_update_dpp(0, w.x, 0x130, 0xf, 0xf, true);
    w.y = __builtin_amdgcn_update_dpp(0, w.y, 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double, w);
#else
    return v;
#endif
}
constexpr double dd_ipow(double x, int k) {
    double v = 1.;
    for (int i = 0; i < k; ++i) v *= x;
    return v;
}

// ---- wave reductions by DPP (row operations within 16 lanes, then row_bcast15 / row_bcast31 across the rows: the total arrives in lane 63) ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dd_dpp(double v) {      // lanes the row mask leaves out keep their own value
#if defined(__HIP_DEVICE_COMPILE__)
    dd_v2i w = __builtin_bit_cast(dd_v2i, v);
    w.x = __builtin_amdgcn_update_dpp(w.x, w.x, CTRL, ROW_MASK, 0xf, false);
    w.y = __builtin_amdgcn_update_dpp(w.y, w.y, CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, w);
#else
    return v;
#endif
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dd_dpp(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
#else
    return v;
#endif
}
__device__ __forceinline__ double wave_max(double v) {      // the maximum over the wave, in every lane (no NaN among the values)
    v = fmax(v, dd_dpp<0xB1, 0xf>(v));      // quad_perm [1, 0, 3, 2]
    v = fmax(v, dd_dpp<0x4E, 0xf>(v));      // quad_perm [2, 3, 0, 1]
    v = fmax(v, dd_dpp<0x141, 0xf>(v));     // row_half_mirror
    v = fmax(v, dd_dpp<0x140, 0xf>(v));     // row_mirror
    v = fmax(v, dd_dpp<0x142, 0xa>(v));     // row_bcast15 into rows 1 and 3
    v = fmax(v, dd_dpp<0x143, 0xc>(v));     // row_bcast31 into rows 2 and 3
    return cp::lane_value(v, 63);
}
__device__ __forceinline__ int wave_min(int v) {
    auto mn = [](int a, int b) { return a < b ? a : b; };
    v = mn(v, dd_dpp<0xB1, 0xf>(v));
    v = mn(v, dd_dpp<0x4E, 0xf>(v));
    v = mn(v, dd_dpp<0x141, 0xf>(v));
    v = mn(v, dd_dpp<0x140, 0xf>(v));
    v = mn(v, dd_dpp<0x142, 0xa>(v));
    v = mn(v, dd_dpp<0x143, 0xc>(v));
    return __builtin_amdgcn_readlane(v, 63);
}

// first index of the maximum of the wave's knots i = own + t with lo <= t < hi (own, lo, hi per lane; the range may be empty in a lane); no NaN among
// the values.  0 when the range is empty in every lane (as wave_merge returns for it).
template <int S>
__device__ __forceinline__ int masked_argmax(const double* m, int own, int lo, int hi) {
    const double ninf = -__builtin_inf();
    double b = ninf;
#pragma unroll
    for (int t = 0; t < S; ++t) b = fmax(b, (t >= lo && t < hi) ? m[t] : ninf);
    const double top = wave_max(b);
    int idx = 0x7fffffff;
#pragma unroll
    for (int t = S - 1; t >= 0; --t) idx = (t >= lo && t < hi && m[t] == top) ? own + t : idx;
    idx = wave_min(idx);
    return idx == 0x7fffffff ? 0 : idx;
}

// m[32]: the lane's second derivatives on exit (SCALED: in the units of the spline's; else short of the factor sqrt 3, which no arg-max sees); the
// sequence in LDS is only read.
template <class LAY = PaddedLayout<32>, bool SCALED = true>
__device__ __forceinline__ void second_derivatives_and_box_recursive(const double* buf, int lane, int margin_first, int margin_second, double* m, int& first,
                                                                     int& second) {
    constexpr int S = 32, N = 64 * S;
    constexpr double P = -DD_CINF, SCALE = 1.7320508075688772935;      // 6 kappa = 6 / (2 sqrt 3)
    auto at = [&](int i) { return buf[LAY::at(i)]; };
    const int own = S * lane;
    {
        double ym = at(lane == 0 ? 1 : own - 1), y0 = at(own);            // (the mirror image of knot 1 stands in front of knot 0)
#pragma unroll
        for (int t = 0; t < S; ++t) {
            const double yp = t < S - 1 ? at(own + t + 1) : at(lane == 63 ? N - 2 : own + S);
            m[t] = (yp - y0) - (y0 - ym);
            ym = y0;
            y0 = yp;
        }
    }
#pragma unroll
    for (int t = S - 2; t >= 0; --t) m[t] = fma(P, m[t + 1], m[t]);
    const double g1 = m[1];
    double gin = dd_from_right(m[0]);
    double f = 0., fs2 = 0.;
#pragma unroll
    for (int t = 0; t < S; ++t) {
        const double e = fma(P, f, m[t]);
        f = t + 1 < S ? fma(-P, m[t + 1], e) : e;
        if (t == S - 2) fs2 = f;
        m[t] = e;
    }
    double fin = dd_from_left(f);
    gin = lane == 63 ? fma(dd_ipow(P, S - 1), fin, fs2) : gin;      // the anti-causal sum at the mirror image of knot n - 2
    fin = lane == 0 ? fma(dd_ipow(P, S - 1), gin, g1) : fin;        // the causal sum at the mirror image of knot 1
    {
        double c = gin;
#pragma unroll
        for (int t = S - 1; t >= 0; --t) {
            c *= P;
            m[t] += c;
        }
        c = fin;
#pragma unroll
        for (int t = 0; t < S; ++t) {
            c *= P;
            m[t] += c;
        }
    }
    // arg-max over [margin_first, n - margin_first) and over [first + margin_second, n - margin_first): first index on ties, NaN as the largest value
    // (numpy's argmax).  Sequences without a NaN -- a wave-uniform test on a sum that any NaN (or infinity) poisons -- take masked_argmax: the lane's
    // maximum as a chain of v_max, the wave's by DPP, the index from one pass of equality tests: a third of the instructions of the general walk,
    // which tracks value and index side by side with NaN-aware comparisons (it stays for the sequences that need it).
    double poison = 0.;
#pragma unroll
    for (int t = 0; t < S; ++t) {
        if (SCALED) m[t] *= SCALE;
        poison = fma(m[t], 0., poison);
    }
    if (__builtin_amdgcn_ballot_w64(poison != 0.) == 0) {      // (NaN != 0 holds: the ballot is zero when every lane's sum is a clean zero)
        const int lo1 = margin_first - own, hi1 = N - margin_first - own;
        first = masked_argmax<S>(m, own, lo1, hi1);
        const int lo2 = first + margin_second - own;
        second = masked_argmax<S>(m, own, lo2, hi1);
        return;
    }
    double best = -__builtin_inf();
    int best_i = 0x7fffffff;
#pragma unroll
    for (int t = S - 1; t >= 0; --t) {      // (the walk runs towards smaller i, so an equal value replaces the one held)
        const int i = own + t;
        const bool inside = i >= margin_first && i < N - margin_first;
        if (inside && ((m[t] != m[t]) || (!(best != best) && m[t] >= best))) {
            best = m[t];
            best_i = i;
        }
    }
    first = wave_merge(best, best_i);
    // ... and over [first + margin_second, n - margin_first): the lanes whose knots all lie inside hold their maximum already
    const int lower = first + margin_second;
    double v2 = best;
    int i2 = best_i;
    if (own < lower) {      // (the lanes in front of the range, and the one it starts in: that one looks at its knots again)
        v2 = -__builtin_inf();
        i2 = 0x7fffffff;
        if (own + S > lower) {
#pragma unroll
            for (int t = S - 1; t >= 0; --t) {
                const int i = own + t;
                const bool inside = i >= lower && i < N - margin_first;
                if (inside && ((m[t] != m[t]) || (!(v2 != v2) && m[t] >= v2))) {
                    v2 = m[t];
                    i2 = i;
                }
            }
        }
    }
    second = wave_merge(v2, i2);
}

// The removal of the box [a, b] (cp_gap_spline, bao_filter.py:395-405): the clamped spline through the x^2-weighted coefficients with the knots
// [a, b] left out returns the datum at every kept knot, so only the box is rewritten; its two end slopes come from eliminations started
// DD_GAP_WINDOW knots to either side (the arithmetic of gap_spline_kernel, cp_spline.hip), run by two lanes side by side on values the wave has
// brought into LDS.  `stage(lo, L, R, hi, zl, zr)` fills zl[i - lo] = y_i (i + 1)^2 for lo <= i <= L and zr[i - R] likewise for R <= i <= hi
// (all lanes take part); `seq` is where the rewritten knots go.  Returns false when nothing is removed (an invalid box: the sequence stays).
template <int S, class Stage>
__device__ __forceinline__ bool remove_box(double* buf, const double* gtab, int lane, int a, int b, Stage stage, double* seq) {
    constexpr int N = 64 * S;
    if (a < 1 || b > N - 2 || b < a) return false;
    const int L = a - 1, R = b + 1;
    const double g = (double)(R - L);
    const int i0 = L - DD_GAP_WINDOW > 0 ? L - DD_GAP_WINDOW : 0, i1 = R + DD_GAP_WINDOW < N - 1 ? R + DD_GAP_WINDOW : N - 1;
    const int lo = i0 > 0 ? i0 - 1 : 0, hi = i1 < N - 1 ? i1 + 1 : N - 1;
    double* zl = buf;                              // z(lo .. L)
    double* zr = buf + DD_GAP_WINDOW + 8;          // z(R .. hi)
    stage(lo, L, R, hi, zl, zr);
    wave_lds_phase();      // zl / zr written by all lanes, read by lanes 0 and 1
    auto z = [&](int i) { return i <= L ? zl[i - lo] : zr[i - R]; };
    double r0 = 0., r1 = 0.;
    if (lane == 0) {            // forward sweep up to L: s_L + cpL s_R = dpL
        double cp = 0., dp = i0 == 0 ? 0. : 0.5 * (z(i0 + 1) - z(i0 - 1));      // clamped: s_0 = 0
        if (L != i0) {
            // (1 / (4 - cp) does not depend on the data: 0, 1/4, 4/15, ... from a table, its limit after 40 knots -- no division in the chain)
            double zm = z(i0), z0 = z(i0 + 1);
#pragma unroll 4
            for (int i = i0 + 1; i < L; ++i) {
                const double zp = z(i + 1);
                cp = gtab[i - i0 < DD_NTAB ? i - i0 : DD_NTAB - 1];
                dp = (3. * (zp - zm) - dp) * cp;
                zm = z0;
                z0 = zp;
            }
            const double d = 3. * (g * (z(L) - z(L - 1)) + (z(R) - z(L)) / g);
            const double den = 2. * (1. + g) - g * cp;
            cp = 1. / den;
            dp = (d - g * dp) / den;
        }
        r0 = cp;
        r1 = dp;
    } else if (lane == 1) {     // backward sweep down to R: s_R + bqR s_L = dqR
        double bq = 0., dq = i1 == N - 1 ? 0. : 0.5 * (z(i1 + 1) - z(i1 - 1));  // clamped: s_{n-1} = 0
        if (R != i1) {
            double zp = z(i1), z0 = z(i1 - 1);
#pragma unroll 4
            for (int i = i1 - 1; i > R; --i) {
                const double zm = z(i - 1);
                bq = gtab[i1 - i < DD_NTAB ? i1 - i : DD_NTAB - 1];
                dq = (3. * (zp - zm) - dq) * bq;
                zp = z0;
                z0 = zm;
            }
            const double d = 3. * ((z(R) - z(L)) / g + g * (z(R + 1) - z(R)));
            const double den = 2. * (g + 1.) - g * bq;
            bq = 1. / den;
            dq = (d - g * dq) / den;
        }
        r0 = bq;
        r1 = dq;
    }
    const double cpL = __shfl(r0, 0), dpL = __shfl(r1, 0), bqR = __shfl(r0, 1), dqR = __shfl(r1, 1);
    const double sL = (dpL - cpL * dqR) / (1. - cpL * bqR);
    const double sR = dqR - bqR * sL;
    const double zL = z(L), zR = z(R);
    const double slope = (zR - zL) / g;
    const double tt = (sL + sR - 2. * slope) / g;
    const double c3 = tt / g, c2 = (slope - sL) / g - tt;
    for (int i = a + lane; i <= b; i += 64) {
        const double u = (double)(i - L), x = (double)(i + 1);
        seq[i] = (zL + u * (sL + u * (c2 + u * c3))) / (x * x);
    }
    wave_lds_phase();      // zl / zr read before the buffer is staged over again
    return true;
}

// The same removal with the two eliminations as REDUCTIONS over the wave: a step of either sweep is an affine map of the running right-hand side,
// d <- (3 (z_{i+1} - z_{i-1}) - d) c_i, and affine maps compose -- each lane builds the map of one knot of the window on either side (64 knots
// each), six ordered butterfly steps compose them, where two lanes walked 64 dependent steps each while 62 waited.  The knots are read from the
// sequence in LDS (layout LAY), nothing is staged.
// INPLACE: the rewritten knots also replace the ones in LDS (the fused tail of the filter, cp_dst.hip: wallish_tail_kernel, transforms the sequence
// back from there), and *all_finite tells whether this lane's were finite.
template <int S, class LAY, bool INPLACE = false>
__device__ __forceinline__ bool remove_box_parallel(const double* ybuf, const double* gtab, int lane, int a, int b, double* seq, bool* all_finite = nullptr) {
    constexpr int N = 64 * S;
    static_assert(DD_GAP_WINDOW == 64, "a knot of either window per lane");
    if (all_finite) *all_finite = true;
    if (a < 1 || b > N - 2 || b < a) return false;
    const int L = a - 1, R = b + 1;
    const double g = (double)(R - L), inv_g = cpmath::recip(g);      // (the quotients below as reciprocals: thirteen IEEE divisions in a row per sequence were a quarter of the kernel's vector instructions, all of them in one dependent chain)
    const int i0 = L - DD_GAP_WINDOW > 0 ? L - DD_GAP_WINDOW : 0, i1 = R + DD_GAP_WINDOW < N - 1 ? R + DD_GAP_WINDOW : N - 1;
    auto z = [&](int i) {
        const double x = (double)(i + 1);
        return ybuf[LAY::at(i)] * (x * x);
    };
    // left: knots i0 + 1 .. L - 1 in this order, lane k the knot i0 + 1 + k; right: knots i1 - 1 .. R + 1 downwards, lane k the knot i1 - 1 - k
    const int nl = L - 1 - i0, nr = i1 - 1 - R;
    double al = 1., bl = 0., ar = 1., br = 0.;
    if (lane < nl) {
        const int i = i0 + 1 + lane;
        const double c = gtab[i - i0 < DD_NTAB ? i - i0 : DD_NTAB - 1];
        al = -c;
        bl = 3. * (z(i + 1) - z(i - 1)) * c;
    }
    if (lane < nr) {
        const int i = i1 - 1 - lane;
        const double c = gtab[i1 - i < DD_NTAB ? i1 - i : DD_NTAB - 1];
        ar = -c;
        br = 3. * (z(i + 1) - z(i - 1)) * c;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double pal = __shfl_xor(al, off), pbl = __shfl_xor(bl, off), par = __shfl_xor(ar, off), pbr = __shfl_xor(br, off);
        const bool lower = !(lane & off);      // this lane's block comes first: the partner's map is applied behind it
        bl = lower ? fma(pal, bl, pbl) : fma(al, pbl, bl);
        al *= pal;
        br = lower ? fma(par, br, pbr) : fma(ar, pbr, br);
        ar *= par;
    }
    // the sweeps' starting values (clamped ends: slope 0; inside the sequence: the centred difference), then the two rows at the gap
    double cpL = 0., dpL = i0 == 0 ? 0. : 0.5 * (z(i0 + 1) - z(i0 - 1));
    if (L != i0) {
        dpL = fma(al, dpL, bl);
        const double cp = nl > 0 ? gtab[nl < DD_NTAB ? nl : DD_NTAB - 1] : 0.;
        const double d = 3. * (g * (z(L) - z(L - 1)) + (z(R) - z(L)) * inv_g);
        const double den = 2. * (1. + g) - g * cp;
        cpL = cpmath::recip(den);
        dpL = (d - g * dpL) * cpL;
    }
    double bqR = 0., dqR = i1 == N - 1 ? 0. : 0.5 * (z(i1 + 1) - z(i1 - 1));
    if (R != i1) {
        dqR = fma(ar, dqR, br);
        const double bq = nr > 0 ? gtab[nr < DD_NTAB ? nr : DD_NTAB - 1] : 0.;
        const double d = 3. * ((z(R) - z(L)) * inv_g + g * (z(R + 1) - z(R)));
        const double den = 2. * (g + 1.) - g * bq;
        bqR = cpmath::recip(den);
        dqR = (d - g * dqR) * bqR;
    }
    const double sL = (dpL - cpL * dqR) * cpmath::recip(1. - cpL * bqR);
    const double sR = dqR - bqR * sL;
    const double zL = z(L), zR = z(R);
    const double slope = (zR - zL) * inv_g;
    const double tt = (sL + sR - 2. * slope) * inv_g;
    const double c3 = tt * inv_g, c2 = (slope - sL) * inv_g - tt;
    bool finite = true;
    for (int i = a + lane; i <= b; i += 64) {
        const double u = (double)(i - L), x = (double)(i + 1);
        const double v = (zL + u * (sL + u * (c2 + u * c3))) * cpmath::recip(x * x);
        if (!INPLACE || seq) seq[i] = v;      // (INPLACE: `seq` may be null -- the rewritten knots are wanted in LDS only)
        if constexpr (INPLACE) {      // ... and into the sequence in LDS (the knots read above lie outside the box: nothing read here is rewritten)
            const_cast<double*>(ybuf)[LAY::at(i)] = v;
            finite &= fabs(v) <= 1.7976931348623157e308;
        }
    }
    if (all_finite) *all_finite = finite;
    return true;
}

}  // namespace cpdd
