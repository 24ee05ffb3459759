// cp_fftlog_tables.h -- host-side construction of the device tables of one FFTLog plan
// (twiddles per pass, Hermitian-extended u in the kernel's thread layout).  Pure host C++.
#pragma once
#include <cmath>
#include <vector>

#include "cp_fft_core.h"

namespace cpfft {

// exp(-2 pi i k / n), n a power of two: the angle is reduced to (-pi/4, pi/4] around the nearest
// multiple of pi/2 and evaluated in long double, so axes are exact and components are correctly
// rounded in practice.
inline cplx unit_root(long long k, long long n) {
    k %= n;
    if (k < 0) k += n;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    const long long q = (4 * k + n / 2) / n;  // nearest multiple of pi/2
    const long long kr = 4 * k - q * n;       // 4k - q n in [-n/2, n/2)
    const long double ang = two_pi * (long double)kr / (4.0L * (long double)n);
    const long double c = cosl(ang), s = sinl(ang);
    long double cr, sr;
    switch (q & 3) {
        case 0: cr = c; sr = s; break;
        case 1: cr = -s; sr = c; break;
        case 2: cr = -c; sr = -s; break;
        default: cr = s; sr = -c; break;
    }
    cplx w;
    w.re = (double)cr;
    w.im = (double)(-sr);
    return w;
}

template <int NP, int P>
inline void build_twiddles(std::vector<cplx>& tw) {
    using PL = Plan<NP, P>;
    tw.assign(PL::TW_TOTAL > 0 ? PL::TW_TOTAL : 1, cplx{1., 0.});
    for (int i = 0; i < PL::NPASS; ++i) {
        const int L = PL::len(i), R = PL::radix(i), M = L / R;
        cplx* t = tw.data() + PL::tw_offset(i);
        for (int s = 0; s < R; ++s)
            for (int j = 0; j < M; ++j) t[s * M + j] = unit_root((long long)j * s, L);
    }
}

// u: (NP/2 + 1) complex of one kernel (reference padded_u, fftlog.py:179-180).  Output: NP complex in the
// thread layout of the middle phase, scaled by 1/NP, Hermitian-extended with real DC / Nyquist bins
// (what numpy's irfft assumes, fftlog.py:544).
template <int NP, int P>
inline void build_u_layout(const double* u_re_im, cplx* out) {
    using PL = Plan<NP, P>;
    constexpr int LASTP = PL::NPASS - 1;
    constexpr int R = PL::radix(LASTP);
    constexpr int T = PL::T, NB = P / R;
    const double inv = 1.0 / (double)NP;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < NB; ++i)
            for (int s = 0; s < R; ++s) {
                const int beta = t + T * i;
                const int pos = beta * R + s;  // last pass: L = R, M = 1
                const int k = dif_freq_of_pos<NP, P>(pos);
                cplx v;
                if (k == 0) {
                    v = cplx{u_re_im[0], 0.};
                } else if (k == NP / 2) {
                    v = cplx{u_re_im[2 * (NP / 2)], 0.};
                } else if (k < NP / 2) {
                    v = cplx{u_re_im[2 * k], u_re_im[2 * k + 1]};
                } else {
                    v = cplx{u_re_im[2 * (NP - k)], -u_re_im[2 * (NP - k) + 1]};
                }
                v.re *= inv;
                v.im *= inv;
                out[(i * R + s) * T + t] = v;
            }
}

}  // namespace cpfft
