// Host emulation of the fused FFTLog kernel's per-thread phases (TEST HARNESS ONLY).
// Runs cp_fftlog_body.h thread-by-thread, phase-by-phase on the CPU (a workgroup barrier becomes the
// end of the loop over threads), with the tables built by the same host code the library uses.
// It validates index / twiddle / digit-order / layout logic without a GPU; it is NOT linked into
// libcosmoprimo_amd.so and the product never calls it.
#include <cstring>
#include <vector>

#include "../../cosmoprimo_amd/csrc/cp_fftlog_body.h"
#include "../../cosmoprimo_amd/csrc/cp_fftlog_dispatch.h"
#include "../../cosmoprimo_amd/csrc/cp_fftlog_tables.h"

using namespace cpfft;

// per-thread register state (tables loaded one phase ahead, prefetched rows) persists across phases and pairs
template <int NP, int P, int IM, int OM, int PH>
static void run_phases(const FftlogArgs& A, const double* ra, const double* rb, double* oa, double* ob, bool has_b, int ker, cplx* lds,
                       const double* nra, const double* nrb, int nxt_ker, typename Fftlog<NP, P, IM, OM>::State* st) {
    using F = Fftlog<NP, P, IM, OM>;
    for (int t = 0; t < F::T; ++t) F::template phase<PH>(t, A, ra, rb, oa, ob, has_b, ker, lds, nra, nrb, nxt_ker, st[t]);
    if constexpr (PH + 1 < F::NPH) run_phases<NP, P, IM, OM, PH + 1>(A, ra, rb, oa, ob, has_b, ker, lds, nra, nrb, nxt_ker, st);
}

template <int NP, int P, int IM, int OM>
static void run_pair(const FftlogArgs& A, const double* ra, const double* rb, double* oa, double* ob, bool has_b, int ker, cplx* lds,
                     const double* nra, const double* nrb, int nxt_ker, void* state, bool first) {
    using F = Fftlog<NP, P, IM, OM>;
    static_assert(sizeof(typename F::State) <= (16 * P + 32 * 8 + 32 + 16 * 8), "state slot too small");
    auto* st = reinterpret_cast<typename F::State*>(state);
    if (first)
        for (int t = 0; t < F::T; ++t) F::init_state(t, A, ra, rb, ker, st[t]);
    run_phases<NP, P, IM, OM, 0>(A, ra, rb, oa, ob, has_b, ker, lds, nra, nrb, nxt_ker, st);
}

template <int NP, int P>
static int emulate(int n, int nker, const double* pre, const double* post, const double* u_re_im, const double* in, double* out,
                   long long nbatch, int ext_l, double val_l, int ext_r, double val_r, int keep_padding) {
    std::vector<cplx> tw, u((size_t)nker * NP), lds(Fftlog<NP, P>::LDS_DATA + Fftlog<NP, P>::LDS_TW_ENTRIES + 1);
    build_twiddles<NP, P>(tw);
    for (int k = 0; k < nker; ++k) build_u_layout<NP, P>(u_re_im + (size_t)k * 2 * (NP / 2 + 1), u.data() + (size_t)k * NP);
    FftlogArgs A;
    const int npad = NP - n;
    A.in = in; A.out = out; A.nbatch = nbatch; A.nker = nker; A.n = n;
    A.in_left = npad / 2;
    A.out_off = keep_padding ? 0 : npad - npad / 2;
    A.n_out = keep_padding ? NP : n;
    A.ext_l = ext_l; A.ext_r = ext_r; A.val_l = val_l; A.val_r = val_r;
    A.stream_rows = 0;
    A.pre = pre; A.post = post; A.u = u.data(); A.tw = tw.data();
    for (int t = 0; t < Plan<NP, P>::T; ++t) Fftlog<NP, P>::fill_lds_tables(t, A, lds.data());
    const long long nhalf = (nbatch + 1) / 2, npairs = nhalf * nker;
    std::vector<char> pfv((size_t)(16 * P + 32 * 8 + 32 + 16 * 8) * Plan<NP, P>::T);  // State is the same size for every variant of (NP, P)
    void* pf = pfv.data();
    auto rows = [&](long long p, const double*& ra, const double*& rb, double*& oa, double*& ob, bool& has_b, int& ker) {
        ker = (int)(p % nker);
        const long long b0 = 2 * (p / nker), b1 = b0 + 1;
        ra = in + (b0 * nker + ker) * n;
        oa = out + (b0 * nker + ker) * A.n_out;
        has_b = b1 < nbatch;
        rb = has_b ? in + (b1 * nker + ker) * n : ra;
        ob = has_b ? out + (b1 * nker + ker) * A.n_out : oa;
    };
    // one emulated workgroup walks all pairs (grid = 1), prefetching the next pair's rows like the kernel does
    for (long long p = 0; p < npairs; ++p) {
        const double *ra, *rb, *nra, *nrb;
        double *oa, *ob, *noa, *nob;
        bool has_b, nhas_b;
        int ker, nker_;
        rows(p, ra, rb, oa, ob, has_b, ker);
        rows(p + 1 < npairs ? p + 1 : p, nra, nrb, noa, nob, nhas_b, nker_);
        const bool first = p == 0;
        // same variant selection as the library (cp_fftlog.hip: select_variant)
        const int v = select_variant(NP, P, n, ext_l, val_l, ext_r, val_r, keep_padding);
        if constexpr ((P == 16 || P == 8) && (NP > 16)) {
            if (v == VAR_HALF_ZERO) { run_pair<NP, P, IN_HALF_ZERO, OUT_HALF>(A, ra, rb, oa, ob, has_b, ker, lds.data(), nra, nrb, nker_, pf, first); continue; }
            if (v == VAR_HALF) { run_pair<NP, P, IN_HALF, OUT_HALF>(A, ra, rb, oa, ob, has_b, ker, lds.data(), nra, nrb, nker_, pf, first); continue; }
        }
        if (v == VAR_LOG) run_pair<NP, P, IN_LOG, OUT_GENERIC>(A, ra, rb, oa, ob, has_b, ker, lds.data(), nra, nrb, nker_, pf, first);
        else run_pair<NP, P, IN_GENERIC, OUT_GENERIC>(A, ra, rb, oa, ob, has_b, ker, lds.data(), nra, nrb, nker_, pf, first);
    }
    return 0;
}

extern "C" int emu_fftlog(int n, int np, int nker, const double* pre, const double* post, const double* u_re_im, const double* in,
                          double* out, long long nbatch, int ext_l, double val_l, int ext_r, double val_r, int keep_padding) {
#define X(NP_, P_) \
    if (np == NP_) return emulate<NP_, P_>(n, nker, pre, post, u_re_im, in, out, nbatch, ext_l, val_l, ext_r, val_r, keep_padding);
    CP_FFTLOG_SIZES(X)
#undef X
    return -1;
}

// plain DFT check of the butterflies
extern "C" void emu_dft(int r, double* re_im) {
    cplx x[16];
    for (int i = 0; i < r; ++i) x[i] = cplx{re_im[2 * i], re_im[2 * i + 1]};
    if (r == 2) Dft<2>::run(x);
    if (r == 4) Dft<4>::run(x);
    if (r == 8) Dft<8>::run(x);
    if (r == 16) Dft<16>::run(x);
    for (int i = 0; i < r; ++i) { re_im[2 * i] = x[i].re; re_im[2 * i + 1] = x[i].im; }
}
