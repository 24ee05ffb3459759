// Host access to the plan builder of the uniform-stretch spliced spline (csrc/cp_splice_uniform_plan.h) for the CPU tests (TEST HARNESS ONLY):
// the tables it hands the kernel, as plain arrays.  tests/test_splice_host.py checks them against a dense solve and runs the kernel's arithmetic in numpy.
#include "../../cosmoprimo_amd/csrc/cp_splice_uniform_plan.h"

// ints: S, nm, wl, wr, src_u, col_u, src_l, col_l, src_r, col_r, gb0, ngb, gfirst, gend, lane_b; doubles: mb0, mb1; win (8, 64); qe / qw as (64 ngb) / (64 ngb, 4),
// at most 64 x 8 entries.  Returns 1 if the scheme fits, 0 if not.
extern "C" int emu_splice_build(int n, const double* x, int npieces, const int* piece_first, const int* piece_src, const int* piece_start, int nq, const double* xq,
                                const int* qj, int u0, int u1, int generic_first, int generic_end, int* ints, double* doubles, double* win, int* qe, double* qw) {
    const cpsu::Built b = cpsu::build(n, x, piece_first, piece_src, piece_start, npieces, nq, xq, qj, u0, u1, generic_first, generic_end);
    if (!b.ok) return 0;
    const cpsu::Tables& T = b.T;
    const int iv[15] = {T.S, T.nm, T.wl, T.wr, T.src_u, T.col_u, T.src_l, T.col_l, T.src_r, T.col_r, T.gb0, T.ngb, T.gfirst, T.gend, T.lane_b};
    for (int i = 0; i < 15; ++i) ints[i] = iv[i];
    doubles[0] = T.mb0;
    doubles[1] = T.mb1;
    for (size_t i = 0; i < b.win.size(); ++i) win[i] = b.win[i];
    for (size_t i = 0; i < b.qe.size(); ++i) qe[i] = b.qe[i];
    for (size_t i = 0; i < b.qw.size(); ++i) qw[i] = b.qw[i];
    return 1;
}
