// cp_fftlog_dispatch.h -- the (padded size NP -> points per thread P) table of the fused kernel.
// One LDS-resident packed pair needs NP * 16 B, so NP <= 8192 (128 KiB of the CU's 160 KiB).
#pragma once

// X(NP, P): T = NP / P threads per pair
#ifdef CP_FFTLOG_ONLY  // development builds: a single size
#define CP_FFTLOG_SIZES(X) X(CP_FFTLOG_ONLY, 16)
#else
#define CP_FFTLOG_SIZES(X) \
    X(4, 4)                \
    X(8, 8)                \
    X(16, 16)              \
    X(32, 16)              \
    X(64, 16)              \
    X(128, 16)             \
    X(256, 16)             \
    X(512, 16)             \
    X(1024, 16)            \
    X(2048, 16)            \
    X(4096, 16)            \
    X(8192, 16)
#endif

#define CP_FFTLOG_MAX_NP 8192

// Kernel variants (cp_fftlog_body.h front / back ends) and the rule that picks one.
// VAR_HALF_ZERO_WINDOW: VAR_HALF_ZERO storing a window of the output columns only (cp_fftlog_execute_window)
enum { VAR_GENERIC = 0, VAR_LOG = 1, VAR_HALF = 2, VAR_HALF_ZERO = 3, VAR_HALF_ZERO_WINDOW = 4, VAR_COUNT = 5 };

// HALF variants exist for P == 16 and NP >= CP_FFTLOG_HALF_MIN_NP
#define CP_FFTLOG_HALF_MIN_NP 512

inline int select_variant(int np, int p, int n, int ext_l, double val_l, int ext_r, double val_r, int keep_padding) {
    if (ext_l == 2 || ext_r == 2) return VAR_LOG;
    if ((p == 16 || p == 8) && np >= CP_FFTLOG_HALF_MIN_NP && 2 * n == np && !keep_padding) {
        if (ext_l == 0 && ext_r == 0 && val_l == 0. && val_r == 0.) return VAR_HALF_ZERO;
        return VAR_HALF;
    }
    return VAR_GENERIC;
}
