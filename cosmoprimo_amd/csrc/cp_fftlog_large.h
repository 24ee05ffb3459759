// cp_fftlog_large.h -- the general-size (Np > 8192) FFTLog path: see cp_fftlog_large.hip
#pragma once
#include <hip/hip_runtime.h>

struct cp_fftlog_large;
// u_re_im: (nker, npad / 2 + 1) complex as (re, im) pairs -- the reference's padded_u, not the thread layout of the fused kernel
int cp_fftlog_large_create(cp_fftlog_large** out, int n, int npad, int nker, const double* pre, const double* post, const double* u_re_im, int device);
int cp_fftlog_large_execute(cp_fftlog_large* p, const double* d_in, double* d_out, long long nbatch, int ext_l, double val_l, int ext_r, double val_r,
                            int keep_padding, hipStream_t stream);
void cp_fftlog_large_destroy(cp_fftlog_large* p);
