// cp_fft_core.h -- in-register radix butterflies and the pass structure of the fused FFTLog kernel.
//
// This header is device code for gfx950 (included by cp_fftlog.hip).  Every function is also a
// valid host function so that tests/host_emu can run the *same* index / twiddle / digit-order
// logic thread-by-thread on the CPU (no GPU in the build container).  The host emulation is a
// unit-test harness only; it is not part of libcosmoprimo_amd.so.
//
// Math (reference cosmoprimo/fftlog.py:228-241, 538-544):
//   g = irfft(conj(rfft(a * pre) * u), n=Np) * post
// irfft(conj X)[n] = irfft(X)[-n], so with U the Hermitian extension of u (DC and Nyquist bins
// real, as numpy's c2r assumes) g = FFT(FFT(a * pre) * U) / Np -- two *forward* complex FFTs and a
// complex-linear map.  Two real rows a, b are therefore packed as z = a + i b and transformed
// together: Re -> g_a, Im -> g_b, with no real-FFT split/merge step.
// FFT #1 is decimation-in-frequency (natural in, digit-reversed out), U is stored in that
// digit-reversed order, FFT #2 is the transposed (decimation-in-time) network: no reordering pass,
// and the last DIF pass / first DIT pass share registers (no LDS round trip in the middle).
#pragma once

#if defined(__HIPCC__)
#define CP_HD __host__ __device__ __forceinline__
#else
#define CP_HD inline
#endif

// scheduling fence: the compiler may not move instructions across it (device only; no code is emitted)
#if defined(__HIP_DEVICE_COMPILE__)
#define CP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define CP_SCHED_FENCE() ((void)0)
#endif

#if defined(CP_STAMPS) && defined(__HIPCC__)
// diagnostic build: wave clock, with the LDS / scalar-memory counter drained on both sides
__device__ __forceinline__ unsigned long long cp_stamp() {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define CP_FS_BEGIN(st) ((st).fs_last = cp_stamp())
#define CP_FS(st, k)                               \
    do {                                           \
        const unsigned long long now_ = cp_stamp(); \
        (st).fs[k] += now_ - (st).fs_last;         \
        (st).fs_last = now_;                       \
    } while (0)
#else
#define CP_FS_BEGIN(st) ((void)0)
#define CP_FS(st, k) ((void)0)
#endif

// Diagnostic builds only (tools/fftlog_microbench.hip): a bit mask that removes one cost at a time so that timing
// differences show what the kernel is waiting on.  Results are wrong when any bit is set; the library is built with 0.
//   1 no LDS traffic   2 no workgroup barriers   4 no table loads   8 no HBM row loads   16 no HBM stores   32 no butterflies
#ifndef CP_ABLATE
#define CP_ABLATE 0
#endif

namespace cpfft {

struct cplx {
    double re, im;
};

// ---- global-memory access through buffer instructions -----------------------------------------------------------
// base is wave-uniform (a kernel argument or a per-pair row pointer, i.e. SGPRs), `voff` the per-lane byte offset and
// `soff` a uniform byte offset: one `buffer_load_dwordx4 v, voff, s[rsrc], soff offen` with NO per-access 64-bit VALU
// address arithmetic (the plain-pointer form costs 2-4 integer VALU instructions per access on a VALU-bound kernel).
#if defined(__HIP_DEVICE_COMPILE__)
typedef int cp_v4i __attribute__((ext_vector_type(4)));
typedef int cp_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cp_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ cplx ld_cplx(const void* base, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(cp_rsrc(base), (int)voff, (int)soff, 0));
}
__device__ __forceinline__ double ld_f64(const void* base, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cp_rsrc(base), (int)voff, (int)soff, 0));
}
__device__ __forceinline__ void st_f64(void* base, unsigned voff, unsigned soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cp_v2i, v), cp_rsrc(base), (int)voff, (int)soff, 0);
}
// the rows themselves, read once and written once: the same accesses with the non-temporal cache policy (aux bit 1 = nt), used when a launch
// moves more bytes than the Infinity Cache holds (FftlogArgs::stream_rows) -- 1.5 % on the 100 000 x 2048 batch
__device__ __forceinline__ double ld_row_f64_nt(const void* base, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(cp_rsrc(base), (int)voff, (int)soff, 2));
}
__device__ __forceinline__ void st_row_f64_nt(void* base, unsigned voff, unsigned soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cp_v2i, v), cp_rsrc(base), (int)voff, (int)soff, 2);
}
__device__ __forceinline__ void st_cplx(void* base, unsigned voff, unsigned soff, cplx v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cp_v4i, v), cp_rsrc(base), (int)voff, (int)soff, 0);
}
// value of the neighbouring lane (lane ^ 1): a DPP quad permute, no LDS
__device__ __forceinline__ double lane_swap1(double v) {
    cp_v2i w = __builtin_bit_cast(cp_v2i, v);
    w.x = __builtin_amdgcn_mov_dpp(w.x, 0xB1, 0xF, 0xF, true);  // quad_perm:[1,0,3,2]
    w.y = __builtin_amdgcn_mov_dpp(w.y, 0xB1, 0xF, 0xF, true);
    return __builtin_bit_cast(double, w);
}
// 2x2 transpose between the lanes of a pair (lane, lane ^ 1): even lanes end with (own lo, neighbour lo), odd lanes with
// (neighbour hi, own hi).  v_cndmask_b32_dpp does the neighbour read and the select in one instruction (4 VALU per pair
// of doubles; the compiler does not form it from mov_dpp + select because the condition has to sit in VCC).
// s_nop 1: the two wait states a DPP read needs after a VALU write of its source.
__device__ __forceinline__ void lane_transpose2(double& lo, double& hi) {
    const cp_v2i l = __builtin_bit_cast(cp_v2i, lo), h = __builtin_bit_cast(cp_v2i, hi);
    cp_v2i a, b;
    asm volatile(
        "s_mov_b64 vcc, %8\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_dpp %0, %4, %6, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %5, %7, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %2, %6, %4, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %7, %5, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(a.x), "=&v"(a.y), "=&v"(b.x), "=&v"(b.y)
        : "v"(h.x), "v"(h.y), "v"(l.x), "v"(l.y), "s"(0x5555555555555555ull)
        : "vcc", "scc");  // s_not_b64 writes SCC: without the clobber the compiler may keep a carry alive across the block
    lo = __builtin_bit_cast(double, a);
    hi = __builtin_bit_cast(double, b);
}
// maxima of two unsigned values over the 64 lanes of a wave, valid in lane 63: four DPP steps inside the rows of 16 lanes (quad_perm
// [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8: every lane of a row then holds the row maximum), then row_bcast:15 into rows 1 and 3 and
// row_bcast:31 into rows 2 and 3 -- twelve v_max_u32_dpp, no lane reads and no scalar arithmetic
__device__ __forceinline__ void wave_max2_u32(unsigned& a, unsigned& b) {
#define CP_WAVE_MAX_STEP(CTRL, ROWS)                                                                  \
    {                                                                                                 \
        const unsigned pa = (unsigned)__builtin_amdgcn_update_dpp((int)a, (int)a, CTRL, ROWS, 0xf, false); \
        const unsigned pb = (unsigned)__builtin_amdgcn_update_dpp((int)b, (int)b, CTRL, ROWS, 0xf, false); \
        a = a > pa ? a : pa;                                                                          \
        b = b > pb ? b : pb;                                                                          \
    }
    CP_WAVE_MAX_STEP(0xB1, 0xf)
    CP_WAVE_MAX_STEP(0x4E, 0xf)
    CP_WAVE_MAX_STEP(0x124, 0xf)
    CP_WAVE_MAX_STEP(0x128, 0xf)
    CP_WAVE_MAX_STEP(0x142, 0xa)  // row_bcast:15 -> rows 1, 3
    CP_WAVE_MAX_STEP(0x143, 0xc)  // row_bcast:31 -> rows 2, 3
#undef CP_WAVE_MAX_STEP
}
// the same for N = 2 or 3 (lo, hi) pairs in one block: VCC is set up once per block instead of once per pair (the kernel is
// bound by instruction issue, scalar instructions included); an asm statement takes at most 30 operands, hence N <= 3
__device__ __forceinline__ void lane_transpose2x2(double* lo, double* hi) {
    cp_v2i l0 = __builtin_bit_cast(cp_v2i, lo[0]), h0 = __builtin_bit_cast(cp_v2i, hi[0]);
    cp_v2i l1 = __builtin_bit_cast(cp_v2i, lo[1]), h1 = __builtin_bit_cast(cp_v2i, hi[1]);
    cp_v2i a0, b0, a1, b1;
    asm volatile(
        "s_mov_b64 vcc, %16\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_dpp %0, %10, %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %11, %9, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %4, %14, %12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %5, %15, %13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %2, %8, %10, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %9, %11, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %6, %12, %14, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %7, %13, %15, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(a0.x), "=&v"(a0.y), "=&v"(b0.x), "=&v"(b0.y), "=&v"(a1.x), "=&v"(a1.y), "=&v"(b1.x), "=&v"(b1.y)
        : "v"(l0.x), "v"(l0.y), "v"(h0.x), "v"(h0.y), "v"(l1.x), "v"(l1.y), "v"(h1.x), "v"(h1.y), "s"(0x5555555555555555ull)
        : "vcc", "scc");
    lo[0] = __builtin_bit_cast(double, a0);
    hi[0] = __builtin_bit_cast(double, b0);
    lo[1] = __builtin_bit_cast(double, a1);
    hi[1] = __builtin_bit_cast(double, b1);
}
__device__ __forceinline__ void lane_transpose2x3(double* lo, double* hi) {
    cp_v2i l0 = __builtin_bit_cast(cp_v2i, lo[0]), h0 = __builtin_bit_cast(cp_v2i, hi[0]);
    cp_v2i l1 = __builtin_bit_cast(cp_v2i, lo[1]), h1 = __builtin_bit_cast(cp_v2i, hi[1]);
    cp_v2i l2 = __builtin_bit_cast(cp_v2i, lo[2]), h2 = __builtin_bit_cast(cp_v2i, hi[2]);
    cp_v2i a0, b0, a1, b1, a2, b2;
    asm volatile(
        "s_mov_b64 vcc, %24\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_dpp %0, %14, %12, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %15, %13, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %4, %18, %16, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %5, %19, %17, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %8, %22, %20, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %9, %23, %21, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_not_b64 vcc, vcc\n\t"
        "v_cndmask_b32_dpp %2, %12, %14, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %13, %15, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %6, %16, %18, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %7, %17, %19, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %10, %20, %22, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %11, %21, %23, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(a0.x), "=&v"(a0.y), "=&v"(b0.x), "=&v"(b0.y), "=&v"(a1.x), "=&v"(a1.y), "=&v"(b1.x), "=&v"(b1.y), "=&v"(a2.x), "=&v"(a2.y),
          "=&v"(b2.x), "=&v"(b2.y)
        : "v"(l0.x), "v"(l0.y), "v"(h0.x), "v"(h0.y), "v"(l1.x), "v"(l1.y), "v"(h1.x), "v"(h1.y), "v"(l2.x), "v"(l2.y), "v"(h2.x), "v"(h2.y),
          "s"(0x5555555555555555ull)
        : "vcc", "scc");
    lo[0] = __builtin_bit_cast(double, a0);
    hi[0] = __builtin_bit_cast(double, b0);
    lo[1] = __builtin_bit_cast(double, a1);
    hi[1] = __builtin_bit_cast(double, b1);
    lo[2] = __builtin_bit_cast(double, a2);
    hi[2] = __builtin_bit_cast(double, b2);
}
// N pairs, N a multiple of 2 or 3 ... in blocks of 3, 3, 2 for the 8 pairs of a P = 16 thread
template <int N>
__device__ __forceinline__ void lane_transpose2_all(double* lo, double* hi) {
    if constexpr (N >= 3 && N != 4) {
        lane_transpose2x3(lo, hi);
        lane_transpose2_all<N - 3>(lo + 3, hi + 3);
    } else if constexpr (N >= 2) {
        lane_transpose2x2(lo, hi);
        lane_transpose2_all<N - 2>(lo + 2, hi + 2);
    } else if constexpr (N == 1) {
        lane_transpose2(lo[0], hi[0]);
    }
}
#else
inline cplx ld_cplx(const void* base, unsigned voff, unsigned soff) {
    return *reinterpret_cast<const cplx*>(reinterpret_cast<const char*>(base) + voff + soff);
}
inline double ld_f64(const void* base, unsigned voff, unsigned soff) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + voff + soff);
}
inline void st_f64(void* base, unsigned voff, unsigned soff, double v) {
    *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + voff + soff) = v;
}
inline double ld_row_f64_nt(const void* base, unsigned voff, unsigned soff) { return ld_f64(base, voff, soff); }
inline void st_row_f64_nt(void* base, unsigned voff, unsigned soff, double v) { st_f64(base, voff, soff, v); }
#endif

// LDS accesses by 32-bit byte address.  On the device the address is an integer in the LDS address space: the buffer's
// own offset is added once per phase (to the thread's base) instead of once per access, which is what pointer arithmetic
// on the generic pointer costs.  The XOR forms of Pass::lds_off rely on the buffer being 4096-byte aligned in LDS
// ((base + l0) ^ k == (base ^ k) + l0 for k < 4096): the kernels declare it so.
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3), aligned(16))) cplx cp_lds_cplx;  // 16-byte slots: ds_read_b128 / ds_write_b128
struct LdsView {
    unsigned l0;
    __device__ __forceinline__ explicit LdsView(const void* lds)
        : l0((unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)lds) {}
    __device__ __forceinline__ cplx read(unsigned addr) const { return *(const cp_lds_cplx*)(unsigned long long)addr; }
    __device__ __forceinline__ void write(unsigned addr, const cplx v) const { *(cp_lds_cplx*)(unsigned long long)addr = v; }
};
#else
struct LdsView {
    static constexpr unsigned l0 = 0u;
    char* p;
    explicit LdsView(const void* lds) : p(const_cast<char*>(static_cast<const char*>(lds))) {}
    cplx read(unsigned addr) const { return *reinterpret_cast<const cplx*>(p + addr); }
    void write(unsigned addr, const cplx v) const { *reinterpret_cast<cplx*>(p + addr) = v; }
};
#endif

CP_HD cplx cmul(const cplx a, const cplx b) {
    cplx r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}

constexpr int cmin(int a, int b) { return a < b ? a : b; }

// ---- row screening (cp_fftlog_body.h, "row independence") ----------------------------------------------------------------
// high dword of |v|: an unsigned integer that orders finite doubles by magnitude, with Inf / NaN above every finite value;
// bits 30..20 are the biased exponent
CP_HD unsigned hi_abs(double v) { return (unsigned)(__builtin_bit_cast(unsigned long long, v) >> 32) & 0x7fffffffu; }

// Pass plan: radices R_0 = P, then min(P, remaining) until the product is NP.
// len(i) = sub-FFT length handled by pass i (L_0 = NP), radix(i) = R_i, M_i = L_i / R_i.
template <int NP, int P>
struct Plan {
    static_assert((NP & (NP - 1)) == 0 && (P & (P - 1)) == 0 && P <= NP && P >= 2, "powers of two");
    static constexpr int T = NP / P;  // threads per packed pair of rows
    static constexpr int count() {
        int n = 0, rem = NP;
        while (rem > 1) {
            rem /= cmin(P, rem);
            ++n;
        }
        return n;
    }
    static constexpr int NPASS = count();
    static constexpr int len(int i) {
        int rem = NP;
        for (int k = 0; k < i; ++k) rem /= cmin(P, rem);
        return rem;
    }
    static constexpr int radix(int i) { return cmin(P, len(i)); }
    // offset of pass i in the concatenated twiddle table (pass i holds len(i) entries [s][j])
    static constexpr int tw_offset(int i) {
        int off = 0;
        for (int k = 0; k < i; ++k) off += len(k);
        return off;
    }
    static constexpr int TW_TOTAL = tw_offset(NPASS);
};

// LDS slot swizzle: XOR the low 4 bits of the element index with a GF(2)-linear function of the higher bits, chosen
// per radix so that with 16-byte complex slots every pass shape of the plan is conflict-free for ds_read_b128 /
// ds_write_b128 (16 lanes are served per LDS cycle and must hit 16 distinct slots mod 16):
//   P = 16 (strides 1 | 16 in blocks of 256 | 16 contiguous per lane):  low4 ^= bits 4..7
//   P = 8  (strides 1 | 8 in blocks of 64 | 8 contiguous per lane):     low3 ^= bits 3..5, bit 3 ^= bit 6
//          (reads are served 16 lanes per cycle over 64 banks, 16-byte writes 8 contiguous lanes over 32 banks:
//          MI355X_MICROARCH.md, LDS; tools/lds_conflict_model.py checks every pass shape against both rules)
// Linearity (swz(a ^ b) == swz(a) ^ swz(b)) is what lets Pass::lds_off split an address into a per-thread base and a
// compile-time per-point constant.
// NP = 16^3, P = 16 (the benchmark shape): a digit-wise PADDED layout instead of the XOR swizzle.  Element (hi, mid, lo) (base-16
// digits of its index) sits in slot lo + 17 mid + 272 hi.  In every pass one digit is the thread's point index r and the other
// two come from the thread index, so an address is thread base + r * constant: the constant goes into the 16-bit offset field
// of ds_read / ds_write and NO per-access address arithmetic is left (the XOR form costs one VALU instruction per access on a
// kernel that is bound by instruction issue).  Bank behaviour (16-byte slots; reads are served 16 lanes at a time in the lane
// groups of MI355X_MICROARCH.md, writes 8 contiguous lanes): the pass-1 and pass-2 shapes and the pass-0 writes are
// conflict-free, and so is the pass-0 shape once the lanes are permuted (Fftlog::pass0_thread).
// MEASURED (MI355X, 100 000 x 2048, tools/mb_variants.sh, round 2): 1.134-1.145 ms against 1.124 ms for the XOR swizzle on the
// same box, although the main loop has 130 fewer VALU instructions -- the kernel is not bound by instruction count.  Kept as a
// build option (-DCP_PADDED_LDS=1), off by default.
#ifndef CP_PADDED_LDS
#define CP_PADDED_LDS 0
#endif
constexpr bool padded_lds(int NP, int P) { return CP_PADDED_LDS && NP == 4096 && P == 16; }
// complex slots of the data region of one packed pair
constexpr int lds_data_slots(int NP, int P) { return padded_lds(NP, P) ? 16 * 272 : NP; }

template <int NP, int P = 16>
CP_HD int swz(int p) {
    if (padded_lds(NP, P)) return (p & 15) + 17 * ((p >> 4) & 15) + 272 * (p >> 8);
    if (NP >= 256 && P == 8) return p ^ ((p >> 3) & 7) ^ (((p >> 6) & 1) << 3);
    if (NP >= 256) return p ^ ((p >> 4) & 15);
    return p;
}

// ---- small DFTs, forward sign exp(-2 pi i r s / R), natural order in and out -------------------
template <int R>
struct Dft;

template <>
struct Dft<2> {
    static CP_HD void run(cplx* x) {
        const cplx a = x[0], b = x[1];
        x[0].re = a.re + b.re;
        x[0].im = a.im + b.im;
        x[1].re = a.re - b.re;
        x[1].im = a.im - b.im;
    }
};

CP_HD void dft4(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {
    const double s0r = x0.re + x2.re, s0i = x0.im + x2.im;
    const double d0r = x0.re - x2.re, d0i = x0.im - x2.im;
    const double s1r = x1.re + x3.re, s1i = x1.im + x3.im;
    const double d1r = x1.re - x3.re, d1i = x1.im - x3.im;
    x0.re = s0r + s1r;
    x0.im = s0i + s1i;
    x2.re = s0r - s1r;
    x2.im = s0i - s1i;
    x1.re = d0r + d1i;  // d0 - i d1
    x1.im = d0i - d1r;
    x3.re = d0r - d1i;  // d0 + i d1
    x3.im = d0i + d1r;
}

template <>
struct Dft<4> {
    static CP_HD void run(cplx* x) { dft4(x[0], x[1], x[2], x[3]); }
};

// multiply by w_16^K = exp(-2 pi i K / 16), K compile-time
template <int K>
CP_HD void mul_w16(cplx& a) {
    constexpr double C1 = 0.92387953251128673848;  // cos(pi/8)
    constexpr double S1 = 0.38268343236508978178;  // sin(pi/8)
    constexpr double H = 0.70710678118654752440;   // sqrt(1/2)
    const double r = a.re, i = a.im;
    if (K == 0) {
    } else if (K == 1) {  // (C1, -S1)
        a.re = r * C1 + i * S1;
        a.im = i * C1 - r * S1;
    } else if (K == 2) {  // (H, -H)
        a.re = (r + i) * H;
        a.im = (i - r) * H;
    } else if (K == 3) {  // (S1, -C1)
        a.re = r * S1 + i * C1;
        a.im = i * S1 - r * C1;
    } else if (K == 4) {  // -i
        a.re = i;
        a.im = -r;
    } else if (K == 6) {  // (-H, -H)
        a.re = (i - r) * H;
        a.im = -(r + i) * H;
    } else if (K == 9) {  // (-C1, S1)
        a.re = -(r * C1 + i * S1);
        a.im = r * S1 - i * C1;
    }
}

template <>
struct Dft<8> {
    // 8 = 2 x 4: r = r0 + 2 r1, s = s0 + 4 s1;  X[s0 + 4 s1] = sum_r0 w8^(r0 s0) w2^(r0 s1) sum_r1 x[r0 + 2 r1] w4^(r1 s0)
    static CP_HD void run(cplx* x) {
        cplx e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
        cplx o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
        dft4(e0, e1, e2, e3);
        dft4(o0, o1, o2, o3);
        mul_w16<2>(o1);  // w8^1
        mul_w16<4>(o2);  // w8^2
        mul_w16<6>(o3);  // w8^3
        x[0].re = e0.re + o0.re; x[0].im = e0.im + o0.im; x[4].re = e0.re - o0.re; x[4].im = e0.im - o0.im;
        x[1].re = e1.re + o1.re; x[1].im = e1.im + o1.im; x[5].re = e1.re - o1.re; x[5].im = e1.im - o1.im;
        x[2].re = e2.re + o2.re; x[2].im = e2.im + o2.im; x[6].re = e2.re - o2.re; x[6].im = e2.im - o2.im;
        x[3].re = e3.re + o3.re; x[3].im = e3.im + o3.im; x[7].re = e3.re - o3.re; x[7].im = e3.im - o3.im;
    }
};

// inner DFT4 with first and last input structurally zero: (0, x1, x2, 0) -> y0..y3 written to the four slots
CP_HD void dft4_mid(cplx& x0, cplx& x1, cplx& x2, cplx& x3) {
    const cplx a = x1, b = x2;
    x0.re = b.re + a.re;
    x0.im = b.im + a.im;
    x2.re = b.re - a.re;
    x2.im = b.im - a.im;
    x1.re = a.im - b.re;  // -b - i a
    x1.im = -b.im - a.re;
    x3.re = -b.re - a.im;  // -b + i a
    x3.im = a.re - b.im;
}

// 16 = 4 x 4: r = r0 + 4 r1, s = s0 + 4 s1;
// X[s0 + 4 s1] = sum_r0 w16^(r0 s0) w4^(r0 s1) sum_r1 x[r0 + 4 r1] w4^(r1 s0)
// After the inner DFT4s (over r1, one per r0) slot x[r0 + 4 s0] holds the partial sum for (r0, s0).
//
// The sqrt(1/2) of the w16^2 / w16^6 twiddles is not applied to the rotated element but folded into the additions that
// consume it (a +- H u is one FMA each): 8 fewer fp64 instructions per radix-16 butterfly.
constexpr double CP_SQRT_HALF = 0.70710678118654752440;
// u = a * w16^2 / H = (re + im, im - re);  u = a * w16^6 / H = (im - re, -re - im)
CP_HD cplx rot_w16_2_unscaled(const cplx a) { return cplx{a.re + a.im, a.im - a.re}; }
CP_HD cplx rot_w16_6_unscaled(const cplx a) { return cplx{a.im - a.re, -a.re - a.im}; }
// DFT4 of (x0, x1, H u2, x3)
CP_HD void dft4_h2(cplx& x0, cplx& x1, cplx& u2, cplx& x3) {
    constexpr double H = CP_SQRT_HALF;
    const double s0r = __builtin_fma(H, u2.re, x0.re), s0i = __builtin_fma(H, u2.im, x0.im);
    const double d0r = __builtin_fma(-H, u2.re, x0.re), d0i = __builtin_fma(-H, u2.im, x0.im);
    const double s1r = x1.re + x3.re, s1i = x1.im + x3.im;
    const double d1r = x1.re - x3.re, d1i = x1.im - x3.im;
    x0.re = s0r + s1r;
    x0.im = s0i + s1i;
    u2.re = s0r - s1r;
    u2.im = s0i - s1i;
    x1.re = d0r + d1i;
    x1.im = d0i - d1r;
    x3.re = d0r - d1i;
    x3.im = d0i + d1r;
}
// DFT4 of (x0, H u1, x2, H u3)
CP_HD void dft4_h13(cplx& x0, cplx& u1, cplx& x2, cplx& u3) {
    constexpr double H = CP_SQRT_HALF;
    const double s0r = x0.re + x2.re, s0i = x0.im + x2.im;
    const double d0r = x0.re - x2.re, d0i = x0.im - x2.im;
    const double s1r = u1.re + u3.re, s1i = u1.im + u3.im;  // unscaled
    const double d1r = u1.re - u3.re, d1i = u1.im - u3.im;
    x0.re = __builtin_fma(H, s1r, s0r);
    x0.im = __builtin_fma(H, s1i, s0i);
    x2.re = __builtin_fma(-H, s1r, s0r);
    x2.im = __builtin_fma(-H, s1i, s0i);
    u1.re = __builtin_fma(H, d1i, d0r);
    u1.im = __builtin_fma(-H, d1r, d0i);
    u3.re = __builtin_fma(-H, d1i, d0r);
    u3.im = __builtin_fma(H, d1r, d0i);
}

// stage S0 of the outer half: twiddles w16^(r0 S0) on x[r0 + 4 S0], then the DFT4 over r0 in place
// (afterwards x[4 S0 + s1] holds X[S0 + 4 s1])
template <int S0>
CP_HD void dft16_outer_stage(cplx* x) {
    if (S0 == 0) {
        dft4(x[0], x[1], x[2], x[3]);
    } else if (S0 == 1) {
        mul_w16<1>(x[4 + 1]);
        x[4 + 2] = rot_w16_2_unscaled(x[4 + 2]);
        mul_w16<3>(x[4 + 3]);
        dft4_h2(x[4], x[5], x[6], x[7]);
    } else if (S0 == 2) {
        x[8 + 1] = rot_w16_2_unscaled(x[8 + 1]);
        mul_w16<4>(x[8 + 2]);
        x[8 + 3] = rot_w16_6_unscaled(x[8 + 3]);
        dft4_h13(x[8], x[9], x[10], x[11]);
    } else {
        mul_w16<3>(x[12 + 1]);
        x[12 + 2] = rot_w16_6_unscaled(x[12 + 2]);
        mul_w16<9>(x[12 + 3]);
        dft4_h2(x[12], x[13], x[14], x[15]);
    }
}

CP_HD void dft16_outer(cplx* x) {
    dft16_outer_stage<0>(x);
    dft16_outer_stage<1>(x);
    dft16_outer_stage<2>(x);
    dft16_outer_stage<3>(x);
    // x[4 s0 + s1] holds X[s0 + 4 s1]: 4x4 transpose (register renaming after unrolling)
    cplx t;
    t = x[1]; x[1] = x[4]; x[4] = t;
    t = x[2]; x[2] = x[8]; x[8] = t;
    t = x[3]; x[3] = x[12]; x[12] = t;
    t = x[6]; x[6] = x[9]; x[9] = t;
    t = x[7]; x[7] = x[13]; x[13] = t;
    t = x[11]; x[11] = x[14]; x[14] = t;
}

CP_HD void dft16_inner(cplx* x) {
    dft4(x[0], x[4], x[8], x[12]);
    dft4(x[1], x[5], x[9], x[13]);
    dft4(x[2], x[6], x[10], x[14]);
    dft4(x[3], x[7], x[11], x[15]);
}
// points 0..3 and 12..15 structurally zero (zero-padded FFTLog input, n = NP/2): 8 instead of 16 additions each
CP_HD void dft16_inner_zero_padded(cplx* x) {
    dft4_mid(x[0], x[4], x[8], x[12]);
    dft4_mid(x[1], x[5], x[9], x[13]);
    dft4_mid(x[2], x[6], x[10], x[14]);
    dft4_mid(x[3], x[7], x[11], x[15]);
}

template <>
struct Dft<16> {
    static CP_HD void run(cplx* x) {
        dft16_inner(x);
        dft16_outer(x);
    }
};

// radix-16 butterfly whose points 0..3 and 12..15 are structural zeros (zero-padded FFTLog input,
// n = NP/2): every inner DFT4 sees (0, x1, x2, 0) and costs 8 instead of 16 additions.
struct Dft16ZeroPadded {
    static CP_HD void run(cplx* x) {
        dft16_inner_zero_padded(x);
        dft16_outer(x);
    }
};

// radix-8 butterfly whose points 0, 1, 6, 7 are structural zeros (zero-padded input at P = 8): both inner DFT4s
// (even and odd points) see (0, a, b, 0)
struct Dft8ZeroPadded {
    static CP_HD void run(cplx* x) {
        cplx e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
        cplx o0 = x[1], o1 = x[3], o2 = x[5], o3 = x[7];
        dft4_mid(e0, e1, e2, e3);
        dft4_mid(o0, o1, o2, o3);
        mul_w16<2>(o1);
        mul_w16<4>(o2);
        mul_w16<6>(o3);
        x[0].re = e0.re + o0.re; x[0].im = e0.im + o0.im; x[4].re = e0.re - o0.re; x[4].im = e0.im - o0.im;
        x[1].re = e1.re + o1.re; x[1].im = e1.im + o1.im; x[5].re = e1.re - o1.re; x[5].im = e1.im - o1.im;
        x[2].re = e2.re + o2.re; x[2].im = e2.im + o2.im; x[6].re = e2.re - o2.re; x[6].im = e2.im - o2.im;
        x[3].re = e3.re + o3.re; x[3].im = e3.im + o3.im; x[7].re = e3.re - o3.re; x[7].im = e3.im - o3.im;
    }
};

// ---- one pass over the thread's P points ---------------------------------------------------------
// Pass I works on sub-FFTs of length L = len(I) with radix R: butterfly beta = (block b, offset j),
// j < M = L / R, touches elements b L + j + M r, r < R.  A thread owns butterflies t + T i.
// Twiddle w_L^(j s) sits at tw[s * M + j] (pass-local table).  DIF: butterfly then twiddle on
// outputs; DIT (the transposed network): twiddle on inputs then butterfly.
template <int NP, int P, int I>
struct Pass {
    using PL = Plan<NP, P>;
    static constexpr int R = PL::radix(I);
    static constexpr int L = PL::len(I);
    static constexpr int M = L / R;
    static constexpr int NB = P / R;
    static constexpr int T = PL::T;

    static CP_HD int elem(int t, int i, int r) {
        const int beta = t + T * i;
        const int b = beta / M, j = beta % M;
        return b * L + j + M * r;
    }
    static CP_HD int joff(int t, int i) { return (t + T * i) % M; }

    // Byte offset of the swizzled slot of element (t, i, r).  The radix-16 shapes reduce to ONE integer instruction per
    // access (or none): the swizzle XORs bits 4..7 of the slot index into bits 0..3, and
    //   M % 256 == 0 : bits 4..7 come from j only           -> base(t, i) + r * 16 M        (an immediate offset)
    //   M == R == 16 : bits 4..7 of the slot are r          -> base(t, i) ^ (r * 0x110)
    //   M == 1, R 16 : bits 4..7 are beta & 15              -> base(t, i) ^ (r * 16)
    // The thread's base address goes through an empty asm so that the compiler cannot re-associate base + buffer address +
    // point offset: the point offset stays the outermost constant addend and lands in the offset field of ds_read / ds_write.
    static CP_HD unsigned opaque_base(unsigned base) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm("" : "+v"(base));
#endif
        return base;
    }
    static CP_HD unsigned lds_base(int t, int i) {
        const unsigned beta = (unsigned)(t + T * i);
        if (padded_lds(NP, P)) {
            return (unsigned)swz<NP, P>(elem(t, i, 0)) * 16u;
        } else if (NP >= 256 && P == 16 && M % 256 == 0) {
            const unsigned b = beta / M, j = beta % M;
            return (b * L + (j ^ ((j >> 4) & 15u))) * 16u;
        } else if (NP >= 256 && M == 16 && R == 16) {
            const unsigned b = beta / M, j = beta % M;
            return b * (L * 16u) + j * 16u;
        } else if (NP >= 256 && M == 1 && R == 16) {
            return beta * 256u + (beta & 15u) * 16u;
        }
        return (unsigned)swz<NP, P>(elem(t, i, 0)) * 16u;
    }
    // base = lds_base(t, i) + the buffer's LDS address l0
    static CP_HD unsigned lds_off(int t, int i, int r, unsigned base, unsigned l0) {
        // padded layout: the point index is a base-16 digit of its own, so its slot offset simply adds (an immediate)
        if (padded_lds(NP, P)) return base + (unsigned)swz<NP, P>(M * r) * 16u;
        if (NP >= 256 && P == 16 && M % 256 == 0) return base + (unsigned)r * (M * 16u);
        if (NP >= 256 && M == 16 && R == 16) return base ^ ((unsigned)r * 0x110u);
        if (NP >= 256 && M == 1 && R == 16) return base ^ ((unsigned)r * 16u);
        // any other shape: element = elem(t, i, 0) | M r (disjoint bits), so by linearity of the swizzle the address is
        // base ^ (low bits of a constant) + (its bits from 4096 up, which the swizzle leaves alone: an immediate offset)
        PointConst c = point_const(r);
        return (base ^ c.lo) + c.hi;
    }
    struct PointConst {
        unsigned lo, hi;
    };
    static CP_HD PointConst point_const(int r) {
        const unsigned c = (unsigned)swz<NP, P>(M * r) * 16u;
        return PointConst{c & 0xFFFu, c & ~0xFFFu};
    }
    // order in which a radix-16 butterfly consumes its points: (0, 8, 4, 12), (1, 9, 5, 13), ...; the LDS reads are issued
    // in that order so that the first additions can start after two of them
    static CP_HD int consume_order(int k) { return R == 16 ? (k >> 2) + 8 * (k & 1) + 4 * ((k >> 1) & 1) : k; }
    static CP_HD void load_lds(int t, const cplx* lds, cplx* x) {
        const LdsView v(lds);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned base = opaque_base(lds_base(t, i) + v.l0);
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int r = consume_order(k);
                if (CP_ABLATE & 1) x[i * R + r] = cplx{1e-3 * t + r, 1. + i};
                else x[i * R + r] = v.read(lds_off(t, i, r, base, v.l0));
            }
        }
    }
    static CP_HD void store_lds(int t, cplx* lds, const cplx* x) {
        const LdsView v(lds);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned base = opaque_base(lds_base(t, i) + v.l0);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (CP_ABLATE & 1) {
                    if (x[i * R + r].re == 1.2345e301) lds[t] = x[i * R + r];  // keeps x alive, never taken
                } else {
                    v.write(lds_off(t, i, r, base, v.l0), x[i * R + r]);
                }
            }
        }
    }
    static CP_HD void twiddle(int t, const cplx* tw, cplx* x) {
        if (M == 1) return;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = joff(t, i);
#pragma unroll
            for (int s = 1; s < R; ++s) x[i * R + s] = cmul(x[i * R + s], tw[s * M + j]);
        }
    }
    // the same in two steps (w holds P entries; slot i R + s), so that a caller can order the table loads
    // against other memory operations
    // SKIP: the first SKIP twiddles (s = 1 .. SKIP) are held elsewhere (Fftlog::State::wpin) and not loaded
    template <int SKIP = 0>
    static CP_HD void twiddle_load(int t, const cplx* tw, cplx* w) {
        if (M == 1) return;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = joff(t, i);
#pragma unroll
            for (int s = 1 + SKIP; s < R; ++s) {
                if (CP_ABLATE & 4) w[i * R + s] = cplx{1. + 1e-9 * t, 0.5 + s};
                else w[i * R + s] = ld_cplx(tw, (unsigned)j * 16u, (unsigned)(s * M) * 16u);
            }
        }
    }
    // twiddles of the middle passes come from a copy of the pass table in LDS (a few KB, loaded once per
    // workgroup): ~10x lower latency than L2 and no traffic on the vector-memory path
    static CP_HD void twiddle_apply_lds(int t, const cplx* ltw, cplx* x) {
        if (M == 1) return;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = joff(t, i);
#pragma unroll
            for (int s = 1; s < R; ++s) {
                cplx w;
                if (CP_ABLATE & 4) w = cplx{1. + 1e-9 * t, 0.5 + s};
                else w = ltw[s * M + j];
                if (CP_ABLATE & 32) x[i * R + s].re += w.re + w.im;
                else x[i * R + s] = cmul(x[i * R + s], w);
            }
        }
    }
    // ... or in two steps like twiddle_load / twiddle_apply: all reads issued up front (behind the data reads of the
    // phase), so that their latency hides under the butterflies instead of stalling every twiddle multiplication
    static CP_HD void twiddle_load_lds(int t, const cplx* ltw, cplx* w) {
        if (M == 1) return;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = joff(t, i);
#pragma unroll
            for (int s = 1; s < R; ++s) {
                if (CP_ABLATE & 4) w[i * R + s] = cplx{1. + 1e-9 * t, 0.5 + s};
                else w[i * R + s] = ltw[s * M + j];
            }
        }
    }
    static CP_HD void twiddle_apply(const cplx* w, cplx* x) {
        if (M == 1) return;
        if (CP_ABLATE & 32) {
            x[1].re += w[1].re + w[R - 1].im;
            return;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int s = 1; s < R; ++s) x[i * R + s] = cmul(x[i * R + s], w[i * R + s]);
    }
    static CP_HD void butterflies(cplx* x) {
        if (CP_ABLATE & 32) return;
#pragma unroll
        for (int i = 0; i < NB; ++i) Dft<R>::run(x + i * R);
    }
    // Butterflies (+ the DIF twiddles w on the outputs when TW) + the LDS writes, in four stages for the single radix-16
    // shape: each outer DFT4 is followed at once by the twiddles and the writes of its four outputs, so the LDS pipe
    // drains under the arithmetic of the next stage instead of in one burst in front of the barrier.
    template <bool ZERO_PADDED, bool TW>
    static CP_HD void butterflies_store(int t, const cplx* w, cplx* lds, cplx* x) {
        if constexpr (R == 16 && NB == 1 && CP_ABLATE == 0) {
            const LdsView v(lds);
            const unsigned base = opaque_base(lds_base(t, 0) + v.l0);
            if (ZERO_PADDED) dft16_inner_zero_padded(x);
            else dft16_inner(x);
            stage_store<0, TW>(t, w, v, base, x);
            stage_store<1, TW>(t, w, v, base, x);
            stage_store<2, TW>(t, w, v, base, x);
            stage_store<3, TW>(t, w, v, base, x);
        } else if constexpr (R == 8 && NB == 1 && CP_ABLATE == 0) {
            // radix 8 = 2 x 4: outputs k and k + 4 come from e_k +- o_k; four stages of two writes
            const LdsView v(lds);
            const unsigned base = opaque_base(lds_base(t, 0) + v.l0);
            cplx e[4] = {x[0], x[2], x[4], x[6]}, o[4] = {x[1], x[3], x[5], x[7]};
            if (ZERO_PADDED) {
                dft4_mid(e[0], e[1], e[2], e[3]);
                dft4_mid(o[0], o[1], o[2], o[3]);
            } else {
                dft4(e[0], e[1], e[2], e[3]);
                dft4(o[0], o[1], o[2], o[3]);
            }
            mul_w16<2>(o[1]);
            mul_w16<4>(o[2]);
            mul_w16<6>(o[3]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                cplx lo{e[k].re + o[k].re, e[k].im + o[k].im}, hi{e[k].re - o[k].re, e[k].im - o[k].im};
                if (TW && M > 1) {
                    if (k > 0) lo = cmul(lo, w[k]);
                    hi = cmul(hi, w[k + 4]);
                }
                v.write(lds_off(t, 0, k, base, v.l0), lo);
                v.write(lds_off(t, 0, k + 4, base, v.l0), hi);
                CP_SCHED_FENCE();
            }
        } else {
            if (ZERO_PADDED && R == 16) Dft16ZeroPadded::run(x);
            else if (ZERO_PADDED && R == 8) Dft8ZeroPadded::run(x);
            else butterflies(x);
            if (TW) twiddle_apply(w, x);
            store_lds(t, lds, x);
        }
    }
    template <int S0, bool TW>
    static CP_HD void stage_store(int t, const cplx* w, const LdsView& v, unsigned base, cplx* x) {
        dft16_outer_stage<S0>(x);
#pragma unroll
        for (int s1 = 0; s1 < 4; ++s1) {
            const int s = S0 + 4 * s1;  // output index of slot 4 S0 + s1
            cplx y = x[4 * S0 + s1];
            if (TW && M > 1 && s > 0) y = cmul(y, w[s]);
            v.write(lds_off(t, 0, s, base, v.l0), y);
        }
        CP_SCHED_FENCE();
    }
    static CP_HD void dif(int t, const cplx* tw, cplx* x) {
        butterflies(x);
        twiddle(t, tw + PL::tw_offset(I), x);
    }
    static CP_HD void dit(int t, const cplx* tw, cplx* x) {
        twiddle(t, tw + PL::tw_offset(I), x);
        butterflies(x);
    }
};

// True when every element a thread reads in pass IR was written (in pass IW, in place) by a thread of the SAME wave: the
// exchange between the two passes then needs no workgroup barrier, because the LDS executes the instructions of one wave in
// order.  (NP = 4096, P = 16: passes 1 <-> 2 trade data inside groups of 16 consecutive threads.)
#ifndef CP_WAVE_LOCAL
#define CP_WAVE_LOCAL 1
#endif
template <int NP, int P>
constexpr bool exchange_is_wave_local(int iw, int ir) {
    using PL = Plan<NP, P>;
    if (!CP_WAVE_LOCAL) return false;
    const int T = PL::T;
    const int Lw = PL::len(iw), Mw = Lw / PL::radix(iw);
    const int Lr = PL::len(ir), Rr = PL::radix(ir), Mr = Lr / Rr;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < P / Rr; ++i)
            for (int r = 0; r < Rr; ++r) {
                const int beta = t + T * i;
                const int e = (beta / Mr) * Lr + beta % Mr + Mr * r;   // Pass<NP, P, ir>::elem(t, i, r)
                const int writer = ((e / Lw) * Mw + e % Mw) % T;        // thread whose pass-iw butterfly holds e
                if (writer / 64 != t / 64) return false;
            }
    return true;
}

// Frequency index held at LDS position pos after the full DIF network (digit reversal for the
// mixed-radix plan).  Host-side only (used to lay out U).
template <int NP, int P>
inline int dif_freq_of_pos(int pos) {
    using PL = Plan<NP, P>;
    int k = 0, mult = 1, rem = pos;
    for (int i = 0; i < PL::NPASS; ++i) {
        const int R = PL::radix(i), M = PL::len(i) / R;
        const int s = rem / M;
        rem = rem % M;
        k += s * mult;
        mult *= R;
    }
    return k;
}

}  // namespace cpfft
