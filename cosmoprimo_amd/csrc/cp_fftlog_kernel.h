// cp_fftlog_kernel.h -- the __global__ fused FFTLog kernel and its per-size launcher table.
// Included by cp_fftlog_inst.hip (one translation unit per size group, see Makefile) and cp_fftlog.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "cp_fftlog_body.h"
#include "cp_fftlog_dispatch.h"
#include "cp_fftlog_tables.h"

namespace cpfft {

struct Launcher {
    // [variant]: kernel entry (nullptr when the variant does not exist for this size)
    const void* func[VAR_COUNT];
    void (*launch)(int variant, const FftlogArgs&, int grid, hipStream_t stream);
    int np, p, block, lds_bytes;
    void (*build_tw)(std::vector<cplx>&);
    void (*build_u)(const double*, cplx*);
};

// per size group (defined in cp_fftlog_inst.hip compiled with -DCP_INST_GROUP=g)
bool find_launcher_g0(int npad, Launcher* out);
bool find_launcher_g1(int npad, Launcher* out);
bool find_launcher_g2(int npad, Launcher* out);
bool find_launcher_g3(int npad, Launcher* out);
bool find_launcher_g4(int npad, Launcher* out);

#if defined(__HIPCC__)
#if defined(CP_STAMPS)
// diagnostic build (tools/fftlog_microbench.hip -DCP_STAMPS): per-wave cycle sums of each phase's work and barrier wait
// (cp_stamp: cp_fft_core.h)
#define CP_STAMP_DECL , unsigned long long* cp_stamp_acc
#define CP_STAMP_ARG , cp_stamp_acc
#else
#define CP_STAMP_DECL
#define CP_STAMP_ARG
#endif
template <int NP, int P, int IM, int OM, int PH>
__device__ __forceinline__ void run_phases(int t, const FftlogArgs& A, const double* ra, const double* rb, double* oa, double* ob, bool has_b,
                                           int ker, cplx* lds, const double* nra, const double* nrb, int nxt_ker,
                                           typename Fftlog<NP, P, IM, OM>::State& st CP_STAMP_DECL) {
    using F = Fftlog<NP, P, IM, OM>;
#if defined(CP_STAMPS)
    const unsigned long long s0 = cp_stamp();
#endif
    F::template phase<PH>(t, A, ra, rb, oa, ob, has_b, ker, lds, nra, nrb, nxt_ker, st);
#if defined(CP_STAMPS)
    const unsigned long long s1 = cp_stamp();
    cp_stamp_acc[2 * PH] += s1 - s0;
#endif
    if constexpr (PH + 1 < F::NPH) {
        if constexpr (F::template barrier_free_after<PH>()) {
            // the next phase reads only what this wave has just written (in place), and the LDS executes one wave's
            // instructions in order: no barrier, the waves of the workgroup drift apart and spread their LDS bursts
            asm volatile("" ::: "memory");
        } else if (!(CP_ABLATE & 2)) {
            __syncthreads();
        }
#if defined(CP_STAMPS)
        cp_stamp_acc[2 * PH + 1] += cp_stamp() - s1;
#endif
        run_phases<NP, P, IM, OM, PH + 1>(t, A, ra, rb, oa, ob, has_b, ker, lds, nra, nrb, nxt_ker, st CP_STAMP_ARG);
    }
}

// Rows of pair p = (q, ker): batch items (2 q, 2 q + 1) of kernel `ker`, p = q nker + ker; an incomplete last pair (odd nbatch,
// q = nhalf - 1) aliases row b to row a.  A workgroup walks p, p + G, p + 2 G, ... (G = gridDim.x) incrementally: ker and the
// element offsets of row a advance by constants computed once per workgroup.  The direct form costs a 64-bit division and six
// 64-bit multiplications in scalar instructions per pair and per wave (~250 instructions); the walk is kept small (32-bit steps)
// because scalar registers spilled to VGPR lanes inside the pair loop cost more than the arithmetic they save.
struct PairRows {
    long long in_off, out_off;  // element offsets of row a in `in` / `out`
    int ker;
    int has_b;  // int, not bool: no padding bytes (a padded struct copy goes through scratch memory)
};

struct PairWalk {
    PairRows cur;
    long long p, p_last_q;              // pair index; first pair index with q = nhalf - 1
    unsigned in_step, out_step;         // per step of G pairs, without the wrap of ker    (G (2 n) < 2^31: see cp_fftlog.hip)
    unsigned in_b, out_b;               // row b - row a = nker rows; also the extra step when ker wraps around nker
    int dker, odd;                      // G mod nker; nbatch odd
};

__device__ __forceinline__ void pair_walk_flags(const FftlogArgs& A, PairWalk& w) {
    w.cur.has_b = !(w.odd && w.p >= w.p_last_q);
}

__device__ __forceinline__ PairWalk pair_walk_begin(const FftlogArgs& A, long long p, unsigned G) {
    PairWalk w;
    const long long q = p / A.nker;
    const unsigned dq = G / (unsigned)A.nker;
    w.p = p;
    w.cur.ker = (int)(p - q * A.nker);
    w.dker = (int)(G - dq * (unsigned)A.nker);
    const long long row = 2 * q * A.nker + w.cur.ker;
    const unsigned drow = 2u * dq * (unsigned)A.nker + (unsigned)w.dker;
    w.cur.in_off = row * A.n;
    w.cur.out_off = row * A.n_out;
    w.in_step = drow * (unsigned)A.n;
    w.out_step = drow * (unsigned)A.n_out;
    w.in_b = (unsigned)A.nker * (unsigned)A.n;
    w.out_b = (unsigned)A.nker * (unsigned)A.n_out;
    w.odd = (int)(A.nbatch & 1);
    w.p_last_q = ((A.nbatch + 1) / 2 - 1) * A.nker;
    pair_walk_flags(A, w);
    return w;
}

__device__ __forceinline__ void pair_walk_next(const FftlogArgs& A, PairWalk& w, unsigned G) {
    w.p += G;
    w.cur.ker += w.dker;
    w.cur.in_off += w.in_step;
    w.cur.out_off += w.out_step;
    if (w.cur.ker >= A.nker) {  // q advances by one more: + 2 nker - nker rows
        w.cur.ker -= A.nker;
        w.cur.in_off += w.in_b;
        w.cur.out_off += w.out_b;
    }
    pair_walk_flags(A, w);
}

// One workgroup = T threads = one packed pair of rows per loop iteration (persistent over pairs).
// Occupancy target: the LDS footprint (16 NP bytes per workgroup) allows 2 workgroups per CU at
// NP = 4096, i.e. 2 waves per SIMD, so the register budget is 256 VGPR+AGPR per lane.
#ifndef CP_WAVES_PER_SIMD
#define CP_WAVES_PER_SIMD 2
#endif
template <int NP, int P, int IM, int OM>
__global__ __launch_bounds__(NP / P, CP_WAVES_PER_SIMD) void fftlog_kernel(const FftlogArgs A) {
    extern __shared__ __attribute__((aligned(4096))) char smem[];  // 4096: see LdsView (cp_fft_core.h)
    cplx* lds = reinterpret_cast<cplx*>(smem);
    using F = Fftlog<NP, P, IM, OM>;
    const int t = threadIdx.x;
    const long long nhalf = (A.nbatch + 1) / 2;
    const long long npairs = nhalf * A.nker;
    typename F::State st;  // tables loaded one phase ahead + prefetched rows (cp_fftlog_body.h)
    long long p = blockIdx.x;
    if (p >= npairs) return;
    PairWalk walk = pair_walk_begin(A, p, gridDim.x);
    PairRows cur = walk.cur;
    {
        const double* ra = A.in + cur.in_off;
        F::init_state(t, A, ra, ra + (cur.has_b ? walk.in_b : 0u), cur.ker, st);
    }
    if constexpr (F::NPASS > 1) {
        F::fill_lds_tables(t, A, lds);
        // row screening of the first pair (HALF variants; later pairs are screened a pair ahead inside the loop)
        F::screen_prefetched(t, st.t0, A, cur.ker, lds, st);
        __syncthreads();
        if constexpr (F::SCREEN_AHEAD) st.info_nxt = F::screen_collect(lds);
    }
    // Drain the one-off loads here.  Otherwise the compiler's wait-count merge at the loop head must also cover this
    // entry path (where the row prefetch is the YOUNGEST operation) and emits vmcnt(0) at the top of every pair, which
    // makes each pair wait for the previous pair's stores to be acknowledged by memory.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
#if defined(CP_START_STAGGER)      // measurements (tools/mb_stagger.sh): the workgroups that share a CU start CP_START_STAGGER x s_sleep(64) apart
    for (int i = 0, n = (int)((blockIdx.x / CP_STAGGER_DIV) % CP_STAGGER_MOD) * CP_START_STAGGER; i < n; ++i) __builtin_amdgcn_s_sleep(64);
#endif
#if defined(CP_STAMPS)
    unsigned long long cp_stamp_acc[2 * F::NPH] = {0};
    for (int i = 0; i < 8; ++i) st.fs[i] = 0;
    const unsigned long long cp_t_begin = cp_stamp();
#endif
    for (;;) {
        const bool more = walk.p + gridDim.x < npairs;
        if (more) pair_walk_next(A, walk, gridDim.x);
        const PairRows nxt = walk.cur;  // == cur on the last pair: its prefetch re-reads the rows it already has
        const double* ra = A.in + cur.in_off;
        double* oa = A.out + cur.out_off;
        const double* nra = A.in + nxt.in_off;
        run_phases<NP, P, IM, OM, 0>(t, A, ra, ra + (cur.has_b ? walk.in_b : 0u), oa, oa + (cur.has_b ? walk.out_b : 0u), cur.has_b, cur.ker,
                                     lds, nra, nra + (nxt.has_b ? walk.in_b : 0u), nxt.ker, st CP_STAMP_ARG);
        if (!more) break;
#if defined(CP_STAMPS)
        const unsigned long long sb = cp_stamp();
#endif
        // no barrier here: the one that protects LDS against the next pair's writes sits in the last phase, right behind
        // its LDS reads (Fftlog::phase), where the waves have just left the previous barrier and are still in step
#if defined(CP_STAMPS)
        cp_stamp_acc[2 * F::NPH - 1] += cp_stamp() - sb;
#endif
        cur = nxt;
    }
#if defined(CP_STAMPS)
    if ((threadIdx.x & 63) == 0) {  // the stamp buffer is aliased onto the (unused) tail of A.post by the microbench
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(const_cast<double*>(A.val_stamp)) +
                                  ((size_t)blockIdx.x * (NP / P / 64) + threadIdx.x / 64) * (2 * F::NPH + 1 + 8);
        for (int i = 0; i < 2 * F::NPH; ++i) dst[i] = cp_stamp_acc[i];
        dst[2 * F::NPH] = cp_stamp() - cp_t_begin;
        for (int i = 0; i < 8; ++i) dst[2 * F::NPH + 1 + i] = st.fs[i];
    }
#endif
}

template <int NP, int P>
constexpr bool has_half() {
    return (P == 16 || P == 8) && NP >= CP_FFTLOG_HALF_MIN_NP;
}

template <int NP, int P>
void launch_impl(int variant, const FftlogArgs& A, int grid, hipStream_t stream) {
    constexpr int T = Plan<NP, P>::T;
    constexpr int lds = Fftlog<NP, P>::LDS_BYTES;
    if constexpr (has_half<NP, P>()) {
        if (variant == VAR_HALF_ZERO) {
            hipLaunchKernelGGL((fftlog_kernel<NP, P, IN_HALF_ZERO, OUT_HALF>), dim3(grid), dim3(T), lds, stream, A);
            return;
        }
        if (variant == VAR_HALF) {
            hipLaunchKernelGGL((fftlog_kernel<NP, P, IN_HALF, OUT_HALF>), dim3(grid), dim3(T), lds, stream, A);
            return;
        }
        if (variant == VAR_HALF_ZERO_WINDOW) {
            hipLaunchKernelGGL((fftlog_kernel<NP, P, IN_HALF_ZERO, OUT_HALF_WINDOW>), dim3(grid), dim3(T), lds, stream, A);
            return;
        }
    }
    if (variant == VAR_LOG)
        hipLaunchKernelGGL((fftlog_kernel<NP, P, IN_LOG, OUT_GENERIC>), dim3(grid), dim3(T), lds, stream, A);
    else
        hipLaunchKernelGGL((fftlog_kernel<NP, P, IN_GENERIC, OUT_GENERIC>), dim3(grid), dim3(T), lds, stream, A);
}

template <int NP, int P>
Launcher make_launcher() {
    Launcher l;
    for (int v = 0; v < VAR_COUNT; ++v) l.func[v] = nullptr;
    l.func[VAR_GENERIC] = reinterpret_cast<const void*>(&fftlog_kernel<NP, P, IN_GENERIC, OUT_GENERIC>);
    l.func[VAR_LOG] = reinterpret_cast<const void*>(&fftlog_kernel<NP, P, IN_LOG, OUT_GENERIC>);
    if constexpr (has_half<NP, P>()) {
        l.func[VAR_HALF] = reinterpret_cast<const void*>(&fftlog_kernel<NP, P, IN_HALF, OUT_HALF>);
        l.func[VAR_HALF_ZERO] = reinterpret_cast<const void*>(&fftlog_kernel<NP, P, IN_HALF_ZERO, OUT_HALF>);
        l.func[VAR_HALF_ZERO_WINDOW] = reinterpret_cast<const void*>(&fftlog_kernel<NP, P, IN_HALF_ZERO, OUT_HALF_WINDOW>);
    }
    l.launch = &launch_impl<NP, P>;
    l.np = NP;
    l.p = P;
    l.block = Plan<NP, P>::T;
    l.lds_bytes = Fftlog<NP, P>::LDS_BYTES;
    l.build_tw = &build_twiddles<NP, P>;
    l.build_u = &build_u_layout<NP, P>;
    return l;
}
#endif  // __HIPCC__

}  // namespace cpfft
