// cp_wallish_tail.h -- wallish2018 behind its forward transform as ONE kernel (reference bao_filter.py:373-431).  Part of cp_dst.hip (included inside its
// anonymous namespace, behind the transform kernels whose pass machinery it uses); the C entry point cp_wallish_tail is at the end of that file.
//
// The three steps that follow the forward sine transform -- second derivatives of the two coefficient sequences, the box between their maxima and its
// removal (:373-405: wallish_dd_box_kernel), the inverse transform with exp(.) / k_lin (:407-413: dst_kernel<..., true>) and the clamped spline through
// the spliced knots at the filter's wavenumbers with the damping (:415-431: splice_uniform_kernel) -- each read what the one before had written:
// 34 + 34 + 33 KB read and 33 KB written per vector for 16 KB that have to move (the 8 KB row of P in, the 8 KB result out), at 5 500 + 2 733 + 1 073
// vector instructions per vector (profiles/r4y_config4_traffic.json, r4y_config4_valu.json).  Here a workgroup takes a PAIR of vectors through all of it
// on the CU: the pair's four sequences of 2048 coefficients fill the 64 KB data region of the 4096-point transform (XOR layout of cp_wallish_dd.h), a wave
// each; the boxes are rewritten there (and in memory, where the filter keeps the sequences); the packed spectrum of the inverse transform is formed
// from LDS; the transformed pair goes back into the same region as two real rows in natural order, on which waves 0 and 1 run the recursions of the
// spliced spline (cp_splice_uniform.h: a lane owns SU consecutive knots of the uniform stretch, its second derivatives overwrite them) and evaluate
// the queries, while waves 2 and 3 copy the columns that do not go through the spline.  Per vector: 32 KB of coefficients + 8 KB of P in, 8 KB out.
struct TailArgs {
    long long nrows;
    double* coef;              // (nrows, 4096) coefficient rows, split layout; the boxes are rewritten in place (what cp_wallish_dd_box leaves there)
    const cplx* tw;            // the transform plan: twiddles, e^{-i pi k / 2N}, 1 / k_lin
    const cplx* rot;
    const double* ikx;
    cpsu::Tables U;            // the uniform-stretch scheme of the splice plan: stretch from the transformed rows (src_u = 1), outer knots from P (array 0)
    const double* pk;          // (nrows, nq)
    const double* tophat;      // (nq) or null
    double* out;               // (nrows, nq)
    int* box;                  // (2 nrows, 2)
    int margin_first, margin_second, off0, off1;
};

#ifndef CP_TAIL_ABLATE      // diagnostic builds (wrong results): 1 no second derivatives / box, 2 no transform, 4 no exponential, 8 no splice, 16 no stores of the result
#define CP_TAIL_ABLATE 0
#endif
#ifndef CP_TAIL_ROT_BATCH      // 1: the rotations of the two spectrum stages requested sixteen at a time instead of where they are used (the compiler keeps ~4
#define CP_TAIL_ROT_BATCH 0    // in flight): measured, 9.72 against 9.63 ms per 125 000 vectors (profiles/r6_geospline_prefilter.txt) -- off
#endif
#ifndef CP_TAIL_KERNARG_RELOAD      // 0: the arguments as an ordinary by-value parameter, live over the whole loop (measurements)
#define CP_TAIL_KERNARG_RELOAD 1
#endif

// The kernel's arguments are read where they are used, through the constant address space from a pointer the compiler cannot see through: held in
// scalar registers for the whole loop over pairs they (and what is derived from them) were ~130 values for 102 registers, and every use of a spilled
// one is a v_readlane -- a quarter of the vector instructions of the spline step, a tenth of the kernel's.
typedef const TailArgs __attribute__((address_space(4))) * TailArgsK;
__device__ __forceinline__ TailArgsK tail_args() {
    TailArgsK p = (TailArgsK)__builtin_amdgcn_kernarg_segment_ptr();      // the kernel's only explicit argument: offset 0 of the segment
    if (CP_TAIL_KERNARG_RELOAD) asm volatile("" : "+s"(p));
    return p;
}

// the queries that go through the spline: v = part + w.z M_j + w.w M_{j+1}, then pk / ((pk / pknow - 1) tophat + 1) as splice_uniform_kernel forms it
__device__ __forceinline__ void tail_evaluate(const double* yu, double* outrow, int lane, const double* part, const double* pkrow) {
    TailArgsK g = tail_args();
    const int ngb = g->U.ngb, gb0 = g->U.gb0, nq = g->U.nq, gfirst = g->U.gfirst, gend = g->U.gend;
    const int* qe = g->U.qe;
    const double4* qw = reinterpret_cast<const double4*>(g->U.qw);
    const double* tophat = g->tophat;
    // Every block of 64 queries requests its plan entries and the row's own values FIRST, all blocks together, then they are consumed: written as one loop
    // (request, use, store per block, the blocks behind wave-uniform branches) each block waited for its own loads -- two memory round trips, the second
    // one (the window) behind the first -- and, the memory counter retiring in order, for the store of the block before: sixteen round trips in a row.
    // Blocks past the last one request the entries of block 0 (valid addresses, results unused).
    int jv[cpsu::NGB];
    double wz[cpsu::NGB], ww[cpsu::NGB], pv[cpsu::NGB], tv[cpsu::NGB];
#pragma unroll
    for (int e = 0; e < cpsu::NGB; ++e) {
        const int slot = 64 * (e < ngb ? e : 0) + lane, q = 64 * gb0 + slot, qc = q < nq ? q : 0;
        jv[e] = qe[slot];
        const double2 w = *reinterpret_cast<const double2*>(reinterpret_cast<const double*>(qw + slot) + 2);
        wz[e] = w.x;
        ww[e] = w.y;
        pv[e] = pkrow[qc];      // the row's own value at the query
        tv[e] = tophat ? tophat[qc] : 0.;
    }
#pragma unroll
    for (int e = 0; e < cpsu::NGB; ++e) {
        if (e < ngb) {
            const int slot = 64 * e + lane, q = 64 * gb0 + slot;
            const int j = jv[e];
            double v = part[e] + fma(wz[e], yu[j], ww[e] * yu[j + 1]);
            const double p = q < nq ? pv[e] : 0.;
            if (tophat) v = p * (v * cpmath::recip(fma(p - v, (q < nq ? tv[e] : 0.), v)));
            if (q >= gfirst && q < gend && !(CP_TAIL_ABLATE & 16)) outrow[q] = v;
        }
    }
}

// the spliced spline of ONE row by one wave: `row` = the transformed row in LDS, natural order (its stretch starts at column col_u; the slot in front of
// the stretch and the slots behind it up to 64 SU belong to the row and are free).  The arithmetic of cpsu::splice_uniform_kernel, statement for statement.
template <int SU>
__device__ __forceinline__ void tail_splice_row(double* row, double* outrow, int lane, const double* pkrow) {
    using cpsu::P;
    using cpsu::WIN_U;
    using cpsu::NGB;
    TailArgsK g0 = tail_args();
    const int nm = g0->U.nm, wl = g0->U.wl;
    double* yu = row + g0->U.col_u;
    // the knots outside the stretch (the row's own values left and right of it), requested with the weights of the junctions below: one wait for both
    // (requested ahead of the stages, with the row's values at the queries, they were eight registers the kernel that evaluates its own spectra does not
    // have through the transform and the exponentials: spilled -- a load, a wait, a scratch store and a scratch reload)
    double gvl = 0., gvr = 0.;
    if (lane < wl) gvl = pkrow[g0->U.col_l + lane];
    if (lane < g0->U.wr) gvr = pkrow[g0->U.col_r + lane];
    // ---- A / p, M_left, B, M_right: weighted sums over the differences of the knots next to the two junctions ----
    double sums[4];
    double gl_last, gr_first;
    {
        const double* win = g0->U.win;
        double ww[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) ww[o] = win[64 * o + lane];
        gl_last = __shfl(gvl, wl > 0 ? wl - 1 : 0);
        gr_first = __shfl(gvr, 0);
        const double ul = yu[lane < WIN_U ? lane : 0], ur = yu[lane < WIN_U ? nm - 1 - lane : 0];
        const double yfirst = yu[0], ylast = yu[nm - 1];
        sums[0] = fma(ww[0], gvl - yfirst, ww[1] * (ul - yfirst));
        sums[1] = fma(ww[2], gvl - gl_last, ww[3] * (ul - gl_last));
        sums[2] = fma(ww[4], ur - ylast, ww[5] * (gvr - ylast));
        sums[3] = fma(ww[6], ur - gr_first, ww[7] * (gvr - gr_first));
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
            for (int o = 0; o < 4; ++o) sums[o] += __shfl_xor(sums[o], off);
    }
    // ---- the queries' knot values, before the second derivatives take their place ----
    double part[NGB];
    {
        const int ngb = g0->U.ngb;
        const int* qe = g0->U.qe;
        const double4* qw = reinterpret_cast<const double4*>(g0->U.qw);
        // (the plan entries of all blocks requested together, then used: block by block behind wave-uniform branches each one waited for its own)
        int jv[NGB];
        double wx[NGB], wy[NGB];
#pragma unroll
        for (int e = 0; e < NGB; ++e) {
            const int slot = 64 * (e < ngb ? e : 0) + lane;
            jv[e] = qe[slot];
            const double2 w = *reinterpret_cast<const double2*>(qw + slot);
            wx[e] = w.x;
            wy[e] = w.y;
        }
#pragma unroll
        for (int e = 0; e < NGB; ++e) {
            part[e] = 0.;
            if (e < ngb) {
                const int j = jv[e];
                const double ya = yu[j], yb = yu[j + 1];      // (j = -1 and j + 1 = nm read slots of the row that hold no knot: replaced)
                part[e] = fma(wx[e], j < 0 ? gl_last : ya, wy[e] * (j + 1 >= nm ? gr_first : yb));
            }
        }
    }
    // ---- the lane's knots: second differences (beyond either end of the stretch its end value again) ----
    const int lane_b = g0->U.lane_b;
    const double m_b = lane == lane_b ? g0->U.mb0 : (lane == lane_b - 1 ? g0->U.mb1 : 0.);
    const double m_a = lane == 0 ? 1. : 0.;
    double g[SU];
    {
        const int base = SU * lane - 1;
        auto yk = [&](int i) { return yu[i < 0 ? 0 : (i > nm - 1 ? nm - 1 : i)]; };
        double y1 = yk(base + 1);
        double dprev = y1 - yk(base);
#pragma unroll
        for (int t = 0; t < SU; ++t) {
            const double y2 = yk(base + t + 2);
            const double dn = y2 - y1;
            g[t] = dn - dprev;
            dprev = dn;
            y1 = y2;
        }
    }
    cp::wave_lds_phase();      // every read of the knot values is done
    {
#pragma unroll
        for (int t = SU - 2; t >= 0; --t) g[t] = fma(P, g[t + 1], g[t]);
        double c = fma(m_b, sums[2], cpdd::dd_from_right(g[0]));      // G at the first knot of the next segment (+ what B adds there)
        double f = 0.;
#pragma unroll
        for (int t = 0; t < SU; ++t) {
            const double e = fma(P, f, g[t]);
            f = t + 1 < SU ? fma(-P, g[t + 1], e) : e;
            g[t] = e;
        }
#pragma unroll
        for (int t = SU - 1; t >= 0; --t) {
            c *= P;
            g[t] += c;
        }
        c = fma(m_a, sums[0], cpdd::dd_from_left(f));                 // F at the last knot of the previous segment (+ A / p in lane 0)
#pragma unroll
        for (int t = 0; t < (SU < 36 ? SU : 36); ++t) {
            c *= P;
            g[t] += c;
        }
        double* mine = yu + SU * lane;
#pragma unroll
        for (int t = 0; t < SU; ++t) mine[t] = g[t];
    }
    cp::wave_lds_phase();
    if (lane == 0) yu[-1] = sums[1];
    if (lane == 1) yu[nm] = sums[3];
    cp::wave_lds_phase();
    tail_evaluate(yu, outrow, lane, part, pkrow);
}

// Stages 2 .. 5 of a pair whose four sequences sit in the data region (XOR layout) with bad_row set for rows that hold a sample that is not finite, behind a
// barrier: second derivatives + box + removal, inverse transform, exp(.) / k_lin, spliced spline + damping.  Shared by wallish_tail_kernel (the sequences
// read from memory) and wallish_full_kernel (produced by the forward transform of the spectra it evaluates).
template <int SU>
__device__ __forceinline__ void tail_stages(long long p, bool has_b, int t, cplx* lds, cplx* ltw, double* dd_tabs, const cpmath::MathTables& mt, int* bad_row) {
    constexpr int N = 4096, P = 16, NS = N / 2, S = 32;
    using PL = Plan<N, P>;
    constexpr int T = PL::T;
    using namespace cpdd;
    const double fn = sqrt(2. / N), fl = sqrt(1. / N);
    const double nan = __builtin_nan("");
    double* seqs = reinterpret_cast<double*>(lds);
        int lane = t;
        asm volatile("" : "+v"(lane));      // (as above: the wave's number and the lane are formed anew for every pair)
        const int wave = __builtin_amdgcn_readfirstlane(lane >> 6);
        lane &= 63;
        const long long myrow = 2 * p + (wave & 1);
        const bool spline_wave = wave < 2 && (wave == 0 || has_b), copy_wave = wave >= 2 && (wave == 2 || has_b);
        // ---- 2. second derivatives, box, box rewritten: wave w <-> sequence w ----
        if ((wave < 2 || has_b) && !(CP_TAIL_ABLATE & 1)) {
            TailArgsK g = tail_args();
            double* buf = seqs + wave * NS;
            const long long row = 2 * p + (wave >> 1), srow = 2 * row + (wave & 1);
            double m[S];
            int first, second;
            second_derivatives_and_box_recursive<Xor32Layout, false>(buf, lane, g->margin_first, g->margin_second, m, first, second);
            const int a = first + g->off0, b = second + g->off1;
            if (lane == 0) {
                int* box = g->box;
                box[2 * srow] = a;
                box[2 * srow + 1] = b;
            }
            bool finite;
            double* coef = g->coef;
            remove_box_parallel<S, Xor32Layout, true>(buf, dd_tabs + DD_NTAB, lane, a, b, coef ? coef + row * N + (wave & 1) * NS : nullptr, &finite);
            if (!finite) bad_row[wave >> 1] = 1;
        }
        __syncthreads();
        // ---- 3. Hermitian-symmetrised, conjugated spectrum of the pair from the sequences (dst_kernel's; coefficient j of a row = knot j >> 1 of its sequence j & 1) ----
        const bool skip_a = bad_row[0] != 0, skip_b = bad_row[1] != 0;
        int tt = t;
        asm volatile("" : "+v"(tt));
        cplx x[P];
        {
            // Where coefficient j of a row sits: sequence j & 1, knot j >> 1 at slot (j >> 1) ^ ((j >> 6) & 31).  With k = tt + 256 r the two coefficients a
            // thread takes, ia = 4095 - k and ib = k - 1, have knots 128 (15 - r) + w and 128 r + w' (w = 127 - (tt >> 1), w' = (tt - 1) >> 1, both below
            // 128), whose slots are a thread constant XOR a constant of r, plus a constant of r -- written out: the index arithmetic of the general
            // expression was three fifths of this stage's vector instructions (38 per coefficient pair)
            const int wa = 127 - (tt >> 1), wb = ((tt + 255) & 255) >> 1;      // (tt = 0: its ib = 256 r - 1 belongs to the column of tt = 256, one r earlier)
            const int base_a = ((tt & 1) ^ 1) * NS + (wa & 96), ha = (wa & 31) ^ (wa >> 5);
            const int base_b = (((tt + 255) & 255) & 1) * NS + (wb & 96), hb = (wb & 31) ^ (wb >> 5);
            const bool first = tt == 0;
            const cplx* rots = tail_args()->rot;
#if CP_TAIL_ROT_BATCH      // the thread's sixteen rotations requested together (the compiler kept ~4 in flight: four memory round trips for the stage instead of one)
            cplx rotv[P];
#pragma unroll
            for (int r = 0; r < P; ++r) rotv[r] = rots[tt + T * r];
            CP_SCHED_FENCE();
#endif
#pragma unroll
            for (int r = 0; r < P; ++r) {
                const int sa = base_a + 128 * (15 - r) + (ha ^ (4 * ((7 - r) & 7)));
                const int rb = (r + 15) & 15;      // the r of ib's column for tt = 0
                const int sb = first ? base_b + 128 * rb + (hb ^ (4 * (rb & 7))) : base_b + 128 * r + (hb ^ (4 * (r & 7)));
                const bool k0 = r == 0 && first;
                const double fa = k0 ? fl : fn;
                const double Aa = fa * seqs[sa], Ab = fa * seqs[2 * NS + sa];
                const double Ba = fa * seqs[sb], Bb = fa * seqs[2 * NS + sb];      // (f_{k-1} = f_n for k >= 1; k = 0 takes f_l and is real)
#if CP_TAIL_ROT_BATCH
                const cplx rot = rotv[r];
#else
                const cplx rot = rots[tt + T * r];
#endif
                const double cs = rot.re, sn = -rot.im;
                cplx Ha = cplx{0.5 * (Aa * cs + Ba * sn), 0.5 * (Aa * sn - Ba * cs)};
                cplx Hb = cplx{0.5 * (Ab * cs + Bb * sn), 0.5 * (Ab * sn - Bb * cs)};
                if (r == 0 && k0) {
                    Ha = cplx{Aa, 0.};
                    Hb = cplx{Ab, 0.};
                }
                if (skip_a) Ha = cplx{0., 0.};
                if (skip_b || !has_b) Hb = cplx{0., 0.};
                x[r].re = Ha.re - Hb.im;
                x[r].im = -(Ha.im + Hb.re);
            }
        }
        __syncthreads();      // every read of the sequences is done: the transform takes the region
        {
            Args A{};      // (the pass machinery of dst_kernel takes its twiddles from here)
            A.tw = tail_args()->tw;
            if (!(CP_TAIL_ABLATE & 2)) dif_all<N, P>(tt, A, x, lds, ltw);
            else {
                Pass<N, P, 0>::store_lds(tt, lds, x);
                __syncthreads();
            }
        }
        asm volatile("" : "+v"(tt));
        // ---- 4. exp(.) / k_lin on the stretch; the pair as two real rows in natural order ----
        {
            double va[P], vb[P];
            TailArgsK g = tail_args();
            const int u_lo = g->U.col_u, u_hi = u_lo + g->U.nm;
            const double* ikx = g->ikx;
            // sample n = tt + 256 s of the pair sits at frequency m = n / 2 (n even) or 4095 - (n - 1) / 2 (n odd) of the network's output, digit-reversed and
            // swizzled: with a = tt >> 1 = a0 + 16 a1 that is slot 256 a0 + 16 a1 + 136 (s & 1) + ((s >> 1) ^ a1) for even tt and its mirror image
            // 256 (15 - a0) + 16 (15 - a1) - 120 (s & 1) + ((s >> 1) ^ a1) for odd tt -- a thread constant, a constant of s and one XOR, where the
            // general expression (pos_of_freq + swz) took fifty integer instructions per sample
            const bool even = (tt & 1) == 0;
            const int a0 = (tt >> 1) & 15, a1 = tt >> 5;
            const int obase = even ? 256 * a0 + 16 * a1 : 256 * (15 - a0) + 16 * (15 - a1), ostep = even ? 136 : -120;
            // (1 / k of the thread's sixteen samples requested together: inside the branch below each one was a memory round trip of its own)
            double ikv[P];
#pragma unroll
            for (int s = 0; s < P; ++s) ikv[s] = ikx[tt + T * s];
#pragma unroll
            for (int s = 0; s < P; ++s) {
                const cplx g = lds[obase + ostep * (s & 1) + ((s >> 1) ^ a1)];
                double ya = even ? g.re : -g.re;
                double yb = even ? -g.im : g.im;
                if (T * s < u_hi && T * s + T > u_lo && !(CP_TAIL_ABLATE & 4)) {      // (uniform over the workgroup: the columns outside the stretch are never looked at)
                    const double ik = ikv[s];
                    ya = cpmath::exp_tab(ya, &mt) * ik;
                    yb = cpmath::exp_tab(yb, &mt) * ik;
                }
                va[s] = skip_a ? nan : ya;
                vb[s] = skip_b ? nan : yb;
            }
            __syncthreads();      // every thread holds its samples: the region is free
#pragma unroll
            for (int s = 0; s < P; ++s) {
                seqs[tt + T * s] = va[s];
                seqs[N + tt + T * s] = vb[s];
            }
        }
        __syncthreads();
        // ---- 5. the spliced spline at the filter's wavenumbers, damping ----
        asm volatile("" : "+v"(lane));
        if (spline_wave && !(CP_TAIL_ABLATE & 8)) {
            TailArgsK g = tail_args();
            tail_splice_row<SU>(seqs + (wave & 1) * N, g->out + myrow * g->U.nq, lane, g->pk + myrow * g->U.nq);
        } else if (copy_wave && !(CP_TAIL_ABLATE & 16)) {
            TailArgsK g = tail_args();
            const int nq = g->U.nq, gfirst = g->U.gfirst, gend = g->U.gend;
            const double* src = g->pk + myrow * nq;
            double* dst = g->out + myrow * nq;
            for (int q0 = 0; q0 < nq; q0 += 256) {      // four loads in flight
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + 64 * u + lane;
                    v[u] = (q < nq && !(q >= gfirst && q < gend)) ? src[q] : 0.;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int q = q0 + 64 * u + lane;
                    if (q < nq && !(q >= gfirst && q < gend)) dst[q] = v[u];
                }
            }
        }
}

template <int SU>
__global__ __launch_bounds__(256, 2) void wallish_tail_kernel(const TailArgs G_by_value) {
    constexpr int N = 4096, P = 16, NS = N / 2;
    using PL = Plan<N, P>;
    constexpr int T = PL::T;
    using namespace cpdd;
    static_assert(!padded_lds(N, P) && lds_data_slots(N, P) == N, "four sequences of 2048 doubles / two real rows of 4096 fill the data region");
    extern __shared__ __attribute__((aligned(4096))) char smem[];
    cplx* lds = reinterpret_cast<cplx*>(smem);
    cplx* ltw = lds + N;
    double* dd_tabs = reinterpret_cast<double*>(ltw + (PL::TW_TOTAL - N));
    const int t = threadIdx.x;
    long long npairs;
    {
        TailArgsK g = tail_args();
        const cplx* tw = g->tw;
        for (int i = t; i < PL::TW_TOTAL - N; i += T) ltw[i] = tw[N + i];
        npairs = (g->nrows + 1) / 2;
    }
    fill_tables(dd_tabs);
    __shared__ int bad_row[2];
    __shared__ cpmath::MathTables mt;      // (the barrier at the top of the first pair covers the fills)
    cpmath::fill_math_tables(&mt);
    double* seqs = reinterpret_cast<double*>(lds);      // sequence w (row w >> 1, parity w & 1) at seqs + w NS, XOR layout; later: row a | row b, natural order
    double na[P], nb[P];
    auto fetch = [&](long long p) {      // the pair's rows as they lie in memory: every coefficient read once
        TailArgsK g = tail_args();
        int tf = t;
        asm volatile("" : "+v"(tf));      // (nothing derived from the thread's number is kept in registers from one pair to the next)
        const double* ra = g->coef + 2 * p * N + tf;
        const double* rb = 2 * p + 1 < g->nrows ? ra + N : ra;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            na[r] = ra[T * r];
            nb[r] = rb[T * r];
        }
    };
    if ((long long)blockIdx.x < npairs) fetch(blockIdx.x);
    for (long long p = blockIdx.x; p < npairs; p += gridDim.x) {
        const bool has_b = 2 * p + 1 < tail_args()->nrows;
        if (t == 0) bad_row[0] = bad_row[1] = 0;
        __syncthreads();      // the data region is free (and the tables are filled)
        // ---- 1. the four sequences into LDS; a row with a sample that is not finite is left out of the transform and comes out as NaN (dst_kernel) ----
        {
            bool bad_a = false, bad_b = false;
            int ts = t;
            asm volatile("" : "+v"(ts));
            // coefficient i = ts + 256 r of a row (split layout: [even-indexed | odd-indexed] coefficients) is knot i & 2047 of sequence i >> 11; its slot
            // (i & 2047) ^ ((i >> 5) & 31) = 256 r' + (ts & 224) + ((ts & 31) ^ (ts >> 5) ^ 8 (r & 3)) with r' = r & 7: a thread constant XOR a constant of r
            const int hs = (ts & 31) ^ (ts >> 5), bs = ts & 224;
#pragma unroll
            for (int r = 0; r < P; ++r) {
                const int slot = (r >> 3) * NS + 256 * (r & 7) + bs + (hs ^ (8 * (r & 3)));
                seqs[slot] = na[r];
                seqs[2 * NS + slot] = nb[r];
                bad_a |= !(fabs(na[r]) <= 1.7976931348623157e308);
                bad_b |= !(fabs(nb[r]) <= 1.7976931348623157e308);
            }
            if (bad_a) bad_row[0] = 1;
            if (bad_b && has_b) bad_row[1] = 1;
        }
        __syncthreads();
        if (p + gridDim.x < npairs) fetch(p + gridDim.x);
        tail_stages<SU>(p, has_b, t, lds, ltw, dd_tabs, mt, bad_row);
    }
}

// ---- wallish2018 of a batch of analytic cosmologies as ONE kernel: dst_generate_kernel's front (the spectra evaluated into the forward transform, cp_dst.hip)
// and the stages above on the same pair, the coefficients never leaving the CU.  Per vector: the 8 KB row of P in, 8 KB out (+ the 32 KB of coefficients
// when the caller wants the rewritten sequences: d_coef).  The two workgroups of a CU are at different points of a pair most of the time: one in the
// evaluation of its 2 x 4096 samples (vector-ALU bound), the other in a transform or a recursion (LDS / latency bound).
struct FullArgs {
    TailArgs tail;      // at offset 0 of the kernel's arguments: read through tail_args()
    GenArgs gen;        // the forward side: parameters of the cosmologies, tables of the evaluation, twiddles (dst.tw, dst.rot); dst.out / box unused
};

#ifndef CP_FULL_KERNARG_RELOAD      // 0: the forward side's arguments as an ordinary by-value parameter, live over the whole loop (measurements)
#define CP_FULL_KERNARG_RELOAD 1
#endif
// ... and gen through gen_args(): the same treatment (tail_args above) for the forward side
__device__ __forceinline__ GenArgsK gen_args() {
    const char __attribute__((address_space(4)))* p = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    GenArgsK g = (GenArgsK)(p + __builtin_offsetof(FullArgs, gen));
    asm volatile("" : "+s"(g));
    return g;
}

template <int SU, int ENGINE>
__global__ __launch_bounds__(256, 2) void wallish_full_kernel(const FullArgs F) {
    constexpr int N = 4096, P = 16, NS = N / 2;
    using PL = Plan<N, P>;
    constexpr int T = PL::T;
    using namespace cpdd;
    static_assert(!padded_lds(N, P) && lds_data_slots(N, P) == N, "the generated samples go through natural-order slots of the data region");
#if CP_FULL_KERNARG_RELOAD
#define CP_GEN() gen_args()
#else
#define CP_GEN() (&F.gen)
#endif
    extern __shared__ __attribute__((aligned(4096))) char smem[];
    cplx* lds = reinterpret_cast<cplx*>(smem);
    cplx* ltw = lds + N;
    double* dd_tabs = reinterpret_cast<double*>(ltw + (PL::TW_TOTAL - N));
    const int t = threadIdx.x;
    {
        const cplx* tw = CP_GEN()->dst.tw;
        for (int i = t; i < PL::TW_TOTAL - N; i += T) ltw[i] = tw[N + i];
    }
    fill_tables(dd_tabs);
    __shared__ int bad_row[2];
    __shared__ cpmath::MathTables mt;      // (the barrier at the top of the first pair covers the fills)
    cpmath::fill_math_tables(&mt);
    const long long npairs = (CP_GEN()->ncosmo + 1) / 2;
    const double fn = sqrt(2. / N), fl = sqrt(1. / N);
    const double nan = __builtin_nan("");
    double* seqs = reinterpret_cast<double*>(lds);
    for (long long p = blockIdx.x; p < npairs; p += gridDim.x) {
        const bool has_b = 2 * p + 1 < CP_GEN()->ncosmo;
        if (t == 0) {
            int zero;
            asm volatile("v_mov_b32 %0, 0" : "=v"(zero));      // (formed here: the compiler kept a pair of zeros in vector registers over the loop, spilled it, and
            bad_row[0] = bad_row[1] = zero;                     // reloaded it here -- a vector-memory wait for the previous pair's stores at the top of every pair)
        }
        __syncthreads();      // the data region is free (and the tables are filled)
        int tt = t;
        asm volatile("" : "+v"(tt));      // (nothing derived from the thread's number is kept in registers from one pair to the next)
        // ---- 0. log(k P_c(k)) of the pair's cosmologies at the 4096 wavenumbers of the linear grid, Makhoul order, into the thread's own slots; forward transform ----
        generate_row<N, P, ENGINE>(CP_GEN(), 2 * p, tt, seqs, cpmath::tables_present(&mt));
        if (has_b) generate_row<N, P, ENGINE>(CP_GEN(), 2 * p + 1, tt, seqs + 1, cpmath::tables_present(&mt));
        {
            cplx x[P];
            bool bad_a = false, bad_b = false;
#pragma unroll
            for (int r = 0; r < P; ++r) {      // the thread's own slots: no barrier
                const int m = tt + T * r;
                const bool lower = m < N / 2;
                const double a = seqs[2 * m], b = has_b ? seqs[2 * m + 1] : 0.;
                bad_a |= !(fabs(a) <= 1.7976931348623157e308);
                bad_b |= !(fabs(b) <= 1.7976931348623157e308);
                x[r].re = lower ? a : -a;
                x[r].im = lower ? b : -b;
            }
            if (bad_a) bad_row[0] = 1;
            if (bad_b) bad_row[1] = 1;
            __syncthreads();      // flags published; every thread has read its slots before the first pass overwrites the region
            const bool skip_a = bad_row[0] != 0, skip_b = bad_row[1] != 0;
            if (skip_a | skip_b) {
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    if (skip_a) x[r].re = 0.;
                    if (skip_b) x[r].im = 0.;
                }
            }
            Args A{};      // (the pass machinery takes its twiddles from here)
            A.tw = CP_GEN()->dst.tw;
            dif_all<N, P>(tt, A, x, lds, ltw);
        }
        asm volatile("" : "+v"(tt));
        // ---- 1. the pair's coefficients (frequency k of the packed transform -> coefficient 4095 - k of either row) into the four sequences, XOR layout ----
        {
            double va[P], vb[P];
            const bool skip_a = bad_row[0] != 0, skip_b = bad_row[1] != 0;
            // frequency k = tt + 256 s sits at slot 256 d0 + 16 d1 + (s ^ d1) (d0, d1: the two low hexadecimal digits of tt), its mirror 4096 - k at
            // 256 e0 + 16 e1 + ((15 - s) ^ e1) with e = 256 - tt (tt = 0: at 16 - s, and at 0 for k = 0)
            const int d1 = tt >> 4, vbase = 256 * (tt & 15) + 16 * d1;
            const int e = 256 - tt, e1 = (e >> 4) & 15, ubase = 256 * (e & 15) + 16 * e1;
            const bool first = tt == 0;
            const cplx* rots = CP_GEN()->dst.rot;
#if CP_TAIL_ROT_BATCH
            cplx rotv[P];
#pragma unroll
            for (int s = 0; s < P; ++s) rotv[s] = rots[tt + T * s];
            CP_SCHED_FENCE();
#endif
#pragma unroll
            for (int s = 0; s < P; ++s) {
                const cplx v = lds[vbase + (s ^ d1)];
                const cplx u = lds[first ? (s == 0 ? 0 : 16 - s) : ubase + ((15 - s) ^ e1)];
#if CP_TAIL_ROT_BATCH
                const cplx rot = rotv[s];
#else
                const cplx rot = rots[tt + T * s];
#endif
                const double f = (s == 0 && first) ? fl : fn;
                va[s] = skip_a ? nan : f * (0.5 * (rot.re * (v.re + u.re) - rot.im * (v.im - u.im)));
                vb[s] = skip_b ? nan : f * (0.5 * (rot.re * (v.im + u.im) - rot.im * (u.re - v.re)));
            }
            __syncthreads();      // every thread has its coefficients: the data region is free
            const int wa = 127 - (tt >> 1);
            const int base_a = ((tt & 1) ^ 1) * NS + (wa & 96), ha = (wa & 31) ^ (wa >> 5);
            double* coef = tail_args()->coef;
#pragma unroll
            for (int s = 0; s < P; ++s) {
                const int slot = base_a + 128 * (15 - s) + (ha ^ (4 * ((7 - s) & 7)));      // coefficient j = 4095 - k: sequence j & 1, knot j >> 1 (tail_stages, stage 3)
                seqs[slot] = va[s];
                seqs[2 * NS + slot] = vb[s];
            }
            if (coef) {      // the sequences as the filter keeps them (the boxes are rewritten there by stage 2): split layout
                const int j = N - 1 - tt;
                double* oa = coef + 2 * p * N + (j & 1) * NS + (j >> 1);
#pragma unroll
                for (int s = 0; s < P; ++s) {
                    oa[-128 * s] = va[s];
                    if (has_b) oa[N - 128 * s] = vb[s];
                }
            }
        }
        __syncthreads();
        tail_stages<SU>(p, has_b, t, lds, ltw, dd_tabs, mt, bad_row);
    }
}
