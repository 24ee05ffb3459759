"""
Headline benchmark (BASELINE.json): batched FFTLog P(k) -> xi(r) transforms/sec at N=2048, fp64, with the achieved
fraction of the HBM roofline.

    python bench.py [--gpus N --steps K --warmup W]          # config 2 (the headline); N > 1 without a launcher: bench.py starts its own N ranks
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W [--config 2|4|5] [--gather]
(under a launcher WORLD_SIZE must equal --gpus: a mismatch exits non-zero)

Workload: per GPU, 100 000 rows x 2048 log-k bins, rows = A_b (k/0.05)^dn_b P_EH(k),
A~U(0.5,2), dn~U(-0.1,0.1), default_rng(rank); PowerToCorrelation(k, ell=0), defaults (lowring, extrap=0) -> Np=4096.
One "step" = one pass of the fused kernel over the resident batch (inputs already in HBM).  Rows are independent, so
ranks shard with no data-path collective (weak scaling: per-GPU work fixed); `value` = rows of all ranks / max time.

Extra objects on the JSON line: "roofline" (algorithmic bytes 2*8*N per row / kernel time measured with HIP events on
the launch stream, against 8 TB/s), "value_api" (the same steps through cp.PowerToCorrelation.__call__), at N=1 "cpu_baseline" (the numpy
oracle -- same math as the reference's numpy path -- on a bounded sample, all host cores through a process pool) and "secondary": BASELINE
configs 3, 4 and 5 at one GPU's share (a few ms to a few tens of ms of GPU time each) with their own roofline figures.

--config 4 / 5: the 8-GPU configs of BASELINE.json as STRONG splits (1 M P(k) vectors through wallish2018 + brieden2022; 10 M background
samples; each rank takes its contiguous block of cosmologies / samples, no data-path collective); --gather adds the final RCCL all_gather.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_K = 2048
ROWS_PER_GPU = 100000
HBM_PEAK_GBS = 8000.          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6       # MI355X_MICROARCH.md: dense fp64, vector = matrix
CONFIG4_CHUNK = 65536         # vectors per pass of config 4; tests/test_full_size_gpu.py runs the one-GPU share at this size (imported from here)
CONFIG4_PROFILE_CHUNKS = 6      # timed chunks of tools/profile_secondary.py 4w / 4b (behind one untimed chunk, which builds the filter's plans)
BYTES_PER_ROW = 2 * 8 * N_K   # read N f64 + write N f64 (tables are batch-shared, excluded)


def _cpu_chunk(args):
    """One worker: run oracle transforms of a (rows, N) chunk until `deadline` (numpy rfft/irfft path, reference fftlog.py:228-241)."""
    seed, rows, seconds = args
    os.environ['OMP_NUM_THREADS'] = '1'
    from oracle import fftlog as ofl
    from oracle.workloads import pk_eh_default, config2_rows
    k, pk = pk_eh_default(N_K)
    t = ofl.power_to_correlation(k, ell=0)
    x = config2_rows(k, pk, seed * rows, (seed + 1) * rows)[:, None, :]
    ofl.apply(t, x[:64])
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ofl.apply(t, x)
        done += rows
    return done, time.perf_counter() - t0


def _host_cores():
    """Usable host cores: scheduler affinity capped by the cgroup CPU quota (containers often expose more CPUs than they may use)."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            cores = max(1, min(cores, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(seconds=10.):
    """Oracle ("port" of the reference's numpy path) on all usable host cores for about `seconds` of wall time."""
    import multiprocessing as mp
    cores = _host_cores()
    rows = 256   # 256 x 4096 f64 temporaries stay in cache (faster per core than larger chunks)
    ctx = mp.get_context('fork')   # before any GPU initialisation in this process
    with ctx.Pool(cores) as pool:
        tic = time.perf_counter()
        res = pool.map(_cpu_chunk, [(i % 300, rows, seconds) for i in range(cores)])
        wall = time.perf_counter() - tic
    total = sum(r[0] for r in res)
    return {'value': total / wall, 'unit': 'transforms/s', 'cores': cores, 'kind': 'port',
            'sample': '%d-row chunks of the config-2 batch (N=%d) repeated for %.0f s on each of %d processes, numpy oracle (%d transforms)'
                      % (rows, N_K, seconds, cores, total),
            'per_core_value': float(np.mean([r[0] / r[1] for r in res]))}


RAMP_S = 0.3


def _ramp(fn, torch, dev, seconds=None):
    """Untimed load in front of a measurement: after an idle period (or lighter work) the first tens of milliseconds run ~10 % and more below the
    sustained rate (see the ramp of the headline measurement in main()); every secondary config is timed behind ``seconds`` of its own work."""
    seconds = RAMP_S if seconds is None else seconds
    fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        fn()
        torch.cuda.synchronize(dev)


def _gpu_ms(fn, reps, torch, dev):
    """(wall ms, HIP-event ms) per call of fn() on the current stream, behind the untimed ramp."""
    _ramp(fn, torch, dev)
    stream = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / reps * 1e3, e0.elapsed_time(e1) / reps


def eh_parameters(n, seed, torch, dev):
    """Omega_m ~ U(.25, .40), Omega_b ~ U(.04, .06), h ~ U(.6, .8), n_s ~ U(.92, 1) for n cosmologies, on the device."""
    rng = np.random.default_rng(seed)
    par = dict(Omega_m=rng.uniform(.25, .40, n), Omega_b=rng.uniform(.04, .06, n), h=rng.uniform(.6, .8, n), n_s=rng.uniform(.92, 1., n))
    return {name: torch.as_tensor(v, device=dev) for name, v in par.items()}


def config3(cp, torch, dev, ncosmo=10000, reps=20):
    """sigma_rz on 256 r x 64 z for a batch of EH cosmologies (method fftlog, nk = 1024): cosmologies/s, HBM fraction on the
    131 072 + 80 algorithmic bytes per cosmology (10 parameters in, 256 x 64 float64 out)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **eh_parameters(ncosmo, 1, torch, dev))
        interp = cosmo.get_fourier().pk_interpolator()
        r, z = torch.as_tensor(np.geomspace(1, 100, 256), device=dev), torch.as_tensor(np.linspace(0, 3, 64), device=dev)
        wall, gpu = _gpu_ms(lambda: interp.sigma_rz(r, z), reps, torch, dev)
        # untimed, after the clock: one sampled cosmology of the timed call's result against the oracle (1e-9)
        from oracle import checks
        par, i = eh_parameters(ncosmo, 1, torch, torch.device('cpu')), 4321 % ncosmo
        ref = checks.config3_sigma_rz({name: float(v[i]) for name, v in par.items()}, r.cpu().numpy(), z.cpu().numpy())
        err = checks.max_relative_error(interp.sigma_rz(r, z)[i].cpu().numpy(), ref)
        assert err < checks.TOLERANCES['config3'], 'config 3 failed its parity spot check: %g' % err
    nbytes = ncosmo * (256 * 64 * 8 + 80)
    return {'workload': 'config 3: sigma_rz 256 r x 64 z, %d EH98 cosmologies, method fftlog nk=1024, through PowerSpectrumInterpolator2D.sigma_rz' % ncosmo,
            'value': ncosmo / (wall * 1e-3), 'unit': 'cosmologies/s', 'ms': wall, 'ms_gpu_events': gpu,
            'roofline': {'bound': 'hbm', 'achieved': nbytes / (wall * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': nbytes / (wall * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_cosmology': 256 * 64 * 8 + 80},
            'parity_spot_check': {'max_rel_err': err, 'tolerance': checks.TOLERANCES['config3'], 'unit_checked': 'cosmology %d, 256 r x 64 z, vs oracle' % i}}


def config3b(cp, torch, dev, ncosmo=10000, reps=10):
    """sigma_rz on 256 r x 64 z for a batch of TABULATED P(k, z) (500 k x 30 z per cosmology) through
    PowerSpectrumInterpolator2D(k, z, pk=(B, nk, nz)).sigma_rz: cosmologies/s and the HBM fraction on the 120 000 + 131 072 algorithmic bytes
    per cosmology.  The tables are the reference's (tests/golden/sigma.npz: the EH98 table of its own test) times one amplitude per cosmology."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'sigma.npz'))
    rng = np.random.default_rng(1)
    amp = torch.as_tensor(rng.uniform(0.5, 2., ncosmo), device=dev)
    tables = amp[:, None, None] * torch.as_tensor(g['table_pk'], device=dev)[None]
    interp = cp.PowerSpectrumInterpolator2D(g['table_k'], g['table_z'], tables)
    r, z = torch.as_tensor(np.geomspace(1, 100, 256), device=dev), torch.as_tensor(np.linspace(0, 3, 64), device=dev)
    wall, gpu = _gpu_ms(lambda: interp.sigma_rz(r, z), reps, torch, dev)
    # untimed, after the clock: one sampled table of the timed call's result against the oracle's RectBivariateSpline route (1e-9)
    from oracle import checks
    i = 4242 % ncosmo
    ref = checks.config3b_sigma_rz(g['table_k'], g['table_z'], float(amp[i]) * g['table_pk'], r.cpu().numpy(), z.cpu().numpy())
    err = checks.max_relative_error(interp.sigma_rz(r, z)[i].cpu().numpy(), ref)
    assert err < checks.TOLERANCES['config3b'], 'config 3B failed its parity spot check: %g' % err
    nbytes = ncosmo * (500 * 30 * 8 + 256 * 64 * 8)
    # fp64 work per table that no implementation of this route can avoid: 64 FFTLogs of Np = 2048 (two complex FFTs per packed pair of rows = one
    # per row, 5 Np log2 Np flop) and the contraction of the 30 tabulated redshifts onto the 64 requested ones at the 1024 wavenumbers of the transform
    flop = 64 * 5 * 2048 * 11 + 2 * 64 * 30 * 1024
    return {'workload': 'config 3B: sigma_rz 256 r x 64 z, %d tabulated P(k, z) of 500 k x 30 z, method fftlog nk=1024 (64 FFTLogs per table), through '
                        'PowerSpectrumInterpolator2D.sigma_rz' % ncosmo,
            'value': ncosmo / (wall * 1e-3), 'unit': 'cosmologies/s', 'ms': wall, 'ms_gpu_events': gpu,
            'roofline': {'bound': 'hbm', 'achieved': nbytes / (wall * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': nbytes / (wall * 1e-3) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_cosmology': 500 * 30 * 8 + 256 * 64 * 8,
                         'note': 'compute-bound on these bytes (arithmetic intensity %.0f flop/B): see roofline_fp64' % (flop / (500 * 30 * 8 + 256 * 64 * 8))},
            'roofline_fp64': {'bound': 'mfma', 'achieved': ncosmo * flop / (wall * 1e-3) / 1e12, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                              'frac': ncosmo * flop / (wall * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 'algorithmic_flop_per_cosmology': flop,
                              'note': '64 FFTLogs (5 Np log2 Np, Np = 2048) + the 30 -> 64 redshift contraction at 1024 wavenumbers; vector and matrix fp64 peaks are equal'},
            'parity_spot_check': {'max_rel_err': err, 'tolerance': checks.TOLERANCES['config3b'], 'unit_checked': 'table %d, 256 r x 64 z, vs oracle' % i}}


# Vector instructions one P(k) vector NEEDS, whatever the implementation (wave-instructions = lane operations / 64): the evaluations of the fit formulae
# (EH98 ~250 vector instructions per wavenumber, its no-wiggle form ~95, with the table-driven logarithm / exponential, the tabulated powers of k and, for
# the no-wiggle form, reciprocals in place of its four divisions (115 before) of round 4 -- rounds 2-3: 390 and 150 with polynomial forms throughout: SQ_INSTS_VALU of sigma8_normalise_kernel per sample net of its per-cosmology
# part, profiles/*_config4_valu.json) and the transforms / solves (a radix-2 FFT count, 5 N log2 N flops per complex transform = N log2 N x 2.5
# multiply-adds; three multiply-adds per knot and sweep for a tridiagonal system).
_EH98, _NOWIGGLE = 250, 95
CONFIG4_ALGORITHMIC = {
    # 4096 wavenumbers of the linear grid + the 1024 of the sigma8 normalisation (the filter's own k); forward and inverse DST of 4096 samples (each
    # half a complex transform of 4096 points per row); clamped splines through 2 x 2048 coefficients and through the 3666 spliced knots, two sweeps
    # each, and the 1024 evaluations of the latter
    'wallish2018': ((4096 + 1024) * _EH98 + 2 * 0.5 * 2.5 * 4096 * 12 + 2 * 3 * (4096 + 3666) + 8 * 1024) / 64.,
    # sigma8 normalisation of both engines (1024 each), EH98 at k_fid / r at the 23 extrema of the fiducial wiggles (all the envelope depends on; up to
    # cp_brieden_smooth it was evaluated at the 341 wavenumbers of k_fid, and this count carried 2 x 341), the no-wiggle template at k_fid r (341); the
    # envelope operator's 23 columns; the per-cosmology spline through 345 knots, its evaluation and 10^x at 341 wavenumbers, 1024 values written
    'brieden2022': ((1024 + 23) * _EH98 + (1024 + 341) * _NOWIGGLE + 23 * 341 + 2 * 3 * 345 + 30 * 341 + 1024) / 64.,
}


def _library_sha():
    import hashlib
    path = os.path.join(ROOT, 'cosmoprimo_amd', 'libcosmoprimo_amd.so')
    return hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16] if os.path.isfile(path) else None


def _config4_valu_roofline(engine, vectors_per_s, chunk):
    """What bounds the filters is instruction issue, not HBM.  `achieved` / `frac`: the ALGORITHMIC vector instructions per vector (CONFIG4_ALGORITHMIC:
    operation counts of the formulae with every EH98 / no-wiggle sample priced at what this package's shortest evaluation costs, _EH98 / _NOWIGGLE --
    those two prices fell between rounds (390 / 150 in round 3), so `frac` compares rounds only at equal prices; `executed.issue_slots_used` is the
    figure that does) x the measured vectors/s against the issue peak of the vector pipes.  Next to it the instructions the
    package's own kernels EXECUTE per vector (committed census profiles/*_config4_valu.json: rocprofv3 --pmc SQ_INSTS_VALU per kernel, framework
    kernels excluded, scripts/profile_config4_valu.sh), flagged when the census was not taken on this library or chunk size."""
    import glob
    peak = 256 * 4 * 2.4e9 / 4
    alg = CONFIG4_ALGORITHMIC[engine]
    out = {'bound': 'vector instruction issue (fp64 and integer VALU, 4 cycles per wave instruction)', 'achieved': alg * vectors_per_s, 'peak': peak,
           'unit': 'wave-instructions/s', 'frac': alg * vectors_per_s / peak, 'algorithmic_wave_instructions_per_vector': alg,
           'algorithmic_prices': {'eh98_sample': _EH98, 'nowiggle_sample': _NOWIGGLE, 'provenance': 'instructions per sample of this package\'s shortest evaluation (sigma8_normalise_kernel census)'}}
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_config4_valu.json')))
    if files:
        with open(files[-1]) as fh:
            census = json.load(fh)
        entry = census.get(engine, {})
        per = entry.get('per_vector', {})
        if 'SQ_INSTS_VALU' in per:
            current = census.get('library_sha256_16') == _library_sha() and entry.get('chunk') == chunk
            out['executed'] = {'valu_wave_instructions_per_vector': per['SQ_INSTS_VALU'], 'mfma_f64_16x16x4_per_vector': per.get('SQ_INSTS_MFMA', 0.),
                               'issue_slots_used': per['SQ_INSTS_VALU'] * vectors_per_s / peak, 'algorithmic_over_executed': alg / per['SQ_INSTS_VALU'],
                               'source': os.path.relpath(files[-1], ROOT), 'census_taken_on_this_library_and_chunk': bool(current)}
    return out


def _headline_valu_roofline(pmc, rows, kernel_ms, source):
    """The second bound of the headline kernel: the vector instructions it executes per row (SQ_INSTS_VALU of the committed rocprofv3 --pmc pass of
    this command) x the rows/s measured live, against the issue peak of the vector pipes (1024 SIMDs, one wave instruction per 4 cycles at the
    nominal 2.4 GHz); and the clock the chip ran the profiled launches at (GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
    counters = pmc.get('counters', {})
    if 'SQ_INSTS_VALU' not in counters:
        return None
    peak = 256 * 4 * 2.4e9 / 4
    per_row = counters['SQ_INSTS_VALU']['mean'] / pmc['rows_per_launch']
    achieved = per_row * rows / (kernel_ms * 1e-3)
    out = {'bound': 'vector instruction issue (fp64 and integer VALU, 4 cycles per wave instruction)', 'achieved': achieved, 'peak': peak,
           'unit': 'wave-instructions/s', 'frac': achieved / peak, 'valu_wave_instructions_per_row': per_row, 'source': source}
    if pmc.get('effective_clock_GHz'):
        clock = pmc['effective_clock_GHz'] * 1e9
        out['effective_clock_GHz'] = clock / 1e9
        out['frac_at_effective_clock'] = achieved / (256 * 4 * clock / 4)
        if 'SQ_WAVE_CYCLES' in counters and 'SQ_BUSY_CYCLES' in counters:      # waves resident per SIMD while the shader engines are busy (32 SE-level busy counters, 1024 SIMDs)
            out['waves_per_simd_while_busy'] = counters['SQ_WAVE_CYCLES']['mean'] / (counters['SQ_BUSY_CYCLES']['mean'] / 32 * 1024)
    return out


def config4(cp, torch, dev, par, chunk=CONFIG4_CHUNK, engines=('wallish2018', 'brieden2022'), spot_check=True):
    """wallish2018 and brieden2022 on EH98 P(k) vectors (nk = 1024) of the cosmologies ``par``, chunk by chunk (P(k) generation and sigma8
    normalisation included, results left on the device): per filter vectors/s, HBM fraction on 16 384 B per vector, HIP-event time.
    chunk : vectors per pass (2 GB per 4096-sample intermediate at 65 536).  The host needs ~1 ms to queue a chunk of brieden2022 whatever its size; the
    device took that long for 16 384 vectors in round 3 and takes it for 32 768 now (tools/chunk_config4.py: 1.6e7 vectors/s in chunks of 16 384,
    2.4-3.0e7 at 32 768, 3.5e7 at 65 536 and for the whole share at once; wallish2018 9.2e6 at any of them)."""
    import warnings
    from cosmoprimo_amd.bao_filter import PowerSpectrumBAOFilter
    n = int(par['Omega_m'].numel())
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        fid = cp.Cosmology(engine='eisenstein_hu')
        for engine in engines:
            kw = dict(cosmo_fid=fid) if engine == 'brieden2022' else {}
            state = {}

            def run(sl):
                cosmo = cp.Cosmology(engine='eisenstein_hu', sigma8=0.8, **{name: v[sl] for name, v in par.items()})
                interp = cosmo.get_fourier().pk_interpolator(z=np.array([0.]))
                if 'filter' not in state:      # made once, called again for every chunk: its _prepare products depend on the fiducial only
                    state['filter'] = PowerSpectrumBAOFilter(interp, engine=engine, **(dict(kw, cosmo=cosmo) if kw else {}))
                else:
                    state['filter'](interp, cosmo=cosmo if kw else None)
                state['cosmo'] = cosmo
                return state['filter']._pknow_rows

            _ramp(lambda: run(slice(0, min(n, chunk))), torch, dev)       # plans, operators and code objects are built on first use; clock ramp
            stream = torch.cuda.current_stream(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for start in range(0, n, chunk):
                res = run(slice(start, min(n, start + chunk)))
            e1.record(stream)
            torch.cuda.synchronize(dev)
            wall = time.perf_counter() - t0
            assert bool(torch.isfinite(res.sum()))      # (after the clock; one reduction: a NaN or an infinity anywhere reaches the sum)
            check = None
            if spot_check:      # untimed, after the clock: one sampled vector of the last timed chunk against the oracle's filter (1e-9 on pknow)
                from oracle import checks
                last = (n - 1) // chunk * chunk
                j = (n - last) // 3
                rsig = float(state['cosmo']._engine._rsigma8.reshape(-1)[j])
                ref = checks.config4_pknow({name: float(v[last + j]) for name, v in par.items()}, rsig, engine)
                err = checks.max_relative_error(res.reshape(n - last, -1)[j].cpu().numpy(), ref)
                assert err < checks.TOLERANCES['config4'], 'config 4 (%s) failed its parity spot check: %g' % (engine, err)
                check = {'max_rel_err': err, 'tolerance': checks.TOLERANCES['config4'], 'unit_checked': 'vector %d (chunk of %d), pknow at 1024 k, vs oracle' % (last + j, chunk)}
            valu = _config4_valu_roofline(engine, n / wall, chunk)
            out[engine] = {'value': n / wall, 'unit': 'vectors/s', 'ms': wall * 1e3, 'ms_gpu_events': e0.elapsed_time(e1), 'vectors': n, 'chunk': chunk,
                           'parity_spot_check': check, 'roofline_valu': valu,
                           'roofline': {'bound': 'hbm', 'achieved': n * 16384 / wall / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                        'frac': n * 16384 / wall / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_vector': 16384,
                                        'note': 'latency / ALU bound by construction (SURVEY.md 8(d)): the HBM fraction is reported, not targeted'}}
    return out


# config 5, per interval of the quadrature grid and per sample: what the formulae ask for, independent of the kernel -- two new ordinates of 1 / E(z) (the
# midpoint and the far end: RK4 on dy/dz = f(z) is Simpson's rule, jax.py:700-710), each E^2 = matter + radiation + curvature (3 multiply-adds in 1 + z)
# + the dark-energy term (2 multiply-adds for the exponent, one exponential, 2 products), then its reciprocal root; the Simpson sum (4) and one row of the
# natural spline's elimination (4: cosmology.py:2027-2042 with jax.py:135-196).  The transcendentals at the shortest forms this hardware has for doubles
# (cp_math.h: exp 11 instructions table-driven, reciprocal square root 8 from the hardware estimate): a count of operations of the algorithm priced in
# instructions, not a census of the kernel.
CONFIG5_ALGORITHMIC_PER_INTERVAL = 2 * (3 + 2 + 11 + 2 + 8) + 4 + 4


def _config5_valu_roofline(samples_per_s, z_samples=None):
    """Vector-instruction roofline of config 5.  `achieved` / `frac`: the ALGORITHMIC instructions per sample -- the intervals a sample needs (those from
    z = 0 up to its own, which the cumulative integral runs over, and the `reach` above it from which the spline's slope at the sample is eliminated:
    what is left of the arbitrary start falls by 0.27 per knot, 1e-13 after 24) x CONFIG5_ALGORITHMIC_PER_INTERVAL -- x samples/s against the issue peak
    of the vector pipes.  Beside it what bg_kernel EXECUTES per sample (committed census profiles/*_config5_valu.json: SQ_INSTS_VALU, tools/census_config5.sh)."""
    import glob
    peak = 256 * 4 * 2.4e9 / 4
    out = {'bound': 'vector instruction issue (fp64 and integer VALU, 4 cycles per wave instruction)', 'peak': peak, 'unit': 'wave-instructions/s'}
    reach = 24
    if z_samples is not None:
        from cosmoprimo_amd.cosmology import get_default_z_interp
        knots = get_default_z_interp('comoving_radial_distance')
        k = np.clip(np.searchsorted(knots, z_samples, side='right') - 1, 0, knots.size - 2)
        needed = float(np.mean(np.minimum(k + 1 + reach, knots.size - 1)))
    else:
        needed = 26. + reach
    alg = needed * CONFIG5_ALGORITHMIC_PER_INTERVAL / 64.      # wave-instructions per sample
    out.update(achieved=alg * samples_per_s, frac=alg * samples_per_s / peak, algorithmic_wave_instructions_per_sample=alg, intervals_needed_per_sample=needed,
               algorithmic_instructions_per_interval_and_thread=CONFIG5_ALGORITHMIC_PER_INTERVAL,
               useful_fp64_flop_per_sample=needed * 2 * 44 + needed * 12, useful_fp64_frac_of_peak=(needed * 2 * 44 + needed * 12) * samples_per_s / (FP64_PEAK_TFLOPS * 1e12))
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_config5_valu.json')))
    if files:
        with open(files[-1]) as fh:
            census = json.load(fh)
        per = census['per_sample']
        out['executed'] = {'valu_wave_instructions_per_sample': per['SQ_INSTS_VALU'], 'valu_instructions_per_thread': 64 * per['SQ_INSTS_VALU'],
                           'issue_slots_used': per['SQ_INSTS_VALU'] * samples_per_s / peak, 'algorithmic_over_executed': alg / per['SQ_INSTS_VALU'],
                           'source': os.path.relpath(files[-1], ROOT),
                           'census_taken_on_this_library': census.get('library_sha256_16') == _library_sha() if 'library_sha256_16' in census else None}
    return out


def config5(torch, dev, om, w0, wa, zz, reps=20, spot_check=True):
    from cosmoprimo_amd import background
    n = int(zz.numel())
    fn = lambda: background.distance('comoving_radial_distance', zz[:, None], dict(w0_fld=w0, wa_fld=wa), Omega_m=om, per_cosmology_z=True)      # noqa: E731
    wall, gpu = _gpu_ms(fn, reps, torch, dev)
    check = None
    if spot_check:      # untimed, after the clock: 300 sampled distances against the oracle (1e-10)
        from oracle import checks
        idx = np.random.default_rng(33).integers(0, n, 300)
        tidx = torch.as_tensor(idx, device=dev)
        ref = checks.config5_distances(*(v[tidx].cpu().numpy() for v in (om, w0, wa, zz)))
        err = checks.max_relative_error(fn()[:, 0][tidx].cpu().numpy(), ref)
        assert err < checks.TOLERANCES['config5'], 'config 5 failed its parity spot check: %g' % err
        check = {'max_rel_err': err, 'tolerance': checks.TOLERANCES['config5'], 'unit_checked': '300 sampled distances vs oracle'}
    return {'workload': 'config 5: comoving_radial_distance for %d (Omega_m, w0, wa, z) samples, one fresh cosmology per sample' % n,
            'value': n / (wall * 1e-3), 'unit': 'samples/s', 'ms': wall, 'ms_gpu_events': gpu, 'parity_spot_check': check,
            # what bounds the kernel is the issue rate of the vector pipes: the algorithm's own operation count against the issue peak, with the
            # instructions the kernel executes (profiler census) beside it
            'roofline': dict(_config5_valu_roofline(n / (wall * 1e-3), zz[:200000].cpu().numpy()), hbm_GBps=n * 40 / (wall * 1e-3) / 1e9)}


def config5_samples(n, seed, torch, dev):
    """Omega_m ~ U(.1, .5), w0 ~ U(-1.5, -.5), wa ~ U(-1, .5) with w0 + wa < 1/3 (re-drawn otherwise), z ~ U(0, 3)."""
    rng = np.random.default_rng(seed)
    om, w0, wa, zz = rng.uniform(0.1, 0.5, n), rng.uniform(-1.5, -0.5, n), rng.uniform(-1., 0.5, n), rng.uniform(0., 3., n)
    bad = w0 + wa >= 1. / 3.
    while bad.any():
        w0[bad], wa[bad] = rng.uniform(-1.5, -0.5, bad.sum()), rng.uniform(-1., 0.5, bad.sum())
        bad = w0 + wa >= 1. / 3.
    return tuple(torch.as_tensor(v, device=dev) for v in (om, w0, wa, zz))


def secondary(cp, torch, dev):
    """BASELINE configs 3, 4, 5 at one GPU's share of the smaller kind (a few ms to tens of ms each), for the N = 1 line."""
    out = {'config3': config3(cp, torch, dev)}
    out['config3b'] = config3b(cp, torch, dev)
    torch.cuda.empty_cache()
    share = 125000      # one GPU's share of the 1 M vectors of BASELINE config 4
    c4 = config4(cp, torch, dev, eh_parameters(share, 2, torch, dev))
    out['config4'] = dict(c4, workload='config 4: wallish2018 and brieden2022 on the %d EH98 P(k) vectors of one GPU (an eighth of 1 M), in chunks of %d queued '
                                         'back to back, P(k) generation and sigma8 normalisation included, results resident' % (share, CONFIG4_CHUNK))
    out['config5'] = config5(torch, dev, *config5_samples(1250000, 3, torch, dev))
    return out


def main_split(args, config):
    """--config 4 / 5: strong split of the BASELINE 8-GPU workloads over the ranks, no data-path collective; rank 0 prints one JSON line."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    launched = world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ)      # under torchrun, also with one rank
    if launched:
        dist.init_process_group('nccl', device_id=dev)
        dist.barrier()      # communicator built here, not at the barrier in front of the timed steps
    line = split_line(args, config, torch, dist, dev, rank, world, launched, args.steps, args.warmup, (args.gather or world > 1) and not args.no_gather)
    if rank == 0:
        print(json.dumps(line))
    if launched:
        dist.destroy_process_group()


def split_line(args, config, torch, dist, dev, rank, world, launched, steps, warmup, gather):
    """The JSON object of a strong split of BASELINE config 4 / 5 over the ranks of an initialised process group (every rank calls this; rank 0 gets
    the object, the others None): the line of ``--config 4 / 5``, and what a bare ``--gpus N > 1`` run embeds under "split_configs"."""
    global RAMP_S
    RAMP_S = 0.     # the W untimed steps of this mode are whole passes over the rank's share (hundreds of ms each): they are the ramp
    import cosmoprimo_amd as cp
    from cosmoprimo_amd.distributed import shard_range, gather_rows
    total = args.rows if args.rows != ROWS_PER_GPU else (1000000 if config == 4 else 10000000)
    start, stop = shard_range(total, rank, world)
    if config == 4:
        full = eh_parameters(total, 2, torch, torch.device('cpu'))
        par = {name: v[start:stop].to(dev) for name, v in full.items()}
        run = lambda check=False: config4(cp, torch, dev, par, spot_check=check)      # noqa: E731
    else:
        full = config5_samples(total, 3, torch, torch.device('cpu'))
        mine = tuple(v[start:stop].to(dev) for v in full)
        run = lambda check=False: config5(torch, dev, *mine, reps=1, spot_check=check)      # noqa: E731
    for _ in range(max(1, warmup)):
        run()
    torch.cuda.synchronize(dev)
    if launched:
        dist.barrier()
    tic = time.perf_counter()
    for _ in range(steps):
        last = run()
    torch.cuda.synchronize(dev)
    t_done = time.perf_counter()
    if launched:
        dist.barrier()
    elapsed = time.perf_counter() - tic
    own_ms = (t_done - tic) / steps * 1e3      # this rank's own steps, before it waited for the others
    rank_ms = [own_ms]
    checked = run(check=True) if rank == 0 else None      # untimed: the rank's share once more with the oracle spot check of each config
    gather_ms = None
    if launched:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])
        every = torch.zeros(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, torch.tensor([own_ms], device=dev, dtype=torch.float64))
        rank_ms = [float(v) for v in every]
        if gather:
            local = torch.zeros((stop - start, 1024 if config == 4 else 1), dtype=torch.float64, device=dev)      # the shape of a rank's results
            gather_rows(local, n_total=total)
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            gather_rows(local, n_total=total)
            torch.cuda.synchronize(dev)
            dist.barrier()
            gather_ms = (time.perf_counter() - t0) * 1e3
    if rank != 0:
        return None
    units = total * (2 if config == 4 else 1)       # config 4: every vector goes through both filters
    line = {'metric': 'BAO-filtered P(k) vectors/sec (wallish2018 + brieden2022, nk=1024)' if config == 4 else 'comoving_radial_distance samples/sec',
            'value': units * steps / elapsed, 'unit': 'filtered vectors/s' if config == 4 else 'samples/s', 'n_gpus': world, 'steps': steps,
            'warmup': warmup, 'ms_per_step': elapsed / steps * 1e3, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'config %d: %d %s split over %d GPU(s) in contiguous blocks, no collective' % (
                config, total, 'EH98 P(k) vectors through both filters' if config == 4 else '(Omega_m, w0, wa, z) samples', world),
                'per_gpu': stop - start, 'rccl_ranks': world if launched else 0},
            'ms_per_step_rank_min': min(rank_ms), 'ms_per_step_rank_max': max(rank_ms), 'ms_per_step_by_rank': rank_ms,
            'parity_spot_check': ({e: checked[e]['parity_spot_check'] for e in checked} if config == 4 else checked['parity_spot_check']),
            'rank0_detail': last}
    if gather_ms is not None:
        line['gather_ms'] = gather_ms
        line['value_with_gather'] = units / (elapsed / steps + gather_ms * 1e-3)      # a step followed by the gather of its results
    return line


def _launch_ranks_if_needed(args):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start the N ranks itself (one process per GPU through
    ``python -m torch.distributed.run``, a CHILD process started before this one has made any GPU call), relay its output and exit with its
    code.  Under a launcher the world size must be the --gpus that was asked for; anything else exits non-zero instead of printing a line
    that claims N GPUs."""
    world = int(os.environ.get('WORLD_SIZE', 1)) if 'RANK' in os.environ else None
    if world is not None:
        if world != args.gpus:
            sys.exit('bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to report a line for %d GPU(s)'
                     % (args.gpus, world, args.gpus))
        return
    if args.gpus <= 1 and not args.launcher:
        return
    import subprocess
    import torch       # device_count() does not initialise the GPU runtime of this process
    have = torch.cuda.device_count()
    if have < args.gpus:
        sys.exit('bench.py: --gpus %d asked for, %d GPU(s) visible on this node; one rank per GPU is the only mode' % (args.gpus, have))
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200, help='timed steps (default 200: 0.2 s of back-to-back launches -- long enough for a device monitor sampling the run to see the GPU busy)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--rows', type=int, default=ROWS_PER_GPU, help='rows per GPU (default: the config-2 batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--gather', action='store_true', help='also time the final RCCL all_gather of the result shards (reported separately); the default for N > 1')
    ap.add_argument('--no-gather', action='store_true', help='N > 1: skip the timing of the final all_gather')
    ap.add_argument('--config', type=int, default=2, choices=[2, 4, 5], help='BASELINE.json config: 2 (headline, weak scaling), 4 or 5 (strong splits)')
    ap.add_argument('--ramp-ms', type=float, default=300., help='untimed load before the warmup steps (and before each secondary config), to reach the sustained device state')
    ap.add_argument('--launcher', action='store_true', help='start the rank(s) through torch.distributed.run even for --gpus 1 (RCCL initialised); N > 1 always does')
    ap.add_argument('--no-secondary', action='store_true', help='skip the configs 3 / 4 / 5 numbers of the N = 1 line (and the split configs 4 / 5 of an N > 1 line)')
    ap.add_argument('--split-configs', action='store_true', help='under a launcher: the strong splits of configs 4 and 5 on the line ("split_configs"); the default for N > 1')
    args = ap.parse_args()
    _launch_ranks_if_needed(args)
    if args.config != 2:
        return main_split(args, args.config)

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    distributed = world > 1 or ('RANK' in os.environ and 'MASTER_ADDR' in os.environ)      # under torchrun, also with one rank: RCCL is then exercised

    def log(msg):
        if rank == 0:
            print('[bench %.1fs] %s' % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)

    T0 = time.perf_counter()
    cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        log('cpu baseline on %d cores ...' % _host_cores())
        cpu = cpu_baseline()   # before the GPU is touched (fork-safe)
        log('cpu baseline: %.0f transforms/s' % cpu['value'])

    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available(), 'bench.py needs a ROCm GPU; there is no CPU path to measure'
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    if distributed:
        dist.init_process_group('nccl', device_id=dev)
        # RCCL builds its communicator on the first collective (hundreds of ms with the device idle): do that now, not at the barrier in
        # front of the timed steps, which would put them back into the clock ramp that the warm-up is there to leave behind
        dist.barrier()

    import cosmoprimo_amd as cp
    from oracle.workloads import pk_eh_default
    k, pk = pk_eh_default(N_K)
    rng = np.random.default_rng(rank)
    nb = args.rows
    amp, dn = rng.uniform(0.5, 2., nb), rng.uniform(-0.1, 0.1, nb)
    tk, tpk = torch.as_tensor(k, device=dev), torch.as_tensor(pk, device=dev)
    rows = (torch.as_tensor(amp, device=dev)[:, None] * (tk[None, :] / 0.05) ** torch.as_tensor(dn, device=dev)[:, None] * tpk[None, :]).contiguous()
    f = cp.PowerToCorrelation(k, ell=0, device=dev)
    log('inputs resident: %d x %d f64' % (nb, N_K))

    # product call path = FFTlog.__call__ -> cp_fftlog_execute; for timing, call the C ABI directly on preallocated buffers
    from cosmoprimo_amd import _lib
    lib = _lib.load()
    plan = f._get_plan(dev)
    out = torch.empty_like(rows)
    stream = torch.cuda.current_stream(dev)

    def step():
        _lib.check(lib.cp_fftlog_execute(plan.handle, rows.data_ptr(), out.data_ptr(), nb, 0, 0., 0, 0., 0, stream.cuda_stream))

    # Bring the device to its sustained state first: after an idle period the first tens of milliseconds of load run ~10 % slower than the
    # steady state that follows (clock / power management ramp; measured with tools/api_probe.py: 1.12 ms for the first 20 launches of a fresh
    # process, 0.975 ms for every later batch of 20).  Untimed, like the W warmup steps that follow it.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_ms * 1e-3:
        for _ in range(10):
            step()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    log('warmup done')
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    # HIP events on the stream the kernel is launched on, around the K back-to-back launches of the timed region (events between the launches would
    # keep each launch from starting under the tail of the one before it, which is how a pipeline queues them)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tic = time.perf_counter()
    e0.record(stream)
    for i in range(args.steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize(dev)
    t_done = time.perf_counter()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - tic
    kernel_ms = e0.elapsed_time(e1) / args.steps
    rank_ms = [(t_done - tic) / args.steps * 1e3]      # this rank's own K steps, before it waited for the others
    # untimed, afterwards: the same K launches bracketed one by one (the spread between launches; the minimum is the quiet-device duration)
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    for i in range(args.steps):
        starts[i].record(stream)
        step()
        ends[i].record(stream)
    torch.cuda.synchronize(dev)
    per_step = [s.elapsed_time(e) for s, e in zip(starts, ends)]
    kernel_ms_each, kernel_ms_min = float(np.mean(per_step)), float(np.min(per_step))

    # the same K steps through the product API (cp.PowerToCorrelation.__call__ on the resident tensor into a resident result: plan
    # lookup, argument checks and the ctypes call included, no allocation, no host synchronisation inside)
    out_api = torch.empty_like(out)      # the caller's result buffer (FFTlog.__call__(out=)): a sampler keeps one, as this loop does
    for _ in range(max(1, args.warmup)):
        f(rows, out=out_api)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    tic_api = time.perf_counter()
    for i in range(args.steps):
        s_api, xi_api = f(rows, out=out_api)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    elapsed_api = time.perf_counter() - tic_api
    assert xi_api.shape == rows.shape and bool(torch.equal(xi_api, out)), 'API path and C-ABI path disagree'

    gather_ms = None
    if distributed:
        tapi = torch.tensor([elapsed_api], device=dev, dtype=torch.float64)
        dist.all_reduce(tapi, op=dist.ReduceOp.MAX)
        elapsed_api = float(tapi[0])
        tmax = torch.tensor([elapsed, kernel_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(tmax[0]), float(tmax[1])
        every = torch.zeros(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, torch.tensor(rank_ms, device=dev, dtype=torch.float64))
        rank_ms = [float(v) for v in every]      # a straggler shows at a glance on the first multi-GPU run
        if (args.gather or world > 1) and not args.no_gather:
            from cosmoprimo_amd.distributed import gather_rows
            full = torch.empty((world * nb, N_K), dtype=out.dtype, device=dev)
            gather_rows(out, n_total=world * nb, out=full)      # equal shards: one all_gather_into_tensor into the preallocated result
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            gather_rows(out, n_total=world * nb, out=full)
            torch.cuda.synchronize(dev)
            dist.barrier()
            gather_ms = (time.perf_counter() - t0) * 1e3

    # sanity: the timed output is the real transform (spot-check one row against the oracle on rank 0)
    if rank == 0:
        from oracle import fftlog as ofl
        t = ofl.power_to_correlation(k, ell=0)
        ref = ofl.apply(t, rows[7].cpu().numpy())[0]
        w = t.y[0] ** 1.5
        err = float(np.abs((out[7].cpu().numpy() - ref) * w).max() / np.abs(ref * w).max())
        assert err < 1e-13, 'bench output failed parity spot-check: %g' % err

        total_rows = nb * world * args.steps
        value = total_rows / elapsed
        achieved = BYTES_PER_ROW * nb / (kernel_ms * 1e-3) / 1e9   # GB/s per GPU, dominant (only) kernel
        grid = __import__('ctypes').c_int()
        block = __import__('ctypes').c_int()
        lds = __import__('ctypes').c_int()
        lib.cp_fftlog_plan_info(plan.handle, nb, grid, block, lds)
        traffic, traffic_src = None, None
        import glob
        prof = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_pmc_summary.json')))
        valu = None
        if prof and nb == ROWS_PER_GPU:   # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command
            with open(prof[-1]) as fh:
                pmc = json.load(fh)
            traffic = pmc['hbm_bytes_per_launch']['total']
            traffic_src = os.path.relpath(prof[-1], ROOT)
            valu = _headline_valu_roofline(pmc, nb, kernel_ms, traffic_src)
        # the committed rocprofv3 --kernel-trace summary of this same command (scripts/gpu_profile.sh, scripts/summarize_steady.py): steady-state
        # dispatches only, next to the HIP-event time measured live above
        rocprof = None
        steady = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_headline_steady.json')))
        if steady and nb == ROWS_PER_GPU:
            with open(steady[-1]) as fh:
                summ = json.load(fh)
            rocprof = dict(summ['steady_state'], source=os.path.relpath(steady[-1], ROOT), kernel_ms_hip_events_same_run=(summ.get('same_run_bench_line') or {}).get('kernel_ms_hip_events'))
        line = {
            'metric': 'batched FFTLog P(k)->xi(r) transforms/sec (N=2048)', 'value': value, 'unit': 'transforms/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'ms_per_step_rank_min': min(rank_ms), 'ms_per_step_rank_max': max(rank_ms), 'ms_per_step_by_rank': rank_ms,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'config 2: %d rows x N_k=%d per GPU, PowerToCorrelation(ell=0), Np=4096, extrap=0' % (nb, N_K),
                       'rows_per_gpu': nb, 'n_k': N_K, 'padded_size': 4096, 'parallelism': 'rows sharded over %d GPU(s), no collective' % world,
                       'grid': grid.value, 'block': block.value, 'lds_bytes': lds.value},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'traffic_source': traffic_src, 'kernel': 'fftlog_kernel<4096,16,IN_HALF_ZERO,OUT_HALF>', 'kernel_ms': kernel_ms,
                         'kernel_ms_bracketed_one_by_one': kernel_ms_each, 'kernel_ms_min': kernel_ms_min, 'rocprof_steady_state': rocprof, 'algorithmic_bytes_per_launch': BYTES_PER_ROW * nb},
            'roofline_valu': valu,
            'parity_spot_check_tilted_err': err,
            'value_api': nb * world * args.steps / elapsed_api,   # same batch through cp.PowerToCorrelation.__call__ (resident tensors)
        }
        line['config']['rccl_ranks'] = world if distributed else 0
        if cpu is not None:
            line['cpu_baseline'] = cpu
        if world == 1 and not args.no_secondary:
            global RAMP_S
            RAMP_S = args.ramp_ms * 1e-3
            del rows, out, out_api, xi_api
            torch.cuda.empty_cache()
            line['secondary'] = secondary(cp, torch, dev)
        if gather_ms is not None:
            line['gather_ms'] = gather_ms
            line['value_with_gather'] = nb * world / (elapsed / args.steps + gather_ms * 1e-3)      # a step followed by the gather of its results
    if distributed and (args.split_configs or (world > 1 and not args.no_secondary)):
        # BASELINE configs 4 and 5 are DEFINED on the 8 GPUs of a node: a multi-GPU run reports their strong splits on the same line (every rank takes part:
        # the barriers and the max over ranks are collectives), a few steps each
        try:
            del rows, out, out_api, xi_api
        except NameError:
            pass
        torch.cuda.empty_cache()
        split = {}
        for config in (4, 5):
            res = split_line(args, config, torch, dist, dev, rank, world, True, min(args.steps, 5), 1, not args.no_gather)
            if rank == 0:
                split['config%d' % config] = res
        if rank == 0:
            line['split_configs'] = split
    if rank == 0:
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
