"""
Power spectrum interpolators on MI355X: :class:`PowerSpectrumInterpolator1D`, :class:`PowerSpectrumInterpolator2D` with
the constructor / ``from_callable`` / ``__call__`` / ``sigma_*`` / ``to_xi`` / ``to_1d`` contracts of the reference
(cosmoprimo/interpolator.py:412-987), and the low-level :class:`Interpolator1D` / :class:`Interpolator2D`
(cosmoprimo/jax.py:135-287).

Data path (all on the GPU, through the C ABI of libcosmoprimo_amd.so):
  * spline fits + evaluations are banded linear operators applied to rows (``cp_spline_*``): natural cubic for
    Interpolator1D (jax.py:172), separable not-a-knot cubic for Interpolator2D (== RectBivariateSpline(kx=ky=3, s=0),
    SURVEY.md App. C5);
  * sigma_r / sigma_rz (default ``method='fftlog'``): P(k) rows on geomspace(kmin, kmax, nk) -> TophatVariance FFTLog
    (fused kernel) -> natural spline in (s, var) at r -> sqrt (interpolator.py:285-291);
  * ``method='simpson'`` and sigma_d: the composite-Simpson weights (jax.py:365-507) times the integrand kernel form a
    fixed operator applied to the P(k) rows (``cp_linop_*``).
Host side (numpy): grids, log-padding (:func:`_pad_log`), masks / NaN / ``bounds_error`` / dtype rules.
Inputs may be numpy (results come back as numpy) or torch CUDA tensors (results stay on the device).
"""
import inspect

import numpy as np

from . import _lib
from . import _device as dv
from .fftlog import CorrelationToPower, PowerToCorrelation, TophatVariance
from .spline import LinearOperator, dense_operator


def get_default_k_callable():
    """Default k grid of interpolators built from callables (reference interpolator.py:18-27)."""
    return np.concatenate([np.logspace(-5, -4, num=20, endpoint=False), np.logspace(-4, -3, num=40, endpoint=False),
                           np.logspace(-3, -2, num=60, endpoint=False), np.logspace(-2, -1, num=80, endpoint=False),
                           np.logspace(-1, 0, num=100, endpoint=False), np.logspace(0, 2, num=240, endpoint=True)])


def get_default_z_callable():
    """Default z grid (reference interpolator.py:34-35)."""
    return np.linspace(0., 10.**0.5, 30)**2


_default_extrap_kmin = 1e-7
_default_extrap_kmax = 1e2


def _pad_log(k, pk, extrap_kmin=_default_extrap_kmin, extrap_kmax=_default_extrap_kmax):
    """Pad (k, pk) with two log-log linearly extrapolated points on each side (reference interpolator.py:42-87). Host numpy; axis 0 = k."""
    logk = np.log10(k)
    logpk = np.log10(pk)
    log_extrap_kmin = np.log10(np.minimum(extrap_kmin, k[0] * (1 - 1e-9)))
    log_extrap_kmax = np.log10(np.maximum(extrap_kmax, k[-1] * (1 + 1e-9)))
    dlogpkdlogk = (logpk[-1] - logpk[-2]) / (logk[-1] - logk[-2])
    padhighk = np.array([logk[-1] * 0.1 + log_extrap_kmax * 0.9, log_extrap_kmax])
    padhighpk = np.array([logpk[-1] + dlogpkdlogk * (padhighk[0] - logk[-1]), logpk[-1] + dlogpkdlogk * (padhighk[1] - logk[-1])])
    dlogpkdlogk = (logpk[1] - logpk[0]) / (logk[1] - logk[0])
    padlowk = np.array([log_extrap_kmin, logk[0] * 0.1 + log_extrap_kmin * 0.9])
    padlowpk = np.array([logpk[0] + dlogpkdlogk * (padlowk[0] - logk[0]), logpk[0] + dlogpkdlogk * (padlowk[1] - logk[0])])
    return np.concatenate([padlowk, logk, padhighk], axis=0), np.concatenate([padlowpk, logpk, padhighpk], axis=0)


def _mask_bounds(x, xlim, bounds_error=False):
    """Masks of in-range values; ``bounds_error`` raises ValueError like the reference (jax.py:120-131). Host numpy."""
    masks = [(xx >= lim[0]) & (xx <= lim[1]) for xx, lim in zip(x, xlim)]
    if bounds_error:
        for mask, xx, lim in zip(masks, x, xlim):
            if not mask.all():
                raise ValueError('input outside of extrapolation range (min: {} vs. {}; max: {} vs. {})'.format(xx.min(), lim[0], xx.max(), lim[1]))
    return masks


_host_copies = {}   # id(tensor) -> (weak reference, tensor version, host copy)


def _host(x):
    """numpy float64 copy of a number / array / tensor (query coordinates are small and define host-built operators).  The copy of a small
    device tensor is remembered while the tensor lives and is not written to: the same grid of radii or redshifts passed call after call
    costs one device-to-host copy (which synchronises the stream), not one per call.  The copy is validated by the tensor's identity and torch
    version counter: writes that bypass that counter (a foreign kernel writing through ``data_ptr()``, a HIP-graph replay into the tensor) are not
    seen -- query grids are inputs, keep them constant or pass a new tensor.  Callers that store the result as a public attribute copy it."""
    if dv.is_torch(x):
        if x.numel() > 65536 or not x.is_cuda:
            return dv.to_host(x).astype('f8', copy=False)
        import weakref
        key = id(x)
        entry = _host_copies.get(key)
        if entry is not None and entry[0]() is x and entry[1] == x._version:
            return entry[2]
        copy = dv.to_host(x).astype('f8', copy=False)
        copy.setflags(write=False)
        _host_copies[key] = (weakref.ref(x, lambda _, key=key: _host_copies.pop(key, None)), x._version, copy)
        return copy
    return np.asarray(x, dtype='f8')


def _finish(t, dtype, like_torch, shape=None):
    """Device tensor -> caller's container (numpy unless any input was a torch tensor) with the reference's dtype rule."""
    torch = dv.torch()
    if shape is not None:
        t = t.reshape(shape)
    if like_torch:
        return t.to(torch.float32 if dtype == np.float32 else torch.float64)
    return dv.to_host(t).astype(dtype, copy=False)


def _call_options(kwargs, from_callable, names=('bounds_error',)):
    """The ``**kwargs`` of the interpolators' ``__call__`` (reference interpolator.py:495, 741, 1142, 1337: they go to the inner evaluation):
    the named options (False by default) and what is left, which a callable-built interpolator hands to its callable, as the reference does;
    for a tabulated one the reference hands it to scipy's spline, which refuses unknown names -- so does this."""
    options = [bool(kwargs.pop(name, False)) for name in names]
    if kwargs and not from_callable:
        raise TypeError('__call__() got an unexpected keyword argument {!r}'.format(sorted(kwargs)[0]))
    return options + [kwargs]


def _simpson_weights(x):
    """Weights w with simpson(y, x) == w @ y for the reference's rule (jax.py:365-507: scipy v1.0.0 simpson, even='avg'). Host numpy."""
    n = x.size

    def pairs(xx):
        h = np.diff(xx)
        h0, h1 = h[0::2], h[1::2]
        hsum, hprod, ratio = h0 + h1, h0 * h1, h0 / h1
        w = np.zeros(xx.size)
        np.add.at(w, np.arange(0, xx.size - 2, 2), hsum / 6.0 * (2 - 1.0 / ratio))
        np.add.at(w, np.arange(1, xx.size - 1, 2), hsum / 6.0 * hsum * hsum / hprod)
        np.add.at(w, np.arange(2, xx.size, 2), hsum / 6.0 * (2 - ratio))
        return w

    if n % 2 == 1:
        return pairs(x)
    w = np.zeros(n)
    w[:-1] += pairs(x[:-1]) / 2.0
    w[1:] += pairs(x[1:]) / 2.0
    w[-1] += 0.25 * (x[-1] - x[-2])
    w[-2] += 0.25 * (x[-1] - x[-2])
    w[0] += 0.25 * (x[1] - x[0])
    w[1] += 0.25 * (x[1] - x[0])
    return w


def kernel_tophat2(x):
    """W(x)^2, W = 3 (sin x - x cos x) / x^3 with the Maclaurin series below 0.1 (reference interpolator.py:90-120). Host numpy."""
    x = np.asarray(x, dtype='f8')
    x2 = x**2
    low = 1. + x2 * (-1.0 / 10.0 + x2 * (1.0 / 280.0 + x2 * (-1.0 / 15120.0 + x2 * (1.0 / 1330560.0 + x2 * (-1.0 / 172972800.0)))))
    with np.errstate(all='ignore'):
        high = 3. * (np.sin(x) - x * np.cos(x)) / x**3
    return np.where(x < 0.1, low, high)**2


_tophat_cache = {}      # (kmin, kmax, nk, device) -> TophatVariance plan; bounded: _tophat_plan


def _tophat_plan(key, k, device):
    """The TophatVariance plan (tables on the device) of a k grid, kept for the next sigma integral on the same grid.  A sampler that varies the range
    or the size of the grid would otherwise leave one plan per value behind (host tables + device memory): at most 16 are kept."""
    fft = _tophat_cache.get(key)
    if fft is None:
        if len(_tophat_cache) >= 16:
            _tophat_cache.clear()
        fft = _tophat_cache[key] = TophatVariance(k, device=device)
    return fft

_op_cache = {}


_fftlog_cache = {}


def _cached_fftlog(cls, x, device, fftlog_kwargs):
    """FFTLog plan (tables on the device) of transform ``cls`` on abscissa ``x``: built once per (grid, options, device) instead of at every
    ``to_xi`` / ``to_pk`` -- a likelihood calls them at every step with the same grid."""
    try:
        key = (cls.__name__, x.tobytes(), tuple(sorted((fftlog_kwargs or {}).items())), device.index)
        hash(key)
    except TypeError:       # an option that cannot be a key (an array ...): no caching
        return cls(x, complex=False, device=device, **(fftlog_kwargs or {}))
    if key not in _fftlog_cache:
        if len(_fftlog_cache) > 16:
            _fftlog_cache.clear()
        _fftlog_cache[key] = cls(x, complex=False, device=device, **(fftlog_kwargs or {}))
    return _fftlog_cache[key]


def _cached_operator(key, build):
    if key not in _op_cache:
        if len(_op_cache) > 64:
            _op_cache.clear()
        _op_cache[key] = build()
    return _op_cache[key]


# The fused kernel reads a query's band of weights from L2 once per PAIR of rows (the separate spline kernel: once per 16 rows) and gives one
# query to one thread: it pays for batches that are small enough to be launch / latency bound and wide enough in queries to keep its threads
# busy (tools/bench_fused_spline.py); large batches (config 3B: 640 000 rows) and single radii (sigma8) take the two kernels.
_FUSED_SPLINE_ROWS = (2, 8192)
_TRANSPOSE_IN_STORE = True
_FUNCTIONAL_RADII = 4          # sigma_rz_analytic: up to this many radii as dot products with the spectrum (cp_sigma_rz_functional)
_DIRECT_K_SPLINE = True        # batches of (k, z) tables: the k splines evaluated from the tables' second derivatives (cp_tables_rows_direct) instead of multiplied
_PAIRED_TABLES = True          # ... from (value, second derivative) pairs in one array, cp_spline_rows_pairs (False: tables and second derivatives as two arrays)


def _fftlog_then_spline(fft, op, rows, device, sqrt=False):
    """The transform of ``rows`` (..., nk) and the spline of every transformed row to the operator's queries as ONE kernel
    (``cp_fftlog_spline_execute``: the transformed rows stay on the CU), for the default transform (1024 samples); None when it does not apply."""
    torch = dv.torch()
    lib = _lib.load()
    if not dv.is_torch(rows) or rows.dtype != torch.float64 or rows.shape[-1] != fft.size or fft.nparallel != 1:
        return None
    plan = fft._get_plan(device)
    if getattr(fft, '_phase', None) is not None or getattr(fft, '_phase_in', None) is not None or not lib.cp_sigma_rz_fused_available(plan.handle, op._handle):
        return None
    rows = rows.contiguous()
    lead = tuple(rows.shape[:-1])
    nrows = int(np.prod(lead, dtype=np.int64))
    if not _FUSED_SPLINE_ROWS[0] <= nrows <= _FUSED_SPLINE_ROWS[1] or op.nq < 32:
        return None
    out = torch.empty(lead + (op.nq,), dtype=torch.float64, device=device)
    if nrows:
        _lib.check(lib.cp_fftlog_spline_execute(plan.handle, op._handle, rows.data_ptr(), out.data_ptr(), nrows, int(bool(sqrt)), dv.stream_of(device)))
    return out


class _GeoSpline(object):

    """Owner of a ``cp_geospline_plan``: the natural spline from a geometric grid (the output grid of an FFTLog) to fixed queries, evaluated inside the
    FFTLog kernel.  With ``fft`` (and :data:`_GEOSPLINE_PREFILTERED`) the plan owns a copy of that transform with the B-spline prefilter folded into
    its ``u`` (``prefiltered`` True: execute takes no FFTLog plan); otherwise, or where the library refuses that (a postfactor that is no power law),
    the spline is solved on the CU from the transform's ordinary output.  ``handle`` is None where the library refuses both (queries near the ends
    of the grid, too wide a span)."""

    def __init__(self, knots, queries, device, fft=None, keep=None):
        import ctypes
        self.handle, self.nq, self.prefiltered, self._keep = None, int(np.size(queries)), False, keep
        handle = ctypes.c_void_p()
        knots, queries = np.ascontiguousarray(knots, dtype='f8'), np.ascontiguousarray(queries, dtype='f8')
        lib = _lib.load()
        if fft is not None and _GEOSPLINE_PREFILTERED and fft.nparallel == 1 and not any(np.iscomplexobj(t) for t in (fft.padded_prefactor, fft.padded_postfactor)):
            pre, post = (np.ascontiguousarray(t, dtype='f8').ravel() for t in (fft.padded_prefactor, fft.padded_postfactor))
            u = np.ascontiguousarray(fft.padded_u, dtype='c16').ravel()
            status = lib.cp_geospline_plan_create_prefiltered(ctypes.byref(handle), fft.size, fft.padded_size, _lib.as_double_p(pre), _lib.as_double_p(post),
                                                              _lib.as_double_p(u.view('f8')), _lib.as_double_p(knots), _lib.as_double_p(queries), queries.size,
                                                              device.index)
            if status == _lib.CP_OK:
                self.handle, self.prefiltered = handle, True
                return
            if status != _lib.CP_EUNSUPPORTED:
                _lib.check(status)
        status = lib.cp_geospline_plan_create(ctypes.byref(handle), _lib.as_double_p(knots), knots.size, _lib.as_double_p(queries), queries.size, device.index)
        if status == _lib.CP_EUNSUPPORTED:
            return
        _lib.check(status)
        self.handle = handle

    def __del__(self):
        try:
            if self.handle:
                _lib.load().cp_geospline_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_GEOSPLINE_MIN_ROWS = 8193      # below: the band-operator kernel of _fftlog_then_spline (launch- / latency-bound regime) ...
_GEOSPLINE_PREFILTERED_MIN_ROWS = 513      # ... unless the prefiltered form applies: it is the faster one from ~500 rows on (tools/bench_spline_crossover.py: 1024 rows
#                                            0.0154 against 0.0193 ms, 8192 rows 0.044 against 0.075)
_GEOSPLINE_PREFILTERED = True      # the spline's solve folded into the transform's u (False: solved on the CU from the ordinary output; measurements)
_SIGMA_RZ_PREFILTERED = True       # ... also in the fused sigma(r, z) kernel of the analytic engines (False: the banded operator out of L2; measurements)
_GEOSPLINE_GROUPED = True      # sigma_rz's (..., nr, nz) layout written by the kernel itself (False: (..., nz, nr) and a transposed view; measurements)


def _fftlog_then_geospline(fft, s, rr, rows, device, sqrt=False, group=0, prefiltered_only=False):
    """The transform of ``rows`` (..., nk) and the natural spline of every transformed row to the radii ``rr`` as ONE kernel with the spline solved
    on the CU (``cp_fftlog_geospline_execute``), for the default transform (1024 samples).  group > 0: rows (..., group, nk) -> (..., nr, group).
    None when it does not apply."""
    torch = dv.torch()
    if not dv.is_torch(rows) or rows.dtype != torch.float64 or rows.shape[-1] != fft.size or fft.nparallel != 1 or fft.size != 1024 or fft.padded_size != 2048:
        return None
    if getattr(fft, '_phase', None) is not None or getattr(fft, '_phase_in', None) is not None:
        return None
    # (keyed by the transform's library plan, rebuilt whenever its tables change; the spline plan holds it, so that its id stays its own)
    native = fft._get_plan(device)
    plan = _cached_operator(('geospline', id(native), _GEOSPLINE_PREFILTERED, s.tobytes(), rr.tobytes(), device.index),
                            lambda: _GeoSpline(s, rr, device, fft=fft, keep=native))
    if plan.handle is None or (prefiltered_only and not plan.prefiltered):
        return None
    rows = rows.contiguous()
    lead = tuple(rows.shape[:-1])
    nrows = int(np.prod(lead, dtype=np.int64))
    if group:
        if len(lead) < 1 or lead[-1] != group or group % 2:
            return None
        oshape = lead[:-1] + (plan.nq, group)
    else:
        oshape = lead + (plan.nq,)
    out = torch.empty(oshape, dtype=torch.float64, device=device)
    if nrows:
        _lib.check(_lib.load().cp_fftlog_geospline_execute(None if plan.prefiltered else native.handle, plan.handle, rows.data_ptr(), out.data_ptr(), nrows, int(group),
                                                           int(bool(sqrt)), dv.stream_of(device)))
    return out


def _with_growth(sigma2, growth_sq):
    """sqrt(sigma2[..., r] growth_sq[..., z]) as (..., nr, nz) (methods other than 'fftlog': elementwise), or sigma2 as it is."""
    if growth_sq is None:
        return sigma2
    return (sigma2[..., :, None] * growth_sq[..., None, :]).sqrt()


def sigma_r2_of_rows(r, pk_rows, kmin=1e-7, kmax=1e2, method='fftlog', nk=None, device=None, growth_sq=None, epsabs=1e-5, epsrel=1e-5, sqrt=False,
                      radii_before_last_axis=False, kernel=None):
    r"""
    :math:`\sigma_r^2 = \frac{1}{2\pi^2}\int dk\,k^2 P(k) W^2(kr)` (reference interpolator.py:200-292) for rows of P(k).

    pk_rows : callable k (numpy, (nk,)) -> device tensor (..., nk): the power spectra sampled at k, k fastest.
    method : 'fftlog' (default), 'simpson', 'leggauss' or 'quad'.  Returns a device tensor (..., nr) holding :math:`\sigma_r^2`.
        'quad' meets ``epsabs`` / ``epsrel`` (the tolerances the reference hands to ``scipy.integrate.quad``, interpolator.py:255-273)
        for every (row, r) by refining a composite Simpson rule in log k for the whole batch at once: see :func:`_refined_rule`.
    growth_sq : optional device tensor (..., nz), one row of factors per row of P(k): the result is then
        :math:`\sqrt{\sigma_r^2\,\mathrm{growth\_sq}(z)}` of shape (..., nr, nz), written once by the interpolation kernel.
    sqrt : return :math:`\sigma_r` instead (the root taken by the kernel that interpolates to ``r``, method 'fftlog').
    radii_before_last_axis : rows (..., nz, nk): return (..., nr, nz) instead of (..., nz, nr) (method 'fftlog', large batches: the transposition
        is part of the spline kernel's store; other cases: a transposed view).
    kernel : host callable x -> W^2(x) of the quadrature methods (default :func:`kernel_tophat2`); 'fftlog' is the top-hat variance whatever it is,
        as in the reference (interpolator.py:285-288).
    """
    device = dv.resolve_device(device)
    window, window_key = (kernel_tophat2, None) if kernel is None or kernel is kernel_tophat2 else (kernel, id(kernel))
    if radii_before_last_axis and (method != 'fftlog' or growth_sq is not None):
        out = sigma_r2_of_rows(r, pk_rows, kmin=kmin, kmax=kmax, method=method, nk=nk, device=device, growth_sq=growth_sq, epsabs=epsabs, epsrel=epsrel, sqrt=sqrt,
                                kernel=kernel)
        return out.transpose(-1, -2) if growth_sq is None and out.ndim >= 2 else out
    rr = _host(r).ravel()
    nk_leggauss = 100 if nk is None else nk
    if nk is None:
        nk = 1024
    if rr.size == 0:
        rows = pk_rows(np.geomspace(kmin, kmax, 4))
        if radii_before_last_axis and rows.ndim >= 2:
            return rows.new_empty(tuple(rows.shape[:-2]) + (0, rows.shape[-2]))
        return rows.new_empty(tuple(rows.shape[:-1]) + (0,))
    if method == 'fftlog':
        k = np.geomspace(kmin, kmax, nk)
        key = (float(kmin), float(kmax), int(nk), device.index)
        fft = _tophat_plan(key, k, device)
        rows = pk_rows(k)
        s = fft.y[0]
        op = _cached_operator(('nat', s.tobytes(), rr.tobytes(), device.index), lambda: LinearOperator.spline(s, rr, bc='natural', device=device))
        # tmp = (2 pi^2) spline(var)(r); sigma^2 = tmp / (2 pi^2)  (interpolator.py:289-291)
        nrows = rows.numel() // max(rows.shape[-1], 1) if dv.is_torch(rows) else 0

        def geospline(prefiltered_only):
            # transform and spline in one kernel, the spline from B-spline coefficients (or solved on the CU), sigma_rz's layout written by it
            group = int(rows.shape[-2]) if radii_before_last_axis and rows.ndim >= 2 and _TRANSPOSE_IN_STORE and _GEOSPLINE_GROUPED else 0
            solved = _fftlog_then_geospline(fft, s, rr, rows, device, sqrt=sqrt, group=group if group % 2 == 0 else 0, prefiltered_only=prefiltered_only)
            if solved is not None and radii_before_last_axis and rows.ndim >= 2 and not (group and group % 2 == 0):
                return solved.transpose(-1, -2)
            return solved

        if growth_sq is None and _GEOSPLINE_PREFILTERED and _GEOSPLINE_PREFILTERED_MIN_ROWS <= nrows < _GEOSPLINE_MIN_ROWS:
            solved = geospline(True)      # medium batches: only the prefiltered form beats the band operator inside the transform's kernel
            if solved is not None:
                return solved
        fused = _fftlog_then_spline(fft, op, rows, device, sqrt=sqrt) if growth_sq is None else None
        if fused is not None:
            return fused.transpose(-1, -2) if radii_before_last_axis and fused.ndim >= 2 else fused
        if growth_sq is None and nrows >= _GEOSPLINE_MIN_ROWS:      # many rows (config 3B: 640 000)
            solved = geospline(False)
            if solved is not None:
                return solved
        if growth_sq is not None:
            var = fft(rows)[1]
            growth_sq = dv.to_device(growth_sq, device)
            if growth_sq.ndim < var.ndim:      # one growth factor for the whole batch
                growth_sq = growth_sq.expand(tuple(var.shape[:-1]) + (growth_sq.shape[-1],)).contiguous()
            return op.outer(var, growth_sq, sqrt=True)
        var = fft(rows, out_window=op.columns)[1]      # the radii see a part of the FFTLog grid: only that part is written
        if radii_before_last_axis and var.ndim >= 2:      # (..., nz, nk) -> (..., nr, nz): sigma_rz's layout written by the spline kernel itself
            if _TRANSPOSE_IN_STORE:
                return op(var, sqrt=sqrt, last_axis_first=True)
            return op(var, sqrt=sqrt).transpose(-1, -2)
        return op(var, sqrt=sqrt)
    if sqrt:
        return sigma_r2_of_rows(r, pk_rows, kmin=kmin, kmax=kmax, method=method, nk=nk if method != 'leggauss' else nk_leggauss, device=device,
                                  growth_sq=growth_sq, epsabs=epsabs, epsrel=epsrel, kernel=kernel).sqrt()
    if method == 'simpson':
        limits = (np.log(kmin * (1. + 1e-9)), np.log(kmax * (1. - 1e-9)))
        logk = np.linspace(*limits, nk)
        k = np.exp(logk)

        def build():
            w = _simpson_weights(logk)
            return LinearOperator.dense(1. / (2. * np.pi**2) * np.asarray(window(k[None, :] * rr[:, None])) * (k**3 * w)[None, :], device=device)

        op = build() if window_key else _cached_operator(('simpson_r', float(kmin), float(kmax), int(nk), rr.tobytes(), device.index), build)
        return _with_growth(op(pk_rows(k)), growth_sq)
    if method == 'leggauss':   # "not accurate" in the reference's own words (interpolator.py:274-280); nk = 100 nodes by default
        nl = nk_leggauss
        limits = (np.log(kmin * (1. + 1e-9)), np.log(kmax * (1. - 1e-9)))
        x, wx = np.polynomial.legendre.leggauss(nl)
        logk = (limits[1] - limits[0]) / 2. * (1. + x) + limits[0]
        k = np.exp(logk)
        w = (limits[1] - limits[0]) / 2. * wx
        build = lambda: LinearOperator.dense(1. / (2. * np.pi**2) * np.asarray(window(k[None, :] * rr[:, None])) * (k**3 * w)[None, :], device=device)  # noqa: E731
        op = build() if window_key else _cached_operator(('leggauss_r', float(kmin), float(kmax), int(nl), rr.tobytes(), device.index), build)
        return _with_growth(op(pk_rows(k)), growth_sq)
    if method == 'quad':
        limits = (np.log(kmin * (1. + 1e-9)), np.log(kmax * (1. - 1e-9)))

        def rule(logk):
            k = np.exp(logk)
            return 1. / (2. * np.pi**2) * np.asarray(window(k[None, :] * rr[:, None])) * (k**3 * _simpson_weights(logk))[None, :]

        return _with_growth(_refined_rule(rule, limits, pk_rows, epsabs / (2. * np.pi**2), epsrel, device, what='sigma_r2'), growth_sq)
    raise NotImplementedError('integrate_sigma_r2 method {} is not available on the GPU path (use "fftlog", "simpson", "leggauss" or "quad")'.format(method))


def _refined_rule(rule, limits, pk_rows, epsabs, epsrel, device, what='integral', start=1025, stop=2**17 + 1):
    """
    ``method='quad'`` of the sigma integrals (reference interpolator.py:167-177, 255-273: one adaptive QUADPACK integration per
    (r, column), sequential and data-dependent).  Here one composite Simpson rule in log k serves every row of the batch and is
    refined globally: the number of intervals doubles until two successive estimates agree within ``max(epsabs, epsrel |I|)`` for
    every (row, r) -- the error of the finer one is then about 1 / 15 of that, inside what ``quad`` promises.  One reduction and
    one host synchronisation per level.

    rule : logk (n,) -> (nout, n) weights of the rule (kernel and Jacobian included);  returns the device tensor (..., nout).
    """
    torch = dv.torch()
    previous, n = None, start
    while True:
        logk = np.linspace(*limits, n)
        current = LinearOperator.dense(rule(logk), device=device)(pk_rows(np.exp(logk)))
        if previous is not None:
            bound = torch.clamp(epsrel * current.abs(), min=epsabs)
            if bool((((current - previous).abs() <= bound) | ~torch.isfinite(current)).all()):   # NaN / Inf entries cannot converge
                return current
        if n >= stop:
            import warnings
            warnings.warn('{}: requested accuracy not reached with {:d} Simpson nodes in log k'.format(what, n))
            return current
        previous, n = current, 2 * n - 1


def sigma_d2_of_rows(pk_rows, kmin=1e-7, kmax=1e2, method='simpson', nk=None, device=None, epsabs=1e-5, epsrel=1e-5):
    r""":math:`\sigma_d^2 = \frac{1}{6\pi^2}\int dk\,P(k)` (reference interpolator.py:123-197, default 'simpson'); device tensor (...,)."""
    device = dv.resolve_device(device)
    limits = (np.log(kmin * (1. + 1e-9)), np.log(kmax * (1. - 1e-9)))
    if method == 'leggauss':   # interpolator.py:183-189
        nl = 100 if nk is None else nk
        x, wx = np.polynomial.legendre.leggauss(nl)
        logk = (limits[1] - limits[0]) / 2. * (1. + x) + limits[0]
        k = np.exp(logk)
        w = (limits[1] - limits[0]) / 2. * wx
        op = _cached_operator(('leggauss_d', float(kmin), float(kmax), int(nl), device.index),
                              lambda: LinearOperator.dense((1. / (6. * np.pi**2) * k * w)[None, :], device=device))
        return op(pk_rows(k))[..., 0]
    if method == 'quad':   # interpolator.py:167-177

        def rule(logk):
            return (1. / (6. * np.pi**2) * np.exp(logk) * _simpson_weights(logk))[None, :]

        return _refined_rule(rule, limits, pk_rows, epsabs / (6. * np.pi**2), epsrel, device, what='sigma_d2')[..., 0]
    if method != 'simpson':
        raise NotImplementedError('integrate_sigma_d2 method {} is not available on the GPU path (use "simpson", "leggauss" or "quad")'.format(method))
    if nk is None:
        nk = 1024
    logk = np.linspace(*limits, nk)
    k = np.exp(logk)
    op = _cached_operator(('simpson_d', float(kmin), float(kmax), int(nk), device.index),
                          lambda: LinearOperator.dense((1. / (6. * np.pi**2) * k * _simpson_weights(logk))[None, :], device=device))
    return op(pk_rows(k))[..., 0]


def _rows_of_callable(pk, device):
    """The reference's callable convention ``pk(k) -> (nk,) + pshape`` as rows for the kernels: k -> device tensor (ncol, nk)."""
    def rows(k):
        p = pk(k)
        p = p if dv.is_torch(p) else np.asarray(p)
        t = dv.to_device(p, device)
        return t.reshape(t.shape[0], -1).T.contiguous()
    return rows


def _finish_sigma2(t, lead, p, *likes):
    """Device tensor (ncol, ...) -> the reference's result ``lead + pshape`` in the float dtype of the inputs (torch if ``pk`` returned torch)."""
    pshape = tuple(p.shape)
    t = t.movedim(0, -1).reshape(tuple(lead) + pshape)
    dtype = dv.float_dtype(*[x for x in likes if x is not None])
    return _finish(t, dtype, dv.is_torch(p))


def integrate_sigma_d2(pk, kmin=1e-7, kmax=1e2, method='simpson', epsabs=1e-5, epsrel=1e-5, nk=None, device=None):
    r"""
    Variance of the displacement field :math:`\sigma_{d}^{2} = \frac{1}{6 \pi^{2}} \int dk P(k)`, by the reference's name, arguments and callable
    convention (interpolator.py:123-197): ``pk`` callable, ``pk(k)`` of shape ``(nk,)`` or ``(nk, ncol)`` for ``k`` of shape ``(nk,)``, a scalar or
    ``(ncol,)`` for a scalar; methods 'simpson' (default, nk = 1024), 'leggauss' (nk = 100), 'quad' (``epsabs`` / ``epsrel``; :func:`_refined_rule`);
    'romberg' raises here as it does upstream with numpy inputs.  Returns an array of the shape of ``pk(kmin)``.
    The integrals run on the device (:func:`sigma_d2_of_rows`, which the interpolators call with device rows directly).
    """
    device = dv.resolve_device(device)
    p = pk(kmin)
    p = p if dv.is_torch(p) else np.asarray(p)
    if not int(np.prod(tuple(p.shape), dtype='i8')):
        return _finish(dv.torch().zeros(tuple(p.shape), dtype=dv.torch().float64, device=device), dv.float_dtype(p), dv.is_torch(p))
    out = sigma_d2_of_rows(_rows_of_callable(pk, device), kmin=kmin, kmax=kmax, method=method, nk=nk, device=device, epsabs=epsabs, epsrel=epsrel)   # (ncol,)
    return _finish_sigma2(out, (), p, p)


def integrate_sigma_r2(r, pk, kmin=1e-7, kmax=1e2, method='fftlog', epsabs=1e-5, epsrel=1e-5, nk=None, kernel=kernel_tophat2, device=None):
    r"""
    Variance of perturbations smoothed by a kernel :math:`W` of radius :math:`r`,
    :math:`\sigma_{r}^{2} = \frac{1}{2 \pi^{2}} \int dk k^{2} P(k) W^{2}(kr)`, by the reference's name, arguments and callable convention
    (interpolator.py:200-292): ``pk`` callable as in :func:`integrate_sigma_d2`; methods 'fftlog' (default: :class:`TophatVariance` on nk = 1024
    wavenumbers + natural spline to ``r``; ``kernel`` is not used, as in the reference), 'simpson' (nk = 1024), 'leggauss' (nk = 100), 'quad';
    ``kernel`` : host callable :math:`x \mapsto W^2(x)` of the quadrature methods.  Returns an array of shape ``r.shape + pk(kmin).shape``.
    The transform / the quadrature run on the device (:func:`sigma_r2_of_rows`).
    """
    device = dv.resolve_device(device)
    p = pk(kmin)
    p = p if dv.is_torch(p) else np.asarray(p)
    rshape = tuple(np.shape(_host(r)))
    if not int(np.prod(tuple(p.shape), dtype='i8')):
        return _finish(dv.torch().zeros(rshape + tuple(p.shape), dtype=dv.torch().float64, device=device), dv.float_dtype(r, p if p.shape else None), dv.is_torch(p))
    out = sigma_r2_of_rows(r, _rows_of_callable(pk, device), kmin=kmin, kmax=kmax, method=method, nk=nk, device=device, epsabs=epsabs, epsrel=epsrel,
                           kernel=kernel)      # (ncol, nr)
    return _finish_sigma2(out.reshape(out.shape[0], *rshape), rshape, p, r, p if p.shape else None)


_MAX_OPERATOR_WEIGHTS = 1 << 29      # 4 GB of float64 on the host


def _check_operator_size(nq, n):
    """Dense (queries x knots) operators are built on the host: beyond 4 GB the caller must come in pieces (a host that runs out of memory is
    killed, it does not raise)."""
    if int(nq) * int(n) > _MAX_OPERATOR_WEIGHTS:
        raise MemoryError('a dense interpolation operator of {:d} queries x {:d} knots: evaluate in pieces'.format(int(nq), int(n)))


def _linear_interp_operator(x, xq, extrap=False):
    """(nq, n) matrix of piecewise-linear interpolation on the knots ``x`` (scipy interp1d kind='linear', reference jax.py:176-177):
    linear continuation of the end intervals if ``extrap``, else NaN rows outside [x[0], x[-1]].  Host numpy."""
    _check_operator_size(xq.size, x.size)
    i = np.clip(np.searchsorted(x, xq, side='right') - 1, 0, x.size - 2)
    t = (xq - x[i]) / (x[i + 1] - x[i])
    w = np.zeros((xq.size, x.size))
    rows = np.arange(xq.size)
    w[rows, i] = 1. - t
    w[rows, i + 1] = t
    if not extrap:
        w[~((xq >= x[0]) & (xq <= x[-1]))] = np.nan
    return w


def _bspline_basis(t, k, x):
    """Dense (len(x), len(t) - k - 1) matrix of B-spline basis values B_j(x) (Cox-de Boor), with polynomial extrapolation of the end pieces."""
    n = t.size - k - 1
    _check_operator_size(x.size, n)
    out = np.zeros((x.size, n))
    for ix, xv in enumerate(x):
        ell = np.searchsorted(t, xv, side='right') - 1
        ell = min(max(ell, k), n - 1)          # interval [t_ell, t_ell+1); clamped: extrapolate with the end polynomial
        b = np.zeros(k + 1)
        b[0] = 1.
        for d in range(1, k + 1):
            saved = 0.
            for r in range(d):
                left, right = t[ell + r + 1 - d], t[ell + r + 1]
                tmp = b[r] / (right - left)
                b[r] = saved + (right - xv) * tmp
                saved = (xv - left) * tmp
            b[d] = saved
        out[ix, ell - k:ell + 1] = b
    return out


def _quadratic_interp_operator(xk, xq, extrap=True):
    """Dense operator of ``interp1d(xk, ., kind=2, fill_value='extrapolate')(xq)`` (scipy make_interp_spline(k=2));
    NaN rows outside [xk[0], xk[-1]] unless ``extrap``."""
    k = 2
    mid = (xk[1:] + xk[:-1]) / 2.
    t = np.concatenate([(xk[0],) * (k + 1), mid[1:-1], (xk[-1],) * (k + 1)])
    colloc = _bspline_basis(t, k, xk)
    w = _bspline_basis(t, k, xq).dot(np.linalg.inv(colloc))
    if not extrap:
        w[~((xq >= xk[0]) & (xq <= xk[-1]))] = np.nan
    return w


def _fitpack_interp_operator(xk, xq, k):
    """Dense (nq, n) operator of the interpolating spline of degree ``k`` (1...5) FITPACK builds with ``s=0`` (``RectBivariateSpline`` along
    one axis, reference jax.py:241-242): interior knots at the data points for odd degrees, half-way between them for even ones (fpregr /
    fpgrre); queries beyond the data are clamped to the end points, as ``bispev`` does."""
    n = xk.size
    if not 1 <= k <= 5:
        raise ValueError('spline degrees 1 <= k <= 5 are supported')       # FITPACK's own limit
    if n <= k:
        raise ValueError('at least {:d} points are needed for degree {:d}'.format(k + 1, k))
    if k % 2:
        interior = xk[(k + 1) // 2:n - (k + 1) // 2]
    else:
        interior = ((xk[1:] + xk[:-1]) / 2.)[k // 2:n - 1 - k // 2]
    t = np.concatenate([(xk[0],) * (k + 1), interior, (xk[-1],) * (k + 1)])
    colloc = _bspline_basis(t, k, xk)
    return np.linalg.solve(colloc.T, _bspline_basis(t, k, np.clip(xq, xk[0], xk[-1])).T).T


def _natural_spline_slopes(x, y):
    """First derivatives at the knots ``x`` of the natural cubic splines through the columns of ``y`` (n, ncol): the tridiagonal system of
    scipy's CubicSpline(bc_type='natural'), solved with the same banded LAPACK routine.  Host numpy; a column holding NaN comes out NaN."""
    from scipy.linalg import solve_banded
    n = x.size
    dx = np.diff(x)
    slope = np.diff(y, axis=0) / dx[:, None]
    ab = np.zeros((3, n))
    ab[1, 1:-1] = 2. * (dx[:-1] + dx[1:])      # diagonal
    ab[0, 2:] = dx[:-1]                          # upper
    ab[2, :-2] = dx[1:]                          # lower
    rhs = np.empty_like(y)
    rhs[1:-1] = 3. * (dx[1:, None] * slope[:-1] + dx[:-1, None] * slope[1:])
    ab[1, 0], ab[0, 1], rhs[0] = 2., 1., 3. * slope[0]
    ab[1, -1], ab[2, -2], rhs[-1] = 2., 1., 3. * slope[-1]
    finite = np.isfinite(rhs).all(axis=0)
    out = np.full_like(y, np.nan)
    if finite.any():
        out[:, finite] = solve_banded((1, 1), ab, rhs[:, finite], overwrite_ab=False, overwrite_b=False, check_finite=False)
    return out


class Interpolator1D(dv.Copyable):

    """1D interpolation along axis 0 of ``fun`` (n, ...) in lin or log10 space; natural cubic spline (reference jax.py:135-209)."""

    def __init__(self, x, fun, k=3, interp_x='lin', interp_fun='lin', extrap=False, assume_sorted=False, device=None):
        if int(k) not in (1, 2, 3):
            raise NotImplementedError('only linear (k=1), quadratic (k=2) and cubic (k=3) interpolation are implemented on the GPU path')
        self.k = int(k)
        self.device = dv.resolve_device(device, fun)
        self.interp_x, self.interp_fun, self.extrap = str(interp_x), str(interp_fun), bool(extrap)
        x = _host(x).ravel()
        fun = dv.to_device(fun, self.device)
        self.shape = tuple(fun.shape[1:])
        if not assume_sorted:
            ix = np.argsort(x)
            x = x[ix]
            fun = fun[dv.upload(ix, self.device)]
        self.xmin, self.xmax = x[0], x[-1]
        self._x = np.log10(x) if self.interp_x == 'log' else x
        fun = fun.reshape(x.size, -1)
        if self.interp_fun == 'log':
            fun = dv.torch().log10(fun)
        self._rows = fun.T.contiguous()   # (ncol, n): one spline per row
        # NaN rule of the reference (jax.py:161-172): columns that are NaN throughout are set aside and stay NaN; if any OTHER column holds a
        # NaN (e.g. the log of a negative P at a few knots) no spline is built at all and every column evaluates to NaN, without raising.
        # +-Inf knots (the log of P = 0) make scipy's CubicSpline raise in the reference; here such a column evaluates to NaN.
        torch = dv.torch()
        nan = torch.isnan(self._rows)
        all_nan, some_nan = nan.all(dim=1), nan.any(dim=1)
        self._nan_rows = some_nan | ~torch.isfinite(self._rows).all(dim=1)
        # ... linear interpolation (scipy's interp1d in the reference, jax.py:176-177) lets a NaN datum spoil the two intervals next to it and nothing else:
        # only the columns that are NaN throughout are NaN throughout; the others are interpolated with the NaN knots masked per query (__call__)
        self._local_nan = None
        if self.k == 1 and bool((some_nan & ~all_nan).any()):
            self._local_nan = nan
            self._nan_rows = all_nan
        if self.k == 3 and bool((some_nan & ~all_nan).any()):
            self._nan_rows = torch.ones_like(self._nan_rows)
        self._any_nan_row = bool(self._nan_rows.any())      # read back once, here: every call asks

    # With few splines (<= 64 columns) every evaluation goes point by point (cp_spline_points): no (queries x knots) operator to build on the
    # host for each new set of queries -- what a likelihood calling with its own redshifts pays at every step.  Many columns on shared queries
    # (batches of spectra) keep the operator form, unless the operator would be larger than this many queries / 4 M entries.
    _npoints_operator = 0
    _operator_chunk = 1 << 16      # queries per pass of the operator routes when the (queries x knots) operator would pass 64 M weights

    def _call_points(self, x, bounds_error, dx):
        """Few splines at very many points (``cp_spline_points``): the queries stay where they are (a device tensor is not read back), the
        splines are given by their values and knot derivatives (the nu = 1 operator at the knots, applied once)."""
        torch = dv.torch()
        like_torch = dv.is_torch(x)
        dtype = dv.float_dtype(x)
        xq = dv.to_device(x, self.device)
        shape = tuple(xq.shape) + self.shape
        xq = xq.reshape(-1)
        if bounds_error:
            lo, hi = torch.aminmax(xq)
            if bool((lo < self.xmin) | (hi > self.xmax)):
                raise ValueError('input outside of extrapolation range ({}, {})'.format(self.xmin, self.xmax))
        outside = None
        if self.interp_x == 'log':
            if not self.extrap:
                # The reference masks in x itself (jax.py:188: xmin <= x <= xmax, xmin = the first knot as handed in -- for the padded P(k) tables
                # 10**log10(extrap_kmin), which need not be extrap_kmin) before its spline refuses what lies outside log10 of the knots.  The kernel
                # compares logarithms; the device's log10 of a query AT an end knot may differ in the last bit from numpy's log10 of the knot: the
                # mask is taken from x, and the logarithm of what it lets through is kept inside the knots
                outside = ~((xq >= self.xmin) & (xq <= self.xmax))
                xq = torch.log10(xq).clamp(float(self._x[0]), float(self._x[-1]))
            else:
                xq = torch.log10(xq)
        slopes = self.__dict__.get('_knot_slopes', None)
        if slopes is None:
            if self._x.size <= 512:     # a small (knots x knots) operator, shared by every spline on these knots
                slope_op = _cached_operator(('i1s', self._x.tobytes(), self.device.index),
                                            lambda: LinearOperator.spline(self._x, self._x, bc='natural', nu=1, device=self.device))
                slopes = slope_op(self._rows).contiguous()
            else:                       # many knots: the tridiagonal system itself, once per interpolator (few columns: a host solve)
                slopes = dv.to_device(_natural_spline_slopes(self._x, dv.to_host(self._rows).T).T, self.device).contiguous()
            self._knot_slopes = slopes
            self._x_device = dv.to_device(self._x, self.device)
        out = torch.empty((self._rows.shape[0], xq.numel()), dtype=torch.float64, device=self.device)
        _lib.check(_lib.load().cp_spline_points(self._x_device.data_ptr(), self._rows.data_ptr(), slopes.data_ptr(), self._x.size, self._rows.shape[0],
                                                xq.data_ptr(), out.data_ptr(), xq.numel(), int(dx), int(self.extrap), self.device.index,
                                                dv.stream_of(self.device)))
        if self._any_nan_row:
            out = torch.where(self._nan_rows[:, None], torch.full_like(out, float('nan')), out)
        if outside is not None:
            out = torch.where(outside, torch.full_like(out, float('nan')), out)
        if self.interp_fun == 'log':
            out = 10**out
        out = out.T if out.shape[0] > 1 else out.reshape(-1, 1)
        return _finish(out, dtype, like_torch, shape)

    def __call__(self, x, bounds_error=False, dx=0):
        like_torch = dv.is_torch(x)
        dtype = dv.float_dtype(x)
        nq = x.numel() if like_torch else np.size(x)
        if self.k == 3 and self._rows.shape[0] <= 64 and (nq > self._npoints_operator or (nq > 1024 and nq * self._x.size > (1 << 22))):
            # few splines, and a (queries x knots) operator that would be large: evaluate point by point
            return self._call_points(x, bounds_error, dx)
        if nq > self._operator_chunk and nq * self._x.size > (1 << 26):
            # a catalogue through one of the operator routes (linear / quadratic interpolation, many columns): in pieces of queries
            xf = x.reshape(-1) if like_torch else np.asarray(x).ravel()
            pieces = [dv.to_device(self(xf[lo:lo + self._operator_chunk], bounds_error=bounds_error, dx=dx), self.device)
                      for lo in range(0, nq, self._operator_chunk)]
            return _finish(dv.torch().cat(pieces, dim=0), dtype, like_torch, tuple(x.shape if like_torch else np.shape(x)) + self.shape)
        xh = _host(x)
        shape = xh.shape + self.shape
        xh = xh.ravel()
        inside, = _mask_bounds([xh], [(self.xmin, self.xmax)], bounds_error=bounds_error)
        if xh.size == 0:
            return _finish(dv.torch().empty((0, self._rows.shape[0]), dtype=dv.torch().float64, device=self.device), dtype, like_torch, shape)
        with np.errstate(all='ignore'):
            xq = np.log10(xh) if self.interp_x == 'log' else xh
        if self.k in (1, 2):     # scipy interp1d(kind='linear' / 'quadratic') of the reference (jax.py:176-177)
            if dx:
                raise TypeError('derivatives are available for cubic interpolation only')
            build = _linear_interp_operator if self.k == 1 else _quadratic_interp_operator
            op = _cached_operator(('i1d-k%d' % self.k, self._x.tobytes(), xq.tobytes(), self.extrap, self.device.index),
                                  lambda: LinearOperator.dense(build(self._x, xq, self.extrap), device=self.device))
        else:
            op = _cached_operator(('i1d', self._x.tobytes(), xq.tobytes(), int(dx), self.extrap, self.device.index),
                                  lambda: LinearOperator.spline(self._x, xq, bc='natural', nu=dx, extrapolate=self.extrap, device=self.device))
        if self.k == 1 and self._local_nan is not None:
            # interp1d takes the interval with x[lo] < xq <= x[hi] (the first one for xq <= x[0]) and returns y[lo] + slope (xq - x[lo]): NaN when either end is
            torch = dv.torch()
            lo = np.clip(np.searchsorted(self._x, xq, side='left'), 1, self._x.size - 1) - 1
            tlo = dv.upload(lo, self.device)
            bad = self._local_nan[:, tlo] | self._local_nan[:, tlo + 1]
            out = torch.where(bad, torch.full((), float('nan'), dtype=torch.float64, device=self.device), op(torch.where(self._local_nan, torch.zeros((), dtype=torch.float64, device=self.device), self._rows)))
        else:
            out = op(self._rows)   # (ncol, nq); NaN outside [xmin, xmax] unless extrap
        if self._any_nan_row:
            out = dv.torch().where(self._nan_rows[:, None], dv.torch().full_like(out, float('nan')), out)
        if not self.extrap and not inside.all():      # the reference's mask in x itself (jax.py:188-192); the operator's own is in the knots' coordinates
            out = dv.torch().where(dv.upload(inside, self.device), out, dv.torch().full_like(out, float('nan')))
        if self.interp_fun == 'log':
            out = 10**out
        return _finish(out.T, dtype, like_torch, shape)


def _fitpack_nan_coefficients(nan, kx, ky):
    """Which B-spline coefficients FITPACK's ``regrid`` (``RectBivariateSpline(s=0)``, reference jax.py:241) leaves NaN for data holding NaN, as a
    boolean tensor like ``nan`` (..., nx, ny), by the order of its eliminations: along an axis of degree >= 2 the Givens rotations and the back
    substitution carry a NaN through the whole axis; along an axis of degree 1 the rotations skip the zero entries and touch nothing, but the back
    substitution ``c[i] = z[i] - 0 * c[i + 1]`` hands a NaN DOWN to every smaller index.  Both degrees >= 2: the whole surface."""
    torch = dv.torch()

    def along(mask, dim, degree):
        if degree == 1:      # suffix "any": index i is NaN if any index >= i is
            return mask.flip(dim).to(torch.int32).cumsum(dim).flip(dim) > 0
        return mask.any(dim=dim, keepdim=True).expand_as(mask)

    return along(along(nan, -2, kx), -1, ky)


def _fitpack_nan_queries(coef_nan, knots_x, kx, xq, knots_y, ky, yq, grid=True):
    """Which evaluations of ``bispev`` are NaN given NaN coefficients (:func:`_fitpack_nan_coefficients`): a query is a sum over the coefficients of
    its knot interval, zero weights included (0 * NaN) -- along an axis of degree 1 those of the two ends of the interval ``t[a] <= q < t[a + 1]`` (the
    last interval closed), along the others the mask is constant anyway.  ``xq``, ``yq``: host queries in the coordinates of the knots; returns a
    device tensor (..., nxq, nyq) (``grid``) or (..., nq)."""
    torch = dv.torch()
    device = coef_nan.device

    def ends(knots, q):
        a = np.clip(np.searchsorted(knots, np.clip(q, knots[0], knots[-1]), side='right') - 1, 0, knots.size - 2)
        return torch.as_tensor(a, dtype=torch.long, device=device)

    if kx == 1:
        a = ends(knots_x, xq)
        mx = coef_nan.index_select(-2, a) | coef_nan.index_select(-2, a + 1)      # (..., nxq, ny)
    else:
        mx = coef_nan[..., :1, :].expand(coef_nan.shape[:-2] + (np.size(xq), coef_nan.shape[-1]))
    if ky != 1:
        m = mx[..., :1]                                                            # constant along y
        return m.expand(mx.shape[:-1] + (np.size(yq),)) if grid else m[..., 0]
    b = ends(knots_y, yq)
    if grid:
        return mx.index_select(-1, b) | mx.index_select(-1, b + 1)                 # (..., nxq, nyq)
    index = b.expand(mx.shape[:-1]).unsqueeze(-1)
    return (mx.gather(-1, index) | mx.gather(-1, index + 1))[..., 0]


class Interpolator2D(dv.Copyable):

    """2D grid interpolation, == RectBivariateSpline(kx, ky, s=0) (reference jax.py:213-287): separable interpolating splines, banded
    not-a-knot cubic operators for degree 3 (the default), dense operators for the other degrees (1, 2, 4, 5)."""

    def __init__(self, x, y, fun, kx=3, ky=3, interp_x='lin', interp_fun='lin', extrap=False, assume_sorted=False, device=None):
        self.kx, self.ky = int(kx), int(ky)
        for k in (self.kx, self.ky):
            if not 1 <= k <= 5:
                raise ValueError('spline degrees 1 <= kx, ky <= 5 are supported')       # as RectBivariateSpline
        torch = dv.torch()
        self.device = dv.resolve_device(device, fun)
        self.interp_x, self.interp_fun, self.extrap = str(interp_x), str(interp_fun), bool(extrap)
        x, y = _host(x).ravel(), _host(y).ravel()
        fun = dv.to_device(fun, self.device)      # (nx, ny), or (batch..., nx, ny): one surface per batch entry on the shared (x, y) grid
        if not assume_sorted:
            ix, iy = np.argsort(x), np.argsort(y)
            x, y = x[ix], y[iy]
            fun = fun.index_select(-2, dv.upload(ix, self.device)).index_select(-1, dv.upload(iy, self.device))
        self.xmin, self.xmax, self.ymin, self.ymax = x[0], x[-1], y[0], y[-1]
        self._x = np.log10(x) if self.interp_x == 'log' else x
        self._y = y
        if self.interp_fun == 'log':
            fun = torch.log10(fun)
        self._fun = fun.contiguous()    # (batch..., nx, ny)
        self._lead = tuple(self._fun.shape[:-2])
        # FITPACK propagates any NaN datum (e.g. the log of a negative P) to the whole surface (reference tests/test_interpolator.py:328-337) -- with
        # cubic (any degree >= 2) splines along both axes.  Linear interpolation along an axis contains it: the coefficients FITPACK leaves NaN are
        # _fitpack_nan_coefficients, the evaluations that see one _fitpack_nan_queries; the others are the numbers FITPACK returns (its solves along
        # a linear axis do not mix rows: with the NaN data replaced by anything finite the clean evaluations come out as they do there)
        self._coef_nan = None
        if (self.kx == 1 or self.ky == 1) and (self._lead or bool(torch.isnan(self._fun).any())):
            nan = torch.isnan(self._fun)
            self._coef_nan = _fitpack_nan_coefficients(nan, self.kx, self.ky)
            self._fun = torch.where(nan, torch.zeros((), dtype=self._fun.dtype, device=self.device), self._fun)
        nan = torch.isnan(self._fun).flatten(-2).any(dim=-1)           # per surface
        if self._lead:
            # a batch of surfaces: the ones holding a NaN are made NaN throughout, on the device -- every interpolated value of theirs is then NaN
            # by arithmetic (the operators are linear), with no flag to read back and no mask to apply to the results
            self._fun = torch.where(nan[..., None, None], torch.full_like(self._fun, float('nan')), self._fun)
            self._has_nan, self._nan_surfaces = False, None
        else:
            self._has_nan, self._nan_surfaces = bool(nan), None

    def _operator(self, axis, q, dense=False):
        """Operator from the knots of ``axis`` ('x': transformed coordinates, 'y') to the queries ``q`` (host, flat); queries outside the
        knots are clamped to the end points (FITPACK's ``bispev``), which only shows with ``extrap=True``: the mask turns them NaN otherwise.
        dense : return the (nq, n) weights on the host instead."""
        knots, k = (self._x, self.kx) if axis == 'x' else (self._y, self.ky)
        q = np.clip(q, knots[0], knots[-1])
        if dense:
            return dense_operator(knots, q, bc='not-a-knot', extrapolate=True) if k == 3 else _fitpack_interp_operator(knots, q, k)
        if k == 3:
            build = lambda: LinearOperator.spline(knots, q, bc='not-a-knot', extrapolate=True, device=self.device)
        else:
            build = lambda: LinearOperator.dense(_fitpack_interp_operator(knots, q, k), device=self.device)
        return _cached_operator(('i2' + axis, k, knots.tobytes(), q.tobytes(), self.device.index), build)

    def rows_y_major(self, xh, yh, exp10=False):
        """The surfaces on the grid of flat host coordinates (xh, yh) as (batch..., ny, nx), x fastest -- the layout of rows of P(k) at every z --
        in two passes: the x operator (banded spline kernel) on the table kept y-major, then the y operator along the middle axis as a GEMM on the
        matrix cores whose epilogue applies 10^x when asked (tables splined in log10 P) and writes the result once.  No mask."""
        with np.errstate(all='ignore'):
            xq = np.log10(xh) if self.interp_x == 'log' else xh
        fun_t = self.__dict__.get('_fun_y_major', None)
        if fun_t is None:
            fun_t = self._fun_y_major = self._fun.transpose(-1, -2).contiguous()       # (batch..., ny, nx), once per object
        knots, k = self._y, self.ky
        yq = np.clip(yh, knots[0], knots[-1])
        opy = _cached_operator(('i2y-dense', k, knots.tobytes(), yq.tobytes(), self.device.index),
                               lambda: LinearOperator.dense(self._operator('y', yq, dense=True), device=self.device))
        lib = _lib.load()
        out = self._rows_y_major_direct(fun_t, xq, opy, exp10) if fun_t.ndim == 3 and self.kx == 3 and _DIRECT_K_SPLINE else None
        if out is not None:
            pass
        elif fun_t.ndim == 3 and lib.cp_tables_rows_available(self._operator('x', xq)._handle, opy._handle):
            opx = self._operator('x', xq)
            # both operators in one kernel (cp_tables_rows): the x-splined tables are never written
            torch = dv.torch()
            out = torch.empty((fun_t.shape[0], opy.nq, opx.nq), dtype=torch.float64, device=self.device)
            if fun_t.shape[0]:
                _lib.check(lib.cp_tables_rows(opx._handle, opy._handle, fun_t.data_ptr(), out.data_ptr(), fun_t.shape[0], 2 if exp10 else 0, 1.,
                                              dv.stream_of(self.device)))
        else:
            out = opy.mid(self._operator('x', xq)(fun_t), post='exp10' if exp10 else None)     # (batch..., ny, nxq) -> (batch..., nyq, nxq)
        if self._nan_surfaces is not None:
            out = dv.torch().where(self._nan_surfaces[..., None, None], dv.torch().full_like(out, float('nan')), out)
        if self._coef_nan is not None:
            bad = _fitpack_nan_queries(self._coef_nan, self._x, self.kx, xq, self._y, self.ky, yq).transpose(-1, -2)
            out = dv.torch().where(bad, dv.torch().full_like(out, float('nan')), out)
        return out

    def _rows_y_major_direct(self, fun_t, xq, opy, exp10):
        """Cubic splines along x for a batch of tables: the second derivatives of every table row at the knots are the tables' own (one
        tridiagonal solve per row, ``cp_spline_rows_second_derivatives``, kept with the object like the coefficients the reference's
        RectBivariateSpline computes when it is built), and a query is four multiply-adds on them -- evaluated by the kernel that contracts the
        y direction on the matrix cores (``cp_tables_rows_direct``).  None where that kernel does not apply (the caller multiplies by operators)."""
        from .spline import SplineRows
        torch = dv.torch()
        lib = _lib.load()
        try:
            kplan = _cached_operator(('i2x-rows', self._x.tobytes(), xq.tobytes(), self.device.index),
                                     lambda: SplineRows(self._x, xq, bc='not-a-knot', extrapolate=True, device=self.device))
            m = self.__dict__.get('_fun_y_major_m', None)
            if m is None:
                ends = _cached_operator(('i2x-rows-m', self._x.tobytes(), self.device.index),
                                        lambda: SplineRows(self._x, self._x[[0, -1]], bc='not-a-knot', device=self.device))
                # (y_j, M_j) pairs: the four numbers of a query are 32 contiguous bytes (kept besides the tables: twice their size)
                m = self._fun_y_major_m = ends.second_derivatives(fun_t, pairs=_PAIRED_TABLES)
            out = torch.empty((fun_t.shape[0], opy.nq, xq.size), dtype=torch.float64, device=self.device)
            if fun_t.shape[0]:
                paired = m.dim() == fun_t.dim() + 1
                _lib.check(lib.cp_tables_rows_direct(kplan._handle, opy._handle, m.data_ptr() if paired else fun_t.data_ptr(), None if paired else m.data_ptr(), out.data_ptr(),
                                                     fun_t.shape[0], 2 if exp10 else 0, 1., dv.stream_of(self.device)))
            return out
        except NotImplementedError:
            return None

    _pairs_chunk = 1 << 16      # pairs of (x, y) evaluated per pass of the pair route

    def _call_many_x(self, x, y, bounds_error):
        """Grid evaluation at very many x (a mesh of wavenumbers) and a few y: the y direction first (operator, few queries), then one spline
        along x per requested y evaluated point by point (``cp_spline_points``); x may live on the device and is not read back."""
        torch = dv.torch()
        like_torch = dv.is_torch(x) or dv.is_torch(y)
        dtype = dv.float_dtype(x, y)
        yh = _host(y)
        xq = dv.to_device(x, self.device)
        shape = tuple(xq.shape) + yh.shape
        xq, yh = xq.reshape(-1), yh.ravel()
        mask_y, = _mask_bounds([yh], [(self.ymin, self.ymax)], bounds_error=bounds_error)
        mask_x = (xq >= self.xmin) & (xq <= self.xmax)
        if bounds_error and not bool(mask_x.all()):
            raise ValueError('input outside of extrapolation range ({}, {})'.format(self.xmin, self.xmax))
        if self._lead:
            raise NotImplementedError('a mesh of x on a batch of surfaces: evaluate on a grid of at most 16 384 x')
        if self._has_nan:
            return _finish(torch.full((xq.numel(), yh.size), float('nan'), dtype=torch.float64, device=self.device), dtype, like_torch, shape)
        if self.interp_x == 'log':
            xq = torch.log10(xq)
        if self.extrap:      # FITPACK's bispev evaluates queries outside the knots at the end knots (as _operator does for the grid route)
            xq = xq.clamp(float(self._x[0]), float(self._x[-1]))
        opy = self._operator('y', yh)
        rows = opy(self._fun).T.contiguous()                      # (nyq, nx): the surface along x at every requested y
        slope_op = _cached_operator(('i2s', self._x.tobytes(), self.device.index),
                                    lambda: LinearOperator.spline(self._x, self._x, bc='not-a-knot', nu=1, extrapolate=True, device=self.device))
        slopes = slope_op(rows).contiguous()
        xk = self.__dict__.get('_x_device', None)
        if xk is None:
            xk = self._x_device = dv.to_device(self._x, self.device)
        out = torch.empty((rows.shape[0], xq.numel()), dtype=torch.float64, device=self.device)
        _lib.check(_lib.load().cp_spline_points(xk.data_ptr(), rows.data_ptr(), slopes.data_ptr(), self._x.size, rows.shape[0], xq.data_ptr(), out.data_ptr(),
                                                xq.numel(), 0, 1, self.device.index, dv.stream_of(self.device)))
        out = out.T
        if self._coef_nan is not None:      # (kx = 3 on this route: the mask does not depend on x)
            bad = _fitpack_nan_queries(self._coef_nan, self._x, self.kx, np.zeros(1), self._y, self.ky, yh)      # (1, nyq)
            out = torch.where(bad, torch.full_like(out, float('nan')), out)
        if self.interp_fun == 'log':
            out = 10**out
        if not self.extrap:
            mask = mask_x[:, None] & dv.upload(mask_y, self.device)[None, :]
            out = torch.where(mask, out, torch.full_like(out, float('nan')))
        return _finish(out, dtype, like_torch, shape)

    def __call__(self, x, y, grid=True, bounds_error=False):
        torch = dv.torch()
        like_torch = dv.is_torch(x) or dv.is_torch(y)
        dtype = dv.float_dtype(x, y)
        nxq, nyq = (v.numel() if dv.is_torch(v) else np.size(v) for v in (x, y))
        if grid and self.kx == 3 and 0 < nyq <= 64 and (nxq > 16384 or (nxq > 1024 and nxq * self._x.size > (1 << 22))):
            return self._call_many_x(x, y, bounds_error)
        if not grid and nxq > self._pairs_chunk and nxq == nyq:
            # very many pairs: in pieces (the pair route multiplies two dense (pairs x knots) operators built on the host)
            xf, yf = (v.reshape(-1) if dv.is_torch(v) else np.asarray(v).ravel() for v in (x, y))
            pieces = [dv.to_device(self(xf[lo:lo + self._pairs_chunk], yf[lo:lo + self._pairs_chunk], grid=False, bounds_error=bounds_error), self.device)
                      for lo in range(0, nxq, self._pairs_chunk)]
            return _finish(torch.cat(pieces, dim=-1), dtype, like_torch, self._lead + tuple(x.shape if dv.is_torch(x) else np.shape(x)))
        xh, yh = _host(x), _host(y)
        shape = self._lead + (xh.shape + yh.shape if grid else xh.shape)
        xh, yh = xh.ravel(), yh.ravel()
        mask_x, mask_y = _mask_bounds([xh, yh], [(self.xmin, self.xmax), (self.ymin, self.ymax)], bounds_error=bounds_error)
        if xh.size == 0 or yh.size == 0 or self._has_nan:
            n = self._lead + ((xh.size, yh.size) if grid else (xh.size,))
            return _finish(torch.full(n, float('nan'), dtype=torch.float64, device=self.device), dtype, like_torch, shape)
        with np.errstate(all='ignore'):
            xq = np.log10(xh) if self.interp_x == 'log' else xh
        if grid:
            opx, opy = self._operator('x', xq), self._operator('y', yh)
            tmp = opx(self._fun.transpose(-1, -2).contiguous())     # rows = y knots: (batch..., ny, nxq)
            out = opy(tmp.transpose(-1, -2).contiguous())           # rows = x queries: (batch..., nxq, nyq)
            mask = mask_x[:, None] & mask_y
        else:
            wx = dv.upload(self._operator('x', xq, dense=True), self.device)   # (nq, nx)
            wy = dv.upload(self._operator('y', yh, dense=True), self.device)   # (nq, ny)
            # sum_i sum_j wx[q, i] f[..., i, j] wy[q, j], one wave per (table, pair) (cp_bilinear_pairs)
            fun = self._fun.contiguous()
            nb = int(np.prod(self._lead, dtype=np.int64)) if self._lead else 1
            out = torch.empty(self._lead + (xq.size,), dtype=torch.float64, device=self.device)
            _lib.check(_lib.load().cp_bilinear_pairs(wx.data_ptr(), wy.data_ptr(), fun.data_ptr(), out.data_ptr(), nb, xq.size, fun.shape[-2], fun.shape[-1],
                                                     self.device.index, dv.stream_of(self.device)))
            mask = mask_x & mask_y
        if self._coef_nan is not None:
            bad = _fitpack_nan_queries(self._coef_nan, self._x, self.kx, xq, self._y, self.ky, yh, grid=grid)
            out = torch.where(bad, torch.full_like(out, float('nan')), out)
        if self.interp_fun == 'log':
            out = 10**out
        if not self.extrap and not mask.all():
            out = torch.where(dv.upload(mask, self.device), out, torch.full_like(out, float('nan')))
        if self._nan_surfaces is not None:
            out = torch.where(self._nan_surfaces.reshape(self._lead + (1,) * (out.ndim - len(self._lead))), torch.full_like(out, float('nan')), out)
        return _finish(out, dtype, like_torch, shape)


def _get_default_kwargs(func, start=0, remove=()):
    """``{argument: default}`` of ``func`` from its ``start``-th argument on, without the names in ``remove`` (what the reference keeps as
    ``default_params`` of its interpolator classes, interpolator.py:296-325)."""
    arguments = list(inspect.signature(func).parameters.values())[start:]
    return {arg.name: arg.default for arg in arguments if arg.name not in remove}


def _sorted_axis(values):
    """A coordinate axis as a private ascending float64 vector (it becomes a public attribute of the interpolator: never the cached, shared host
    copy of a device tensor) and the permutation that sorts it."""
    axis = np.array(_host(values), dtype='f8').ravel()
    order = np.argsort(axis)
    return axis[order], order


class _BasePowerSpectrumInterpolator(dv.Copyable):

    """Base class for power spectrum interpolators (reference interpolator.py:327-407)."""

    def _prepare(self, k, pk, z=None, interp_k='log', extrap_pk='log', extrap_kmin=_default_extrap_kmin, extrap_kmax=_default_extrap_kmax):
        """Sorted axes and table as attributes (``k``, ``z``, ``_pk`` of shape (k,) or (k, z)), the extrapolation range, and the (k, P) the splines
        are built on: with log-log extrapolation the table continued as power laws down to ``extrap_kmin`` and up to ``extrap_kmax``
        (reference interpolator.py:331-351)."""
        self.interp_k, self.extrap_pk = str(interp_k), str(extrap_pk)
        self.k, order_k = _sorted_axis(k)
        table = np.asarray(_host(pk), dtype='f8')
        if z is not None or table.ndim > 1:
            table = table.reshape(self.k.size, -1)
        table = table[order_k]
        if z is not None:
            self.z, order_z = _sorted_axis(z)
            table = table[:, order_z]
        self._pk = table
        if self.extrap_pk != 'log':
            self.extrap_kmin, self.extrap_kmax = self.k[0], self.k[-1]
            return self.k, table
        if self.interp_k != 'log':
            self.extrap_kmin, self.extrap_kmax = self.k[0], self.k[-1]
            raise ValueError('log-log extrapolation requires log-x interpolation')
        self.extrap_kmin, self.extrap_kmax = extrap_kmin, extrap_kmax
        with np.errstate(all='ignore'):   # negative P -> NaN everywhere, without raising (reference tests/test_interpolator.py:328-337)
            logk, logpk = _pad_log(self.k, table, extrap_kmin=extrap_kmin, extrap_kmax=extrap_kmax)
            return 10**logk, 10**logpk

    def params(self):
        """Return interpolator parameter dictionary."""
        return {name: getattr(self, name) for name in self.default_params}

    def as_dict(self):
        """Return interpolator as a dictionary."""
        state = self.params()
        for name in ['k', 'pk']:
            state[name] = getattr(self, name)
        if hasattr(self, 'z'):
            state['z'] = self.z
        return state

    def clone(self, **kwargs):
        """Clone interpolator, i.e. return a deepcopy with (possibly) other attributes in ``kwargs``."""
        return self.__class__(**{**self.as_dict(), **kwargs})

    def deepcopy(self):
        """Deep copy interpolator (interpolators built from callables are re-tabulated at ``k``, as in the reference)."""
        return self.__class__(**self.as_dict())

    @property
    def kmin(self):
        """Minimum (interpolated) ``k`` value."""
        return self.k[0]

    @property
    def kmax(self):
        """Maximum (interpolated) ``k`` value."""
        return self.k[-1]


class PowerSpectrumInterpolator1D(_BasePowerSpectrumInterpolator):

    """1D power spectrum interpolator with ``sigma_r``, ``sigma_d``, ``to_xi`` (reference interpolator.py:412-605)."""

    def __init__(self, k, pk, interp_k='log', extrap_pk='log', extrap_kmin=_default_extrap_kmin, extrap_kmax=_default_extrap_kmax, interp_order_k=3,
                 device=None):
        self._rsigma8sq = 1.
        self.device = dv.resolve_device(device, pk)
        k, pk = self._prepare(k, pk, interp_k=interp_k, extrap_pk=extrap_pk, extrap_kmin=extrap_kmin, extrap_kmax=extrap_kmax)
        self.interp_order_k = int(interp_order_k)
        self._interp = Interpolator1D(k, pk, k=self.interp_order_k, interp_x=self.interp_k, interp_fun=self.extrap_pk, assume_sorted=True, device=self.device)
        self.is_from_callable = False

    default_params = _get_default_kwargs(__init__, start=3, remove=('device',))

    @property
    def pk(self):
        """Power spectrum array (evaluated at ``k`` if built from a callable), with normalisation."""
        if self.is_from_callable:
            return self(self.k)
        return self._pk * self._rsigma8sq

    @classmethod
    def from_callable(cls, k=None, pk_callable=None, extrap_kmin=_default_extrap_kmin, extrap_kmax=_default_extrap_kmax, device=None):
        """Build from ``pk_callable(k)`` -> array / device tensor of shape (nk,) or (nk, ncol) (reference interpolator.py:462-493)."""
        if k is None:
            k = get_default_k_callable()
        self = cls.__new__(cls)
        self.__dict__.update(self.default_params)
        self._rsigma8sq = 1.
        self.device = dv.resolve_device(device)
        self.k = np.sort(_host(k).ravel())
        self.extrap_kmin, self.extrap_kmax = extrap_kmin, extrap_kmax
        self.is_from_callable = True
        self._interp = pk_callable
        return self

    def _eval_device(self, kh, bounds_error=False, extra=None):
        """P(k) at host wavenumbers ``kh`` (flat) as a device tensor (nk,) + trailing column shape."""
        torch = dv.torch()
        if self.is_from_callable:
            mask_k, = _mask_bounds([kh], [(self.extrap_kmin, self.extrap_kmax)], bounds_error=bounds_error)
            out = dv.to_device(self._interp(kh, **(extra or {})), self.device)
            mask = dv.upload(mask_k, self.device).reshape((-1,) + (1,) * (out.ndim - 1))
            out = torch.where(mask, out, torch.full_like(out, float('nan')))
        else:
            out = self._interp(dv.upload(kh, self.device), bounds_error=bounds_error)
        return out * self._rsigma8sq

    def __call__(self, k, **kwargs):
        """Evaluate the power spectrum at wavenumbers ``k``; NaN outside [extrap_kmin, extrap_kmax], ``bounds_error=True`` raises instead
        (reference interpolator.py:495-520; other ``kwargs`` reach the callable of an interpolator built by :meth:`from_callable`)."""
        bounds_error, extra = _call_options(kwargs, self.is_from_callable)
        like_torch = dv.is_torch(k)
        dtype = dv.float_dtype(k)
        kh = _host(k)
        out = self._eval_device(kh.ravel(), bounds_error=bounds_error, extra=extra)
        return _finish(out, dtype, like_torch, kh.shape + tuple(out.shape[1:]))

    def _rows(self, kh):
        """P(k) as rows (ncol..., nk) for the sigma integrals."""
        out = self._eval_device(kh)
        return out.reshape(kh.size, -1).T.contiguous() if out.ndim > 1 else out[None, :]

    def _colshape(self):
        return tuple(self._eval_device(np.array([self.k[0]])).shape[1:])

    def sigma_d(self, **kwargs):
        r"""R.m.s. of the displacement field :math:`\sqrt{\frac{1}{6\pi^2}\int dk P(k)}` (reference interpolator.py:523-545)."""
        cs = self._colshape()
        out = sigma_d2_of_rows(self._rows, kmin=self.extrap_kmin, kmax=self.extrap_kmax, device=self.device, **kwargs)**0.5
        return _finish(out, np.dtype('f8'), False, cs)

    def sigma_r(self, r, **kwargs):
        r"""R.m.s. of perturbations in a sphere of radius :math:`r` (reference interpolator.py:547-573); shape r.shape (+ columns)."""
        like_torch = dv.is_torch(r)
        dtype = dv.float_dtype(r)
        rh = _host(r)
        cs = self._colshape()
        out = sigma_r2_of_rows(rh, self._rows, kmin=self.extrap_kmin, kmax=self.extrap_kmax, device=self.device, **kwargs)**0.5   # (ncol, nr)
        return _finish(out.T, dtype, like_torch, rh.shape + cs)

    def sigma8(self, **kwargs):
        """R.m.s. of perturbations in a sphere of 8."""
        return self.sigma_r(8., **kwargs)

    def rescale_sigma8(self, sigma8=1.):
        """Rescale power spectrum to the provided ``sigma8`` normalisation."""
        self._rsigma8sq = 1.
        self._rsigma8sq = sigma8**2 / self.sigma8()**2

    def to_xi_arrays(self, nk=1024, fftlog_kwargs=None):
        """FFTLog transform to the correlation function on the FFTLog grid: ``(s, xi)`` arrays, xi of shape (nk,) + columns."""
        k = np.geomspace(self.extrap_kmin, self.extrap_kmax, nk)
        s, xi = _cached_fftlog(PowerToCorrelation, k, self.device, fftlog_kwargs)(self._rows(k))
        cs = self._colshape()
        return s.cpu().numpy(), xi.T.reshape((nk,) + cs).cpu().numpy()

    def to_xi(self, nk=1024, fftlog_kwargs=None, **kwargs):
        """Transform into a :class:`CorrelationFunctionInterpolator1D` with FFTLog (reference interpolator.py:584-605)."""
        s, xi = self.to_xi_arrays(nk=nk, fftlog_kwargs=fftlog_kwargs)
        default_params = dict(interp_s='log', interp_order_s=self.interp_order_k)
        default_params.update(kwargs)
        return CorrelationFunctionInterpolator1D(s, xi=xi, device=self.device, **default_params)


def sigma_rz_analytic(engine, bg, pk, r, growth_sq, device, kmin=1e-7, kmax=1e2, blocks=0, keep_spectra=False):
    """``cp_sigma_rz_analytic``: sqrt(sigma^2(r) growth_sq) of a batch of cosmologies of an analytic engine, (batch, nr, nz), with P(k) evaluated
    inside the call on the 1024 wavenumbers geomspace(kmin, kmax) of the default transform.  ``keep_spectra``: also return those spectra,
    (batch, 1024) (the sigma8 normalisation keeps them: the filters ask for P on the same wavenumbers next).  None when the parameter arrays do
    not match the batch of ``growth_sq`` (batch, nz)."""
    torch = dv.torch()
    from .background import DEFAULTS as bg_defaults
    from .power import PK_DEFAULTS
    nb, nz = growth_sq.shape
    cbg, n1, keep1 = dv.pack_params(_lib.BG_PARAMS, bg, bg_defaults, device)
    cpk, n2, keep2 = dv.pack_params(_lib.PK_PARAMS, pk, PK_DEFAULTS, device)
    if {n for n in (n1, n2) if n is not None} - {nb}:
        return None
    nk = 1024
    k = np.geomspace(kmin, kmax, nk)
    rr = np.asarray(r, dtype='f8').ravel()
    key = (float(kmin), float(kmax), nk, device.index)
    fft = _tophat_plan(key, k, device)
    s = fft.y[0]
    op = _cached_operator(('nat', s.tobytes(), rr.tobytes(), device.index), lambda: LinearOperator.spline(s, rr, bc='natural', device=device))
    lib = _lib.load()
    out = torch.empty((nb, rr.size, nz), dtype=torch.float64, device=device)
    spectra = torch.empty((nb, nk), dtype=torch.float64, device=device) if keep_spectra else None
    work = torch.empty(int(lib.cp_sigma_rz_workspace_bytes(nb, nk)), dtype=torch.uint8, device=device)
    growth_sq = growth_sq.contiguous()
    nu, keep_nu = dv.ncdm_arg(bg, nb)
    if 1 <= rr.size <= _FUNCTIONAL_RADII and blocks == 0:
        # few radii (sigma8: one): transform and spline are linear in P(k) -- sigma^2(r_q) = sum_j F[q, j] P(k_j), F = what the two return for unit
        # spectra, computed once per (grid, radii) -- and the kernel is the evaluation of P(k) with a dot product behind it
        functional = _cached_operator(('sigma_functional', key, rr.tobytes()),
                                      lambda: op(fft(torch.eye(nk, dtype=torch.float64, device=device))[1]).transpose(0, 1).contiguous())
        _lib.check(lib.cp_sigma_rz_functional(_lib.ENGINES[engine], nb, dv.as_void_p(cbg), 0, nu, dv.as_void_p(cpk), nk, dv.upload(k, device).data_ptr(),
                                              functional.data_ptr(), rr.size, growth_sq.data_ptr(), nz, out.data_ptr(),
                                              spectra.data_ptr() if keep_spectra else None, work.data_ptr(), device.index, dv.stream_of(device)))
        return out, spectra, k
    if blocks == 0 and _GEOSPLINE_PREFILTERED and _SIGMA_RZ_PREFILTERED:
        # the spline's solve folded into the transform (radii well inside its output grid): the kernel's tail is four coefficients per radius
        native = fft._get_plan(device)
        geo = _cached_operator(('geospline', id(native), True, s.tobytes(), rr.tobytes(), device.index), lambda: _GeoSpline(s, rr, device, fft=fft, keep=native))
        if geo.prefiltered:
            _lib.check(lib.cp_sigma_rz_analytic_prefiltered(_lib.ENGINES[engine], nb, dv.as_void_p(cbg), 0, nu, dv.as_void_p(cpk), nk,
                                                            dv.upload(k, device).data_ptr(), geo.handle, growth_sq.data_ptr(), nz, out.data_ptr(),
                                                            spectra.data_ptr() if keep_spectra else None, work.data_ptr(), device.index, dv.stream_of(device)))
            return out, spectra, k
    _lib.check(lib.cp_sigma_rz_analytic(_lib.ENGINES[engine], nb, dv.as_void_p(cbg), 0, nu, dv.as_void_p(cpk), nk, dv.upload(k, device).data_ptr(),
                                        fft._get_plan(device).handle, op._handle, growth_sq.data_ptr(), nz, out.data_ptr(),
                                        spectra.data_ptr() if keep_spectra else None, work.data_ptr(), blocks, device.index, dv.stream_of(device)))
    return out, spectra, k


def sigma8_normalise(engine, bg, pk, sigma8, device, kmin=1e-7, kmax=1e2):
    """``cp_sigma8_normalise``: the sigma8 normalisation of a batch of cosmologies of an analytic engine in one kernel (reference
    eisenstein_hu.py:94-103, 331-342).  ``pk`` carries the first-guess amplitudes; ``sigma8`` a float or a (batch,) device tensor.  Returns
    (rsigma8 (batch,), A_s rsigma8^2 (batch,), spectra without growth at the normalised amplitude (batch, 1024), their wavenumbers), or None when
    the parameters are not a batch."""
    torch = dv.torch()
    from .background import DEFAULTS as bg_defaults
    from .power import PK_DEFAULTS
    cbg, n1, keep1 = dv.pack_params(_lib.BG_PARAMS, bg, bg_defaults, device)
    cpk, n2, keep2 = dv.pack_params(_lib.PK_PARAMS, pk, PK_DEFAULTS, device)
    sizes = {n for n in (n1, n2) if n is not None}
    if len(sizes) != 1:
        return None
    nb, nk = sizes.pop(), 1024
    k = np.geomspace(kmin, kmax, nk)
    key = (float(kmin), float(kmax), nk, device.index)
    fft = _tophat_plan(key, k, device)
    s, rr = fft.y[0], np.array([8.])
    op = _cached_operator(('nat', s.tobytes(), rr.tobytes(), device.index), lambda: LinearOperator.spline(s, rr, bc='natural', device=device))
    functional = _cached_operator(('sigma_functional', key, rr.tobytes()),
                                  lambda: op(fft(torch.eye(nk, dtype=torch.float64, device=device))[1]).transpose(0, 1).contiguous())
    target = _lib.cp_param()
    if (dv.is_torch(sigma8) and sigma8.ndim) or (not dv.is_torch(sigma8) and np.ndim(sigma8)):
        st = dv.to_device(sigma8, device, cache=False).reshape(-1)
        if st.numel() != nb:
            return None
        target.ptr, target.value = st.data_ptr(), 0.
    else:
        target.ptr, target.value = None, float(sigma8)
    lib = _lib.load()
    rsigma8 = torch.empty(nb, dtype=torch.float64, device=device)
    amplitude = torch.empty(nb, dtype=torch.float64, device=device)
    spectra = torch.empty((nb, nk), dtype=torch.float64, device=device)
    work = torch.empty(int(lib.cp_sigma_rz_workspace_bytes(nb, nk)), dtype=torch.uint8, device=device)
    nu, keep_nu = dv.ncdm_arg(bg, nb)
    _lib.check(lib.cp_sigma8_normalise(_lib.ENGINES[engine], nb, dv.as_void_p(cbg), 0, nu, dv.as_void_p(cpk), nk, dv.upload(k, device).data_ptr(),
                                       functional.data_ptr(), target, rsigma8.data_ptr(), amplitude.data_ptr(), spectra.data_ptr(), work.data_ptr(),
                                       device.index, dv.stream_of(device)))
    return rsigma8, amplitude, spectra, k


class PowerSpectrumInterpolator2D(_BasePowerSpectrumInterpolator):

    """2D power spectrum interpolator with ``sigma_rz``, ``sigma_dz``, ``to_1d``, ``to_xi`` (reference interpolator.py:609-987)."""

    def __init__(self, k, z, pk, interp_k='log', extrap_pk='log', extrap_kmin=_default_extrap_kmin, extrap_kmax=_default_extrap_kmax,
                 interp_order_k=3, interp_order_z=3, growth_factor_sq=None, device=None):
        self._rsigma8sq = 1.
        self.growth_factor_sq = growth_factor_sq
        self.device = dv.resolve_device(device, pk)
        self._tables_batched = len(pk.shape if hasattr(pk, 'shape') else np.shape(pk)) == 3
        if self._tables_batched:
            # (batch, nk, nz): one (k, z) table per cosmology on shared grids, kept on the device (extension of the reference's (nk, nz) table);
            # growth_factor_sq(z) -> (nz,) or (batch, nz) multiplies the interpolated tables as it does the reference's single table
            k, pk = self._prepare_tables(k, z, pk, interp_k=interp_k, extrap_pk=extrap_pk, extrap_kmin=extrap_kmin, extrap_kmax=extrap_kmax)
        else:
            k, pk = self._prepare(k, pk, z=z, interp_k=interp_k, extrap_pk=extrap_pk, extrap_kmin=extrap_kmin, extrap_kmax=extrap_kmax)
        self.interp_order_k, self.interp_order_z = int(interp_order_k), int(interp_order_z)
        is2d = self._is2d()
        if is2d:
            self._interp = Interpolator2D(k, self.z, pk, kx=self.interp_order_k, ky=self.interp_order_z, interp_x=self.interp_k, interp_fun=self.extrap_pk,
                                          assume_sorted=True, device=self.device)
        else:
            if self.growth_factor_sq is None:
                raise ValueError('provide either 2D pk array or growth_factor_sq')
            # one column of P(k) (a batch: one column per cosmology, (batch, nk, 1) -> the columns (nk, batch) of one 1D interpolator) x growth factor
            self._interp = Interpolator1D(k, pk[:, :, 0].T if self._tables_batched else pk[:, 0], k=self.interp_order_k, interp_x=self.interp_k,
                                          interp_fun=self.extrap_pk, assume_sorted=True, device=self.device)
        self.is_from_callable = False

    default_params = _get_default_kwargs(__init__, start=4, remove=('device',))

    def _is2d(self):
        """A (k, z) table (or a batch of them), as opposed to one column of P(k) (per cosmology) with a growth factor."""
        return self._pk.shape[-1] > 1

    def _table_k_limits(self):
        """Range of wavenumbers a tabulated interpolator returns numbers on: [extrap_kmin, extrap_kmax] (reference interpolator.py:802) AND the range of
        the spline's own knots (jax.py:250: the first and last knot as handed in -- with log-log extrapolation ``10**log10(extrap_kmin)``, which is not
        always ``extrap_kmin``: a query at such an end is NaN in the reference, and here)."""
        return max(self.extrap_kmin, self._interp.xmin), min(self.extrap_kmax, self._interp.xmax)

    def _prepare_tables(self, k, z, pk, interp_k='log', extrap_pk='log', extrap_kmin=_default_extrap_kmin, extrap_kmax=_default_extrap_kmax):
        """``_prepare`` + ``_pad_log`` (reference interpolator.py:329-351, 42-87) for a batch of tables (batch, nk, nz), on the device: sorted grids,
        and with log-log extrapolation two linearly extrapolated points of log10 P against log10 k on either side."""
        torch = dv.torch()
        self.k, self.z = np.array(_host(k), dtype='f8').ravel(), np.array(_host(z), dtype='f8').ravel()      # private copies: public attributes
        pk = dv.to_device(pk, self.device)
        if tuple(pk.shape[1:]) != (self.k.size, self.z.size):
            raise ValueError('pk must be (batch, {:d}, {:d}), got {}'.format(self.k.size, self.z.size, tuple(pk.shape)))
        ik, iz = np.argsort(self.k), np.argsort(self.z)
        if np.any(ik[1:] < ik[:-1]) or np.any(iz[1:] < iz[:-1]):
            pk = pk.index_select(1, dv.upload(ik, self.device)).index_select(2, dv.upload(iz, self.device))
        self.k, self.z, self._pk = self.k[ik], self.z[iz], pk.contiguous()
        self.interp_k, self.extrap_pk = str(interp_k), str(extrap_pk)
        self.extrap_kmin, self.extrap_kmax = self.k[0], self.k[-1]
        if self.extrap_pk != 'log':
            return self.k, self._pk
        if self.interp_k != 'log':
            raise ValueError('log-log extrapolation requires log-x interpolation')
        self.extrap_kmin, self.extrap_kmax = extrap_kmin, extrap_kmax
        logk, logpk = np.log10(self.k), torch.log10(self._pk)            # negative P -> NaN for that table, without raising
        lo, hi = np.log10(np.minimum(extrap_kmin, self.k[0] * (1 - 1e-9))), np.log10(np.maximum(extrap_kmax, self.k[-1] * (1 + 1e-9)))
        knots_lo, knots_hi = np.array([lo, logk[0] * 0.1 + lo * 0.9]), np.array([logk[-1] * 0.1 + hi * 0.9, hi])
        slope_lo = (logpk[:, 1] - logpk[:, 0]) / (logk[1] - logk[0])          # (batch, nz)
        slope_hi = (logpk[:, -1] - logpk[:, -2]) / (logk[-1] - logk[-2])
        pad_lo = torch.stack([logpk[:, 0] + slope_lo * float(kk - logk[0]) for kk in knots_lo], dim=1)
        pad_hi = torch.stack([logpk[:, -1] + slope_hi * float(kk - logk[-1]) for kk in knots_hi], dim=1)
        return 10**np.concatenate([knots_lo, logk, knots_hi]), 10**torch.cat([pad_lo, logpk, pad_hi], dim=1)

    def _rescaled(self, out, nlead):
        """``out`` (batch..., ...) times the sigma8 rescaling factor (a float, or one value per batch entry)."""
        rs = self._rsigma8sq
        if isinstance(rs, float) and rs == 1.:     # nothing to rescale: no pass over the result
            return out
        if np.ndim(rs) == 0 and not dv.is_torch(rs):
            return out * float(rs)
        rs = dv.to_device(rs, self.device)
        return out * rs.reshape(tuple(rs.shape) + (1,) * (out.ndim - rs.ndim))

    @property
    def pk(self):
        """Power spectrum array (evaluated on (k, z) if built from a callable), without growth factor, with normalisation."""
        if self.is_from_callable:
            kwargs = {'ignore_growth': True} if self.growth_factor_sq is not None else {}
            return self(self.k, self.z, **kwargs)
        if getattr(self, '_tables_batched', False):
            return self._rescaled(self._pk, 1)
        return self._pk * self._rsigma8sq

    @property
    def zmin(self):
        """Minimum (spline-interpolated) redshift."""
        return self.z[0]

    @property
    def zmax(self):
        """Maximum (spline-interpolated) redshift."""
        return self.z[-1]

    @classmethod
    def from_callable(cls, k=None, z=None, pk_callable=None, growth_factor_sq=None, extrap_kmin=_default_extrap_kmin, extrap_kmax=_default_extrap_kmax,
                      device=None):
        """
        Build from callables (reference interpolator.py:696-739).  ``pk_callable(k)`` -> (..., nk) and ``growth_factor_sq(z)`` ->
        (..., nz) device tensors / arrays (leading dimensions = a batch of cosmologies), or ``pk_callable(k, z, grid=True)`` -> (..., nk, nz).
        """
        if k is None:
            k = get_default_k_callable()
        if z is None:
            z = get_default_z_callable()
        self = cls.__new__(cls)
        self.__dict__.update(self.default_params)
        self._rsigma8sq = 1.
        self.device = dv.resolve_device(device)
        self.k, self.z = np.sort(_host(k).ravel()), np.sort(_host(z).ravel())
        self.growth_factor_sq = growth_factor_sq
        self.extrap_kmin, self.extrap_kmax = extrap_kmin, extrap_kmax
        self.is_from_callable = True
        self._interp = pk_callable
        return self

    def _eval_device(self, kh, zh, grid=True, ignore_growth=False, bounds_error=False, extra=None):
        """P(k, z) at flat host coordinates as a device tensor (batch..., nk, nz) (grid) or (batch..., nk) (pairs)."""
        torch = dv.torch()
        extra = extra or {}
        if self.is_from_callable:
            mask_k, mask_z = _mask_bounds([kh, zh], [(self.extrap_kmin, self.extrap_kmax), (self.zmin, self.zmax)], bounds_error=bounds_error)
            if dv.is_torch(mask_k):      # wavenumbers that live on the device (a mesh): masks there too
                mask_z = dv.upload(mask_z, mask_k.device)
            mask = mask_k[:, None] & mask_z if grid else mask_k & mask_z
            if self.growth_factor_sq is not None:
                tmp = dv.to_device(self._interp(kh, **extra), self.device)            # (..., nk)
                if not ignore_growth:
                    growth = dv.to_device(self.growth_factor_sq(zh), self.device)     # (..., nz)
                    tmp = tmp[..., :, None] * growth[..., None, :] if grid else tmp * growth
                elif grid:
                    tmp = tmp[..., :, None].expand(tmp.shape + (zh.size,))
            else:
                tmp = dv.to_device(self._interp(kh, zh, grid=grid, **extra), self.device)
            out = tmp if bool(mask.all()) else torch.where(dv.upload(mask, self.device), tmp, torch.full_like(tmp, float('nan')))
        else:
            is2d = self._is2d()
            mask_k, mask_z = _mask_bounds([kh, zh], [self._table_k_limits(), (self.zmin, self.zmax)], bounds_error=bounds_error)
            if not is2d:
                mask_z = mask_z | True    # ignore input z
            mask = mask_k[:, None] & mask_z if grid else mask_k & mask_z
            if is2d:
                interp = self._interp
                saved = interp.extrap
                interp.extrap = True      # the P(k, z) mask below uses the extrapolation range, as the reference does
                tmp = interp(dv.upload(kh, self.device), dv.upload(zh, self.device), grid=grid)
                interp.extrap = saved
            else:
                tmp = self._interp(dv.upload(kh, self.device))
                if tmp.ndim > 1:      # a batch of columns: the batch axis leads
                    tmp = tmp.T
                if grid:
                    tmp = tmp[..., :, None].expand(tuple(tmp.shape) + (zh.size,))
            if self.growth_factor_sq is not None and not ignore_growth:
                growth = dv.to_device(self.growth_factor_sq(zh), self.device)
                tmp = tmp * (growth[..., None, :] if grid and growth.ndim > 1 else growth)      # (batch, nz) against (batch, nk, nz)
            out = tmp if mask.all() else torch.where(dv.upload(mask, self.device), tmp, torch.full_like(tmp, float('nan')))
        if getattr(self, '_tables_batched', False):
            return self._rescaled(out, 1)
        if isinstance(self._rsigma8sq, float) and self._rsigma8sq == 1.:     # (nothing to rescale: no pass over the result)
            return out
        return out * self._rsigma8sq

    def __call__(self, k, z, grid=True, **kwargs):
        """Evaluate at wavenumbers ``k`` and redshifts ``z``: shape (batch...) + k.shape + z.shape (``grid``) or + k.shape (pairs).
        ``kwargs``: ``ignore_growth`` (leave the growth factor out), ``bounds_error`` (raise outside the ranges instead of NaN), both False by
        default (reference interpolator.py:741-817); anything else reaches the callable of an interpolator built by :meth:`from_callable`."""
        ignore_growth, bounds_error, extra = _call_options(kwargs, self.is_from_callable, ('ignore_growth', 'bounds_error'))
        like_torch = dv.is_torch(k) or dv.is_torch(z)
        dtype = dv.float_dtype(k, z)
        zh = _host(z)
        if self.is_from_callable and self.growth_factor_sq is not None and dv.is_torch(k) and k.is_cuda and k.numel() > 16384:
            kh = k.to(dv.torch().float64)      # a mesh of wavenumbers on the device goes to the callable as it is (no copy to the host and back)
        else:
            kh = _host(k)
        out = self._eval_device(kh.reshape(-1), zh.ravel(), grid=grid, ignore_growth=ignore_growth, bounds_error=bounds_error, extra=extra)
        nlead = out.ndim - (2 if grid else 1)
        shape = tuple(out.shape[:nlead]) + (kh.shape + zh.shape if grid else kh.shape)
        return _finish(out, dtype, like_torch, shape)

    def _rows_z(self, zh, ignore_growth=False):
        """Callable k -> rows (batch..., nz, nk) of P(k, z) for the sigma integrals."""
        def rows(kh):
            if not self.is_from_callable and self._is2d() and kh.size and zh.size:
                # (k, z) tables: the surfaces come out of the two spline operators z-major already (z contraction first), no transposed copy
                torch = dv.torch()
                mask_k, mask_z = _mask_bounds([kh, zh], [self._table_k_limits(), (self.zmin, self.zmax)])
                out = self._interp.rows_y_major(kh, zh, exp10=self._interp.interp_fun == 'log')
                mask = mask_z[:, None] & mask_k
                if not mask.all():
                    out = torch.where(dv.upload(mask, self.device), out, torch.full_like(out, float('nan')))
                if self.growth_factor_sq is not None and not ignore_growth:
                    out = out * dv.to_device(self.growth_factor_sq(zh), self.device)[..., :, None]
                return self._rescaled(out, out.ndim - 2) if getattr(self, '_tables_batched', False) else (
                    out if isinstance(self._rsigma8sq, float) and self._rsigma8sq == 1. else out * self._rsigma8sq)
            return self._eval_device(kh, zh, grid=True, ignore_growth=ignore_growth).transpose(-1, -2).contiguous()
        return rows

    def _separable(self):
        """P(k, z) = P(k) x growth_factor_sq(z) exactly: interpolators built from (callable or one tabulated column) + growth factor."""
        if self.growth_factor_sq is None:
            return False
        return self.is_from_callable or not self._is2d()

    def _growth_sq_device(self, zh):
        """growth_factor_sq at flat host redshifts as a device tensor (batch..., nz), NaN outside the redshift range of a callable."""
        torch = dv.torch()
        growth = dv.to_device(self.growth_factor_sq(zh), self.device)
        if self.is_from_callable:
            _, mask_z = _mask_bounds([self.z[:1], zh], [(self.zmin, self.zmax)] * 2)
            if not mask_z.all():
                growth = torch.where(dv.upload(mask_z, self.device), growth, torch.full_like(growth, float('nan')))
        return growth

    def _sigma_separable(self, integrate, zh):
        """sigma^2(..., z) = growth_factor_sq(z) x sigma^2 of the z-independent spectrum: the k integral is linear in P, so ONE transform
        per cosmology replaces one per (cosmology, z) -- same numbers as the reference's per-z integrals to rounding.  Returns the two
        square roots, (batch..., n) and (batch..., nz) (NaN outside the redshift range), for the caller to multiply out in the layout
        it returns: the (batch, n, nz) result is then written once instead of being multiplied, masked, rooted and transposed in four passes."""
        torch = dv.torch()
        z0 = np.array([self.z[0]])                                 # any redshift inside the table: the growth factor is left out

        def rows(kh):
            return self._eval_device(kh, z0, grid=True, ignore_growth=True).transpose(-1, -2).contiguous()      # (batch..., 1, nk)

        base = integrate(rows)[..., 0, :]                          # (batch..., n)
        growth = dv.to_device(self.growth_factor_sq(zh), self.device)          # (batch..., nz)
        if self.is_from_callable:                                  # NaN outside the redshift range (tabulated single columns ignore z)
            _, mask_z = _mask_bounds([z0, zh], [(self.zmin, self.zmax)] * 2)
            growth = torch.where(dv.upload(mask_z, self.device), growth, torch.full_like(growth, float('nan')))
        return base.sqrt(), growth.sqrt()

    def sigma_dz(self, z, **kwargs):
        r""":math:`\sigma_d(z) = \sqrt{\frac{1}{6\pi^2}\int dk P(k, z)}` (reference interpolator.py:819-844); shape (batch...) + z.shape."""
        like_torch = dv.is_torch(z)
        dtype = dv.float_dtype(z)
        zh = _host(z)
        if self._separable():
            base, growth = self._sigma_separable(lambda rows: sigma_d2_of_rows(rows, kmin=self.extrap_kmin, kmax=self.extrap_kmax, device=self.device,
                                                                                 **kwargs)[..., None], zh.ravel())
            out = base * growth                                    # (batch..., 1) x (batch..., nz)
        else:
            out = sigma_d2_of_rows(self._rows_z(zh.ravel()), kmin=self.extrap_kmin, kmax=self.extrap_kmax, device=self.device, **kwargs)**0.5
        return _finish(out, dtype, like_torch, tuple(out.shape[:-1]) + zh.shape)

    def sigma_rz(self, r, z, **kwargs):
        r"""
        R.m.s. of perturbations in spheres of radius :math:`r` at redshifts :math:`z` (reference interpolator.py:846-875):
        shape (batch...) + r.shape + z.shape.
        """
        like_torch = dv.is_torch(r) or dv.is_torch(z)
        dtype = dv.float_dtype(r, z)
        rh, zh = _host(r), _host(z)
        if self._separable() and rh.size and zh.size:
            # sigma^2(r, z) = growth_factor_sq(z) x sigma^2(r) of the z-independent spectrum (the k integral is linear in P): ONE transform per
            # cosmology instead of one per (cosmology, z), and the (batch, nr, nz) result written once by the kernel that interpolates in r
            growth_sq = self._growth_sq_device(zh.ravel())
            out = self._sigma_rz_fused(rh, growth_sq, kwargs)
            if out is not None:
                return _finish(out, dtype, like_torch, tuple(out.shape[:-2]) + rh.shape + zh.shape)

            def rows(kh):
                return self._eval_device(kh, self.z[:1], grid=True, ignore_growth=True)[..., 0]      # (batch..., nk)

            out = sigma_r2_of_rows(rh, rows, kmin=self.extrap_kmin, kmax=self.extrap_kmax, device=self.device, growth_sq=growth_sq, **kwargs)
        elif self._separable():
            base, growth = self._sigma_separable(lambda rows: sigma_r2_of_rows(rh, rows, kmin=self.extrap_kmin, kmax=self.extrap_kmax, device=self.device,
                                                                                 **kwargs), zh.ravel())
            out = base[..., :, None] * growth[..., None, :]
        else:
            # (batch..., nz, nk) rows -> (batch..., nr, nz): the transposition is part of the store of the kernel that splines to r
            out = sigma_r2_of_rows(rh, self._rows_z(zh.ravel()), kmin=self.extrap_kmin, kmax=self.extrap_kmax, device=self.device, sqrt=True,
                                     radii_before_last_axis=True, **kwargs)
        return _finish(out, dtype, like_torch, tuple(out.shape[:-2]) + rh.shape + zh.shape)

    _two_stream_min_bytes = 1 << 20        # results smaller than this take the three separate calls (a few microseconds of kernels either way)
    _two_stream_blocks = 0       # 0: the fused kernel (cp_sigma.hip); n > 0: n blocks on two streams with the three separate kernels (measurements)

    def _sigma_rz_fused(self, rh, growth_sq, kwargs):
        """Batches of cosmologies built from an analytic engine's callable, default method: ONE library call, ``cp_sigma_rz_analytic``, which runs
        the whole chain -- P(k) evaluated into the registers of the FFTLog, transform in LDS, spline to the radii out of LDS, product with the growth
        factors, root, the (nr x nz) results written once -- as one kernel (csrc/cp_sigma.hip).  The ALU-bound and the store-bound stage then
        overlap across workgroups instead of adding up as three launches, and the spectra and variances never leave the CU.  Same arithmetic in
        the same order as the separate kernels.  Returns None when the conditions are not met (the caller takes the separate calls)."""
        torch = dv.torch()
        call = self._interp if self.is_from_callable else None
        rs = self._rsigma8sq
        if call is None or not hasattr(call, 'analytic_engine') or kwargs or growth_sq.ndim != 2 or not (isinstance(rs, float) and rs == 1.):
            return None
        nb, nz = growth_sq.shape
        if nb * rh.size * nz * 8 < self._two_stream_min_bytes:
            return None
        engine, bg, pk = call.analytic_engine()
        res = sigma_rz_analytic(engine, bg, pk, rh.ravel(), growth_sq, self.device, kmin=self.extrap_kmin, kmax=self.extrap_kmax, blocks=self._two_stream_blocks)
        return None if res is None else res[0]

    def sigma8_z(self, z=0, **kwargs):
        """R.m.s. of perturbations in a sphere of 8."""
        return self.sigma_rz(8., z=z, **kwargs)

    def rescale_sigma8(self, sigma8=1.):
        """Rescale power spectrum to the provided ``sigma8`` normalisation at z = 0."""
        self._rsigma8sq = 1.
        self._rsigma8sq = sigma8**2 / self.sigma8_z(z=0)**2

    def growth_rate_rz(self, r, z, dz=1e-3, **kwargs):
        r""":math:`f(r, z) = d\ln\sigma_r(z) / d\ln a` by finite differences in z (reference interpolator.py:886-936)."""
        hdz = dz / 2.
        dtype = dv.float_dtype(r, z)
        rh, zh = _host(r), _host(z)
        shape = rh.shape + zh.shape
        if not all(shape):
            return np.zeros(shape, dtype=dtype)
        zf = zh.ravel()

        def fun(zz):
            return np.log(np.asarray(self.sigma_rz(rh.ravel(), zz, **kwargs), dtype='f8'))

        feval = [f.reshape(f.shape[:-2] + (-1, zf.size)) for f in [fun(zf - dz), fun(zf - hdz), fun(zf), fun(zf + hdz), fun(zf + dz)]]
        toret = np.where(zf < self.zmin + hdz, -feval[4] + 4 * feval[3] - 3 * feval[2], feval[3] - feval[1])
        toret = np.where(zf > self.zmax - hdz, -(-feval[0] + 4 * feval[1] - 3 * feval[2]), toret)
        dsigdlna = -(toret / dz) * (1 + zf)
        return dsigdlna.astype(dtype).reshape(dsigdlna.shape[:-2] + shape)

    def to_1d(self, z, **kwargs):
        """:class:`PowerSpectrumInterpolator1D` at redshift(s) ``z`` (reference interpolator.py:938-963)."""
        if self.is_from_callable:
            zh = _host(z)

            def pk_callable(kh):
                out = self._eval_device(np.asarray(kh, dtype='f8').ravel(), zh.ravel(), grid=True) / self._rsigma8sq
                nbatch = out.ndim - 2
                if nbatch:      # a batch of cosmologies becomes columns of the 1D interpolator: (nk, batch..., nz), as Primordial.pk_interpolator
                    out = out.permute(nbatch, *range(nbatch), out.ndim - 1)
                out = out.reshape(out.shape[:-1] + zh.shape) if zh.ndim else out[..., 0]
                return out * self._rsigma8sq

            return PowerSpectrumInterpolator1D.from_callable(self.k, pk_callable=pk_callable, extrap_kmin=self.extrap_kmin, extrap_kmax=self.extrap_kmax,
                                                             device=self.device)
        default_params = dict(extrap_pk=self.extrap_pk, extrap_kmin=self.extrap_kmin, extrap_kmax=self.extrap_kmax, interp_order_k=self.interp_order_k)
        default_params.update(kwargs)
        self.extrap_kmin, self.extrap_kmax = -np.inf, np.inf    # in case self.k > self.extrap_kmax
        pk = self(self.k, z=z)
        self.extrap_kmin, self.extrap_kmax = default_params['extrap_kmin'], default_params['extrap_kmax']
        return PowerSpectrumInterpolator1D(self.k, pk, device=self.device, **default_params)

    def to_xi_arrays(self, nk=1024, fftlog_kwargs=None):
        """FFTLog transform to the correlation function on the (s, z) grid, growth factor left out as in the reference
        (interpolator.py:965-987, ``ignore_growth=True``): ``(s, z, xi)`` arrays with xi (batch..., nk, nz)."""
        k = np.geomspace(self.extrap_kmin, self.extrap_kmax, nk)
        rows = self._rows_z(self.z, ignore_growth=True)(k)     # (batch..., nz, nk)
        s, xi = _cached_fftlog(PowerToCorrelation, k, self.device, fftlog_kwargs)(rows)
        return s.cpu().numpy(), self.z, xi.transpose(-1, -2).cpu().numpy()

    def to_xi(self, nk=1024, fftlog_kwargs=None, **kwargs):
        """Transform into a :class:`CorrelationFunctionInterpolator2D` with FFTLog (reference interpolator.py:965-987).
        A batch of cosmologies (leading dimensions) gives an interpolator of the batch: xi(s, z) -> (batch..., ns, nz), see :meth:`_to_xi_batch`."""
        default_params = dict(interp_s='log', interp_order_s=self.interp_order_k, interp_order_z=self.interp_order_z, growth_factor_sq=self.growth_factor_sq)
        default_params.update(kwargs)
        batch = self._to_xi_batch(nk, fftlog_kwargs, default_params)
        if batch is not None:
            return batch
        s, z, xi = self.to_xi_arrays(nk=nk, fftlog_kwargs=fftlog_kwargs)
        return CorrelationFunctionInterpolator2D(s, z=z, xi=xi, device=self.device, **default_params)

    def _to_xi_batch(self, nk, fftlog_kwargs, params):
        """:meth:`to_xi` of a batch of cosmologies, or None for one cosmology: a :class:`CorrelationFunctionInterpolator2D` of a batch of tables kept on
        the device, (batch, ns, nz) -- or (batch, ns, 1) for P(k) x growth factor: ONE transform per cosmology, the growth factor handed on."""
        k = np.geomspace(self.extrap_kmin, self.extrap_kmax, nk)
        fft = _cached_fftlog(PowerToCorrelation, k, self.device, fftlog_kwargs)
        if self._separable():
            rows = self._eval_device(k, self.z[:1], grid=True, ignore_growth=True)[..., 0]      # (batch..., nk)
            if rows.ndim < 2:
                return None
            rows = rows[..., None, :]
        elif getattr(self, '_tables_batched', False) or (self.is_from_callable and self._eval_device(k[:1], self.z[:1]).ndim > 2):
            rows = self._rows_z(self.z, ignore_growth=True)(k)      # (batch..., nz, nk)
        else:
            return None
        s, xi = fft(rows.contiguous())
        xi = xi.transpose(-1, -2)
        return CorrelationFunctionInterpolator2D(s.cpu().numpy(), z=self.z, xi=xi.reshape((-1,) + tuple(xi.shape[-2:])), device=self.device, **params)


def get_default_s_callable():
    """Default separations of interpolators built from callables (reference interpolator.py:30-31)."""
    return np.logspace(-6., 2., 500)


class _BaseCorrelationFunctionInterpolator(dv.Copyable):

    """Base class for correlation function interpolators (reference interpolator.py:990-1071)."""

    def _prepare(self, s, xi, z=None, interp_s='log'):
        self.s = np.array(_host(s), dtype='f8').ravel()      # private copies: public attributes
        self._xi = np.array(_host(xi), dtype='f8')
        if self._xi.ndim > 1:
            self._xi = self._xi.reshape(self.s.shape + (-1,))
        ix = np.argsort(self.s)
        self.s, self._xi = self.s[ix], self._xi[ix]
        if z is not None:
            self.z = np.array(_host(z), dtype='f8').ravel()
            ix = np.argsort(self.z)
            self.z, self._xi = self.z[ix], self._xi[:, ix]
        self.interp_s = str(interp_s)
        return self.s, self._xi

    def params(self):
        """Return interpolator parameter dictionary."""
        return {name: getattr(self, name) for name in self.default_params}

    def as_dict(self):
        """Return interpolator as a dictionary."""
        state = self.params()
        for name in ['s', 'xi']:
            state[name] = getattr(self, name)
        if hasattr(self, 'z'):
            state['z'] = self.z
        return state

    def clone(self, **kwargs):
        """Clone interpolator, i.e. return a deepcopy with (possibly) other attributes in ``kwargs``."""
        return self.__class__(**{**self.as_dict(), **kwargs})

    def deepcopy(self):
        """Deep copy interpolator (interpolators built from callables are re-tabulated at ``s``, as in the reference)."""
        return self.__class__(**self.as_dict())

    @property
    def smin(self):
        """Minimum (interpolated) ``s`` value."""
        return self.s[0]

    @property
    def smax(self):
        """Maximum (interpolated) ``s`` value."""
        return self.s[-1]

    @property
    def extrap_smin(self):
        """Minimum (extrapolated) ``s`` value (same as minimum interpolated value)."""
        return self.s[0]

    @property
    def extrap_smax(self):
        """Maximum (extrapolated) ``s`` value (same as maximum interpolated value)."""
        return self.s[-1]


class CorrelationFunctionInterpolator1D(_BaseCorrelationFunctionInterpolator):

    """1D correlation function interpolator: lin-y natural spline in (log) s, no extrapolation (reference interpolator.py:1074-1222)."""

    def __init__(self, s, xi, interp_s='log', interp_order_s=3, device=None):
        self._rsigma8sq = 1.
        self.device = dv.resolve_device(device, xi)
        s, xi = self._prepare(s, xi, interp_s=interp_s)
        self.interp_order_s = int(interp_order_s)
        self._interp = Interpolator1D(s, xi, k=self.interp_order_s, interp_x=self.interp_s, assume_sorted=True, device=self.device)
        self.is_from_callable = False

    default_params = _get_default_kwargs(__init__, start=3, remove=('device',))

    @property
    def xi(self):
        """Correlation function array (evaluated at ``s`` if built from a callable), with normalisation."""
        if self.is_from_callable:
            return self(self.s)
        return self._xi * self._rsigma8sq

    @classmethod
    def from_callable(cls, s=None, xi_callable=None, device=None):
        """Build from ``xi_callable(s)`` -> array / device tensor of shape (ns,) or (ns, ncol) (reference interpolator.py:1109-1139)."""
        if s is None:
            s = get_default_s_callable()
        self = cls.__new__(cls)
        self.__dict__.update(self.default_params)
        self._rsigma8sq = 1.
        self.device = dv.resolve_device(device)
        self.s = np.sort(_host(s).ravel())
        self.is_from_callable = True
        self._interp = xi_callable
        return self

    def _eval_device(self, sh, bounds_error=False, extra=None):
        """xi(s) at host separations ``sh`` (flat) as a device tensor (ns,) + trailing column shape."""
        torch = dv.torch()
        if self.is_from_callable:
            mask_s, = _mask_bounds([sh], [(self.smin, self.smax)], bounds_error=bounds_error)
            out = dv.to_device(self._interp(sh, **(extra or {})), self.device)
            mask = dv.upload(mask_s, self.device).reshape((-1,) + (1,) * (out.ndim - 1))
            out = torch.where(mask, out, torch.full_like(out, float('nan')))
        else:
            out = self._interp(dv.upload(sh, self.device), bounds_error=bounds_error)
        return out * self._rsigma8sq

    def __call__(self, s, **kwargs):
        """Evaluate the correlation function at separations ``s``; NaN outside [smin, smax], ``bounds_error=True`` raises instead
        (reference interpolator.py:1142-1165; other ``kwargs`` reach the callable of an interpolator built by :meth:`from_callable`)."""
        bounds_error, extra = _call_options(kwargs, self.is_from_callable)
        like_torch = dv.is_torch(s)
        dtype = dv.float_dtype(s)
        sh = _host(s)
        out = self._eval_device(sh.ravel(), bounds_error=bounds_error, extra=extra)
        return _finish(out, dtype, like_torch, sh.shape + tuple(out.shape[1:]))

    def sigma_d(self, **kwargs):
        """R.m.s. of the displacement field, through :meth:`to_pk` (reference interpolator.py:1169-1175)."""
        return self.to_pk().sigma_d(**kwargs)

    def sigma_r(self, r, **kwargs):
        """R.m.s. of perturbations in a sphere of radius r, through :meth:`to_pk` (reference interpolator.py:1177-1183)."""
        return self.to_pk().sigma_r(r, **kwargs)

    def sigma8(self, **kwargs):
        """R.m.s. of perturbations in a sphere of 8."""
        return self.sigma_r(8., **kwargs)

    def rescale_sigma8(self, sigma8=1.):
        """Rescale the correlation function to the provided ``sigma8`` normalisation."""
        self._rsigma8sq = 1.
        self._rsigma8sq = sigma8**2 / self.sigma8()**2

    def to_pk(self, ns=1024, fftlog_kwargs=None, **kwargs):
        """Transform into a :class:`PowerSpectrumInterpolator1D` with FFTLog (reference interpolator.py:1201-1222)."""
        s = np.geomspace(self.extrap_smin, self.extrap_smax, ns)
        out = self._eval_device(s)
        cs = tuple(out.shape[1:])
        rows = out.reshape(ns, -1).T.contiguous() if out.ndim > 1 else out[None, :]
        k, pk = _cached_fftlog(CorrelationToPower, s, self.device, fftlog_kwargs)(rows)
        default_params = dict(interp_k='log', interp_order_k=self.interp_order_s)
        default_params.update(kwargs)
        return PowerSpectrumInterpolator1D(k.cpu().numpy(), pk=pk.T.reshape((ns,) + cs), device=self.device, **default_params)


class CorrelationFunctionInterpolator2D(_BaseCorrelationFunctionInterpolator):

    """2D correlation function interpolator (reference interpolator.py:1225-1498)."""

    def __init__(self, s, z, xi=None, interp_s='log', interp_order_s=3, interp_order_z=None, growth_factor_sq=None, device=None):
        self._rsigma8sq = 1.
        self.growth_factor_sq = growth_factor_sq
        self.device = dv.resolve_device(device, xi)
        self._tables_batched = len(xi.shape if hasattr(xi, 'shape') else np.shape(xi)) == 3
        if self._tables_batched:
            # (batch, ns, nz): one (s, z) table per cosmology on shared grids, kept on the device (extension of the reference's (ns, nz) table, what
            # to_xi() of a batch of cosmologies returns); (batch, ns, 1) with several z: xi(s) x growth_factor_sq(z), z being the range of validity
            s, xi = self._prepare_tables(s, z, xi, interp_s=interp_s)
        else:
            s, xi = self._prepare(s, xi, z=z, interp_s=interp_s)
        is2d = self._xi.shape[-1] > 1
        # int() of the default None raises TypeError, as in the reference (interpolator.py:1262: its own default does not construct; to_xi() always
        # passes the order of the P(k, z) interpolator it comes from)
        self.interp_order_s, self.interp_order_z = int(interp_order_s), int(interp_order_z)
        if is2d:
            self._interp = Interpolator2D(s, self.z, xi, kx=self.interp_order_s, ky=self.interp_order_z, interp_x=self.interp_s, assume_sorted=True,
                                          device=self.device)
        else:
            if self.growth_factor_sq is None:
                raise ValueError('provide either 2D pk array or growth_factor_sq')
            self._interp = Interpolator1D(s, xi[:, :, 0].T if self._tables_batched else xi[:, 0], k=self.interp_order_s, interp_x=self.interp_s, assume_sorted=True,
                                          device=self.device)
        self.is_from_callable = False

    def _prepare_tables(self, s, z, xi, interp_s='log'):
        """``_prepare`` for a batch of tables (batch, ns, nz) -- or (batch, ns, 1): constant in z over the range of ``z`` -- on the device."""
        self.s, self.z = np.array(_host(s), dtype='f8').ravel(), np.array(_host(z), dtype='f8').ravel()
        xi = dv.to_device(xi, self.device)
        if tuple(xi.shape[1:]) not in [(self.s.size, self.z.size), (self.s.size, 1)]:
            raise ValueError('xi must be (batch, {0:d}, {1:d}) or (batch, {0:d}, 1), got {2}'.format(self.s.size, self.z.size, tuple(xi.shape)))
        i_s, i_z = np.argsort(self.s), np.argsort(self.z)
        if np.any(i_s[1:] < i_s[:-1]):
            xi = xi.index_select(1, dv.upload(i_s, self.device))
        if xi.shape[2] > 1 and np.any(i_z[1:] < i_z[:-1]):
            xi = xi.index_select(2, dv.upload(i_z, self.device))
        self.s, self.z, self._xi = self.s[i_s], self.z[i_z], xi.contiguous()
        self.interp_s = str(interp_s)
        return self.s, self._xi

    def _rescaled(self, out):
        """``out`` (batch..., ...) times the sigma8 rescaling factor (a float, or one value per batch entry)."""
        rs = self._rsigma8sq
        if np.ndim(rs) or dv.is_torch(rs):
            rs = dv.to_device(rs, self.device)
            rs = rs.reshape(tuple(rs.shape) + (1,) * (out.ndim - rs.ndim))
        return out * rs

    default_params = _get_default_kwargs(__init__, start=4, remove=('device',))

    @property
    def xi(self):
        """Correlation function array (evaluated on (s, z) if built from a callable), without growth factor, with normalisation."""
        if self.is_from_callable:
            kwargs = {'ignore_growth': True} if self.growth_factor_sq is not None else {}
            return self(self.s, self.z, **kwargs)
        return self._rescaled(self._xi)

    @property
    def zmin(self):
        """Minimum (spline-interpolated) redshift."""
        return self.z[0]

    @property
    def zmax(self):
        """Maximum (spline-interpolated) redshift."""
        return self.z[-1]

    @classmethod
    def from_callable(cls, s=None, z=None, xi_callable=None, growth_factor_sq=None, device=None):
        """Build from ``xi_callable(s)`` + ``growth_factor_sq(z)``, or ``xi_callable(s, z, grid=True)`` (reference interpolator.py:1300-1337)."""
        if s is None:
            s = get_default_s_callable()
        if z is None:
            z = get_default_z_callable()
        self = cls.__new__(cls)
        self.__dict__.update(self.default_params)
        self._rsigma8sq = 1.
        self.device = dv.resolve_device(device)
        self.s, self.z = np.sort(_host(s).ravel()), np.sort(_host(z).ravel())
        self.growth_factor_sq = growth_factor_sq
        self.is_from_callable = True
        self._interp = xi_callable
        return self

    def _eval_device(self, sh, zh, grid=True, ignore_growth=False, bounds_error=False, extra=None):
        """xi(s, z) at flat host coordinates as a device tensor (ns, nz) (grid) or (ns,) (pairs)."""
        torch = dv.torch()
        extra = extra or {}
        mask_s, mask_z = _mask_bounds([sh, zh], [(self.smin, self.smax), (self.zmin, self.zmax)], bounds_error=bounds_error)
        if self.is_from_callable:
            mask = mask_s[:, None] & mask_z if grid else mask_s & mask_z
            if self.growth_factor_sq is not None:
                tmp = dv.to_device(self._interp(sh, **extra), self.device)
                if not ignore_growth:
                    growth = dv.to_device(self.growth_factor_sq(zh), self.device)
                    tmp = tmp[..., :, None] * growth[..., None, :] if grid else tmp * growth
                elif grid:
                    tmp = tmp[..., :, None].expand(tmp.shape + (zh.size,))
            else:
                tmp = dv.to_device(self._interp(sh, zh, grid=grid, **extra), self.device)
        else:
            is2d = self._xi.shape[-1] > 1
            if not is2d and not (getattr(self, '_tables_batched', False) and self.z.size > 1):
                mask_z = mask_z | True    # ignore input z
            mask = mask_s[:, None] & mask_z if grid else mask_s & mask_z
            if is2d:
                tmp = self._interp(dv.upload(sh, self.device), dv.upload(zh, self.device), grid=grid)
            else:
                tmp = self._interp(dv.upload(sh, self.device))
                if tmp.ndim > 1:      # a batch of columns: the batch axis leads
                    tmp = tmp.T
                if grid:
                    tmp = tmp[..., :, None].expand(tuple(tmp.shape) + (zh.size,))
            if self.growth_factor_sq is not None and not ignore_growth:
                growth = dv.to_device(self.growth_factor_sq(zh), self.device)
                tmp = tmp * (growth[..., None, :] if grid and growth.ndim > 1 else growth)      # (batch, nz) against (batch, ns, nz)
        out = torch.where(dv.upload(mask, self.device), tmp, torch.full_like(tmp, float('nan')))
        return self._rescaled(out)

    def __call__(self, s, z, grid=True, **kwargs):
        """Evaluate at separations ``s`` and redshifts ``z``: shape (batch...) + s.shape + z.shape (``grid``) or + s.shape (pairs).
        ``kwargs``: ``ignore_growth``, ``bounds_error``, both False by default (reference interpolator.py:1337-1406); anything else reaches the
        callable of an interpolator built by :meth:`from_callable`."""
        ignore_growth, bounds_error, extra = _call_options(kwargs, self.is_from_callable, ('ignore_growth', 'bounds_error'))
        like_torch = dv.is_torch(s) or dv.is_torch(z)
        dtype = dv.float_dtype(s, z)
        sh, zh = _host(s), _host(z)
        out = self._eval_device(sh.ravel(), zh.ravel(), grid=grid, ignore_growth=ignore_growth, bounds_error=bounds_error, extra=extra)
        lead = tuple(out.shape[:out.ndim - (2 if grid else 1)])
        return _finish(out, dtype, like_torch, lead + (sh.shape + zh.shape if grid else sh.shape))

    def sigma_dz(self, z, **kwargs):
        """R.m.s. of the displacement field at ``z``, through :meth:`to_pk` (reference interpolator.py:1409-1415)."""
        return self.to_pk().sigma_dz(z=z, **kwargs)

    def sigma_rz(self, r, z, **kwargs):
        """R.m.s. of perturbations in spheres of radius r at ``z``, through :meth:`to_pk` (reference interpolator.py:1417-1423)."""
        return self.to_pk().sigma_rz(r, z=z, **kwargs)

    def sigma8_z(self, z, **kwargs):
        """R.m.s. of perturbations in a sphere of 8 at ``z``."""
        return self.sigma_rz(8., z=z, **kwargs)

    def rescale_sigma8(self, sigma8=1.):
        """Rescale the correlation function to the provided ``sigma8`` normalisation at z = 0."""
        self._rsigma8sq = 1.
        self._rsigma8sq = sigma8**2 / self.sigma8_z(z=0)**2

    def growth_rate_rz(self, r, z, **kwargs):
        """Growth rate from the log-derivative of sigma_r(z), through :meth:`to_pk` (reference interpolator.py:1438-1444)."""
        return self.to_pk().growth_rate_rz(r, z=z, **kwargs)

    def to_1d(self, z, **kwargs):
        """:class:`CorrelationFunctionInterpolator1D` at redshift ``z`` (reference interpolator.py:1446-1467)."""
        if self.is_from_callable:
            return CorrelationFunctionInterpolator1D.from_callable(self.s, lambda s, **kw: self(s, z=z, **kw), device=self.device)
        if getattr(self, '_tables_batched', False):
            zh = _host(z)

            def xi_callable(sh):      # the cosmologies of the batch become columns of the 1D interpolator: (ns, batch, nz), as PowerSpectrumInterpolator2D.to_1d
                out = self._eval_device(np.asarray(sh, dtype='f8').ravel(), zh.ravel(), grid=True).permute(1, 0, 2)
                return out.reshape(out.shape[:-1] + zh.shape) if zh.ndim else out[..., 0]

            return CorrelationFunctionInterpolator1D.from_callable(self.s, xi_callable, device=self.device)
        default_params = dict(interp_order_s=self.interp_order_s)
        default_params.update(kwargs)
        return CorrelationFunctionInterpolator1D(self.s, self(self.s, z=z), device=self.device, **default_params)

    def to_pk(self, ns=1024, fftlog_kwargs=None, **kwargs):
        """Transform into a :class:`PowerSpectrumInterpolator2D` with FFTLog, growth factor left out (reference interpolator.py:1469-1498)."""
        s = np.geomspace(self.extrap_smin, self.extrap_smax, ns)
        constant = self.growth_factor_sq is not None and (self.is_from_callable or self._xi.shape[-1] == 1)      # xi(s) x growth factor: one transform (per cosmology)
        zz = self.z[:1] if constant else self.z
        rows = self._eval_device(s, zz, grid=True, ignore_growth=True).transpose(-1, -2).contiguous()     # (batch..., nz, ns)
        k, pk = _cached_fftlog(CorrelationToPower, s, self.device, fftlog_kwargs)(rows)
        default_params = dict(interp_k='log', extrap_pk='log', interp_order_k=self.interp_order_s, interp_order_z=self.interp_order_z,
                              growth_factor_sq=self.growth_factor_sq)
        default_params.update(kwargs)
        pk = pk.transpose(-1, -2)
        if constant:
            pk = pk.expand(tuple(pk.shape[:-1]) + (self.z.size,))
        if pk.ndim > 2:      # a batch of cosmologies: a batch of (k, z) tables on the device (leading dimensions flattened)
            pk = pk.reshape((-1,) + tuple(pk.shape[-2:]))
        return PowerSpectrumInterpolator2D(k.cpu().numpy(), z=self.z, pk=pk, device=self.device, **default_params)
