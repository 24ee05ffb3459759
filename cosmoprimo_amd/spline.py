"""
Cubic splines (and any fixed linear map) from fixed knots to fixed query points, applied on the GPU to batches of rows
(``cp_spline_*`` / ``cp_linop_*``).  The operator is built once per (knots, queries, boundary condition) on the host.
"""
import ctypes

import numpy as np

from . import _lib
from . import _device as dv


class LinearOperator(object):

    """out[..., q] = post(scale * sum_j W[q, j] y[..., j]) with W banded; owns a ``cp_spline_plan``."""

    def __init__(self, handle, n, nq, device):
        self._handle, self.n, self.nq, self.device = handle, n, nq, device

    @classmethod
    def spline(cls, x, xq, bc='natural', nu=0, extrapolate=False, device=None):
        """Cubic spline through knots ``x`` evaluated (derivative order ``nu``) at ``xq``; bc in 'natural', 'clamped', 'not-a-knot'."""
        device = dv.resolve_device(device)
        x = np.ascontiguousarray(x, dtype='f8').ravel()
        xq = np.ascontiguousarray(xq, dtype='f8').ravel()
        handle = ctypes.c_void_p()
        _lib.check(_lib.load().cp_spline_plan_create(ctypes.byref(handle), x.size, _lib.as_double_p(x), xq.size, _lib.as_double_p(xq),
                                                     _lib.SPLINE_BC[bc], int(nu), int(bool(extrapolate)), device.index))
        return cls(handle, x.size, xq.size, device)

    @classmethod
    def dense(cls, w, device=None):
        """Any dense (nq, n) operator (e.g. quadrature weights)."""
        device = dv.resolve_device(device)
        w = np.ascontiguousarray(w, dtype='f8')
        handle = ctypes.c_void_p()
        _lib.check(_lib.load().cp_linop_plan_create(ctypes.byref(handle), w.shape[1], w.shape[0], _lib.as_double_p(w), device.index))
        return cls(handle, w.shape[1], w.shape[0], device)

    @property
    def bandwidth(self):
        bw = ctypes.c_int()
        _lib.check(_lib.load().cp_spline_plan_info(self._handle, None, None, ctypes.byref(bw)))
        return bw.value

    @property
    def columns(self):
        """(first, count): the entries of an input row that :meth:`__call__` reads, whichever kernel it runs (cp_spline_plan_columns) -- the
        producer of the rows may leave the others unwritten (``FFTlog.__call__(..., out_window=op.columns)``)."""
        first, count = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.load().cp_spline_plan_columns(self._handle, ctypes.byref(first), ctypes.byref(count)))
        return first.value, count.value

    _PATHS = {None: 0, 'valu': 16, 'mfma': 32}     # CP_SPLINE_PATH_*: force one kernel (measurements); default: the library's choice

    def __call__(self, y, sqrt=False, scale=1., path=None, last_axis_first=False):
        """y : torch tensor (..., n) on the operator's device -> (..., nq).  Dense operators run as a float64 GEMM on the matrix cores,
        banded ones (splines) on the vector ALUs; ``path`` = 'valu' / 'mfma' forces one of the two kernels.
        last_axis_first : y (..., m, n) -> (..., nq, m): the last-but-one axis of y becomes the fastest of the result (the transposition is
        part of the kernel's store, ``cp_spline_apply_grouped``)."""
        torch = dv.torch()
        y = dv.to_device(y, self.device)
        if y.shape[-1] != self.n:
            raise ValueError('last dimension must be {:d}, got {}'.format(self.n, tuple(y.shape)))
        lead = tuple(y.shape[:-1])
        nrows = int(np.prod(lead, dtype=np.int64))
        group = 0
        if last_axis_first:
            if len(lead) < 1:
                raise ValueError('last_axis_first needs y of at least two dimensions')
            group, oshape = int(lead[-1]), lead[:-1] + (self.nq, lead[-1])
        else:
            oshape = lead + (self.nq,)
        out = torch.empty(oshape, dtype=torch.float64, device=self.device)
        if nrows:
            _lib.check(_lib.load().cp_spline_apply_grouped(self._handle, y.data_ptr(), out.data_ptr(), nrows, group, int(bool(sqrt)) | self._PATHS[path],
                                                           float(scale), dv.stream_of(self.device)))
        return out

    _POSTS = {None: 0, 'sqrt': 1, 'exp10': 2}       # CP_SPLINE_POST_*

    def mid(self, y, post=None, scale=1.):
        """y : (..., n, m) device tensor -> (..., nq, m) = post(scale x sum_j W[q, j] y[..., j, :]): the operator along the last-but-one axis, the
        last axis left contiguous (``cp_linop_apply_mid``, matrix cores; operators built with :meth:`dense` only).  post : None, 'sqrt' or 'exp10'."""
        torch = dv.torch()
        y = dv.to_device(y, self.device)
        if y.ndim < 2 or y.shape[-2] != self.n:
            raise ValueError('last-but-one dimension must be {:d}, got {}'.format(self.n, tuple(y.shape)))
        lead, m = tuple(y.shape[:-2]), int(y.shape[-1])
        nb = int(np.prod(lead, dtype=np.int64))
        out = torch.empty(lead + (self.nq, m), dtype=torch.float64, device=self.device)
        if nb and m:
            _lib.check(_lib.load().cp_linop_apply_mid(self._handle, y.data_ptr(), out.data_ptr(), nb, m, self._POSTS[post], float(scale),
                                                      dv.stream_of(self.device)))
        return out

    def outer(self, y, g, sqrt=False, scale=1., out=None):
        """y : (..., n), g : (..., nz) device tensors with the same leading shape -> (..., nq, nz) = f(scale x (W y)[..., q] x g[..., z]), f = sqrt or
        identity, written once by the kernel that interpolates (``cp_spline_apply_outer``)."""
        torch = dv.torch()
        y, g = dv.to_device(y, self.device), dv.to_device(g, self.device)
        if y.shape[-1] != self.n or tuple(y.shape[:-1]) != tuple(g.shape[:-1]):
            raise ValueError('need y (..., {:d}) and g (..., nz) with the same leading shape, got {} and {}'.format(self.n, tuple(y.shape), tuple(g.shape)))
        lead, nz = tuple(y.shape[:-1]), int(g.shape[-1])
        nrows = int(np.prod(lead, dtype=np.int64))
        if out is None:
            out = torch.empty(lead + (self.nq, nz), dtype=torch.float64, device=self.device)
        elif tuple(out.shape) != lead + (self.nq, nz) or not out.is_contiguous() or out.dtype != torch.float64:
            raise ValueError('out must be a contiguous float64 tensor of shape {}'.format(lead + (self.nq, nz)))
        if nrows and nz:
            _lib.check(_lib.load().cp_spline_apply_outer(self._handle, y.data_ptr(), g.data_ptr(), nz, out.data_ptr(), nrows, int(bool(sqrt)), float(scale),
                                                         dv.stream_of(self.device)))
        return out

    def __del__(self):
        try:
            if self._handle:
                _lib.load().cp_spline_plan_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


def dense_operator(x, xq, bc='natural', nu=0, extrapolate=False):
    """The dense (nq, n) spline operator on the host (numpy); rows of out-of-range queries are NaN unless ``extrapolate``."""
    x = np.ascontiguousarray(x, dtype='f8').ravel()
    xq = np.ascontiguousarray(xq, dtype='f8').ravel()
    if xq.size * x.size > (1 << 29):      # 4 GB of weights: a catalogue / mesh of queries is evaluated point by point or in pieces, never as one matrix
        raise MemoryError('a dense spline operator of {:d} queries x {:d} knots'.format(xq.size, x.size))
    w = np.empty((xq.size, x.size))
    _lib.check(_lib.load().cp_spline_operator(x.size, _lib.as_double_p(x), xq.size, _lib.as_double_p(xq), _lib.SPLINE_BC[bc], int(nu),
                                              int(bool(extrapolate)), _lib.as_double_p(w), None))
    return w


class SplicedClampedSpline(object):

    """Clamped cubic spline through knots whose values are contiguous pieces of the rows of two arrays, evaluated at fixed queries: the
    tridiagonal system of ``scipy.interpolate.CubicSpline(knots, values, bc_type='clamped')`` solved for every row in LDS (``cp_splice_*``),
    with the wiggle damping of wallish2018 as an optional last step of the same kernel (reference bao_filter.py:415-431).

    pieces : up to three (source, start, count): knots take the columns [start, start + count) of the rows of array ``source`` (0 or 1).
    Raises NotImplementedError when the knots do not fit the kernel's scheme (the caller applies the spline as operators then)."""

    def __init__(self, knots, pieces, xq, device=None):
        self.device = dv.resolve_device(device)
        knots = np.ascontiguousarray(knots, dtype='f8').ravel()
        xq = np.ascontiguousarray(xq, dtype='f8').ravel()
        src, start, count = (np.ascontiguousarray([p[i] for p in pieces], dtype=np.int32) for i in range(3))
        as_int_p = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
        self._handle = ctypes.c_void_p()
        _lib.check(_lib.load().cp_splice_plan_create(ctypes.byref(self._handle), knots.size, _lib.as_double_p(knots), len(pieces), as_int_p(src), as_int_p(start),
                                                     as_int_p(count), xq.size, _lib.as_double_p(xq), self.device.index))
        self.nq = xq.size

    @property
    def scheme(self):
        """0: elimination in LDS; 1: recursions on a uniform stretch of knots (the default where the knots and queries fit it)."""
        return int(_lib.load().cp_splice_plan_scheme(self._handle))

    @scheme.setter
    def scheme(self, scheme):
        _lib.check(_lib.load().cp_splice_plan_set_scheme(self._handle, int(scheme)))

    def __call__(self, rows0, rows1=None, tophat=None):
        """rows0 (nrows, n0), rows1 (nrows, n1) device tensors -> (nrows, nq); ``tophat`` (nq,): rows0 / ((rows0 / spline - 1) tophat + 1)."""
        torch = dv.torch()
        rows0 = dv.to_device(rows0, self.device).contiguous()
        if rows1 is not None:
            rows1 = dv.to_device(rows1, self.device).contiguous()
            if rows1.shape[0] != rows0.shape[0]:
                raise ValueError('the two arrays must hold the same rows')
        out = torch.empty((rows0.shape[0], self.nq), dtype=torch.float64, device=self.device)
        _lib.check(_lib.load().cp_splice_apply(self._handle, rows0.data_ptr(), rows0.shape[1], rows1.data_ptr() if rows1 is not None else None,
                                               rows1.shape[1] if rows1 is not None else 0, rows0.shape[0], tophat.data_ptr() if tophat is not None else None,
                                               out.data_ptr(), dv.stream_of(self.device)))
        return out

    def __del__(self):
        try:
            if self._handle:
                _lib.load().cp_splice_plan_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class SplineRows(object):

    """Natural, clamped or not-a-knot cubic spline through fixed knots ``x`` for very many rows, evaluated at fixed queries ``xq``, by elimination in LDS
    (``cp_spline_rows_*``): the function ``LinearOperator.spline(x, xq, bc=...)`` applies as a banded operator, at a tenth of the arithmetic and
    reading only the knots the queries can see.  Raises NotImplementedError where the scheme does not fit (very long windows): use the operator then."""

    def __init__(self, x, xq, bc='natural', extrapolate=False, device=None):
        self.device = dv.resolve_device(device)
        x = np.ascontiguousarray(x, dtype='f8').ravel()
        xq = np.ascontiguousarray(xq, dtype='f8').ravel()
        self._handle = ctypes.c_void_p()
        _lib.check(_lib.load().cp_spline_rows_plan_create(ctypes.byref(self._handle), x.size, _lib.as_double_p(x), _lib.SPLINE_BC[bc], int(bool(extrapolate)), xq.size,
                                                          _lib.as_double_p(xq), self.device.index))
        self.n, self.nq = x.size, xq.size

    @property
    def window(self):
        """(first knot, number of knots, rows per wave, halo) of the plan."""
        vals = [ctypes.c_int() for _ in range(4)]
        _lib.check(_lib.load().cp_spline_rows_plan_info(self._handle, *[ctypes.byref(v) for v in vals]))
        return tuple(v.value for v in vals)

    def __call__(self, y, sqrt=False, scale=1., last_axis_first=False):
        """y (..., n) device tensor -> (..., nq); ``last_axis_first``: y (..., m, n) -> (..., nq, m), the transposition being part of the store."""
        torch = dv.torch()
        y = dv.to_device(y, self.device).contiguous()
        if y.shape[-1] != self.n:
            raise ValueError('last dimension must be {:d}, got {}'.format(self.n, tuple(y.shape)))
        lead = tuple(y.shape[:-1])
        nrows = int(np.prod(lead, dtype=np.int64))
        group = 0
        if last_axis_first:
            if len(lead) < 1:
                raise ValueError('last_axis_first needs rows of at least two dimensions')
            group = int(lead[-1])
            shape = lead[:-1] + (self.nq, group)
        else:
            shape = lead + (self.nq,)
        out = torch.empty(shape, dtype=torch.float64, device=self.device)
        if nrows:
            _lib.check(_lib.load().cp_spline_rows_apply(self._handle, y.data_ptr(), nrows, int(bool(sqrt)), float(scale), group, out.data_ptr(),
                                                        dv.stream_of(self.device)))
        return out

    def second_derivatives(self, y, pairs=False):
        """y (..., n) -> (..., n): the second derivatives of the spline at its knots (needs queries that span the knots);
        pairs=True: (..., n, 2), the knot values with their second derivatives (y_j, M_j) -- what ``cp_tables_rows_direct`` reads fastest."""
        torch = dv.torch()
        y = dv.to_device(y, self.device).contiguous()
        if y.shape[-1] != self.n:
            raise ValueError('last dimension must be {:d}, got {}'.format(self.n, tuple(y.shape)))
        out = torch.empty(tuple(y.shape) + (2,), dtype=torch.float64, device=self.device) if pairs else torch.empty_like(y)
        nrows = y.numel() // self.n
        if nrows:
            fun = _lib.load().cp_spline_rows_pairs if pairs else _lib.load().cp_spline_rows_second_derivatives
            _lib.check(fun(self._handle, y.data_ptr(), nrows, out.data_ptr(), dv.stream_of(self.device)))
        return out

    def __del__(self):
        try:
            if self._handle:
                _lib.load().cp_spline_rows_plan_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
