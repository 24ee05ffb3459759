"""
Utilities with the reference's names (cosmoprimo/utils.py): :class:`LeastSquareSolver` (:144-272) and
:class:`DistanceToRedshift` (:275-316).

``LeastSquareSolver`` is host-side numpy: it acts on <= 10 parameters x <= 1024 samples once per filter (SURVEY.md 8(a) a18:
"negligible"); the BAO filters turn its solution into a dense operator that is applied to all columns on the device
(:func:`cosmoprimo_amd.bao_filter._constrained_lsq_operator`).  ``DistanceToRedshift`` is a device spline (:class:`Interpolator1D`).
"""
import functools
import inspect
import logging
import os

import numpy as np

from ._device import Copyable as _Copyable, float_dtype as _float_dtype, is_torch as _is_torch
from .interpolator import Interpolator1D


logger = logging.getLogger('Utils')


def mkdir(dirname):
    """Try to create ``dirname`` and catch :class:`OSError` (reference utils.py:13-18)."""
    try:
        os.makedirs(dirname)
    except OSError:
        return


def savefig(filename, fig=None, bbox_inches='tight', pad_inches=0.1, dpi=200, **kwargs):
    """Save figure ``fig`` (default: the current one) to ``filename``, creating its directory; ``kwargs`` go to
    :meth:`matplotlib.figure.Figure.savefig` (reference utils.py:322-351).  Returns the figure."""
    from matplotlib import pyplot as plt
    mkdir(os.path.dirname(filename))
    logger.info('Saving figure to {}.'.format(filename))
    if fig is None:
        fig = plt.gcf()
    fig.savefig(filename, bbox_inches=bbox_inches, pad_inches=pad_inches, dpi=dpi, **kwargs)
    return fig


class BaseClass(_Copyable):
    """Base class with the reference's ``copy()`` / ``__copy__`` (utils.py:51-64): a shallow copy."""

    def __copy__(self):
        other = self.__class__.__new__(self.__class__)
        other.__dict__.update(self.__dict__)
        return other

    def copy(self):
        return self.__copy__()


def addproperty(*attrs):
    """Class decorator of the reference (utils.py:67-87): read-only properties ``attr`` for values stored as ``_attr``."""
    def decorate(cls):
        for attr in attrs:
            setattr(cls, attr, property(functools.partial(lambda self, name: getattr(self, '_' + name), name=attr)))
        return cls
    return decorate


def flatarray(iargs=[0], dtype=np.float64):
    """Method decorator of the reference (utils.py:98-138): the array arguments ``iargs`` (positions behind ``self``) reach the method flattened --
    they must share one shape --, and its result (an array or a dict of arrays, last axis = the flat one) is given that shape back, float32 only
    if every such input was float32.  The sections of this package do the same inside their kernels' wrappers; this is for code written
    against the reference's helper."""
    positions = tuple(iargs)

    def decorate(method):
        signature = inspect.signature(method)

        @functools.wraps(method)
        def wrapper(*args, **kwargs):
            bound = signature.bind_partial(*args, **kwargs)
            bound.apply_defaults()
            obj, rest = bound.args[0], list(bound.args[1:])
            out_dtype = _float_dtype(*[rest[i] for i in positions])
            in_dtype = out_dtype if dtype is None else dtype
            shape = None
            for i in positions:
                value = rest[i]
                value = value.detach().cpu().numpy() if _is_torch(value) else value
                flat = np.asarray(value, dtype=in_dtype)
                if shape is None:
                    shape = flat.shape
                elif flat.shape != shape:
                    raise ValueError('input arrays must have same shape, found {}, {}'.format(shape, flat.shape))
                rest[i] = flat.ravel()
            result = method(obj, *rest, **bound.kwargs)

            def restore(value):
                value = np.asarray(value, dtype=out_dtype)
                return value.reshape(value.shape[:-1] + shape)

            if isinstance(result, dict):
                return {key: restore(value) for key, value in result.items()}
            return restore(result)

        return wrapper

    return decorate


class LeastSquareSolver(_Copyable):
    r"""
    Weighted linear least squares, optionally under linear equality constraints: the coefficients :math:`\mathbf{p}` that minimise
    :math:`\chi^2 = (\delta - \mathbf{p} G)^T F (\delta - \mathbf{p} G)` subject to :math:`\mathbf{p} C = c`
    (same constructor, call, ``model`` and ``chi2`` as the reference's class, utils.py:144-272).

    The stationarity conditions with Lagrange multipliers are one linear system, the Karush-Kuhn-Tucker matrix
    :math:`K = \begin{pmatrix} G F G^T & -C \\ C^T & 0 \end{pmatrix}` acting on (coefficients, multipliers) with right-hand side
    :math:`(G F \delta, c)`; the data enter linearly, so the solve is a fixed matrix applied to (data, constraint values), kept as
    ``projector`` when ``compute_inverse`` is set (many data vectors) and redone with ``numpy.linalg.solve`` per call otherwise.
    """
    def __init__(self, gradient, precision=1., constraint_gradient=None, compute_inverse=True):
        basis = np.asarray(gradient, dtype='f8')
        if basis.ndim > 2:
            raise ValueError('gradient must be at most 2D')
        self.isscalar = basis.ndim < 2          # one template: the coefficient is returned without its axis
        self.gradient = basis.reshape(-1, basis.shape[-1]) if basis.ndim else basis.reshape(1, 1)
        nparams, ndata = self.gradient.shape
        self.precision = np.asarray(precision, dtype='f8')
        # G F: a full precision matrix multiplies from the right, a scalar or a diagonal scales the columns
        weighted = self.gradient.dot(self.precision) if self.precision.ndim == 2 else self.gradient * self.precision
        normal = weighted.dot(self.gradient.T)
        self.nconstraints = 0
        if constraint_gradient is not None:
            columns = np.atleast_2d(np.asarray(constraint_gradient, dtype='f8'))
            if columns.ndim != 2 or columns.shape[0] != nparams:
                raise ValueError('constraint_gradient must be 2D, of first dimension the number of model parameters (gradient first dimension)')
            nc = self.nconstraints = columns.shape[1]
            kkt = np.zeros((nparams + nc,) * 2)
            kkt[:nparams, :nparams], kkt[:nparams, nparams:], kkt[nparams:, :nparams] = normal, -columns, columns.T
            load = np.zeros((nparams + nc, ndata + nc))      # (data, constraint values) -> right-hand side
            load[:nparams, :ndata] = weighted
            load[nparams:, ndata:] = np.eye(nc)
            normal, weighted = kkt, load
        self.inverse_fisher, self.gradient_precision = normal, weighted
        if compute_inverse:
            inverse = np.linalg.inv(normal)
            defect = np.abs(inverse.dot(normal) - np.eye(normal.shape[0])).max()
            if not defect <= 1e-4:
                import warnings
                warnings.warn('Numerically inaccurate inverse matrix, max absolute diff {:.6f}.'.format(defect))
            self.projector = inverse.dot(weighted).T

    def compute(self, delta, constraint=None):
        """Solve for the data vector(s) ``delta`` (..., ndata), with the constraint values ``constraint`` if the solver has constraints."""
        self.delta = np.atleast_1d(np.asarray(delta, dtype='f8'))
        loaded = self.delta
        if constraint is not None:
            values = np.atleast_1d(np.asarray(constraint, dtype='f8'))
            loaded = np.concatenate([loaded, np.broadcast_to(values, loaded.shape[:-1] + values.shape[-1:])], axis=-1)
        if 'projector' in self.__dict__:
            solution = loaded.dot(self.projector)
        else:
            solution = np.linalg.solve(self.inverse_fisher, self.gradient_precision.dot(loaded.T)).T
        self.params = solution[..., :self.gradient.shape[0]]      # (the multipliers behind them are not kept)

    def __call__(self, delta, constraint=None):
        """Best-fit coefficients, (..., nparams); (...) for a single template."""
        self.compute(delta, constraint=constraint)
        return self.params[..., 0] if self.isscalar else self.params

    def model(self):
        """The fitted model, (..., ndata)."""
        return self.params.dot(self.gradient)

    def chi2(self):
        r""":math:`\chi^2` of the fit, (...)."""
        residual = self.delta - self.model()
        weighted = residual.dot(self.precision) if self.precision.ndim == 2 else residual * self.precision
        return (weighted * residual).sum(axis=-1)


class DistanceToRedshift(_Copyable):

    """Distance -> redshift conversion by spline interpolation of a tabulated redshift -> distance relation (reference utils.py:275-316)."""

    def __init__(self, distance, zmax=100., nz=512, interp_order=3, device=None):
        zgrid = 1. / np.geomspace(1. / (1. + zmax), 1., nz)[::-1] - 1.
        rgrid = np.asarray(distance(zgrid), dtype='f8')
        self._interp = Interpolator1D(rgrid, zgrid, k=interp_order, device=device)

    def __call__(self, distance, bounds_error=True):
        """(Interpolated) redshift at ``distance`` (scalar or array)."""
        return self._interp(distance, bounds_error=bounds_error)
