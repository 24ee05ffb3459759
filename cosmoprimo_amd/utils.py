"""
Utilities with the reference's names (cosmoprimo/utils.py): :class:`LeastSquareSolver` (:144-272) and
:class:`DistanceToRedshift` (:275-316).

``LeastSquareSolver`` is host-side numpy: it acts on <= 10 parameters x <= 1024 samples once per filter (SURVEY.md 8(a) a18:
"negligible"); the BAO filters turn its solution into a dense operator that is applied to all columns on the device
(:func:`cosmoprimo_amd.bao_filter._constrained_lsq_operator`).  ``DistanceToRedshift`` is a device spline (:class:`Interpolator1D`).
"""
import numpy as np

from ._device import Copyable as _Copyable
from .interpolator import Interpolator1D


class LeastSquareSolver(_Copyable):
    r"""
    Solve :math:`d\chi^2 / d\mathbf{p} = 0` for
    :math:`\chi^2 = (\delta - \mathbf{p} \cdot \mathrm{grad})^T \mathbf{F} (\delta - \mathbf{p} \cdot \mathrm{grad})`, optionally
    under linear equality constraints :math:`\mathbf{p} \cdot \mathrm{cgrad} = c` (reference utils.py:144-272).
    """
    def __init__(self, gradient, precision=1., constraint_gradient=None, compute_inverse=True):
        self.gradient = np.atleast_1d(np.asarray(gradient, dtype='f8'))
        self.isscalar = self.gradient.ndim == 1
        if self.isscalar:
            self.gradient = self.gradient[None, :]
        elif self.gradient.ndim != 2:
            raise ValueError('gradient must be at most 2D')
        self.precision = np.asarray(precision, dtype='f8')
        hv = self.gradient * self.precision if self.precision.ndim < 2 else self.gradient.dot(self.precision)
        invfisher = hv.dot(self.gradient.T)
        if constraint_gradient is None:
            self.nconstraints = 0
        else:
            cg = np.atleast_2d(np.asarray(constraint_gradient, dtype='f8'))
            self.nconstraints = cg.shape[-1]
            if cg.ndim != 2 or cg.shape[0] != self.gradient.shape[0]:
                raise ValueError('constraint_gradient must be 2D, of first dimension the number of model parameters (gradient first dimension)')
            nc = self.nconstraints
            invfisher = np.block([[invfisher, -cg], [cg.T, np.zeros((nc, nc))]])     # bordered normal matrix (reference :211-214)
            hv = np.block([[hv, np.zeros(cg.shape)], [np.zeros((nc, self.gradient.shape[-1])), np.eye(nc)]])
        self.inverse_fisher = invfisher
        self.gradient_precision = hv
        if compute_inverse:
            fisher = np.linalg.inv(invfisher)
            tmp = fisher.dot(invfisher)
            if not np.allclose(tmp, np.eye(tmp.shape[0]), rtol=1e-04, atol=1e-04):
                import warnings
                warnings.warn('Numerically inaccurate inverse matrix, max absolute diff {:.6f}.'.format(np.max(np.abs(tmp - np.eye(tmp.shape[0])))))
            self.projector = fisher.dot(hv).T

    def compute(self, delta, constraint=None):
        """Solve the least-square problem for ``delta`` (..., ndata)."""
        self.delta = delta = np.atleast_1d(np.asarray(delta, dtype='f8'))
        if constraint is not None:
            constraint = np.atleast_1d(np.asarray(constraint, dtype='f8'))
            delta = np.concatenate([self.delta, np.broadcast_to(constraint, self.delta.shape[:-1] + constraint.shape[-1:])], axis=-1)
        if hasattr(self, 'projector'):
            params = delta.dot(self.projector)
        else:
            params = np.linalg.solve(self.inverse_fisher, self.gradient_precision.dot(delta.T)).T
        self.params = params[..., :self.gradient.shape[0]]

    def __call__(self, delta, constraint=None):
        """Best-fit parameters."""
        self.compute(delta, constraint=constraint)
        if self.isscalar:
            return self.params[..., 0]
        return self.params

    def model(self):
        """Model at the best fit."""
        return self.params.dot(self.gradient)

    def chi2(self):
        r""":math:`\chi^2` at the best fit."""
        delta = self.delta - self.model()
        if self.precision.ndim < 2:
            return ((delta * self.precision) * delta).sum(axis=-1)
        return (delta.dot(self.precision) * delta).sum(axis=-1)


class DistanceToRedshift(_Copyable):

    """Distance -> redshift conversion by spline interpolation of a tabulated redshift -> distance relation (reference utils.py:275-316)."""

    def __init__(self, distance, zmax=100., nz=512, interp_order=3, device=None):
        zgrid = 1. / np.geomspace(1. / (1. + zmax), 1., nz)[::-1] - 1.
        rgrid = np.asarray(distance(zgrid), dtype='f8')
        self._interp = Interpolator1D(rgrid, zgrid, k=interp_order, device=device)

    def __call__(self, distance, bounds_error=True):
        """(Interpolated) redshift at ``distance`` (scalar or array)."""
        return self._interp(distance, bounds_error=bounds_error)
