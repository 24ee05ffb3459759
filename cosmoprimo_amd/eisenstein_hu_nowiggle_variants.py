"""
Eisenstein & Hu 1997 (astro-ph/9710252) analytic formulae with massive neutrinos and scale-dependent growth on MI355X
(reference cosmoprimo/eisenstein_hu_nowiggle_variants.py).  The only analytic engine that copes with massive species: the transfer
function depends on (k, z), so P(k, z) comes from one kernel (``cp_power_eval_variants``) rather than from P(k) x growth(z).
"""
import warnings

import numpy as np

from . import _device as dv
from . import power as pwmod
from .cosmology import BaseEngine, BaseSection, CosmologyError, _out
from .eisenstein_hu import Background, Thermodynamics, Primordial  # noqa: F401  (sections discovered by name)
from .eisenstein_hu import Fourier as EHFourier
from .interpolator import PowerSpectrumInterpolator2D, _host, sigma_r2_of_rows


class EisensteinHuNoWiggleVariantsEngine(BaseEngine):

    """Eisenstein & Hu & variants analytic formulae (reference eisenstein_hu_nowiggle_variants.py:13-76)."""
    name = 'eisenstein_hu_nowiggle_variants'
    _copes_with_ncdm = True

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.batch_size is None and self._has_fld:
            warnings.warn('{} cannot cope with non-constant dark energy'.format(self.__class__.__name__))
        self.compute()
        self._A_s = self._get_A_s_fid()

    def compute(self):
        """The engine's attributes -- densities and fractions, z_eq, k_eq, z_drag, rs_drag [Mpc], p_c, p_cb, gamma_ncdm, beta_c (what the
        reference's ``_set_rsdrag`` and ``compute`` leave on the engine, eisenstein_hu_nowiggle_variants.py:32-76) -- read back from
        ``cp_variants_scalars``, i.e. from the device function the evaluation kernel takes them from: floats for one cosmology, (B,) numpy arrays
        for a batch."""
        scalars = pwmod.variants_scalars(self.bg_params(), ncdm=self.get_background()._ncdm, device=self.device)
        table = dv.torch().stack([scalars[name].reshape(-1) for name in scalars], dim=0).cpu().numpy()      # one copy for all of them
        for name, row in zip(scalars, table):
            setattr(self, name, float(row[0]) if self.batch_size is None else row.copy())
        self.N_ncdm = self['N_ncdm']

    def pk_params(self, rsigma8=None):
        rs = self._rsigma8 if rsigma8 is None else rsigma8
        if rs is None:
            rs = 1.
        if dv.is_torch(rs):      # the normalised amplitudes of a batch: one product per normalisation, not one per evaluation of P(k)
            cached = self.__dict__.get('_A_s_normalised')
            if cached is None or cached[0] is not rs:
                cached = self.__dict__['_A_s_normalised'] = (rs, dv.to_device(self._A_s, self.device) * rs**2)
            A_s = cached[1]
        else:
            A_s = self._A_s * rs**2
        return dict(A_s=A_s, n_s=self['n_s'], alpha_s=self['alpha_s'], beta_s=self['beta_s'], k_pivot=self['k_pivot'])


class Transfer(BaseSection):

    """Matter transfer function T(k, z) (reference eisenstein_hu_nowiggle_variants.py:79-154)."""

    def __init__(self, engine):
        super().__init__(engine)
        self.ba = engine.get_background()

    def _device(self, what, kh, zh, of, rsigma8=None):
        if of not in ('delta_m', 'delta_cb'):
            raise CosmologyError('No {} transfer function can be computed (choices are ["delta_cb", "delta_m"]).'.format(of))
        e = self._engine
        return pwmod.variants(what, kh, zh, of=of, bg=e.bg_params(), pk=e.pk_params(rsigma8=rsigma8), ncdm=self.ba._ncdm, device=self.device)

    def transfer_kz(self, k, z=0., of='delta_m', grid=True):
        """Transfer function of 'delta_m' or 'delta_cb' at ``k`` [h/Mpc] and ``z``: k.shape + z.shape (``grid``) or k.shape (pairs)."""
        kh, zh = _host(k), _host(z)
        out = self._device('transfer', kh.ravel(), zh.ravel(), of, rsigma8=1.)      # (batch..., nz, nk)
        out = out.transpose(-1, -2)
        if grid:
            return _out(out.reshape(out.shape[:-2] + kh.shape + zh.shape), k)
        if kh.shape != zh.shape:
            raise ValueError('k and z must have the same shape with grid=False')
        return _out(dv.torch().diagonal(out, dim1=-2, dim2=-1).reshape(out.shape[:-2] + kh.shape), k)


class Fourier(EHFourier):

    """Matter power spectrum (reference eisenstein_hu_nowiggle_variants.py:157-193)."""

    def _pk_device(self, kh, zh, of):
        """P(k, z) of the pair ``of`` (delta_m / delta_cb) as a device tensor (batch..., nk, nz), growth included."""
        if of[0] == of[1]:
            return self.tr._device('matter', kh, zh, of[0]).transpose(-1, -2)
        # cross spectrum: T_a T_b instead of T_a^2 (reference :178-181)
        pa = self.tr._device('matter', kh, zh, of[0]).transpose(-1, -2)
        ta = self.tr._device('transfer', kh, zh, of[0], rsigma8=1.).transpose(-1, -2)
        tb = self.tr._device('transfer', kh, zh, of[1], rsigma8=1.).transpose(-1, -2)
        return pa * tb / ta

    def pk_interpolator(self, of='delta_m', **kwargs):
        """:class:`PowerSpectrumInterpolator2D` of the pair ``of`` among 'delta_m', 'delta_cb', 'theta_m', 'theta_cb' (reference :159-193)."""
        if not isinstance(of, (tuple, list)):
            of = (of, of)
        ntheta = sum(of_.startswith('theta_') for of_ in of)
        of = tuple(of_.replace('theta_', 'delta_') for of_ in of)
        ba, device = self.ba, self.device

        def pk_callable(k, z, grid=True):
            kh, zh = np.asarray(k, dtype='f8').ravel(), np.asarray(z, dtype='f8').ravel()
            out = self._pk_device(kh, zh, of)
            if ntheta:
                out = out * dv.to_device(ba.growth_rate(dv.to_device(zh, device)), device)[..., None, :]**ntheta
            if not grid:
                out = dv.torch().diagonal(out, dim1=-2, dim2=-1)
            return out

        return PowerSpectrumInterpolator2D.from_callable(pk_callable=pk_callable, growth_factor_sq=None, device=device, **kwargs)

    def _sigma8_m_device(self):
        """sigma8 of the current normalisation as a device tensor: P(k, z=0) -> TophatVariance FFTLog -> natural spline at r = 8."""
        def rows(kh):
            return self._pk_device(kh, np.zeros(1), ('delta_m', 'delta_m'))[..., 0]
        return sigma_r2_of_rows(8., rows, kmin=1e-7, kmax=1e2, device=self.device)[..., 0]**0.5
