"""
Eisenstein & Hu 1998 engine on MI355X (reference cosmoprimo/eisenstein_hu.py): fit coefficients, transfer function,
primordial spectrum, P(k, z) and the sigma8 normalisation, all evaluated by the HIP kernels of ``cp_power.hip`` /
``cp_fftlog*.hip`` / ``cp_spline.hip`` for one cosmology or a batch.
"""

import numpy as np

from . import _device as dv
from . import power as pwmod, utils
from .cosmology import BaseEngine, BaseSection, DefaultBackground, _out
from .interpolator import PowerSpectrumInterpolator1D, PowerSpectrumInterpolator2D, sigma_r2_of_rows, _host


class EisensteinHuEngine(BaseEngine):

    """Eisenstein & Hu analytic formulae, https://arxiv.org/abs/astro-ph/9709112 (reference eisenstein_hu.py:11-103)."""
    name = 'eisenstein_hu'
    _transfer = 'eisenstein_hu'

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # (no warning for massive neutrinos, curvature or a dark-energy fluid: the reference's are commented out, eisenstein_hu.py:24-32 -- the
        # species enter through the background the sections read: Omega0_m, Omega_m(z), Omega_de(z))
        self.compute()
        self._A_s = self._get_A_s_fid()

    def compute(self):
        """Fit coefficients (eisenstein_hu.py:34-92) as attributes: rs_drag [Mpc], z_drag, k_eq, alpha_c, ... (floats or (B,) tensors)."""
        sc = pwmod.eh_scalars(self.bg_params(), device=self.device)
        for name, v in sc.items():
            setattr(self, name, float(v) if v.ndim == 0 else v)

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


class Background(DefaultBackground):
    """Background quantities with closed-form growth (reference eisenstein_hu.py:106-152)."""

    def growth_factor(self, z, znorm=None):
        """CPT92 approximation of the growth factor (eisenstein_hu.py:115-140)."""
        growthz = self._eval('growth_cpt', z)
        if znorm is not None:
            return (1. + znorm) * growthz
        g0 = self._eval('growth_cpt', np.zeros(()))
        if np.ndim(g0) and np.ndim(growthz) > np.ndim(g0):
            g0 = g0.reshape(g0.shape + (1,) * (np.ndim(growthz) - np.ndim(g0)))
        return growthz / g0

    def growth_rate(self, z):
        """Approximation of the growth rate Omega_m(z)^(0.55 + 0.05 (1 + w(z=1))) (eisenstein_hu.py:143-152)."""
        return self._eval('growth_rate', z)


@utils.addproperty('rs_drag', 'z_drag')
class Thermodynamics(BaseSection):

    """rs_drag [Mpc/h] and z_drag (reference eisenstein_hu.py:155-162)."""

    def __init__(self, engine):
        super().__init__(engine)
        self._rs_drag = engine.rs_drag * engine['h'] if not dv.is_torch(engine.rs_drag) else engine.rs_drag * dv.to_device(engine['h'], engine.device)
        self._z_drag = engine.z_drag


@utils.addproperty('k_pivot', 'n_s', 'alpha_s', 'beta_s')
class Primordial(BaseSection):

    """Primordial power spectrum (reference eisenstein_hu.py:165-230)."""

    def __init__(self, engine):
        super().__init__(engine)
        self._rsigma8 = engine._rescale_sigma8()
        self._n_s, self._alpha_s, self._beta_s = engine['n_s'], engine['alpha_s'], engine['beta_s']
        self._k_pivot = engine['k_pivot'] / self._h

    @property
    def A_s(self):
        r"""Scalar amplitude of the primordial power spectrum at :math:`k_\mathrm{pivot}`, unitless."""
        return self._engine._A_s * self._rsigma8**2

    @property
    def ln_1e10_A_s(self):
        return np.log(1e10 * self.A_s) if not dv.is_torch(self.A_s) else dv.torch().log(1e10 * self.A_s)

    def pk_k(self, k, mode='scalar'):
        r"""Primordial spectrum of curvature perturbations at ``k`` [h/Mpc], in (Mpc/h)^3 (eisenstein_hu.py:189-215)."""
        ['scalar'].index(mode)
        kh = _host(k)
        out = pwmod.analytic(getattr(self._engine, '_transfer', 'eisenstein_hu'), 'primordial', kh.ravel(), bg=self._engine.bg_params(), pk=self._engine.pk_params(), device=self.device)
        return _out(out.reshape(out.shape[:-1] + kh.shape), k)

    def pk_interpolator(self, mode='scalar'):
        return PowerSpectrumInterpolator1D.from_callable(pk_callable=lambda k: self.pk_k(dv.upload(k, self.device), mode=mode).T
                                                         if self._engine.batch_size else self.pk_k(dv.upload(k, self.device), mode=mode),
                                                         device=self.device)


class Transfer(BaseSection):

    """Matter transfer function (reference eisenstein_hu.py:233-283)."""

    def transfer_k(self, k):
        kh = _host(k)
        out = pwmod.analytic(self._engine._transfer, 'transfer', kh.ravel(), bg=self._engine.bg_params(), pk=self._engine.pk_params(rsigma8=1.),
                             device=self.device)
        return _out(out.reshape(out.shape[:-1] + kh.shape), k)


class Fourier(BaseSection):

    """Matter power spectrum (reference eisenstein_hu.py:286-342)."""

    def __init__(self, engine):
        super().__init__(engine)
        self.pm = engine.get_primordial()   # triggers the sigma8 normalisation, as in the reference
        self.tr = engine.get_transfer()
        self.ba = engine.get_background()

    def _pk0_device(self, kh, kscale=None):
        """P(k, z) without growth: device tensor (batch..., nk); ``kscale``: per-cosmology factors applied to k.

        A batch of cosmologies keeps the last spectra it evaluated at its FIDUCIAL amplitude (the sigma8 normalisation evaluates them on the 1024
        wavenumbers every filter and every sigma integral asks for next): P is linear in A_s, so the normalised spectra on the same
        wavenumbers are those rows times (sigma8 / sigma8_fid)^2 -- one multiplication per sample instead of an EH98 evaluation (~390 fp64
        instructions).  Held by the engine, released with it."""
        e = self._engine
        rs = e._rsigma8
        if kscale is None and e.batch_size is not None and not dv.is_torch(kh) and np.size(kh) <= 2048 and (rs is None or dv.is_torch(rs) or rs == 1.):
            kh = np.asarray(kh, dtype='f8')
            key = (kh.shape, kh.tobytes())
            ready = e.__dict__.get('_pk0_normalised')      # left by the sigma8 normalisation kernel: these wavenumbers at the normalised amplitude
            if ready is not None and ready[0] == key and ready[1] is rs:
                return ready[2]
            held = e.__dict__.get('_pk0_fiducial')
            if held is None or held[0] != key:
                unit = pwmod.analytic(e._transfer, 'matter', kh, bg=e.bg_params(), pk=e.pk_params(rsigma8=1.), device=self.device)
                held = e.__dict__['_pk0_fiducial'] = (key, unit)
            if rs is None or not dv.is_torch(rs):
                return held[1]
            return held[1] * (rs**2).reshape(-1, 1)
        return pwmod.analytic(e._transfer, 'matter', kh, bg=e.bg_params(), pk=e.pk_params(), device=self.device, kscale=kscale)

    def pk_interpolator(self, of='delta_m', **kwargs):
        """:class:`PowerSpectrumInterpolator2D` of the pair ``of`` ('delta_m', 'theta_m'), built from callables (eisenstein_hu.py:295-329)."""
        if isinstance(of, str):
            of = (of,)
        of = list(of)
        of = of + [of[0]] * (2 - len(of))
        ntheta = sum(of_.startswith('theta_') for of_ in of)
        ba, device = self.ba, self.device

        growth_cache = {}      # host redshift grid (bytes) -> device result: the cosmology behind this closure does not change

        def growth_factor_sq(z):
            key = None
            if not dv.is_torch(z) and np.size(z) <= 4096:
                key = (np.shape(z), np.asarray(z, dtype='f8').tobytes())
                if key in growth_cache:
                    return growth_cache[key]
            zt = dv.to_device(z, device)
            g = dv.to_device(ba.growth_factor(zt, znorm=0.), device)**2
            if ntheta:
                g = g * dv.to_device(ba.growth_rate(zt), device)**ntheta
            if key is not None:
                if len(growth_cache) >= 8:
                    growth_cache.clear()
                growth_cache[key] = g
            return g

        def pk_callable(k):
            return self._pk0_device(k if dv.is_torch(k) else np.asarray(k, dtype='f8'))

        # what the one-call sigma(r, z) pipeline of the library needs to evaluate the same spectra itself (cp_sigma_rz_analytic)
        pk_callable.analytic_engine = lambda: (self._engine._transfer, self._engine.bg_params(), self._engine.pk_params())

        interp = PowerSpectrumInterpolator2D.from_callable(pk_callable=pk_callable, growth_factor_sq=growth_factor_sq, device=device, **kwargs)
        # batched cosmologies: P_c(k * kscale_c) in one launch (used by the brieden2022 filter, one rs_drag ratio per cosmology)
        def pk_scaled(k, kscale):
            rows = self._pk0_device(np.asarray(k, dtype='f8'), kscale=kscale)
            rs = interp._rsigma8sq
            return rows if isinstance(rs, float) and rs == 1. else rows * rs      # (no pass over the batch for a factor of one)

        interp._pk_scaled = pk_scaled
        return interp

    def sigma_rz(self, r, z, of='delta_m', **kwargs):
        r"""R.m.s. of `of` perturbations in spheres of :math:`r` Mpc/h."""
        return self.pk_interpolator(of=of, **kwargs).sigma_rz(r, z)

    def sigma8_z(self, z, of='delta_m'):
        r"""R.m.s. of `of` perturbations in spheres of 8 Mpc/h."""
        return self.sigma_rz(8., z, of=of)

    @property
    def sigma8_m(self):
        r"""Current r.m.s. of matter perturbations in a sphere of 8 Mpc/h, unitless."""
        out = self.sigma8_z(0., of='delta_m')
        return out

    def _sigma8_m_device(self):
        """sigma8 of the current normalisation as a device tensor (batch...,): P(k) -> TophatVariance FFTLog -> natural spline at r = 8."""
        g0 = dv.to_device(self.ba.growth_factor(dv.torch().zeros((), dtype=dv.torch().float64, device=self.device), znorm=0.), self.device)**2   # device z: no trip to the host

        def rows(kh):
            p0 = self._pk0_device(kh)
            return p0 * (g0[..., None] if g0.ndim else g0)

        e = self._engine
        rs = e._rsigma8
        if e.batch_size is not None and g0.ndim == 1 and (rs is None or (not dv.is_torch(rs) and rs == 1.)):
            # a batch at its fiducial amplitude (the sigma8 normalisation, eisenstein_hu.py:94-103): ONE kernel evaluates the spectra, transforms them
            # and reads sigma8 off -- and leaves the spectra where _pk0_device looks for them
            from .interpolator import sigma_rz_analytic
            res = sigma_rz_analytic(e._transfer, e.bg_params(), e.pk_params(rsigma8=1.), 8., g0[:, None], self.device, keep_spectra=True)
            if res is not None:
                out, spectra, k = res
                e.__dict__['_pk0_fiducial'] = ((k.shape, k.tobytes()), spectra)
                return out[:, 0, 0]
        return sigma_r2_of_rows(8., rows, kmin=1e-7, kmax=1e2, device=self.device)[..., 0]**0.5
