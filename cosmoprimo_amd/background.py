"""
Background E(z) and distances on MI355X for batches of cosmologies: the data-parallel core behind
``DefaultBackground`` (reference cosmoprimo/cosmology.py:1954-2042, 1855-1912, 1751-1759).

:func:`distance` is the batch entry point (one HIP thread per (cosmology, z) sample, ``cp_background_distance``);
the section class :class:`cosmoprimo_amd.cosmology.DefaultBackground` wraps it with the reference's method names.
"""
import ctypes

import numpy as np

from . import _lib
from . import _device as dv

KINDS = tuple(_lib.BG_KINDS)
DEFAULTS = dict(h=0.7, Omega_cdm=0.25, Omega_b=0.05, Omega_k=0., T_cmb=2.7255, N_ur=3.044, w0_fld=-1., wa_fld=0.)


def _is_torch(x):
    return type(x).__module__.startswith('torch')


def _cparam(v, device, keep):
    """cp_param of a float or a per-cosmology array / tensor (kept alive in ``keep``); returns (cp_param, length or None)."""
    import torch
    if np.ndim(v) == 0 and not _is_torch(v):
        return _lib.cp_param(None, float(v)), None
    t = v.to(device=device, dtype=torch.float64) if _is_torch(v) else dv.upload(np.asarray(v, dtype='f8'), device, cache=False)
    t = t.reshape(-1).contiguous()
    keep.append(t)
    return _lib.cp_param(t.data_ptr(), 0.), t.numel()


class NcdmTables(object):

    """
    Massive neutrinos for the background kernels (``cp_ncdm_tables``): per cosmology and species the natural cubic splines of the
    comoving density and pressure on the reference's 119 knots (DefaultBackground.rho_ncdm / p_ncdm caches, cosmology.py:1961-1998,
    built from _compute_ncdm_momenta :74-137 with numpy's 100-point Gauss-Laguerre rule).  ``m_ncdm`` [eV], ``T_ncdm_over_cmb``:
    one entry per species, each a float or an array of shape (ncosmo,), like ``h`` and ``T_cmb``.
    """
    def __init__(self, m_ncdm, T_ncdm_over_cmb, h=DEFAULTS['h'], T_cmb=DEFAULTS['T_cmb'], ncosmo=1, device=None):
        import torch
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        self.device = torch.device(device)
        self.nspecies, self.ncosmo = len(m_ncdm), int(ncosmo)
        if len(T_ncdm_over_cmb) != self.nspecies:
            raise TypeError('T_ncdm_over_cmb and m_ncdm must be of same length, found {:d} != {:d}'.format(len(T_ncdm_over_cmb), self.nspecies))
        self.tab = torch.zeros((self.ncosmo, self.nspecies, 4, _lib.NCDM_NKNOTS), dtype=torch.float64, device=self.device)
        if not self.nspecies:
            return
        keep = []
        ch, cT = _cparam(h, self.device, keep)[0], _cparam(T_cmb, self.device, keep)[0]
        cm = (_lib.cp_param * self.nspecies)(*[_cparam(v, self.device, keep)[0] for v in m_ncdm])
        ct = (_lib.cp_param * self.nspecies)(*[_cparam(v, self.device, keep)[0] for v in T_ncdm_over_cmb])
        for t in keep:
            if t.numel() != self.ncosmo:
                raise ValueError('per-cosmology arrays must have length ncosmo = {:d}, got {:d}'.format(self.ncosmo, t.numel()))
        nodes, weights = np.polynomial.laguerre.laggauss(100)
        nodes, weights = np.ascontiguousarray(nodes, dtype='f8'), np.ascontiguousarray(weights, dtype='f8')
        _lib.background_init(self.device.index)      # (the knot tables of the device: once, not inside an asynchronous entry point)
        _lib.check(_lib.load().cp_ncdm_tables(self.ncosmo, self.nspecies, ch, cT, ctypes.cast(cm, ctypes.c_void_p), ctypes.cast(ct, ctypes.c_void_p), 100,
                                              _lib.as_double_p(nodes), _lib.as_double_p(weights), self.tab.data_ptr(), self.device.index,
                                              torch.cuda.current_stream(self.device).cuda_stream))

    def struct(self, species=None):
        return _lib.cp_ncdm(self.nspecies, -1 if species is None else int(species), self.tab.data_ptr())


def growth_ode_tables(params=None, mass='m', ncdm=None, ncosmo=1, device=None):
    """Linear growth D and D'/D on the 201 knots of the reference's ODE solution (``cp_growth_ode_tables``; DefaultBackground.growth_factor,
    cosmology.py:2044-2093): returns (knots (201,) numpy, ascending z; device tensor (ncosmo, 2, 201))."""
    import torch
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    device = torch.device(device)
    params = dict(params or {})
    keep = []
    cparams = (_lib.cp_param * len(_lib.BG_PARAMS))()
    for i, name in enumerate(_lib.BG_PARAMS):
        cparams[i], n = _cparam(params.get(name, DEFAULTS[name]), device, keep)
        if n is not None and n != ncosmo:
            raise ValueError('per-cosmology arrays must have length ncosmo = {:d}, got {:d}'.format(ncosmo, n))
    tab = torch.empty((ncosmo, 2, _lib.GROWTH_NKNOTS), dtype=torch.float64, device=device)
    cn = ncdm.struct() if ncdm is not None and ncdm.nspecies else None
    _lib.background_init(device.index)
    _lib.check(_lib.load().cp_growth_ode_tables(ncosmo, ctypes.cast(cparams, ctypes.c_void_p), 0, ctypes.byref(cn) if cn is not None else None,
                                                {'m': 0, 'cb': 1}[mass], tab.data_ptr(), device.index, torch.cuda.current_stream(device).cuda_stream))
    knots = np.empty(_lib.GROWTH_NKNOTS)
    _lib.check(_lib.load().cp_growth_ode_knots(_lib.as_double_p(knots), knots.size))
    return knots, tab


def distance(kind, z, params=None, Omega_m=None, per_cosmology_z=False, device=None, ncdm=None, species=None):
    """
    Evaluate ``kind`` (one of :data:`KINDS`: 'comoving_radial_distance', 'comoving_transverse_distance',
    'angular_diameter_distance', 'luminosity_distance' [Mpc/h], 'efunc', 'hubble_function' [km/s/Mpc]).

    params : dict of ``h, Omega_cdm, Omega_b, Omega_k, T_cmb, N_ur, w0_fld, wa_fld``; each a float (shared) or an array /
        torch CUDA tensor of shape (ncosmo,).  Missing entries take the reference defaults (cosmology.py:730-733).
    Omega_m : float or array, optional; replaces ``Omega_cdm`` (Omega_cdm = Omega_m - Omega_b, cosmology.py:1163-1165).
    z : array or torch tensor.  ``per_cosmology_z=False``: any shape, shared by all cosmologies -> output (ncosmo,) + z.shape
        (or z.shape for scalar parameters).  ``per_cosmology_z=True``: shape (ncosmo, ...) -> output of the same shape.

    Returns numpy for numpy / float inputs, a torch tensor on the same device if ``z`` is a torch tensor.  float32 ``z`` gives
    float32 output (reference utils.flatarray, utils.py:98-138); computation is float64.
    """
    import torch
    if kind not in _lib.BG_KINDS:
        raise ValueError('unknown quantity {}; choose one of {}'.format(kind, KINDS))
    params = dict(params or {})
    for name in params:
        if name not in _lib.BG_PARAMS:
            raise ValueError('unknown background parameter {}'.format(name))
    if Omega_m is not None:
        params['Omega_cdm'] = Omega_m
    z_torch = _is_torch(z)
    if device is None:
        if z_torch and z.is_cuda:
            device = z.device
        else:
            for v in params.values():
                if _is_torch(v) and v.is_cuda:
                    device = v.device
                    break
    if device is None:
        if not torch.cuda.is_available():
            raise RuntimeError('cosmoprimo_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path')
        device = torch.device('cuda', torch.cuda.current_device())
    device = torch.device(device)
    if device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    out_dtype = None
    if z_torch:
        out_dtype = z.dtype if z.dtype in (torch.float32, torch.float64) else torch.float64
        tz = z.to(device=device, dtype=torch.float64)
    else:
        z = np.asarray(z)
        np_dtype = z.dtype if z.dtype in (np.float32, np.float64) else np.float64
        tz = dv.upload(np.asarray(z, dtype='f8'), device)
    ncosmo, batched = 1, False
    tensors = {}
    cparams = (_lib.cp_param * len(_lib.BG_PARAMS))()
    for i, name in enumerate(_lib.BG_PARAMS):
        v = params.get(name, DEFAULTS[name])
        if np.ndim(v) == 0 and not _is_torch(v):
            cparams[i].ptr, cparams[i].value = None, float(v)
            continue
        t = v.to(device=device, dtype=torch.float64) if _is_torch(v) else dv.upload(np.asarray(v, dtype='f8'), device, cache=False)
        t = t.reshape(-1).contiguous()
        if t.numel() == 1 and not batched and np.ndim(v) == 0:
            cparams[i].ptr, cparams[i].value = None, float(t)
            continue
        if batched and t.numel() != ncosmo:
            raise ValueError('parameter arrays must share one length, got {} and {}'.format(ncosmo, t.numel()))
        ncosmo, batched = t.numel(), True
        tensors[name] = t
        cparams[i].ptr, cparams[i].value = t.data_ptr(), 0.
    zshape = tuple(tz.shape)
    if per_cosmology_z:
        if not batched or len(zshape) < 1 or zshape[0] != ncosmo:
            raise ValueError('per_cosmology_z needs z of shape (ncosmo, ...) = ({:d}, ...), got {}'.format(ncosmo, zshape))
        nz = int(np.prod(zshape[1:], dtype=np.int64))
        oshape = zshape
    else:
        nz = int(np.prod(zshape, dtype=np.int64))
        oshape = ((ncosmo,) if batched else ()) + zshape
    tz = tz.reshape(-1).contiguous()
    out = torch.empty(ncosmo * nz, dtype=torch.float64, device=device)
    if ncosmo * nz:
        stream = torch.cuda.current_stream(device).cuda_stream
        cn = None
        if ncdm is not None and ncdm.nspecies:
            if ncdm.ncosmo != ncosmo:
                raise ValueError('massive-neutrino tables hold {:d} cosmologies, the parameters {:d}'.format(ncdm.ncosmo, ncosmo))
            cn = ncdm.struct(species)
        _lib.background_init(device.index)
        _lib.check(_lib.load().cp_background_eval(ncosmo, nz, ctypes.cast(cparams, ctypes.c_void_p), int(Omega_m is not None),
                                                  ctypes.byref(cn) if cn is not None else None, tz.data_ptr(), int(not per_cosmology_z), out.data_ptr(),
                                                  _lib.BG_KINDS[kind], device.index, stream))
    out = out.reshape(oshape)
    if z_torch:
        return out.to(out_dtype)
    return dv.to_host(out).astype(np_dtype, copy=False)
