"""
Analytic matter power spectra on MI355X for batches of cosmologies: the data-parallel core behind the engines
'eisenstein_hu', 'eisenstein_hu_nowiggle' and 'bbks' (reference eisenstein_hu.py, eisenstein_hu_nowiggle.py, bbks.py).
"""
import numpy as np

from . import _lib
from . import _device as dv
from .background import DEFAULTS as BG_DEFAULTS

PK_DEFAULTS = dict(A_s=2.43e-9 * (0.8 / 0.87659)**2, n_s=0.96, alpha_s=0., beta_s=0., k_pivot=0.05)


def A_s_fid(sigma8):
    """First guess for A_s given sigma8 (reference BaseEngine._get_A_s_fid, cosmology.py:505-510)."""
    return 2.43e-9 * (sigma8 / 0.87659)**2


def analytic(engine, what, k, z=None, bg=None, pk=None, Omega_m=None, device=None, kscale=None):
    """
    ``what`` in ('matter', 'transfer', 'primordial', 'log_k_matter' = log(k P), z=None) for engine in ('eisenstein_hu', 'eisenstein_hu_nowiggle', 'bbks').

    k : (nk,) wavenumbers [h/Mpc] (numpy or torch), shared by the batch; z : (nz,) redshifts or None (no growth factor).
    bg : background parameters (see :func:`cosmoprimo_amd.background.distance`), pk : ``A_s, n_s, alpha_s, beta_s, k_pivot``;
    floats or arrays of shape (ncosmo,).  ``bg['ncdm']``: :class:`cosmoprimo_amd.background.NcdmTables` of the same cosmologies, for massive neutrinos
    (they enter through Omega0_m, the growth factor and, for BBKS, Omega_m: reference eisenstein_hu.py:322, 134-135, bbks.py:38).

    kscale : optional (ncosmo,) factors: cosmology i is evaluated at ``k * kscale[i]``.

    Returns a torch tensor on the device of shape (ncosmo,) (if batched) + ((nz,) if z is given) + (nk,), k fastest.
    """
    torch = dv.torch()
    bg, pk = dict(bg or {}), dict(pk or {})
    if Omega_m is not None:
        bg['Omega_cdm'] = Omega_m
    device = dv.resolve_device(device, k, z, *bg.values(), *pk.values())
    cbg, n1, keep1 = dv.pack_params(_lib.BG_PARAMS, bg, BG_DEFAULTS, device)
    cpk, n2, keep2 = dv.pack_params(_lib.PK_PARAMS, pk, PK_DEFAULTS, device)
    if n1 is not None and n2 is not None and n1 != n2:
        raise ValueError('parameter arrays must share one length, got {} and {}'.format(n1, n2))
    batched = n1 is not None or n2 is not None
    ncosmo = n1 or n2 or 1
    tk = dv.to_device(k, device).reshape(-1)
    nk = tk.numel()
    if what == 'log_k_matter' and z is not None:
        raise ValueError("'log_k_matter' is the spectrum without growth factor: z must be None")
    with_z = z is not None and what == 'matter'
    tz = dv.to_device(z, device).reshape(-1) if with_z else None
    nz = tz.numel() if with_z else 0
    out = torch.empty((ncosmo, max(nz, 1), nk), dtype=torch.float64, device=device)
    tks = None
    if kscale is not None:
        tks = dv.to_device(kscale, device).reshape(-1)
        if tks.numel() != ncosmo:
            raise ValueError('kscale must have one entry per cosmology ({:d}), got {:d}'.format(ncosmo, tks.numel()))
    lib = _lib.load()
    work = torch.empty(max(int(lib.cp_power_workspace_bytes(ncosmo)), 8), dtype=torch.uint8, device=device)    # fit coefficients: torch's caching allocator
    nu, keep_nu = dv.ncdm_arg(bg, ncosmo)
    _lib.check(lib.cp_power_eval(_lib.ENGINES[engine], _lib.PK_WHAT[what], ncosmo, dv.as_void_p(cbg), int(Omega_m is not None), nu, dv.as_void_p(cpk), nk,
                                 tk.data_ptr(), tks.data_ptr() if tks is not None else None, nz, tz.data_ptr() if with_z else None, out.data_ptr(),
                                 work.data_ptr(), device.index, dv.stream_of(device)))
    shape = ((ncosmo,) if batched else ()) + ((nz,) if with_z else ()) + (nk,)
    return out.reshape(shape)


def eh_scalars(bg=None, Omega_m=None, device=None):
    """dict of the EH / no-wiggle / BBKS fit coefficients (reference eisenstein_hu.py:34-92), torch tensors of shape (ncosmo,) or ()."""
    torch = dv.torch()
    bg = dict(bg or {})
    if Omega_m is not None:
        bg['Omega_cdm'] = Omega_m
    device = dv.resolve_device(device, *bg.values())
    cbg, n, keep = dv.pack_params(_lib.BG_PARAMS, bg, BG_DEFAULTS, device)
    ncosmo = n or 1
    out = torch.empty((ncosmo, len(_lib.EH_SCALARS)), dtype=torch.float64, device=device)
    nu, keep_nu = dv.ncdm_arg(bg, ncosmo)
    _lib.check(_lib.load().cp_eh_scalars(ncosmo, dv.as_void_p(cbg), int(Omega_m is not None), nu, out.data_ptr(), device.index, dv.stream_of(device)))
    return {name: (out[:, i] if n is not None else out[0, i]) for i, name in enumerate(_lib.EH_SCALARS)}


def variants_scalars(bg=None, ncdm=None, device=None):
    """dict of the attributes of the 'eisenstein_hu_nowiggle_variants' engine (reference eisenstein_hu_nowiggle_variants.py:32-76) through
    ``cp_variants_scalars``: torch tensors of shape (ncosmo,) or ().  ``ncdm``: :class:`cosmoprimo_amd.background.NcdmTables` or None."""
    import ctypes
    torch = dv.torch()
    bg = dict(bg or {})
    device = dv.resolve_device(device, *bg.values())
    cbg, n, keep = dv.pack_params(_lib.BG_PARAMS, bg, BG_DEFAULTS, device)
    ncosmo = n or 1
    cn = None
    if ncdm is not None and ncdm.nspecies:
        if ncdm.ncosmo != ncosmo:
            raise ValueError('massive-neutrino tables hold {:d} cosmologies, the parameters {:d}'.format(ncdm.ncosmo, ncosmo))
        cn = ncdm.struct()
    out = torch.empty((ncosmo, len(_lib.VARIANTS_SCALARS)), dtype=torch.float64, device=device)
    _lib.check(_lib.load().cp_variants_scalars(ncosmo, dv.as_void_p(cbg), 0, ctypes.byref(cn) if cn is not None else None, out.data_ptr(), device.index,
                                               dv.stream_of(device)))
    return {name: (out[:, i] if n is not None else out[0, i]) for i, name in enumerate(_lib.VARIANTS_SCALARS)}


def variants(what, k, z, of='delta_m', bg=None, pk=None, ncdm=None, device=None):
    """
    ``what`` in ('matter', 'transfer') of the 'eisenstein_hu_nowiggle_variants' engine (reference eisenstein_hu_nowiggle_variants.py:
    Eisenstein & Hu 1997 with massive neutrinos) through ``cp_power_eval_variants``.  ``ncdm``: :class:`cosmoprimo_amd.background.NcdmTables`
    of the same cosmologies, or None.  Returns a device tensor (ncosmo,) (if batched) + (nz, nk).
    """
    import ctypes
    torch = dv.torch()
    bg, pk = dict(bg or {}), dict(pk or {})
    device = dv.resolve_device(device, k, z, *bg.values(), *pk.values())
    cbg, n1, keep1 = dv.pack_params(_lib.BG_PARAMS, bg, BG_DEFAULTS, device)
    cpk, n2, keep2 = dv.pack_params(_lib.PK_PARAMS, pk, PK_DEFAULTS, device)
    if n1 is not None and n2 is not None and n1 != n2:
        raise ValueError('parameter arrays must share one length, got {} and {}'.format(n1, n2))
    batched = n1 is not None or n2 is not None
    ncosmo = n1 or n2 or 1
    tk, tz = dv.to_device(k, device).reshape(-1), dv.to_device(z, device).reshape(-1)
    out = torch.empty((ncosmo, tz.numel(), tk.numel()), dtype=torch.float64, device=device)
    cn = None
    if ncdm is not None and ncdm.nspecies:
        if ncdm.ncosmo != ncosmo:
            raise ValueError('massive-neutrino tables hold {:d} cosmologies, the parameters {:d}'.format(ncdm.ncosmo, ncosmo))
        cn = ncdm.struct()
    _lib.check(_lib.load().cp_power_eval_variants(_lib.PK_WHAT[what], {'delta_m': 0, 'delta_cb': 1}[of], ncosmo, dv.as_void_p(cbg), 0,
                                                  ctypes.byref(cn) if cn is not None else None, dv.as_void_p(cpk), tk.numel(), tk.data_ptr(), tz.numel(),
                                                  tz.data_ptr(), out.data_ptr(), device.index, dv.stream_of(device)))
    return out if batched else out[0]
