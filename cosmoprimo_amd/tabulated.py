"""
'tabulated' engine (reference cosmoprimo/tabulated.py): background quantities interpolated linearly in a (z, quantity...) table -- what
``fiducial.TabulatedDESI()`` is for: redshift -> E(z), D_C(z) for whole catalogues.  The table lives in HBM, the interpolation is
``cp_interp_linear`` (bit-identical to the reference's ``numpy.interp``); torch redshifts on the device are interpolated in place.

extra_params: ``filename`` (ASCII table, '#' comments, first column z) and ``names`` (the other columns, default ['efunc',
'comoving_radial_distance']) as in the reference, or -- not in the reference -- ``table``: a dictionary {'z': ..., name: ...} of arrays.
"""
import numpy as np

from . import _lib
from . import _device as dv
from .cosmology import BaseEngine, BaseSection, CosmologyError


class TabulatedEngine(BaseEngine):

    """Engine using tabulated values from an ASCII file (reference tabulated.py:6-17)."""
    name = 'tabulated'
    _copes_with_ncdm = True     # nothing is computed from the parameters

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        table = self._extra_params.get('table', None)
        if table is None:
            self._names = list(self._extra_params.get('names', ['efunc', 'comoving_radial_distance']))
            arrays = np.loadtxt(self._extra_params['filename'], comments='#', usecols=range(len(self._names) + 1), unpack=True)
            table = dict(zip(['z'] + self._names, arrays))
        else:
            self._names = [name for name in table if name != 'z']
        self.z = np.ascontiguousarray(table['z'], dtype='f8')
        if self.z.ndim != 1 or self.z.size < 1 or np.any(np.diff(self.z) < 0.):
            raise CosmologyError('tabulated redshifts must be a 1D ascending array')
        self._tables = {}
        for name in self._names:
            array = np.ascontiguousarray(table[name], dtype='f8')
            if array.shape != self.z.shape:
                raise CosmologyError('tabulated {} must have the shape of z'.format(name))
            setattr(self, name, array)
            self._tables[name] = _InterpTable(self.z, array, self.device)


class _InterpTable(object):

    """Owner of a ``cp_interp_table``: one column of the table with its redshifts on the device, and the law of the redshift grid (uniform, uniform in
    the logarithm behind a few leading knots as data/desi.dat, neither) from which the kernel guesses a sample's interval instead of bisecting."""

    def __init__(self, x, f, device):
        import ctypes
        self._args = (x, f, device)
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.load().cp_interp_table_create(ctypes.byref(self.handle), x.size, _lib.as_double_p(x), _lib.as_double_p(f), device.index))
        law, first = ctypes.c_int(), ctypes.c_longlong()
        _lib.check(_lib.load().cp_interp_table_law(self.handle, ctypes.byref(law), ctypes.byref(first)))
        self.law = (law.value, first.value)      # (0 none / 1 uniform / 2 uniform in the logarithm, the knot it holds from)

    def __copy__(self):      # a handle has one owner: copies build their own table
        return self.__class__(*self._args)

    def __deepcopy__(self, memo):
        return self.__class__(*self._args)

    def __reduce__(self):
        return (self.__class__, self._args)

    def __call__(self, tz, out, stream):
        """out = numpy.interp(tz) on the device (float64, or float32 samples and results: computed in double); True when a sample lies outside the
        table (or is NaN)."""
        import ctypes
        outside = ctypes.c_int(0)
        apply = _lib.load().cp_interp_table_apply if tz.dtype == dv.torch().float64 else _lib.load().cp_interp_table_apply_f32
        _lib.check(apply(self.handle, tz.data_ptr(), out.data_ptr(), tz.numel(), ctypes.byref(outside), stream))
        return bool(outside.value)

    def __del__(self):
        try:
            if self.handle:
                _lib.load().cp_interp_table_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Background(BaseSection):

    """Tabulated background quantities (reference tabulated.py:20-27)."""

    def __init__(self, engine):
        super().__init__(engine)
        self.ba = engine

    def _interp(self, name, z):
        torch = dv.torch()
        ba = self.ba
        like_torch = dv.is_torch(z)
        dtype = dv.float_dtype(z)
        if like_torch and z.is_cuda and z.device == self.device and z.dtype == torch.float32 and ba._tables[name].law[0]:
            tz = z.contiguous()      # a float32 catalogue on the device: read and written as it is (no widened copies)
        else:
            tz = dv.to_device(z, self.device).contiguous()
        out = torch.empty_like(tz)
        if tz.numel():
            # the kernel itself reports samples outside the table (a NaN redshift is "outside" as well, unlike numpy's comparison): no pass over the
            # redshifts before it, one flag read back behind it
            if ba._tables[name](tz, out, dv.stream_of(self.device)):
                raise CosmologyError('Input z outside of tabulated range.')
        if like_torch:
            return out.to(torch.float32 if dtype == np.float32 else torch.float64)
        return dv.to_host(out).astype(dtype, copy=False)

    def __getattr__(self, name):
        # one method per tabulated column (the reference defines efunc and comoving_radial_distance)
        if not name.startswith('_') and name in self.__dict__.get('ba', {}).__dict__.get('_tables', {}):
            return lambda z: self._interp(name, z)
        raise AttributeError('{} has no attribute {}'.format(self.__class__.__name__, name))

    def efunc(self, z):
        r"""Function giving :math:`E(z)`, where the Hubble parameter is defined as :math:`H(z) = H_{0} E(z)`, unitless."""
        return self._interp('efunc', z)

    def comoving_radial_distance(self, z):
        r"""Comoving radial distance, in :math:`\mathrm{Mpc}/h`."""
        return self._interp('comoving_radial_distance', z)
