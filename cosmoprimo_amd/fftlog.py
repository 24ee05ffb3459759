"""
FFTLog on MI355X: same classes and call signatures as the reference ``cosmoprimo/fftlog.py``
(:class:`FFTlog`, :class:`HankelTransform`, :class:`PowerToCorrelation`, :class:`CorrelationToPower`,
:class:`TophatVariance`, :class:`GaussianVariance`, :func:`pad`, the Mellin kernels), with
``FFTlog.__call__`` executed by ONE fused HIP kernel (``csrc/cp_fftlog*.{h,hip}``) through the C ABI
of ``libcosmoprimo_amd.so``.

Host side (this file): argument handling and the shape / dtype bookkeeping of ``__call__`` (fftlog.py:198-241); the
plan tables (what reference ``FFTlog._setup`` computes, fftlog.py:144-184, plus the conventions of the named transforms)
come from the library's host routine ``cp_fftlog_tables`` (``csrc/cp_fftlog_setup.cpp``).
Inputs may be numpy arrays (copied to the GPU and back) or torch CUDA tensors (zero copy; results
stay on the device).  There is no CPU compute path.
"""
import numpy as np

from . import _lib
from . import _device as dv


# ---------------------------------------------------------------------------------------------
# Mellin-transform kernels (reference fftlog.py:666-766)
# ---------------------------------------------------------------------------------------------
class BaseKernel(object):

    """Base kernel: ``kernel(z)`` returns the Mellin transform U_K(z) for complex ``z``."""
    _kind = None
    _param = 0.

    def __call__(self, z):
        return self.eval(z)

    def eval(self, z):
        shape = np.shape(z)
        return _lib.kernel_eval(self._kind, self._param, z).reshape(shape)

    def __eq__(self, other):
        return other.__class__ == self.__class__


class BaseBesselKernel(BaseKernel):

    """Base Bessel kernel (reference fftlog.py:677-685)."""

    def __init__(self, nu):
        self.nu = nu

    @property
    def _param(self):
        return float(self.nu)

    def __eq__(self, other):
        return other.__class__ == self.__class__ and other.nu == self.nu


class BesselJKernel(BaseBesselKernel):
    """(Mellin transform of) Bessel kernel (reference fftlog.py:688-695)."""
    _kind = _lib.KERNEL_BESSEL_J


class SphericalBesselJKernel(BaseBesselKernel):
    """(Mellin transform of) spherical Bessel kernel (reference fftlog.py:698-705)."""
    _kind = _lib.KERNEL_SPHERICAL_BESSEL_J


class BaseTophatKernel(BaseKernel):

    """Base tophat kernel (reference fftlog.py:708-716)."""

    def __init__(self, ndim=1):
        self.ndim = ndim

    @property
    def _param(self):
        return float(self.ndim)

    def __eq__(self, other):
        return other.__class__ == self.__class__ and other.ndim == self.ndim


class TophatKernel(BaseTophatKernel):
    """(Mellin transform of) tophat kernel (reference fftlog.py:719-726)."""
    _kind = _lib.KERNEL_TOPHAT


class TophatSqKernel(BaseTophatKernel):
    """(Mellin transform of) square of tophat kernel (reference fftlog.py:729-746)."""
    _kind = _lib.KERNEL_TOPHAT_SQ


class GaussianKernel(BaseKernel):
    """(Mellin transform of) Gaussian kernel (reference fftlog.py:749-756)."""
    _kind = _lib.KERNEL_GAUSSIAN


class GaussianSqKernel(BaseKernel):
    """(Mellin transform of) square of Gaussian kernel (reference fftlog.py:759-766)."""
    _kind = _lib.KERNEL_GAUSSIAN_SQ


# ---------------------------------------------------------------------------------------------
# pad (host, for coordinate grids; the data path pads inside the kernel) -- reference fftlog.py:436-505
# ---------------------------------------------------------------------------------------------
def _split_pair(value):
    try:
        left, right = value
    except (TypeError, ValueError):
        left = right = value
    return left, right


def pad(array, pad_width, axis=-1, extrap=0):
    """
    Pad ``array`` along ``axis`` (same contract as reference ``pad``, fftlog.py:436-505).

    extrap : 'log' (log-log extrapolation), 'edge' (repeat edge value) or a number; a tuple differentiates left / right.
    """
    array = np.asarray(array)
    wl, wr = _split_pair(pad_width)
    el, er = _split_pair(extrap)
    axis = axis % array.ndim
    a = np.moveaxis(array, axis, -1)

    def side(e, w, left):
        if isinstance(e, str) and e == 'edge':
            return np.repeat(a[..., :1] if left else a[..., -1:], w, axis=-1)
        if isinstance(e, str) and e == 'log':
            if left:
                return a[..., :1] * (a[..., 1:2] / a[..., :1]) ** np.arange(-w, 0)
            return a[..., -1:] / (a[..., -2:-1] / a[..., -1:]) ** np.arange(1, w + 1)
        return np.full(a.shape[:-1] + (w,), e)

    out = np.concatenate([side(el, wl, True), a, side(er, wr, False)], axis=-1)
    return np.moveaxis(out, -1, axis)


_EXTRAP_CODES = {'edge': _lib.EXTRAP_EDGE, 'log': _lib.EXTRAP_LOGLOG}


def _extrap_code(e):
    if isinstance(e, str):
        if e not in _EXTRAP_CODES:
            raise ValueError('Unknown extrapolation {}'.format(e))
        return _EXTRAP_CODES[e], 0.
    return _lib.EXTRAP_CONSTANT, float(e)


class _Plan(object):

    """Owner of a ``cp_fftlog_plan`` (device tables); freed with the object."""

    def __init__(self, n, npad, pre, post, u, device):
        import ctypes
        self._handle = ctypes.c_void_p()
        nker = pre.shape[0]
        pre = np.ascontiguousarray(pre, dtype='f8')
        post = np.ascontiguousarray(post, dtype='f8')
        u = np.ascontiguousarray(u, dtype='c16')
        lib = _lib.load()
        _lib.check(lib.cp_fftlog_plan_create(ctypes.byref(self._handle), n, npad, nker, _lib.as_double_p(pre), _lib.as_double_p(post),
                                             _lib.as_double_p(u.view('f8')), device))

    @property
    def handle(self):
        return self._handle

    def __del__(self):
        try:
            if self._handle:
                _lib.load().cp_fftlog_plan_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


def _torch():
    import torch
    return torch


def _is_torch(x):
    return type(x).__module__.startswith('torch')


def _unit_phase_split(table, what):
    """Split a (nker, npad) table into a real table and one unit complex factor per row (tables of ``complex=True`` transforms and of
    their inverses are of that form); real tables return ``(table, None)``."""
    table = np.asarray(table)
    if not np.iscomplexobj(table):
        return np.ascontiguousarray(table, dtype='f8'), None
    pivot = np.abs(table).argmax(axis=-1)
    lead = table[np.arange(table.shape[0]), pivot]
    phase = lead / np.abs(lead)
    real = table / phase[:, None]
    if np.abs(real.imag).max() > 1e-12 * np.abs(real.real).max():
        raise NotImplementedError('{} must be a real table times one complex factor per transform'.format(what))
    return np.ascontiguousarray(real.real), phase


class FFTlog(dv.Copyable):
    r"""
    FFTLog algorithm (https://jila.colorado.edu/~ajsh/FFTLog/) for :math:`G(y) = \int_0^\infty x dx F(x) K(xy)`,
    same constructor and call contract as the reference (fftlog.py:31-248).

    The named transforms below differ from the bare algorithm only by a *convention* (class attribute ``_convention``): an offset
    added to the tilt ``q``, a power law ``c x^p`` multiplied into the prefactor and a phase per multipole for the postfactor.
    All tables come from one call to the library's host routine ``cp_fftlog_tables``.
    """
    # (tilt offset, prefactor power p, prefactor constant c, phase of order ell or None)
    _convention = (0., 0., 1., None)

    def __init__(self, x, kernel, q=0, minfolds=2, lowring=True, xy=1, check_level=0, engine='numpy', device=None, **engine_kwargs):
        r"""
        Parameters are those of the reference (fftlog.py:49-92): ``x`` log-spaced input coordinates (1D or one row per
        kernel), ``kernel`` callable(s) returning the Mellin transform, ``q`` tilt(s), ``minfolds``, ``lowring``, ``xy``,
        ``check_level``.

        engine : string, default='numpy'
            The reference's default by name; every named engine (``'numpy'``, ``'fftw'``, ``'mi355x'`` / ``'hip'``) runs the fused HIP kernel
            (this package has no CPU path); ``engine_kwargs`` of the FFTW engine (``nthreads``, ``wisdom``, ``plan``) are ignored.
            An object with the reference's ``forward`` / ``backward`` methods (fftlog.py:508-544) is honoured as well: the
            transform then runs un-fused around that engine (prefactor, engine.forward, x u, engine.backward, postfactor).

        device : int, string, torch.device, default=None
            GPU holding the plan; defaults to the device of the first input (current CUDA device for numpy inputs).

        rescale_rows : bool (``engine_kwargs``)
            Accepted for compatibility with earlier versions of this package and ignored: the kernel now always keeps the rounding
            of a row relative to its own magnitude (rows that differ by more than a factor 32 from their pair partner are rescaled
            by exact powers of two inside the kernel).
        """
        engine_kwargs.pop('rescale_rows', None)
        self.inparallel = isinstance(kernel, (tuple, list))
        kernels = list(kernel) if self.inparallel else [kernel]
        nker = len(kernels)
        x = np.array(x, dtype='f8')
        if check_level and self.inparallel and x.ndim == 2 and x.shape[0] != nker:
            raise ValueError('x and kernel must of same length')
        for name, value in (('q', q), ('xy', xy)):
            if check_level and np.ndim(value) and len(value) != nker:
                raise ValueError('{} and kernel must be lists of same length'.format(name))
        self.x = np.ascontiguousarray(np.broadcast_to(x, (nker, x.shape[-1])))     # one row of coordinates per kernel
        self._device = device
        self._phase = None       # unit complex factor per kernel on the way out (complex=True transforms), else None
        self._phase_in = None    # ... and on the way in (inverse of a complex=True transform)
        self._setup(kernels, np.broadcast_to(np.asarray(q, dtype='f8'), (nker,)), minfolds=minfolds, lowring=lowring,
                    xy=np.broadcast_to(np.asarray(xy, dtype='f8'), (nker,)), check_level=check_level)
        self.set_fft_engine(engine, **engine_kwargs)

    def _device_copy(self, name, array, dev):
        """Device tensor of a host table of this plan (output coordinates, phases), copied once per device instead of on every call."""
        cache = self.__dict__.setdefault('_device_tables', {})
        key = (name, dev.index)
        if key not in cache:
            cache[key] = dv.upload(array, dev)
        return cache[key]

    def set_fft_engine(self, engine='numpy', **engine_kwargs):
        """Select the engine (reference fftlog.py:119-132); see :func:`get_fft_engine`."""
        self._engine = get_fft_engine(engine, size=self.padded_size, nparallel=self.nparallel, **engine_kwargs)

    @property
    def nparallel(self):
        """Number of transforms performed in parallel."""
        return self.x.shape[0]

    @property
    def size(self):
        """Size of x-coordinates."""
        return self.x.shape[-1]

    def _setup(self, kernels, qs, minfolds=2, lowring=True, xy=1., check_level=0):
        """All tables of the plan (what reference fftlog.py:144-184 sets up, plus the convention of the named transform), built on the
        host by ``cp_fftlog_tables``; kernels that are arbitrary Python callables are evaluated here at the arguments the tables need."""
        import ctypes
        lib = _lib.load()
        n, nker = self.size, self.nparallel
        npad = lib.cp_fftlog_padded_size(n, int(minfolds))
        if npad < 0:
            raise ValueError('minfolds = {} and size = {:d} do not give a valid padded size'.format(minfolds, n))
        self.padded_size = npad
        left = (npad - n) // 2
        self.padded_size_in_left, self.padded_size_in_right = left, npad - n - left
        self.padded_size_out_left, self.padded_size_out_right = npad - n - left, left
        q_offset, pre_power, pre_const, _ = self._convention
        spec = (_lib.FFTlogSpec * nker)()
        custom = [not isinstance(kernel, BaseKernel) or kernel._kind is None for kernel in kernels]
        for ik, kernel in enumerate(kernels):
            spec[ik].kind = _lib.KERNEL_CUSTOM if custom[ik] else kernel._kind
            spec[ik].param = 0. if custom[ik] else kernel._param
            spec[ik].q, spec[ik].xy = float(qs[ik]), float(np.broadcast_to(xy, (nker,))[ik])
            spec[ik].pre_power, spec[ik].pre_const, spec[ik].post_sign = pre_power, pre_const, 1.
        nu = npad // 2 + 1
        u_custom = lowring_custom = None
        if any(custom):
            delta = np.log(self.x[:, -1] / self.x[:, 0]) / (n - 1)
            u_custom, lowring_custom = np.zeros((nker, nu), dtype='c16'), np.zeros(nker, dtype='c16')
            for ik, kernel in enumerate(kernels):
                if custom[ik]:
                    u_custom[ik] = kernel(qs[ik] + 2j * np.pi / npad / delta[ik] * np.arange(nu))   # same rounding as the library's arguments
                    if lowring:
                        lowring_custom[ik] = kernel(qs[ik] + 1j * np.pi / delta[ik])
        out = {name: np.empty(shape, dtype='f8') for name, shape in [('delta', nker), ('lnxy', nker), ('y', (nker, n)), ('padded_x', (nker, npad)),
                                                                     ('padded_y', (nker, npad)), ('pre', (nker, npad)), ('post', (nker, npad)),
                                                                     ('u', (nker, nu, 2))]}
        P = _lib.as_double_p
        _lib.check(lib.cp_fftlog_tables(n, nker, P(self.x), spec, int(minfolds), int(bool(lowring)), int(check_level),
                                        None if u_custom is None else P(u_custom.view('f8')), None if lowring_custom is None else P(lowring_custom.view('f8')),
                                        *[P(out[name]) for name in ('delta', 'lnxy', 'y', 'padded_x', 'padded_y', 'pre', 'post', 'u')]))
        self.delta, self.lnxy, self.y, self.padded_x, self.padded_y = (out[name] for name in ('delta', 'lnxy', 'y', 'padded_x', 'padded_y'))
        self._pre, self._post, self._u = out['pre'], out['post'], out['u'][..., 0] + 1j * out['u'][..., 1]
        self._plan = None

    # The three tables are plain arrays that user code may rescale (as the reference's subclasses do, fftlog.py:280): assigning
    # one of them drops the device plan.  A complex table must be a real table times one unit factor per transform.
    @property
    def padded_prefactor(self):
        return self._pre

    @padded_prefactor.setter
    def padded_prefactor(self, value):
        self._pre, self._plan = np.asarray(value), None

    @property
    def padded_postfactor(self):
        return self._post

    @padded_postfactor.setter
    def padded_postfactor(self, value):
        self._post, self._plan = np.asarray(value), None

    @property
    def padded_u(self):
        return self._u

    @padded_u.setter
    def padded_u(self, value):
        self._u, self._plan = np.asarray(value), None

    # -- device plan --------------------------------------------------------------------------
    def _resolve_device(self, tensor=None):
        torch = _torch()
        if self._device is not None:
            dev = torch.device(self._device if not isinstance(self._device, int) else 'cuda:{:d}'.format(self._device))
        elif tensor is not None:
            dev = tensor.device
        else:
            if not torch.cuda.is_available():
                raise RuntimeError('cosmoprimo_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path')
            dev = torch.device('cuda', torch.cuda.current_device())
        if dev.type != 'cuda':
            raise ValueError('FFTlog runs on a GPU; got device {}'.format(dev))
        if dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        return dev

    def _get_plan(self, dev):
        """Build (once per device) the library plan from the current tables."""
        if self._plan is not None and self._plan[0] == dev.index:
            return self._plan[1]
        pre, self._phase_in = _unit_phase_split(self._pre, 'the prefactor')
        post, self._phase = _unit_phase_split(self._post, 'the postfactor')
        plan = _Plan(self.size, self.padded_size, pre, post, self._u, dev.index)
        self._plan = (dev.index, plan)
        self.__dict__.pop('_device_tables', None)
        return plan

    def __call__(self, fun, extrap=0, keep_padding=False, out_window=None, out=None):
        """
        Perform the transforms (reference fftlog.py:198-241).

        fun : numpy array or torch CUDA tensor; last dimensions must broadcast against (:attr:`nparallel`, len(x)).
        extrap : 0 (default), number, 'edge', 'log', or a (left, right) tuple.
        keep_padding : return the padded transform.
        out_window : (first, count), not in the reference: the caller reads these entries of every output row only; the others are unspecified
            (the default transform then does not write them: cp_fftlog_execute_window).
        out : not in the reference: a float64, contiguous tensor on the plan's device with the number of elements of the result, which the kernel
            writes (nothing is allocated: a sampler calling with the same batch shape step after step keeps one result buffer); the returned
            transform is a view of it.  Real transforms only (a ``complex=True`` result is a product formed after the kernel).

        Returns ``(y, fftloged)``, numpy for numpy input, torch (same device) for torch input.  Output is float64
        (complex128 for ``complex=True`` transforms) as in the reference.
        """
        if not isinstance(self._engine, MI355XFFTEngine):
            return self._call_unfused(fun, extrap=extrap, keep_padding=keep_padding)
        torch = _torch()
        is_torch = _is_torch(fun)
        dev = self._resolve_device(fun if is_torch else None)
        plan = self._get_plan(dev)           # also splits the unit phases off complex tables
        if not is_torch:
            fun = dv.upload(fun, dev)
        if self._phase_in is not None:       # complex prefactor: the FFT sees the real part of fun x prefactor (numpy.fft.rfft, fftlog.py:540)
            tfun = (fun.to(torch.complex128) * self._device_copy('phase_in', self._phase_in, dev)[:, None]).real
        else:
            tfun = (fun.real if fun.is_complex() else fun).to(device=dev, dtype=torch.float64)
        n, nker, npad = self.size, self.nparallel, self.padded_size
        fshape = tuple(tfun.shape)
        if len(fshape) < 1 or fshape[-1] != n:
            raise ValueError('fun last dimension must be {:d}, got shape {}'.format(n, fshape))
        # broadcast against (nker, n), as fun * padded_prefactor does in the reference
        if len(fshape) == 1:
            tfun = tfun[None, :]
        if nker > 1 and tfun.shape[-2] not in (1, nker):
            raise ValueError('fun shape {} does not broadcast against ({:d}, {:d})'.format(fshape, nker, n))
        bshape = tuple(tfun.shape[:-2]) + (nker, n) if nker > 1 else tuple(tfun.shape)
        tin = tfun.expand(bshape).contiguous()
        (cl, vl), (cr, vr) = (_extrap_code(e) for e in _split_pair(extrap))
        # Rows stay independent inside the kernel (non-finite rows give NaN rows, rows of very different magnitude are rescaled by exact
        # powers of two: csrc/cp_fftlog_body.h, "row independence"), as with numpy's row-by-row FFTs: nothing to screen here, no extra
        # pass over the batch, no host synchronisation.
        nbatch = int(np.prod(bshape[:-2] if nker > 1 else bshape[:-1], dtype='i8'))
        nout = npad if keep_padding else n
        if out is None:
            tout = torch.empty(bshape[:-1] + (nout,), dtype=torch.float64, device=dev)
        else:
            if self._phase is not None:
                raise ValueError('out= is for real transforms (complex=True multiplies the result by a phase behind the kernel)')
            if not (_is_torch(out) and out.dtype == torch.float64 and out.device == dev and out.is_contiguous() and out.numel() == nbatch * nker * nout):
                raise ValueError('out must be a contiguous float64 tensor of {:d} elements on {}'.format(nbatch * nker * nout, dev))
            tout = out.view(bshape[:-1] + (nout,))
        if nbatch > 0 and out_window is not None:
            _lib.check(_lib.load().cp_fftlog_execute_window(plan.handle, tin.data_ptr(), tout.data_ptr(), nbatch, cl, vl, cr, vr, int(bool(keep_padding)),
                                                            int(out_window[0]), int(out_window[1]), torch.cuda.current_stream(dev).cuda_stream))
        elif nbatch > 0:
            _lib.check(_lib.load().cp_fftlog_execute(plan.handle, tin.data_ptr(), tout.data_ptr(), nbatch, cl, vl, cr, vr, int(bool(keep_padding)),
                                                     torch.cuda.current_stream(dev).cuda_stream))
        if self._phase is not None:
            tout = tout * self._device_copy('phase', self._phase, dev)[:, None]
        y = self.padded_y if keep_padding else self.y
        if not self.inparallel:
            y = y[0]
            tout = tout.reshape(fshape[:-1] + (nout,))
        if is_torch:
            return self._device_copy(('y_padded' if keep_padding else 'y') + ('' if self.inparallel else '0'), y, dev), tout
        return y, dv.to_host(tout)

    def _call_unfused(self, fun, extrap=0, keep_padding=False):
        """The transform around a user-supplied engine object (reference fftlog.py:198-241 with the protocol of :508-544): host numpy
        for the elementwise steps, ``engine.forward`` (real -> half complex spectrum) and ``engine.backward`` (which must compute
        ``irfft(conj(.))``) for the FFTs.  This is the reference's plug point for foreign FFT libraries; the fused kernel is not involved."""
        fun = np.asarray(fun.cpu() if _is_torch(fun) else fun)
        spectrum = self._engine.forward(pad(fun, (self.padded_size_in_left, self.padded_size_in_right), axis=-1, extrap=extrap) * self.padded_prefactor)
        out = self._engine.backward(spectrum * self.padded_u) * self.padded_postfactor
        if keep_padding:
            y = self.padded_y
        else:
            y, out = self.y, out[..., self.padded_size_out_left:self.padded_size_out_left + self.size]
        if not self.inparallel:
            y, out = y[0], out.reshape(fun.shape[:-1] + out.shape[-1:])
        return y, out

    def inv(self):
        """Turn the transform into its inverse, in place (reference fftlog.py:243-248): input and output grids trade places (the padded
        grids become the un-padded ones, as there), each factor becomes the reciprocal of its counterpart and u -> 1 / conj(u)."""
        tables = dict(x=self.y, y=self.x, padded_x=self.y, padded_y=self.x, padded_prefactor=1. / self._post, padded_postfactor=1. / self._pre,
                      padded_u=1. / np.conj(self._u))
        for name, value in tables.items():
            setattr(self, name, value)
        self.__dict__.pop('_device_tables', None)     # device copies of the output coordinates and phases
        self._plan = None


def _kernels_of(cls, order):
    """One kernel for a scalar order, a list (transforms in parallel) for a sequence of orders."""
    return cls(order) if np.ndim(order) == 0 else [cls(o) for o in order]


class _MultipoleTransform(FFTlog):
    """Transforms between multipoles of order ``ell`` with spherical Bessel kernels: tilt offset 1.5, prefactor x^3 c, and the phase
    ``_phase_unit ** ell`` -- kept as such for ``complex=True`` (complex128 output), reduced to its real sign (-1)^(ell // 2) otherwise."""
    _phase_unit = 1.

    def __init__(self, x, ell=0, q=0, complex=False, **kwargs):
        FFTlog.__init__(self, x, _kernels_of(SphericalBesselJKernel, ell), q=self._convention[0] + np.asarray(q), **kwargs)
        ell = np.atleast_1d(ell)
        phase = self._phase_unit**ell if complex else (-1.)**(ell // 2)
        self.padded_postfactor = self._post * phase[:, None]


class HankelTransform(FFTlog):
    """Hankel transform with Bessel kernels of order ``nu`` (reference fftlog.py:252-280): prefactor x^2."""
    _convention = (0., 2., 1., None)

    def __init__(self, x, nu=0, **kwargs):
        FFTlog.__init__(self, x, _kernels_of(BesselJKernel, nu), **kwargs)


class PowerToCorrelation(_MultipoleTransform):
    r"""
    Power spectrum to correlation function (reference fftlog.py:284-330):
    :math:`\xi_\ell(s) = \frac{(-i)^\ell}{2\pi^2} \int dk k^2 P_\ell(k) j_\ell(ks)`.
    """
    _convention = (1.5, 3., (2. * np.pi)**-1.5, None)
    _phase_unit = -1j

    def __init__(self, k, ell=0, q=0, complex=False, **kwargs):
        """``k``: input log-spaced wavenumbers, 1-D (broadcast to every ``ell``) or one row per ``ell``; ``q``: tilt on top of the 1.5 of the
        convention; ``complex``: keep the phase (-i)^ell instead of its real sign; ``kwargs`` for :class:`FFTlog` (reference fftlog.py:292-330)."""
        _MultipoleTransform.__init__(self, k, ell=ell, q=q, complex=complex, **kwargs)


class CorrelationToPower(_MultipoleTransform):
    r"""
    Correlation function to power spectrum (reference fftlog.py:334-377):
    :math:`P_\ell(k) = 4\pi i^\ell \int ds s^2 \xi_\ell(s) j_\ell(ks)`.
    """
    _convention = (1.5, 3., (2. * np.pi)**1.5, None)
    _phase_unit = 1j

    def __init__(self, s, ell=0, q=0, complex=False, **kwargs):
        """``s``: input log-spaced separations, 1-D (broadcast to every ``ell``) or one row per ``ell``; ``q``: tilt on top of the 1.5 of the
        convention; ``complex``: keep the phase i^ell instead of its real sign; ``kwargs`` for :class:`FFTlog` (reference fftlog.py:342-377)."""
        _MultipoleTransform.__init__(self, s, ell=ell, q=q, complex=complex, **kwargs)


class _WindowVariance(FFTlog):
    r"""Variance of the field smoothed by a window: :math:`\sigma^2(r) = \frac{1}{2\pi^2} \int dk k^2 P(k) W^2(kr)` (tilt offset 1.5)."""
    _convention = (1.5, 3., 1. / (2. * np.pi**2), None)
    _window = None

    def __init__(self, k, q=0, **kwargs):
        FFTlog.__init__(self, k, self._window(), q=self._convention[0] + q, **kwargs)


class TophatVariance(_WindowVariance):
    """Variance in a tophat window (reference fftlog.py:381-405)."""
    _window = staticmethod(lambda: TophatSqKernel(ndim=3))


class GaussianVariance(_WindowVariance):
    """Variance in a Gaussian window (reference fftlog.py:409-433)."""
    _window = staticmethod(GaussianSqKernel)


# ---------------------------------------------------------------------------------------------
# engines (reference fftlog.py:508-663).  The reference's protocol splits a transform into forward / backward FFTs done by an engine
# object; here FFTlog.__call__ is one fused kernel whatever the engine's name.  The engine classes keep the reference's names and
# methods: an instance handed to FFTlog(engine=...) selects the fused kernel, and forward / backward called by hand are device FFTs
# of this package (cp_rfft_forward / cp_rfft_backward), not numpy's or FFTW's.
# ---------------------------------------------------------------------------------------------
class BaseFFTEngine(object):

    """FFT engine descriptor (reference fftlog.py:508-531); does not touch OMP_NUM_THREADS (no host threads are used)."""

    def __init__(self, size, nparallel=1, nthreads=None):
        self.size = size
        self.nparallel = nparallel
        self.nthreads = nthreads


class MI355XFFTEngine(BaseFFTEngine):

    """The fused HIP FFTLog kernel (pad, prefactor, FFT, u, inverse FFT, postfactor, crop in one launch) when given to :class:`FFTlog`;
    ``forward`` / ``backward`` are the two real FFTs of the reference's protocol on the device, for code that calls them itself."""
    name = 'mi355x'

    def __init__(self, size, nparallel=1, nthreads=None, device=None, **kwargs):
        super(MI355XFFTEngine, self).__init__(size, nparallel=nparallel, nthreads=nthreads)
        _lib.load()  # fail loudly when the HIP library is missing
        self._device = device
        self._rfft_plans = {}

    def _rfft_plan(self, device):
        import ctypes
        key = device.index
        if key not in self._rfft_plans:
            handle = ctypes.c_void_p()
            _lib.check(_lib.load().cp_rfft_plan_create(ctypes.byref(handle), int(self.size), key))
            self._rfft_plans[key] = handle
        return self._rfft_plans[key]

    @staticmethod
    def _native_size(size):
        """Sizes the row kernels of ``cp_rfft.hip`` take: powers of two from 8 to 16 384; every other size goes through Bluestein's algorithm on them."""
        return 8 <= size <= 16384 and size & (size - 1) == 0

    def _dft_any_size(self, a):
        """DFT along the last axis of the complex device tensor ``a`` (rows, n) for ANY n, by Bluestein's chirp-z identity: with w_j = exp(-i pi j^2 / n),
        X_k = w_k sum_j (a_j w_j) conj(w)_{k-j} -- a cyclic convolution of length M >= 2 n - 1, taken as a power of two and run on this package's own real
        FFTs (a complex transform of length M = the real transforms of its real and imaginary parts; the inverse = the forward transform of the conjugate)."""
        torch = _torch()
        n = a.shape[-1]
        dev = a.device
        M = 8
        while M < 2 * n - 1:
            M *= 2
        if not self._native_size(M):      # (the convolution must itself be a native transform: no recursion)
            raise NotImplementedError('FFT engine of size {:d}: sizes that are not a power of two go up to 8192 (their convolution of length {:d} must fit '
                                      'the row kernels, 16 384 samples); powers of two up to 16 384'.format(n, M))
        cache = self.__dict__.setdefault('_bluestein', {})
        key = (n, dev.index)
        if key not in cache:
            j = np.arange(n, dtype='i8')
            w = np.exp(-1j * np.pi * ((j * j) % (2 * n)) / n)      # (j^2 reduced modulo 2 n: the phase keeps its digits for long rows)
            b = np.zeros(M, dtype='c16')
            b[:n] = np.conj(w)
            b[M - n + 1:] = np.conj(w[1:][::-1])
            cache[key] = (torch.as_tensor(w).to(dev), torch.as_tensor(np.fft.fft(b)).to(dev), MI355XFFTEngine(M, device=dev))
        w, B, sub = cache[key]

        def fft(c):      # complex (rows, M) -> complex (rows, M)
            halves = [sub.forward(part.contiguous()) for part in (c.real, c.imag)]      # (rows, M // 2 + 1) each
            full = [torch.cat([h, torch.conj(h[..., 1:M // 2]).flip(-1)], dim=-1) for h in halves]
            return full[0] + 1j * full[1]

        padded = torch.zeros(a.shape[:-1] + (M,), dtype=torch.complex128, device=dev)
        padded[..., :n] = a * w
        conv = torch.conj(fft(torch.conj(fft(padded) * B))) / M
        return conv[..., :n] * w

    def _run_any_size(self, fun, backward):
        torch = _torch()
        is_torch = _is_torch(fun)
        dev = dv.resolve_device(self._device, fun)
        n = int(self.size)
        nh = n // 2 + 1
        if backward:      # irfft(conj(fun), n): the DFT of the Hermitian extension of fun is real; the imaginary parts of its DC (and Nyquist) bins are ignored, as numpy does
            x = (fun if is_torch else torch.as_tensor(np.ascontiguousarray(fun, dtype='c16'))).to(device=dev, dtype=torch.complex128)
            if x.ndim < 1 or x.shape[-1] != nh:
                raise ValueError('last dimension must be {:d}, got shape {}'.format(nh, tuple(x.shape)))
            x = x.reshape(-1, nh).clone()
            x[:, 0] = x[:, 0].real
            if n % 2 == 0:
                x[:, -1] = x[:, -1].real
            full = torch.cat([x, torch.conj(x[:, 1:(n + 1) // 2]).flip(-1)], dim=-1)
            out = (self._dft_any_size(full).real / n).reshape(tuple(fun.shape[:-1]) + (n,))
        else:
            x = (fun if is_torch else torch.as_tensor(np.ascontiguousarray(fun, dtype='f8'))).to(device=dev, dtype=torch.float64)
            if x.ndim < 1 or x.shape[-1] != n:
                raise ValueError('last dimension must be {:d}, got shape {}'.format(n, tuple(x.shape)))
            out = self._dft_any_size(x.reshape(-1, n).to(torch.complex128))[:, :nh].reshape(tuple(fun.shape[:-1]) + (nh,))
        return out if is_torch else dv.to_host(out)

    def _run(self, fun, backward):
        if not self._native_size(int(self.size)):
            return self._run_any_size(fun, backward)
        torch = _torch()
        is_torch = _is_torch(fun)
        dev = dv.resolve_device(self._device, fun)
        nin = self.size // 2 + 1 if backward else self.size
        nout = self.size if backward else self.size // 2 + 1
        dtype = torch.complex128 if backward else torch.float64
        if is_torch:
            tin = fun.to(device=dev, dtype=dtype)
        else:
            tin = torch.as_tensor(np.ascontiguousarray(fun, dtype='c16' if backward else 'f8')).to(dev)
        if tin.ndim < 1 or tin.shape[-1] != nin:
            raise ValueError('last dimension must be {:d}, got shape {}'.format(nin, tuple(tin.shape)))
        tin = tin.contiguous()
        lead = tuple(tin.shape[:-1])
        nrows = int(np.prod(lead, dtype='i8'))
        tout = torch.empty(lead + (nout,), dtype=torch.float64 if backward else torch.complex128, device=dev)
        if nrows:
            lib, plan, stream = _lib.load(), self._rfft_plan(dev), torch.cuda.current_stream(dev).cuda_stream
            if backward:
                _lib.check(lib.cp_rfft_backward(plan, tin.data_ptr(), tout.data_ptr(), nrows, 1, stream))
            else:
                _lib.check(lib.cp_rfft_forward(plan, tin.data_ptr(), tout.data_ptr(), nrows, stream))
        return tout if is_torch else dv.to_host(tout)

    def forward(self, fun):
        """``rfft(fun, axis=-1)``: (..., size) real -> (..., size // 2 + 1) complex (reference fftlog.py:536-539); numpy in, numpy out;
        device tensor in, device tensor out."""
        return self._run(fun, backward=False)

    def backward(self, fun):
        """``irfft(conj(fun), n=size, axis=-1)`` (reference fftlog.py:541-544): (..., size // 2 + 1) complex -> (..., size) real."""
        return self._run(fun, backward=True)

    def __del__(self):
        try:
            for handle in self.__dict__.get('_rfft_plans', {}).values():
                _lib.load().cp_rfft_plan_destroy(handle)
            self._rfft_plans = {}
        except Exception:
            pass


class NumpyFFTEngine(MI355XFFTEngine):

    """The reference's default engine by name (fftlog.py:533-544), for code written against it: same constructor, same ``forward`` /
    ``backward`` contract, served by this package's device FFTs."""
    name = 'numpy'


class FFTWEngine(MI355XFFTEngine):

    """The reference's :mod:`pyfftw` engine by name (fftlog.py:567-638): same constructor; ``wisdom`` and ``plan`` are validated as there
    and otherwise unused (there is nothing to plan), ``nthreads`` is recorded.  Parity with FFTW itself is unpinned (pyfftw is not in the image)."""
    name = 'fftw'

    def __init__(self, size, nparallel=1, nthreads=None, wisdom=None, plan='measure', **kwargs):
        if str(plan).lower() not in ('estimate', 'measure', 'patient', 'exhaustive'):
            raise ValueError('Plan {} unknown'.format(plan))
        super(FFTWEngine, self).__init__(size, nparallel=nparallel, nthreads=nthreads, **kwargs)
        self.wisdom, self.plan = wisdom, str(plan).lower()


def apply_along_last_axes(func, array, naxes=1, toret=None):
    """``func`` on every block of the last ``naxes`` axes of ``array`` (reference fftlog.py:547-560), results gathered in ``toret`` (made like
    ``array`` if not given; it must have the same leading axes).  The reference reshapes both arrays in place around its loop; this one
    walks views and leaves the callers' arrays alone."""
    array = np.asarray(array)
    if toret is None:
        toret = np.empty_like(array)
    nlead = array.ndim - naxes
    if nlead < 0 or toret.shape[:toret.ndim - naxes] != array.shape[:nlead]:
        raise ValueError('array {} and toret {} must share the axes in front of the last {:d}'.format(array.shape, toret.shape, naxes))
    for index in np.ndindex(*array.shape[:nlead]):
        toret[index] = func(array[index])
    return toret


_ENGINE_NAMES = {'mi355x': MI355XFFTEngine, 'hip': MI355XFFTEngine, 'numpy': NumpyFFTEngine, 'fftw': FFTWEngine}


def get_fft_engine(engine, *args, **kwargs):
    """
    Return the engine (reference fftlog.py:641-663): ``'numpy'`` -> :class:`NumpyFFTEngine`, ``'fftw'`` -> :class:`FFTWEngine` as there, plus
    ``'mi355x'`` / ``'hip'``; every one of them makes :meth:`FFTlog.__call__` the fused HIP kernel.  Anything that is not a string is passed
    through, as in the reference: a foreign object with ``forward`` / ``backward`` methods (protocol of fftlog.py:508-544) makes
    :meth:`FFTlog.__call__` run un-fused around it.
    """
    if isinstance(engine, str):
        cls = _ENGINE_NAMES.get(engine.lower())
        if cls is None:
            raise ValueError('FFT engine {} is unknown'.format(engine))
        return cls(*args, **kwargs)
    if not isinstance(engine, MI355XFFTEngine) and not (callable(getattr(engine, 'forward', None)) and callable(getattr(engine, 'backward', None))):
        raise ValueError('FFT engine {!r} has no forward / backward methods'.format(engine))
    return engine
