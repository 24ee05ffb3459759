"""
FFTLog on MI355X: same classes and call signatures as the reference ``cosmoprimo/fftlog.py``
(:class:`FFTlog`, :class:`HankelTransform`, :class:`PowerToCorrelation`, :class:`CorrelationToPower`,
:class:`TophatVariance`, :class:`GaussianVariance`, :func:`pad`, the Mellin kernels), with
``FFTlog.__call__`` executed by ONE fused HIP kernel (``csrc/cp_fftlog*.{h,hip}``) through the C ABI
of ``libcosmoprimo_amd.so``.

Host side (this file): table setup exactly as reference ``FFTlog._setup`` (fftlog.py:144-184) in
numpy, with the complex log-gamma of the Mellin kernels evaluated by the library's own host routine
(``cp_kernel_eval``), and shape / dtype bookkeeping of ``__call__`` (fftlog.py:198-241).
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


class FFTlog(dv.Copyable):
    r"""
    FFTLog algorithm (https://jila.colorado.edu/~ajsh/FFTLog/) for :math:`G(y) = \int_0^\infty x dx F(x) K(xy)`,
    same constructor and call contract as the reference (fftlog.py:31-248).
    """
    def __init__(self, x, kernel, q=0, minfolds=2, lowring=True, xy=1, check_level=0, engine='mi355x', device=None, **engine_kwargs):
        r"""
        Parameters are those of the reference (fftlog.py:49-92): ``x`` log-spaced input coordinates (1D or one row per
        kernel), ``kernel`` callable(s) returning the Mellin transform, ``q`` tilt(s), ``minfolds``, ``lowring``, ``xy``,
        ``check_level``.

        engine : string, default='mi355x'
            The fused HIP kernel.  The reference's names ``'numpy'`` and ``'fftw'`` are accepted and run the same kernel
            (this package has no CPU path); ``engine_kwargs`` of the FFTW engine (``nthreads``, ``wisdom``, ``plan``) are ignored.

        device : int, string, torch.device, default=None
            GPU holding the plan; defaults to the device of the first input (current CUDA device for numpy inputs).

        rescale_rows : bool (``engine_kwargs``)
            Accepted for compatibility with earlier versions of this package and ignored: the kernel now always keeps the rounding
            of a row relative to its own magnitude (rows that differ by more than a factor 32 from their pair partner are rescaled
            by exact powers of two inside the kernel).
        """
        engine_kwargs.pop('rescale_rows', None)
        self.inparallel = isinstance(kernel, (tuple, list))
        if not self.inparallel:
            kernel = [kernel]
        kernel = list(kernel)
        if np.ndim(q) == 0:
            q = [q] * len(kernel)
        q = list(q)
        self.x = np.asarray(x, dtype='f8')
        if not self.inparallel:
            self.x = self.x[None, :]
        elif self.x.ndim == 1:
            self.x = np.tile(self.x[None, :], (len(kernel), 1))
        if np.ndim(xy) == 0:
            xy = [xy] * len(kernel)
        xy = list(xy)
        if check_level:
            if len(self.x) != len(kernel):
                raise ValueError('x and kernel must of same length')
            if len(q) != len(kernel):
                raise ValueError('q and kernel must be lists of same length')
            if len(xy) != len(kernel):
                raise ValueError('xy and kernel must be lists of same length')
        self._device = device
        self._plan = None
        self._phase = None  # complex phase applied after the (real) postfactor, see PowerToCorrelation(complex=True)
        self._setup(kernel, q, minfolds=minfolds, lowring=lowring, xy=xy, check_level=check_level)
        self.set_fft_engine(engine, **engine_kwargs)

    def _device_copy(self, name, array, dev):
        """Device tensor of a host table of this plan (output coordinates, phases), copied once per device instead of on every call."""
        cache = self.__dict__.setdefault('_device_tables', {})
        key = (name, dev.index)
        if key not in cache:
            cache[key] = _torch().as_tensor(array, device=dev)
        return cache[key]

    def set_fft_engine(self, engine='mi355x', **engine_kwargs):
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
        """Set up u functions and pre/post factors: host numpy, operation for operation as reference fftlog.py:144-184."""
        self.delta = np.log(self.x[:, -1] / self.x[:, 0]) / (self.size - 1)

        nfolds = (self.size * minfolds - 1).bit_length()
        self.padded_size = 2**nfolds
        npad = self.padded_size - self.size
        self.padded_size_in_left, self.padded_size_in_right = npad // 2, npad - npad // 2
        self.padded_size_out_left, self.padded_size_out_right = npad - npad // 2, npad // 2

        if check_level:
            if not np.allclose(np.log(self.x[:, 1:] / self.x[:, :-1]), self.delta[:, None], rtol=1e-3):
                raise ValueError('Input x must be log-spaced')
            if self.padded_size < self.size:
                raise ValueError('Convolution size must be larger than input x size')

        if lowring:
            self.lnxy = np.array([delta / np.pi * np.angle(kernel(q + 1j * np.pi / delta)) for kernel, delta, q in zip(kernels, self.delta, qs)], dtype='f8')
        else:
            self.lnxy = np.log(xy) + self.delta

        self.y = np.exp(self.lnxy - self.delta)[:, None] / self.x[:, ::-1]

        m = np.arange(0, self.padded_size // 2 + 1)
        self.padded_u, self.padded_prefactor, self.padded_postfactor = [], [], []
        self.padded_x = pad(self.x, (self.padded_size_in_left, self.padded_size_in_right), axis=-1, extrap='log')
        self.padded_y = pad(self.y, (self.padded_size_out_left, self.padded_size_out_right), axis=-1, extrap='log')
        prev_kernel, prev_q, prev_delta, prev_u = None, None, None, None
        for kernel, padded_x, padded_y, lnxy, delta, q in zip(kernels, self.padded_x, self.padded_y, self.lnxy, self.delta, qs):
            self.padded_prefactor.append(padded_x**(-q))
            self.padded_postfactor.append(padded_y**(-q))
            if kernel is prev_kernel and q == prev_q and delta == prev_delta:
                u = prev_u
            else:
                u = prev_u = np.asarray(kernel(q + 2j * np.pi / self.padded_size / delta * m))
            self.padded_u.append(u * np.exp(-2j * np.pi * lnxy / self.padded_size / delta * m))
            prev_kernel, prev_q, prev_delta = kernel, q, delta
        self.padded_u = np.array(self.padded_u)
        self.padded_prefactor = np.array(self.padded_prefactor)
        self.padded_postfactor = np.array(self.padded_postfactor)
        self._plan = None

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
        post = self.padded_postfactor
        pre = self.padded_prefactor
        if np.iscomplexobj(pre):
            raise NotImplementedError('complex prefactor (inverse of a complex=True transform) is not supported')
        if np.iscomplexobj(post):
            # complex=True transforms: postfactor = real table x unit phase per kernel; the kernel applies the real
            # table and the phase is applied on the way out
            if self._phase is None:
                raise NotImplementedError('complex postfactor without a per-kernel phase')
            post = np.real(post / self._phase[:, None])
        plan = _Plan(self.size, self.padded_size, pre, post, self.padded_u, dev.index)
        self._plan = (dev.index, plan)
        return plan

    def __call__(self, fun, extrap=0, keep_padding=False):
        """
        Perform the transforms (reference fftlog.py:198-241).

        fun : numpy array or torch CUDA tensor; last dimensions must broadcast against (:attr:`nparallel`, len(x)).
        extrap : 0 (default), number, 'edge', 'log', or a (left, right) tuple.
        keep_padding : return the padded transform.

        Returns ``(y, fftloged)``, numpy for numpy input, torch (same device) for torch input.  Output is float64
        (complex128 for ``complex=True`` transforms) as in the reference.
        """
        torch = _torch()
        is_torch = _is_torch(fun)
        if is_torch:
            dev = self._resolve_device(fun)
            tfun = fun.to(device=dev, dtype=torch.float64)
        else:
            fun = np.asarray(fun)
            if np.iscomplexobj(fun):
                fun = fun.real  # numpy.fft.rfft discards the imaginary part (fftlog.py:540)
            dev = self._resolve_device(None)
            tfun = torch.from_numpy(np.ascontiguousarray(fun, dtype='f8')).to(dev)
        n, nker, npad = self.size, self.nparallel, self.padded_size
        fshape = tuple(tfun.shape)
        if len(fshape) < 1 or fshape[-1] != n:
            raise ValueError('fun last dimension must be {:d}, got shape {}'.format(n, fshape))
        # broadcast against (nker, n), as fun * padded_prefactor does in the reference
        if nker > 1:
            if len(fshape) == 1:
                tfun = tfun[None, :]
            if tfun.shape[-2] not in (1, nker):
                raise ValueError('fun shape {} does not broadcast against ({:d}, {:d})'.format(fshape, nker, n))
            bshape = tuple(tfun.shape[:-2]) + (nker, n)
        else:
            bshape = fshape if len(fshape) >= 2 else (1, n)
            if len(fshape) == 1:
                tfun = tfun[None, :]
        tin = tfun.expand(bshape).contiguous()
        el, er = _split_pair(extrap)
        (cl, vl), (cr, vr) = _extrap_code(el), _extrap_code(er)
        # Rows stay independent inside the kernel (non-finite rows give NaN rows, rows of very different magnitude are rescaled by exact
        # powers of two: csrc/cp_fftlog_body.h, "row independence"), as with numpy's row-by-row FFTs: nothing to screen here, no extra
        # pass over the batch, no host synchronisation.
        nbatch = 1
        for s in bshape[:-2] if nker > 1 else bshape[:-1]:
            nbatch *= s
        nout = npad if keep_padding else n
        oshape = bshape[:-1] + (nout,)
        tout = torch.empty(oshape, dtype=torch.float64, device=dev)
        if nbatch > 0:
            plan = self._get_plan(dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.load().cp_fftlog_execute(plan.handle, tin.data_ptr(), tout.data_ptr(), nbatch, cl, vl, cr, vr, int(bool(keep_padding)),
                                                     stream))
        if self._phase is not None:
            tout = tout * self._device_copy('phase', self._phase, dev)[:, None]
        y = self.padded_y if keep_padding else self.y
        if not self.inparallel:
            y = y[0]
            tout = tout.reshape(fshape[:-1] + (nout,))
        if is_torch:
            return self._device_copy('y_padded' if keep_padding else 'y' + ('' if self.inparallel else '0'), y, dev), tout
        return y, dv.to_host(tout)

    def inv(self):
        """Inverse the transform, in place (reference fftlog.py:243-248, including its padded_x / padded_y quirk)."""
        self.x, self.y = self.y, self.x
        self.__dict__.pop('_device_tables', None)     # device copies of the output coordinates
        self.padded_x, self.padded_y = self.y, self.x
        self.padded_prefactor, self.padded_postfactor = 1 / self.padded_postfactor, 1 / self.padded_prefactor
        self.padded_u = 1 / self.padded_u.conj()
        if self._phase is not None:
            raise NotImplementedError('inv() of a complex=True transform is not supported')
        self._plan = None


class HankelTransform(FFTlog):
    """Hankel transform with Bessel kernels (reference fftlog.py:252-280)."""
    def __init__(self, x, nu=0, **kwargs):
        if np.ndim(nu) == 0:
            kernel = BesselJKernel(nu)
        else:
            kernel = [BesselJKernel(nu_) for nu_ in nu]
        FFTlog.__init__(self, x, kernel, **kwargs)
        self.padded_prefactor *= self.padded_x**2


class PowerToCorrelation(FFTlog):
    r"""
    Power spectrum to correlation function (reference fftlog.py:284-330):
    :math:`\xi_\ell(s) = \frac{(-i)^\ell}{2\pi^2} \int dk k^2 P_\ell(k) j_\ell(ks)`.
    """
    def __init__(self, k, ell=0, q=0, complex=False, **kwargs):
        if np.ndim(ell) == 0:
            kernel = SphericalBesselJKernel(ell)
        else:
            kernel = [SphericalBesselJKernel(ell_) for ell_ in ell]
        FFTlog.__init__(self, k, kernel, q=1.5 + q, **kwargs)
        self.padded_prefactor *= self.padded_x**3 / (2 * np.pi)**1.5
        ell = np.atleast_1d(ell)
        if complex:
            phase = (-1j)**ell
            self._phase = phase
        else:
            phase = (-1)**(ell // 2)
        self.padded_postfactor = self.padded_postfactor * phase[:, None]


class CorrelationToPower(FFTlog):
    r"""
    Correlation function to power spectrum (reference fftlog.py:334-377):
    :math:`P_\ell(k) = 4\pi i^\ell \int ds s^2 \xi_\ell(s) j_\ell(ks)`.
    """
    def __init__(self, s, ell=0, q=0, complex=False, **kwargs):
        if np.ndim(ell) == 0:
            kernel = SphericalBesselJKernel(ell)
        else:
            kernel = [SphericalBesselJKernel(ell_) for ell_ in ell]
        FFTlog.__init__(self, s, kernel, q=1.5 + q, **kwargs)
        self.padded_prefactor *= self.padded_x**3 * (2 * np.pi)**1.5
        ell = np.atleast_1d(ell)
        if complex:
            phase = (1j)**ell
            self._phase = phase
        else:
            phase = (-1)**(ell // 2)
        self.padded_postfactor = self.padded_postfactor * phase[:, None]


class TophatVariance(FFTlog):
    """Variance in a tophat window (reference fftlog.py:381-405)."""
    def __init__(self, k, q=0, **kwargs):
        kernel = TophatSqKernel(ndim=3)
        FFTlog.__init__(self, k, kernel, q=1.5 + q, **kwargs)
        self.padded_prefactor *= self.padded_x**3 / (2 * np.pi**2)


class GaussianVariance(FFTlog):
    """Variance in a Gaussian window (reference fftlog.py:409-433)."""
    def __init__(self, k, q=0, **kwargs):
        kernel = GaussianSqKernel()
        FFTlog.__init__(self, k, kernel, q=1.5 + q, **kwargs)
        self.padded_prefactor *= self.padded_x**3 / (2 * np.pi**2)


# ---------------------------------------------------------------------------------------------
# engines (reference fftlog.py:508-663): the reference's engine protocol splits the transform into
# forward / backward FFTs; here the whole of FFTlog.__call__ is one kernel, so the engine object only
# names the backend.
# ---------------------------------------------------------------------------------------------
class BaseFFTEngine(object):

    """FFT engine descriptor (reference fftlog.py:508-531); does not touch OMP_NUM_THREADS."""

    def __init__(self, size, nparallel=1, nthreads=None):
        self.size = size
        self.nparallel = nparallel
        self.nthreads = nthreads


class MI355XFFTEngine(BaseFFTEngine):

    """The fused HIP FFTLog kernel (pad, prefactor, FFT, u, inverse FFT, postfactor, crop in one launch)."""
    name = 'mi355x'

    def __init__(self, size, nparallel=1, nthreads=None, **kwargs):
        super(MI355XFFTEngine, self).__init__(size, nparallel=nparallel, nthreads=nthreads)
        _lib.load()  # fail loudly when the HIP library is missing


def get_fft_engine(engine, *args, **kwargs):
    """
    Return the engine (reference fftlog.py:641-663).  ``'mi355x'`` / ``'hip'`` and, for drop-in use of code written
    against the reference, ``'numpy'`` / ``'fftw'`` all select the fused HIP kernel.  Engine *objects* implementing the
    reference's forward / backward protocol cannot be fused and are rejected.
    """
    if isinstance(engine, str):
        if engine.lower() in ('mi355x', 'hip', 'numpy', 'fftw'):
            return MI355XFFTEngine(*args, **kwargs)
        raise ValueError('FFT engine {} is unknown'.format(engine))
    if isinstance(engine, MI355XFFTEngine):
        return engine
    raise NotImplementedError('custom forward/backward FFT engines cannot be used with the fused MI355X kernel')
