"""
BAO filters on MI355X: ``wallish2018`` and ``brieden2022`` behind the reference's registry / factory
(cosmoprimo/bao_filter.py: metaclass registry :22-31, base class :34-169, Wallish2018 :345-431, Brieden2022 :434-509,
factory :912-921).

Data path (device, through the C ABI): the input interpolator evaluated as rows ``(ncol, nk)``; for wallish2018 the fused
``log(k P)`` -> batched DST-II (``cp_dst``), clamped-spline second derivatives (``cp_spline`` operator), per-column peak box
(arg-max reductions), gap spline (``cp_gap_spline``), batched DST-III with fused ``exp(.)/k``, final clamped splice as a
fixed operator; for brieden2022 the envelope interpolation and the re-sampling are fixed operators applied to all columns.
Host side (numpy, small, once per filter): grids, masks, the least-squares fit and the peak search of ``_prepare`` (341 samples
of ONE fiducial cosmology), construction of the operators.
"""
import numpy as np

from . import _lib
from . import _device as dv
from . import power as pwmod
from .cosmology import Cosmology, Fourier
from .dst import DST
from .interpolator import _quadratic_interp_operator, _simpson_weights, _bspline_basis  # noqa: F401
from .interpolator import (PowerSpectrumInterpolator1D, PowerSpectrumInterpolator2D, CorrelationFunctionInterpolator1D,
                           CorrelationFunctionInterpolator2D)
from .spline import LinearOperator, SplicedClampedSpline, dense_operator


def _host_value(x):
    """float or numpy array from a float / array / device tensor."""
    return dv.to_host(x) if dv.is_torch(x) else np.asarray(x, dtype='f8')


class RegisteredPowerSpectrumBAOFilter(type):

    """Metaclass registering :class:`BasePowerSpectrumBAOFilter`-derived classes by ``name`` (reference bao_filter.py:22-31)."""
    _registry = {}

    def __new__(meta, name, bases, class_dict):
        cls = super().__new__(meta, name, bases, class_dict)
        meta._registry[cls.name] = cls
        return cls


class BasePowerSpectrumBAOFilter(dv.Copyable, metaclass=RegisteredPowerSpectrumBAOFilter):

    """Base BAO filter for power spectrum (reference bao_filter.py:34-169)."""
    name = 'base'

    def __init__(self, pk_interpolator, cosmo=None, cosmo_fid=None, **kwargs):
        self._cosmo_fid = cosmo_fid
        self._cosmo = cosmo
        self.pk_interpolator = pk_interpolator
        self.device = pk_interpolator.device
        self.set_k(**kwargs)
        self.set_pk(pk_interpolator, cosmo=cosmo)
        self._prepare()
        self._compute()
        self._finalize()

    def _prepare(self):
        """Anything that can be done once."""

    def set_k(self, nk=1024):
        """Wavenumbers where the power spectrum is evaluated (reference bao_filter.py:81-90)."""
        self.k = np.geomspace(self.pk_interpolator.extrap_kmin, self.pk_interpolator.extrap_kmax, nk)

    def _rows(self, k):
        """The input power spectrum at ``k`` as device rows (ncol, nk) + the reference output shape (nk, ...)."""
        interp = self.pk_interpolator
        if isinstance(interp, PowerSpectrumInterpolator2D):
            rows = interp._rows_z(interp.z, ignore_growth=True)(np.asarray(k, dtype='f8'))      # (batch..., nz, nk)
        else:
            rows = interp._rows(np.asarray(k, dtype='f8'))                                       # (ncol, nk)
        lead = tuple(rows.shape[:-1])
        return rows.reshape(-1, rows.shape[-1]), lead

    def set_pk(self, pk_interpolator, cosmo=None):
        """Set input power spectrum to remove BAO wiggles from (reference bao_filter.py:92-102)."""
        if cosmo is not None:
            self._cosmo = cosmo
        self.pk_interpolator = pk_interpolator
        self._pk_rows, self._lead = self._rows(self.k)
        if isinstance(pk_interpolator, PowerSpectrumInterpolator2D):
            self.shape = (self.k.size,) + self._lead[-1:] if len(self._lead) == 1 else self._lead[:-1] + (self.k.size, self._lead[-1])
        else:
            cs = pk_interpolator._colshape()
            self.shape = (self.k.size,) + cs

    def _finalize(self):
        """The host copies are made when ``pk`` / ``pknow`` are first read: a caller that keeps working on the device (``pknow_rows``) or
        wants ``pknow`` only does not pay for two device-to-host copies of the batch."""
        self._host = {}

    def _to_host(self, name):
        """Device rows -> array of the reference's shape (k first, then columns)."""
        if name not in self._host:
            a = dv.to_host(getattr(self, '_{}_rows'.format(name)))
            if isinstance(self.pk_interpolator, PowerSpectrumInterpolator2D) and len(self._lead) > 1:
                self._host[name] = np.moveaxis(a.reshape(self._lead + (self.k.size,)), -1, -2)
            else:
                self._host[name] = a.T.reshape(self.shape)
        return self._host[name]

    @property
    def pk(self):
        """Input power spectrum at :attr:`k`."""
        return self._to_host('pk')

    @property
    def pknow(self):
        """Power spectrum without BAO wiggles at :attr:`k`."""
        return self._to_host('pknow')

    @property
    def pknow_rows(self):
        """:attr:`pknow` as it sits in HBM: tensor (ncolumns, nk), one row per input power spectrum."""
        return self._pknow_rows

    def __call__(self, pk_interpolator, cosmo=None):
        self.set_pk(pk_interpolator, cosmo=cosmo)
        self._compute()
        self._finalize()
        return self

    @property
    def wiggles(self):
        """Extracted wiggles."""
        return self.pk / self.pknow

    def smooth_pk_interpolator(self, **kwargs):
        """Smooth (no-wiggle) power spectrum interpolator (reference bao_filter.py:115-129)."""
        return self.pk_interpolator.clone(k=self.k, pk=self.pknow, **kwargs)

    def smooth_xi_interpolator(self, **kwargs):
        """Smooth correlation function through FFTLog (reference bao_filter.py:131-146): the ``to_xi`` output of the smooth interpolator."""
        return self.smooth_pk_interpolator().to_xi(**kwargs)

    @property
    def cosmo(self):
        """Cosmology."""
        if self._cosmo is None:
            self._cosmo = Cosmology(engine='eisenstein_hu', device=self.device)
        return self._cosmo

    @property
    def cosmo_fid(self):
        """Reference cosmology."""
        if self._cosmo_fid is None:
            self._cosmo_fid = Cosmology(engine='eisenstein_hu', device=self.device)
        return self._cosmo_fid

    def rs_drag_ratio(self):
        """Ratio of ``cosmo.rs_drag`` to the fiducial one, 1 if no ``cosmo`` (reference bao_filter.py:161-169, hard-coded fiducial included)."""
        if self._cosmo is None:
            return 1.
        if self._cosmo_fid is None:
            rs_drag_fid = 100.91463132327911
        else:
            key = id(self._cosmo_fid)           # one fiducial cosmology per filter object, as a rule: its sound horizon is read back once
            if getattr(self, '_rs_drag_fid', (None, None))[0] != key:
                self._rs_drag_fid = (key, _host_value(self.cosmo_fid.rs_drag))
            rs_drag_fid = self._rs_drag_fid[1]
        rs_drag = self.cosmo.rs_drag
        if dv.is_torch(rs_drag) and rs_drag.ndim and np.ndim(rs_drag_fid) == 0:
            return rs_drag / float(rs_drag_fid)     # a batch of cosmologies: one ratio per cosmology, left on the device
        ratio = _host_value(rs_drag) / _host_value(rs_drag_fid)
        return float(ratio) if np.ndim(ratio) == 0 else ratio

    def _batch_size(self):
        """Number of cosmologies whose spectra the input holds row by row, None for one cosmology (the reference's case, any number of columns): the
        batch size of ``cosmo`` when a batched cosmology is given (one rs_drag ratio, one no-wiggle template per cosmology)."""
        nb = getattr(self._cosmo, 'batch_size', None) if self._cosmo is not None else None
        if nb is None:
            return None
        ncol = self._pk_rows.shape[0]
        if ncol % nb:
            raise ValueError('the input holds {:d} spectra, the cosmology {:d} parameter sets'.format(ncol, nb))
        return nb

    def _columns_per_cosmology(self):
        return self._pk_rows.shape[0] // self._batch_size()

    def _per_row(self, values):
        """Per-cosmology host values (B,) -> one per row of the input spectra."""
        return np.repeat(np.asarray(values), self._columns_per_cosmology())

    def _scalar_rs_drag_ratio(self):
        """The ratio for filters whose operator is built for ONE ratio (a dense matrix per value): a batch of cosmologies is refused."""
        ratio = self.rs_drag_ratio()
        if np.ndim(ratio):
            raise NotImplementedError('{} builds one dense operator per rs_drag ratio: give it one cosmology at a time (wallish2018 and brieden2022 '
                                      'take batches of cosmologies)'.format(self.__class__.__name__))
        return float(ratio)


# ... whose epilogue can also run the second-derivative / box step (cp_dst_forward_analytic_box).  NOT the default: measured on 32 768-vector chunks the
# transform grows from 2.35 to 3.11 ms for the 0.62 ms of cp_wallish_dd_box it replaces (the solve is a chain of dependent steps: at the two
# workgroups per CU of the transform it is not hidden, and it holds the transform's registers and LDS while it runs) -- profiles/r4_wallish_box_in_transform.txt
_TRANSFORM_FINDS_BOXES = False
# everything behind the forward transform (second derivatives + box, inverse transform, spliced spline + damping) as ONE kernel, the transformed rows never
# leaving the CU (cp_wallish_tail): batches too large for the second derivatives to be kept (they are not written then)
_TAIL_IN_ONE_KERNEL = True
# ... and the forward transform with its spectra in the same kernel (cp_wallish_full): the filter of a batch of analytic cosmologies is ONE launch behind the
# sigma8 normalisation; the coefficient sequences are not written then (like the second derivatives, they are kept for small batches only)
_ALL_IN_ONE_KERNEL = True
_TRANSFORM_EVALUATES_SPECTRA = True      # wallish2018 on batches of analytic cosmologies: cp_dst_forward_analytic (False: evaluation kernel, then transform)


_DONE = object()      # second slot of Wallish2018PowerSpectrumBAOFilter._log_k_rows: the whole filter has run (cp_wallish_full)


class Wallish2018PowerSpectrumBAOFilter(BasePowerSpectrumBAOFilter):

    """
    Sine-transform the power spectrum to real space, cut the BAO peak, re-interpolate with a spline (reference bao_filter.py:345-431;
    https://arxiv.org/pdf/1810.02800.pdf App. D), with the reference's hand-tuned margins.
    """
    name = 'wallish2018'
    _nlin = 4096
    _margin_first, _margin_second, _offset = 20, 5, (-10, 20)

    _ops_cache = {}

    def _operators(self):
        """Plans that depend on the grids only (built once per (k range, nk, device) and shared by all filter instances)."""
        key = (float(self.pk_interpolator.extrap_kmin), float(self.pk_interpolator.extrap_kmax), self.k.size, self.device.index)
        if getattr(self, '_ops', None) is not None and getattr(self, '_ops_key', None) == key:
            return self._ops          # (a call with another interpolator may bring another k range: the reference takes it from the current one, :363)
        self._ops_key = key
        if key in self._ops_cache:
            self._ops = self._ops_cache[key]
            return self._ops
        kmin = self.pk_interpolator.extrap_kmin
        klin = np.linspace(kmin, 2., self._nlin)
        mask = (klin > 1e-2) & (klin < 1.5)                                                   # :415
        mask_left, mask_right = self.k < 5e-4, self.k > 2.                                    # :417
        knots = np.concatenate([self.k[mask_left], klin[mask], self.k[mask_right]], axis=0)
        # The clamped spline through the spliced knots (:420): their values are contiguous pieces of two arrays that already sit in HBM -- P at
        # self.k (left and right pieces) and the smoothed spectrum on the linear grid (middle piece).  One kernel solves the spline's tridiagonal
        # system per vector in LDS, evaluates it at self.k and applies the damping (SplicedClampedSpline); where its scheme does not fit, the
        # spline is applied as one dense operator on each array (its columns moved to the positions of the pieces, zero elsewhere).
        n_left, n_mid = int(mask_left.sum()), int(mask.sum())
        try:
            pieces = [(0, 0, n_left), (1, int(np.flatnonzero(mask)[0]), n_mid), (0, int(np.flatnonzero(mask_right)[0]), int(mask_right.sum()))]
            splice = SplicedClampedSpline(knots, pieces, self.k, device=self.device)
        except (NotImplementedError, IndexError):
            w = dense_operator(knots, self.k, bc='clamped')                                    # (nk, nknots)
            w_pk, w_lin = np.zeros((self.k.size, self.k.size)), np.zeros((self.k.size, self._nlin))
            w_pk[:, np.flatnonzero(mask_left)] = w[:, :n_left]
            w_pk[:, np.flatnonzero(mask_right)] = w[:, n_left + n_mid:]
            w_lin[:, np.flatnonzero(mask)] = w[:, n_left:n_left + n_mid]
            splice = (LinearOperator.dense(w_pk, device=self.device), LinearOperator.dense(w_lin, device=self.device))
        tophat = np.ones_like(self.k)
        m = self.k > 1.
        tophat[m] *= np.exp(-20.**2 * (self.k[m] / 1. - 1.)**2)                                # :426-431
        self._ops = dict(klin=klin, dst=DST(self._nlin, kx=klin, device=self.device), dd=None, splice=splice,
                         tophat=dv.to_device(tophat, self.device))
        if len(self._ops_cache) >= 8:      # (a sampler that varies the k range or nk: the plans of the grids it has left go with their last filter)
            self._ops_cache.clear()
        self._ops_cache[key] = self._ops
        return self._ops

    def _box(self, dd):
        """Per-column index box to cut, from the two maxima of the second derivative (reference bao_filter.py:390-394). dd : (ncol, n)."""
        torch = dv.torch()
        mf, ms, off = self._margin_first, self._margin_second, self._offset
        dd = dd.contiguous()
        box = torch.empty((dd.shape[0], 2), dtype=torch.int32, device=dd.device)
        _lib.check(_lib.load().cp_wallish_box(dd.data_ptr(), dd.shape[0], dd.shape[1], mf, ms, off[0], off[1], box.data_ptr(), self.device.index,
                                              dv.stream_of(self.device)))
        return box

    _keep_second_derivatives = 256      # sequences up to which the second derivatives are kept as ``_dd`` (they are only looked at, never used again)

    def _second_derivatives_and_box(self, y, ops):
        """``spline(x, nu=2)`` of the clamped spline through each sequence (reference bao_filter.py:377-382) and the box between its two maxima
        (:390-394), as one kernel that solves the tridiagonal system in LDS (``cp_wallish_dd_box``), which also rewrites the boxes in place (:395-405).  y : (nseq, n).  Returns (dd or None, box, whether the boxes are
        rewritten already)."""
        torch = dv.torch()
        nseq, n = y.shape
        mf, ms, off = self._margin_first, self._margin_second, self._offset
        if n in (1024, 2048):
            dd = torch.empty_like(y) if nseq <= self._keep_second_derivatives else None
            box = torch.empty((nseq, 2), dtype=torch.int32, device=y.device)
            if nseq:
                _lib.check(_lib.load().cp_wallish_dd_box(y.data_ptr(), nseq, n, mf, ms, off[0], off[1], box.data_ptr(), dd.data_ptr() if dd is not None else None,
                                                         y.data_ptr(), self.device.index, dv.stream_of(self.device)))      # (the boxes are rewritten in place, :395-405)
            return dd, box, True
        if ops['dd'] is None:       # other lengths: the second derivatives as a spline operator, then the searches
            x = 1. + np.arange(n)
            ops['dd'] = LinearOperator.spline(x, x, bc='clamped', nu=2, device=self.device)
        dd = ops['dd'](y)
        return dd, self._box(dd), False

    def _full(self, engine, bg, pk, dst):
        """The whole filter of a batch of analytic cosmologies as one kernel (``cp_wallish_full``: the spectra evaluated into the forward transform, the
        coefficients taken through the rest on the CU); False where the plans or the parameters are not the ones the kernel is written for."""
        torch = dv.torch()
        from .background import DEFAULTS as bg_defaults
        from .power import PK_DEFAULTS
        ops = self._operators()
        if dst.n != 4096 or engine not in _lib.ENGINES or not isinstance(ops['splice'], SplicedClampedSpline):
            return False
        rows = self._pk_rows.contiguous()
        cbg, n1, keep1 = dv.pack_params(_lib.BG_PARAMS, bg, bg_defaults, self.device)
        cpk, n2, keep2 = dv.pack_params(_lib.PK_PARAMS, pk, PK_DEFAULTS, self.device)
        sizes = {n for n in (n1, n2) if n is not None}
        if len(sizes) != 1 or sizes.pop() != rows.shape[0]:
            return False
        ncosmo = rows.shape[0]
        lib = _lib.load()
        mf, ms, off = self._margin_first, self._margin_second, self._offset
        box = torch.empty((2 * ncosmo, 2), dtype=torch.int32, device=self.device)
        out = torch.empty_like(rows)
        work = torch.empty(int(lib.cp_dst_forward_analytic_workspace_bytes(ncosmo)), dtype=torch.uint8, device=self.device)
        nu, keep_nu = dv.ncdm_arg(bg, ncosmo)
        status = lib.cp_wallish_full(dst._handle, ops['splice']._handle, _lib.ENGINES[engine], ncosmo, dv.as_void_p(cbg), 0, nu, dv.as_void_p(cpk), rows.data_ptr(), rows.shape[1],
                                     mf, ms, off[0], off[1], ops['tophat'].data_ptr(), box.data_ptr(), None, out.data_ptr(), work.data_ptr(), dv.stream_of(self.device))
        if status == _lib.CP_EUNSUPPORTED:
            return False
        _lib.check(status)
        self._dd, self._boxes = None, [box[0::2], box[1::2]]
        self._even_now = self._odd_now = None      # (not written for batches this large: the sequences never leave the CU)
        self._pknow_rows = out
        return True

    def _tail(self, ffted, y, ops):
        """Everything behind the forward transform in one kernel (``cp_wallish_tail``, reference bao_filter.py:373-431); False where the plans are not the
        ones the kernel is written for (the three separate calls then)."""
        torch = dv.torch()
        pk = self._pk_rows.contiguous()
        if pk.shape[0] != ffted.shape[0] or not ffted.is_contiguous():
            return False
        mf, ms, off = self._margin_first, self._margin_second, self._offset
        box = torch.empty((y.shape[0], 2), dtype=torch.int32, device=y.device)
        out = torch.empty_like(pk)
        status = _lib.load().cp_wallish_tail(ops['dst']._handle, ops['splice']._handle, ffted.data_ptr(), pk.data_ptr(), pk.shape[1], pk.shape[0], mf, ms, off[0], off[1],
                                             ops['tophat'].data_ptr(), box.data_ptr(), out.data_ptr(), dv.stream_of(self.device))
        if status == _lib.CP_EUNSUPPORTED:
            return False
        _lib.check(status)
        self._dd, self._boxes = None, [box[0::2], box[1::2]]
        self._even_now, self._odd_now = y[0::2], y[1::2]      # (the boxes are rewritten in place)
        self._pknow_rows = out
        return True

    def _log_k_rows(self, klin, dst=None):
        """log(k_lin P(k_lin)) rows (ncol, 4096) of a batch of cosmologies of an analytic engine, written by the evaluation kernel term by term
        (``cp_power_eval``, CP_PK_LOG_K_MATTER) -- the transform then reads its input as it is, without 4096 logarithms per vector.  None for any
        other input (tabulated spectra, several redshifts, rescaled amplitudes): the transform takes the logarithm itself."""
        interp = self.pk_interpolator
        call = getattr(interp, '_interp', None) if getattr(interp, 'is_from_callable', False) else None
        rs = getattr(interp, '_rsigma8sq', None)
        if not isinstance(interp, PowerSpectrumInterpolator2D) or not hasattr(call, 'analytic_engine') or np.size(interp.z) != 1 or not (
                isinstance(rs, float) and rs == 1.):
            return None, None
        engine, bg, pk = call.analytic_engine()
        if dst is not None and _TRANSFORM_EVALUATES_SPECTRA:      # a batch of cosmologies: the transform kernel evaluates the spectra itself
            ncol = self._pk_rows.shape[0]
            if _ALL_IN_ONE_KERNEL and _TAIL_IN_ONE_KERNEL and 2 * ncol > self._keep_second_derivatives and self._full(engine, bg, pk, dst):
                return None, _DONE
            if _TRANSFORM_FINDS_BOXES and 2 * ncol > self._keep_second_derivatives and dst.n == 4096:
                # ... and runs the next step on the coefficients it holds (second derivatives, boxes, boxes rewritten): nothing is kept of
                # the second derivatives, as for every large batch
                res = dst.forward_analytic(engine, bg, pk, split=True, box=(self._margin_first, self._margin_second, self._offset[0], self._offset[1]))
                if res is not None and res[0].shape[0] == ncol:
                    return None, res
            ffted = dst.forward_analytic(engine, bg, pk, split=True)
            if ffted is not None and ffted.shape[0] == ncol:
                return None, ffted
        rows = pwmod.analytic(engine, 'log_k_matter', klin, bg=bg, pk=pk, device=self.device)
        return (rows if rows.ndim == 2 and rows.shape[0] == self._pk_rows.shape[0] else None), None

    def _compute(self):
        torch = dv.torch()
        ops = self._operators()
        lib = _lib.load()
        # dst(log(k P)), type 2, ortho, written as [even-indexed | odd-indexed] coefficients: seen as (2 ncol, 2048) the two sequences of
        # every vector are consecutive rows, and share the operators (x_even = x_odd = 1 + arange(2048), bao_filter.py:374-375)
        logkp, ffted = self._log_k_rows(ops['klin'], dst=ops['dst'])
        if ffted is _DONE:      # the whole filter ran in one kernel (cp_wallish_full)
            return
        solved = None
        if isinstance(ffted, tuple):                                      # ... which has also found and rewritten the boxes
            ffted, solved = ffted
        if ffted is not None:                                             # a batch of analytic cosmologies: evaluated inside the transform
            pass
        elif logkp is not None:                                           # analytic engine: log(k_lin P) straight from the evaluation kernel
            ffted = ops['dst'](logkp, split=True)
        else:
            rows, _ = self._rows(ops['klin'])                             # P(k_lin), (ncol, 4096)
            ffted = ops['dst'](rows, fused=True, split=True)
        y = ffted.view(2 * ffted.shape[0], ffted.shape[1] // 2)
        if (_TAIL_IN_ONE_KERNEL and solved is None and isinstance(ops['splice'], SplicedClampedSpline) and ffted.shape[1] == 4096 and
                y.shape[0] > self._keep_second_derivatives and self._tail(ffted, y, ops)):
            return
        dd, box, removed = (None, solved, True) if solved is not None else self._second_derivatives_and_box(y, ops)
        out = y      # in place: the kept coefficients stay where they are, only the boxes are rewritten
        if not removed:
            _lib.check(lib.cp_gap_spline(y.data_ptr(), box.data_ptr(), out.data_ptr(), y.shape[0], y.shape[1], self.device.index, dv.stream_of(self.device)))
        self._dd, self._boxes = None if dd is None else [dd[0::2], dd[1::2]], [box[0::2], box[1::2]]
        self._even_now, self._odd_now = out[0::2], out[1::2]
        pknow_lin = ops['dst'](out.view(ffted.shape), inverse=True, fused=True, split=True)          # exp(idst(.)) / k_lin
        pk = self._pk_rows.contiguous()
        if isinstance(ops['splice'], SplicedClampedSpline):     # spline through the spliced knots at self.k, then pk / ((pk / pknow - 1) tophat + 1) (:420-431)
            out = ops['splice'](pk, pknow_lin, tophat=ops['tophat'])
        else:
            from_pk, from_lin = ops['splice'][0](pk), ops['splice'][1](pknow_lin)
            out = torch.empty_like(pk)
            _lib.check(lib.cp_wallish_finish(pk.data_ptr(), from_pk.data_ptr(), from_lin.data_ptr(), ops['tophat'].data_ptr(), out.data_ptr(), pk.shape[0],
                                             pk.shape[1], self.device.index, dv.stream_of(self.device)))
        self._pknow_rows = out


def _local_maxima(x):
    """Indices of strict local maxima, plateaus reported at their midpoint (what scipy.signal.find_peaks returns without conditions)."""
    out, i, n = [], 1, x.size
    while i < n - 1:
        if x[i - 1] < x[i]:
            j = i
            while j < n - 1 and x[j + 1] == x[i]:
                j += 1
            if j < n - 1 and x[j + 1] < x[i]:
                out.append((i + j) // 2)
            i = j + 1
        else:
            i += 1
    return np.array(out, dtype=int)


def _fiducial_wiggles(cosmo_fid, k_fid):
    """
    Wiggles of the fiducial cosmology on ``k_fid`` (z = 0), as brieden2022 and peakaverage define them (reference bao_filter.py:461-472,
    536-548): the ratio of its P(k) to its Eisenstein-Hu no-wiggle P(k), divided by a broad-band correction -- the cubic in k
    times 1/k (powers k^-1 .. k^2) fitted to the ratio with weights k^2 and pinned to its value and first difference at both ends.
    Returns ``(ratio, correction)``; ``ratio / correction`` oscillates around 1 and equals 1 (to rounding) at the two first and two
    last samples.
    """
    # (a sampler builds a filter per step with ONE fiducial cosmology: its wiggles are kept with the cosmology object, per set of wavenumbers)
    try:
        kept = _fiducial_wiggles_kept.setdefault(cosmo_fid, {})
    except TypeError:      # (an object that takes no weak reference: nothing kept)
        kept = {}
    # (another engine set on the same object: computed again -- the entry holds a weak reference to the engine it was computed with and is valid
    # for that very object only: an id() alone is reused once the engine is freed)
    import weakref
    engine = getattr(cosmo_fid, '_engine', None)
    key = k_fid.tobytes()
    entry = kept.get(key)
    if entry is None or entry[0] is None or entry[0]() is not engine:
        if len(kept) > 8:
            kept.clear()
        pk = np.asarray(Fourier(cosmo_fid).pk_interpolator()(k_fid, z=0.), dtype='f8')
        pknow = np.asarray(Fourier(cosmo_fid, engine='eisenstein_hu_nowiggle', set_engine=False).pk_interpolator()(k_fid, z=0.), dtype='f8')
        ratio = pk / pknow
        powers = k_fid[None, :]**np.arange(-1., 3.)[:, None]                      # (4, n)
        ends = _end_constraints(k_fid.size, order=2)                              # value and first difference at either end, as rows
        correction = _constrained_lsq_operator(powers, k_fid**2, powers.dot(ends.T), ends).dot(ratio)
        try:
            kept[key] = (weakref.ref(engine), ratio, correction)
        except TypeError:
            kept[key] = (None, ratio, correction)      # (never valid again: computed at every call)
    _, ratio, correction = kept[key]
    return ratio.copy(), correction.copy()


import weakref      # noqa: E402
_fiducial_wiggles_kept = weakref.WeakKeyDictionary()


def _wiggle_extrema(residual, start):
    """
    Indices of the local maxima and of the local minima of ``residual[start:]`` (what ``scipy.signal.find_peaks`` reports for the
    series and for its negative), with the tie at the end of the series settled by rule instead of by rounding: the fit behind
    ``residual`` pins its last two samples to the same value, so in exact arithmetic the series ends on a two-sample plateau,
    and whether the first of the two counts as an extremum depends on the last bit of the fit (in the reference as well: its lists
    for the default fiducial cosmology hold that sample as a maximum, tests/golden/bao.npz; the knot moves the smooth P(k) by up
    to 4e-4 above k = 0.36 h/Mpc).  Here it is an extremum of the kind the series approaches the plateau from: a maximum when it
    rises into it, a minimum when it falls into it -- which is what the reference finds for its default fiducial cosmology.
    """
    x = np.array(residual[start:], dtype='f8')
    n = x.size
    x[n - 2] = x[n - 1] = 0.5 * (x[n - 2] + x[n - 1])                          # the plateau the constraints impose
    found = []
    for sign in (1., -1.):
        ix = _local_maxima(sign * x)
        if sign * x[n - 3] < sign * x[n - 2] and PLATEAU_EXTREMUM:
            ix = np.append(ix, n - 2)
        found.append(ix + start)
    return found


# The sample in front of the plateau as an extremum (see _wiggle_extrema): True, the rule above -- what the reference finds for its default fiducial
# cosmology and for about half of any others (for the rest its fit's last bit falls the other way and it has no extremum there: pknow differs by ~1e-4 above
# k = 0.36 h/Mpc); False: never.  tests/test_filter_fuzz_gpu.py checks both readings against the reference's outputs for random fiducial cosmologies.
PLATEAU_EXTREMUM = True


def _envelope_operator(k_fid, peaks):
    """
    The mean of the quadratic ``interp1d`` through the maxima and through the minima (reference bao_filter.py:482-488) as a matrix acting on the
    samples at ``k_fid``: (n, n), zero but for the columns of the extrema; those columns (ascending positions in ``k_fid``) and the operator
    restricted to them, one row per extremum, (len(columns), n).
    """
    n = k_fid.size
    M = np.zeros((n, n))
    for ix in peaks:
        ix = np.asarray(ix) % n
        M[:, ix] += 0.5 * _quadratic_interp_operator(k_fid[ix], k_fid)
    columns = np.unique(np.concatenate([np.asarray(ix) % n for ix in peaks]))
    return M, columns, np.ascontiguousarray(M[:, columns].T)


_envelope_operators = {}


class Brieden2022PowerSpectrumBAOFilter(BasePowerSpectrumBAOFilter):

    """
    Average the minima and maxima envelopes of the wiggles (reference bao_filter.py:434-509; https://arxiv.org/abs/2204.11868 App. D).
    ``cosmo_fid`` must be provided, with an engine.
    """
    name = 'brieden2022'

    @property
    def cosmo_fid(self):
        """Reference cosmology."""
        if self._cosmo_fid is None:
            raise ValueError('cosmo_fid must be provided, with an engine')
        return self._cosmo_fid

    def _prepare(self):
        """Fiducial products (reference bao_filter.py:461-480), host numpy on the 341 samples of 1e-3 <= k <= 1: the wiggles of the
        fiducial cosmology, the knots of their two envelopes (first and last sample added as end knots) and the envelope operator."""
        self.kmask_fid = (self.k >= 1e-3) & (self.k <= 1.)
        self.k_fid = self.k[self.kmask_fid]
        ratio, correction = _fiducial_wiggles(self.cosmo_fid, self.k_fid)
        self.pknow_correction = correction[:, None]
        self.ratio_fid = (ratio / correction)[:, None]
        start = np.searchsorted(self.k_fid, 0.02, side='right') + 1            # extrema are looked for above k = 0.02 only
        last = self.k_fid.size - 1
        self.ik_fid_peaks = [np.concatenate([[0] if ix[0] > 0 else [], ix, [-1] if ix[-1] < last else []]).astype(int)
                             for ix in _wiggle_extrema(self.ratio_fid[:, 0], start)]
        self._set_envelope_operator()

    def _kfid_index(self):
        """Positions of ``k_fid`` in ``k`` as a device index list (a boolean mask would make torch count its entries on the host at every use)."""
        if getattr(self, '_kfid_index_cache', None) is None:
            self._kfid_index_cache = dv.upload(np.flatnonzero(self.kmask_fid), self.device)
        return self._kfid_index_cache

    def _set_envelope_operator(self):
        """``_interp`` (reference bao_filter.py:482-488) is linear in y for fixed peak indices: one dense (341 x 341) operator."""
        # (the operator depends on the wavenumbers and the positions of the fiducial extrema only: filter objects of one fiducial cosmology share it)
        key = (self.k_fid.tobytes(), tuple(np.asarray(ix).tobytes() for ix in self.ik_fid_peaks), self.device.index)
        if key not in _envelope_operators:
            if len(_envelope_operators) > 16:
                _envelope_operators.clear()
            M, columns, rows = _envelope_operator(self.k_fid, self.ik_fid_peaks)
            _envelope_operators[key] = (M, LinearOperator.dense(M, device=self.device), columns, rows)
        M, self._envelope, columns, rows = _envelope_operators[key]
        self._envelope_columns = (columns, rows)      # what a batch is run with (cp_brieden_smooth)
        self.ratio_now_fid = M.dot(self.ratio_fid)

    def _compute(self):
        torch = dv.torch()
        rescale = self.rs_drag_ratio()
        if dv.is_torch(rescale) or np.ndim(rescale):
            return self._compute_batched(dv.to_device(rescale, self.device).reshape(-1))
        rescale = float(rescale)
        rows, _ = self._rows(self.k_fid / rescale)                                               # (ncol, 341)
        pknow = Fourier(self.cosmo, engine='eisenstein_hu_nowiggle', set_engine=False).pk_interpolator()(self.k_fid * rescale, z=0.)
        pknow = dv.to_device(np.asarray(pknow, dtype='f8') * self.pknow_correction[:, 0], self.device)      # (341,)
        ratio = rows / pknow / dv.to_device(self.ratio_fid[:, 0], self.device)
        pknow_cols = self._envelope(ratio) * pknow * dv.to_device(self.ratio_now_fid[:, 0], self.device)   # (ncol, 341)
        # the input interpolator cloned on (k_fid / rescale, pknow) and evaluated at k_fid (reference bao_filter.py:503-509)
        interp = self.pk_interpolator
        knots = self.k_fid / rescale
        if isinstance(interp, PowerSpectrumInterpolator2D):
            z = interp.z
            pk2 = pknow_cols.reshape(self._lead + (knots.size,))
            # (nz, nknots) -> the (nknots, nz) table of the clone; a batch of tables (batch, nz, nknots) -> (batch, nknots, nz): one clone for all of them
            clone = PowerSpectrumInterpolator2D(knots, z, pk2.transpose(-1, -2).contiguous() if pk2.ndim > 2 else pk2.T, interp_k=interp.interp_k, extrap_pk=interp.extrap_pk, extrap_kmin=interp.extrap_kmin,
                                                extrap_kmax=interp.extrap_kmax, interp_order_k=interp.interp_order_k, interp_order_z=interp.interp_order_z,
                                                growth_factor_sq=interp.growth_factor_sq, device=self.device)
            new = clone._rows_z(z, ignore_growth=True)(self.k_fid).reshape(-1, self.k_fid.size)
        else:
            clone = PowerSpectrumInterpolator1D(knots, pknow_cols.T, interp_k=interp.interp_k, extrap_pk=interp.extrap_pk, extrap_kmin=interp.extrap_kmin,
                                                extrap_kmax=interp.extrap_kmax, interp_order_k=interp.interp_order_k, device=self.device)
            new = clone._rows(self.k_fid)
        out = self._pk_rows.clone()
        out.index_copy_(1, self._kfid_index(), new)
        self._pknow_rows = out


# brieden2022 over a batch, what follows the two P(k) evaluations: 2 = one kernel (cp_brieden_smooth: P taken at the extrema of the fiducial wiggles only),
# 1 = ratio kernel, dense envelope operator, cp_brieden_resample, 0 = ratio kernel, operator, knots / per-column spline / final pass
_RESAMPLE_IN_ONE_KERNEL = 2


def _brieden_compute_batched(self, rescale):
    """
    One rs_drag ratio per cosmology (``cosmo`` is a batch): same steps as :meth:`Brieden2022PowerSpectrumBAOFilter._compute`, with the
    spectra evaluated at per-cosmology wavenumbers in one launch (``kscale``) and the final re-sampling -- a log-log natural spline on the
    per-cosmology knots k_fid / rescale, extended by ``_pad_log`` -- done by the per-column spline kernel (``cp_spline_columns``).
    Input: the 2D interpolator of a batched analytic engine built with a single redshift, e.g. ``Fourier(cosmo).pk_interpolator(z=[0.])``.
    """
    torch = dv.torch()
    interp = self.pk_interpolator
    if not (isinstance(interp, PowerSpectrumInterpolator2D) and hasattr(interp, '_pk_scaled') and interp.z.size == 1):
        raise NotImplementedError('brieden2022 over a batch of cosmologies needs the pk_interpolator(z=[z0]) of a batched analytic engine')
    nb, n = rescale.numel(), self.k_fid.size
    lib, stream = _lib.load(), dv.stream_of(self.device)
    rescale = rescale.contiguous()
    # the one-kernel routes solve the re-sampling spline with the constant coefficients of a geometric k_fid: any other spacing takes the general route
    steps = np.diff(np.log(self.k_fid))
    geometric = bool(np.all(np.abs(steps - steps[0]) <= 1e-9 * abs(steps[0])))
    fused = _RESAMPLE_IN_ONE_KERNEL == 2 and geometric and 129 <= n <= 512 and self._envelope_columns[0].size <= 64
    # P_c(k_fid / r_c), (B, 341) -- or at the extrema only, (B, 23): all the envelope depends on
    rows = interp._pk_scaled(self.k_fid[self._envelope_columns[0]] if fused else self.k_fid, 1. / rescale).contiguous()
    now = Fourier(self.cosmo, engine='eisenstein_hu_nowiggle', set_engine=False).pk_interpolator(z=np.array([0.]))
    g0 = dv.to_device(now.growth_factor_sq(np.array([0.])), self.device).reshape(nb).contiguous()
    raw = now._pk_scaled(self.k_fid, rescale).contiguous()
    const = self.__dict__.get('_device_constants')
    if const is None:      # fiducial products and the k_fid range: uploaded once per filter object
        first = int(np.flatnonzero(self.kmask_fid)[0])
        assert np.array_equal(np.flatnonzero(self.kmask_fid), first + np.arange(n))      # 1e-3 <= k <= 1 is a contiguous range of k
        const = self._device_constants = dict(correction=dv.to_device(self.pknow_correction[:, 0], self.device), ratio_fid=dv.to_device(self.ratio_fid[:, 0], self.device),
                                              ratio_now_fid=dv.to_device(self.ratio_now_fid[:, 0], self.device), k_fid=dv.to_device(self.k_fid, self.device),
                                              log_k_fid=dv.to_device(np.log10(self.k_fid), self.device), first=first)
    pk = self._pk_rows.contiguous()
    if fused:
        # pknow = P_nowiggle x growth x correction, ratio = P / pknow / ratio_fid at the extrema, the envelope from it, log10 of envelope x pknow x ratio_now_fid
        # on the per-cosmology knots k_fid / rescale with the two extrapolated knots of _pad_log on either side, its natural spline at k_fid, 10^x written
        # over the k_fid range of P (bao_filter.py:493-509): one kernel, a wave per cosmology
        if 'peaks' not in const:
            const['peaks'] = dv.upload(self._envelope_columns[0].astype(np.int32), self.device)
            const['operator'] = dv.to_device(np.ascontiguousarray(self._envelope_columns[1]), self.device)
        res = torch.empty_like(pk)
        _lib.check(lib.cp_brieden_smooth(rows.data_ptr(), raw.data_ptr(), g0.data_ptr(), const['correction'].data_ptr(), const['ratio_fid'].data_ptr(), const['peaks'].data_ptr(),
                                         const['operator'].data_ptr(), int(self._envelope_columns[0].size), const['ratio_now_fid'].data_ptr(), const['k_fid'].data_ptr(),
                                         const['log_k_fid'].data_ptr(), rescale.data_ptr(), float(interp.extrap_kmin), float(interp.extrap_kmax), pk.data_ptr(), res.data_ptr(),
                                         nb, n, pk.shape[1], const['first'], self.device.index, stream))
        self._pknow_rows = res
        return
    # pknow = P_nowiggle x growth x correction; ratio = P / pknow / ratio_fid (reference bao_filter.py:493-499): one pass
    pknow, ratio = torch.empty_like(rows), torch.empty_like(rows)
    _lib.check(lib.cp_brieden_ratio(rows.data_ptr(), raw.data_ptr(), g0.data_ptr(), const['correction'].data_ptr(), const['ratio_fid'].data_ptr(),
                                    pknow.data_ptr(), ratio.data_ptr(), nb, n, self.device.index, stream))
    envelope = self._envelope(ratio)                                                                  # (B, 341)
    if _RESAMPLE_IN_ONE_KERNEL and geometric and 129 <= n <= 512:
        # log10 of envelope x pknow x ratio_now_fid on the per-cosmology knots k_fid / rescale with the two extrapolated knots of _pad_log on either side,
        # its natural spline at k_fid, 10^x written over the k_fid range of P (bao_filter.py:500-509): one kernel, a wave per cosmology (k_fid is a
        # range of the filter's geometric grid: the spline's system has constant coefficients)
        res = torch.empty_like(pk)
        _lib.check(lib.cp_brieden_resample(envelope.data_ptr(), pknow.data_ptr(), const['ratio_now_fid'].data_ptr(), const['k_fid'].data_ptr(), const['log_k_fid'].data_ptr(),
                                           rescale.data_ptr(), float(interp.extrap_kmin), float(interp.extrap_kmax), pk.data_ptr(), res.data_ptr(), nb, n, pk.shape[1],
                                           const['first'], self.device.index, stream))
        self._pknow_rows = res
        return
    # log10 of envelope x pknow x ratio_now_fid on the per-cosmology knots k_fid / rescale, knot-major (345, B), with the two extrapolated knots of
    # _pad_log (interpolator.py:42-87) on either side: one pass
    xk, yk = torch.empty((n + 4, nb), dtype=torch.float64, device=self.device), torch.empty((n + 4, nb), dtype=torch.float64, device=self.device)
    _lib.check(lib.cp_brieden_knots(envelope.data_ptr(), pknow.data_ptr(), const['ratio_now_fid'].data_ptr(), const['k_fid'].data_ptr(), rescale.data_ptr(),
                                    float(interp.extrap_kmin), float(interp.extrap_kmax), xk.data_ptr(), yk.data_ptr(), nb, n, self.device.index, stream))
    out = torch.empty((n, nb), dtype=torch.float64, device=self.device)
    scratch = torch.empty(int(lib.cp_spline_columns_scratch_doubles(nb, n + 4)), dtype=torch.float64, device=self.device)
    _lib.check(lib.cp_spline_columns(xk.data_ptr(), yk.data_ptr(), nb, n + 4, const['log_k_fid'].data_ptr(), n, out.data_ptr(), scratch.data_ptr(),
                                     self.device.index, stream))
    # the input spectra with 10^(re-sampled) written over the k_fid range (bao_filter.py:509): one pass
    res = torch.empty_like(pk)
    _lib.check(lib.cp_brieden_finish(pk.data_ptr(), out.data_ptr(), res.data_ptr(), nb, pk.shape[1], const['first'], n, self.device.index, stream))
    self._pknow_rows = res


Brieden2022PowerSpectrumBAOFilter._compute_batched = _brieden_compute_batched


def PowerSpectrumBAOFilter(pk_interpolator, engine='wallish2018', cosmo=None, cosmo_fid=None, **kwargs):
    """
    Run power spectrum BAO filter ``engine`` (reference bao_filter.py:912-921); every filter of the reference's registry except
    'bspline' (which fails in the reference itself under numpy 2).
    """
    engine = engine.lower()
    if engine not in RegisteredPowerSpectrumBAOFilter._registry or engine == 'base':
        raise ValueError('BAO filter {} is not available on the MI355X path; choose one of {}'.format(
            engine, sorted(name for name in RegisteredPowerSpectrumBAOFilter._registry if name != 'base')))
    return RegisteredPowerSpectrumBAOFilter._registry[engine](pk_interpolator, cosmo=cosmo, cosmo_fid=cosmo_fid, **kwargs)


# ---- remaining P(k) filters of the registry ------------------------------------------------------------
# For a fixed cosmology each of them is a fixed linear map of a transformed spectrum (log10 P, log(k P), or the ratio to the
# Eisenstein & Hu no-wiggle spectrum): the map is built once on the host (least-squares projectors / Savitzky-Golay weights /
# products of spline operators: numpy on <= 1024 samples) and applied to all columns on the device by the dense operator kernel.

def _constrained_lsq_operator(gradient, precision, constraint_gradient, constraint_matrix, inverse=False):
    """
    ``utils.LeastSquareSolver(gradient, precision (1D), constraint_gradient)`` (reference utils.py:161-272) followed by ``model()``,
    as one (ndata, ndata) matrix A: model = A . delta, for constraints that are themselves linear in delta
    (constraint = constraint_matrix . delta).  ``inverse`` = the reference's ``compute_inverse`` (explicit inverse vs solve).
    """
    nparams, ndata = gradient.shape
    nc = constraint_gradient.shape[-1]
    hv = gradient * precision
    invfisher = np.block([[hv.dot(gradient.T), -constraint_gradient], [constraint_gradient.T, np.zeros((nc, nc))]])
    hv = np.block([[hv, np.zeros(constraint_gradient.shape)], [np.zeros((nc, ndata)), np.eye(nc)]])
    proj = np.linalg.inv(invfisher).dot(hv) if inverse else np.linalg.solve(invfisher, hv)      # (nparams + nc, ndata + nc)
    to_params = proj[:nparams, :ndata] + proj[:nparams, ndata:].dot(constraint_matrix)          # (nparams, ndata)
    return gradient.T.dot(to_params)


def _constrained_lsq_steps(gradient, precision, constraint_gradient, constraint_matrix):
    """
    ``utils.LeastSquareSolver(..., compute_inverse=True)`` in the operation order of its ``compute`` and ``model`` (reference utils.py:218-232, 246-272):
    constraint values c = C . delta, parameters = delta . P1 + c . P2 with (P1 | P2) the explicitly inverted Karush-Kuhn-Tucker matrix applied to
    (G F | 1), model = parameters . G.  Returns the four host matrices (C, P1^T, P2^T, G^T) for dense operators applied in that order.

    Why not the one (ndata, ndata) matrix of :func:`_constrained_lsq_operator`: for hinton2017 (degree 12: condition 5e12, the inverse off by 5e-6)
    the reference's numbers follow from THIS inverse evaluated this way; the parameters are sums with 4e4 times their size in terms and the model
    3e2, so any summation order reproduces them to 1e-12 -- whereas the products G^T P that form the single matrix cancel to 1e-10 of their terms
    in its ENTRIES (numpy itself is 3e-10 from the reference through it, the device kernel's blocked sums 1e-7).
    """
    nparams, ndata = gradient.shape
    nc = constraint_gradient.shape[-1]
    hv = gradient * precision
    invfisher = np.block([[hv.dot(gradient.T), -constraint_gradient], [constraint_gradient.T, np.zeros((nc, nc))]])
    hv = np.block([[hv, np.zeros(constraint_gradient.shape)], [np.zeros((nc, ndata)), np.eye(nc)]])
    proj = np.linalg.inv(invfisher).dot(hv)                  # (nparams + nc, ndata + nc): the reference's projector, transposed
    return constraint_matrix, proj[:nparams, :ndata], proj[:nparams, ndata:], gradient.T


def _end_constraints(n, order=2):
    """Rows picking delta[0], delta[1] - delta[0] (, second difference) and the same at the other end (reference bao_filter.py:226-229, 335)."""
    rows = []
    for sign in (1, -1):
        idx = [0, 1, 2] if sign > 0 else [-1, -2, -3]
        r0 = np.zeros(n); r0[idx[0]] = 1.
        r1 = np.zeros(n); r1[idx[1]] = 1.; r1[idx[0]] = -1.
        rows += [r0, r1]
        if order > 2:
            r2 = np.zeros(n); r2[idx[2]] = 1.; r2[idx[1]] = -2.; r2[idx[0]] = 1.
            rows.append(r2)
    return np.array(rows)


def _savgol_operator(n, window, polyorder=4):
    """``scipy.signal.savgol_filter(x, window, polyorder, mode='interp')`` as an (n, n) matrix: the centre weights of a local
    polynomial fit inside, and the polynomial fitted to the first / last ``window`` samples for the half-window at either end."""
    h = window // 2
    powers = np.arange(polyorder + 1)
    x = np.arange(-h, h + 1, dtype='f8')
    centre = np.linalg.pinv(x[:, None]**powers)[0]
    S = np.zeros((n, n))
    for i in range(h, n - h):
        S[i, i - h:i + h + 1] = centre
    V = x[:, None]**powers                               # edge fits on a centred abscissa (same polynomial, better conditioned)
    fit = V.dot(np.linalg.pinv(V))                       # fitted values at the window's own samples
    S[:h, :window] = fit[:h]
    S[n - h:, n - window:] = fit[window - h:]
    return S


class _OperatorFilterMixin(object):

    def _eh_nowiggle(self, k):
        """Eisenstein & Hu no-wiggle P(k, z=0) of ``cosmo`` on the device (reference: Fourier(cosmo, engine='eisenstein_hu_nowiggle')): (nk,) for one
        cosmology; for a batch of cosmologies one row per row of the input spectra, (ncol, nk) (the columns of a cosmology share its row)."""
        fo = Fourier(self.cosmo, engine='eisenstein_hu_nowiggle', set_engine=False)
        if self._batch_size() is None:
            return dv.upload(np.asarray(fo.pk_interpolator()(k, z=0.), dtype='f8'), self.device)
        rows = fo.pk_interpolator()._rows_z(np.zeros(1))(np.asarray(k, dtype='f8'))      # (B, 1, nk), on the device
        rows = rows.reshape(rows.shape[0], rows.shape[-1])
        return rows.repeat_interleave(self._columns_per_cosmology(), dim=0) if self._columns_per_cosmology() > 1 else rows



class Hinton2017PowerSpectrumBAOFilter(BasePowerSpectrumBAOFilter):

    """High-degree polynomial fitted to the power spectrum in log-log space (reference bao_filter.py:172-241)."""
    name = 'hinton2017'

    def __init__(self, pk_interpolator, degree=12, sigma=0.5, weight=0.9, **kwargs):
        self.degree, self.sigma, self.weight = degree, sigma, weight
        super(Hinton2017PowerSpectrumBAOFilter, self).__init__(pk_interpolator, **kwargs)

    def _fit_operator(self, imax):
        """The weighted, end-pinned polynomial fit in log-log space as dense operators (:func:`_constrained_lsq_steps`), for a spectrum whose maximum sits
        at sample ``imax`` of the fitted range (reference bao_filter.py:215-235: the weights dip around the maximum); kept per position."""
        cache = self.__dict__.setdefault('_fit_operators', {})
        if imax not in cache:
            logk = np.log10(self.k[self.kmask])
            w = 1. - self.weight * np.exp(-0.5 * ((logk - logk[imax]) / self.sigma)**2)
            gradient = np.array([((logk - np.mean(logk)) / np.std(logk))**i for i in range(self.degree + 1)])
            cg = np.column_stack([gradient[..., 0], gradient[..., 1] - gradient[..., 0], gradient[..., 2] - 2. * gradient[..., 1] + gradient[..., 0],
                                  gradient[..., -1], gradient[..., -2] - gradient[..., -1], gradient[..., -3] - 2. * gradient[..., -2] + gradient[..., -1]])
            steps = _constrained_lsq_steps(gradient, w**2, cg, _end_constraints(logk.size, order=3))
            cache[imax] = tuple(LinearOperator.dense(np.ascontiguousarray(m), device=self.device) for m in steps)
        return cache[imax]

    @staticmethod
    def _apply_fit(ops, logpk):
        """The reference's ``solver(logpk, constraint=...)`` + ``solver.model()`` (bao_filter.py:229-235) in its order of operations, on rows of log10 P."""
        to_constraint, from_data, from_constraint, to_model = ops
        params = from_data(logpk) + from_constraint(to_constraint(logpk).contiguous())      # (ncol, degree + 1)
        return to_model(params.contiguous())

    def _prepare(self):
        """The fit's weights follow the maximum of the FIRST column of the spectrum ("approximation", :219).  A batch of cosmologies (a batched ``cosmo``, or a
        batched 2D interpolator: one cosmology per leading index) is that many inputs of the reference: each takes the weights of its own first column -- the
        cosmologies whose maxima fall on the same sample share an operator (the positions are read back once: one integer per cosmology)."""
        torch = dv.torch()
        self.kmask = (self.k > 1e-4) & (self.k < 5.)
        ncosmo = self._batch_size() or (int(np.prod(self._lead[:-1])) if isinstance(self.pk_interpolator, PowerSpectrumInterpolator2D) and len(self._lead) > 1 else 1)
        per = self._pk_rows.shape[0] // ncosmo
        first = self._pk_rows[::per][:, dv.upload(self.kmask, self.device)]      # the first column of every cosmology, fitted range
        imax = torch.log10(first).argmax(dim=1).cpu().numpy()
        imax = np.repeat(imax, per)
        self._groups = [(self._fit_operator(int(i)), None if ncosmo == 1 else dv.upload(np.flatnonzero(imax == i), self.device)) for i in np.unique(imax)]

    def _compute(self):
        torch = dv.torch()
        mask = dv.upload(self.kmask, self.device)
        res = self._pk_rows.clone()
        logpk = torch.log10(self._pk_rows[:, mask]).contiguous()
        fitted = torch.empty_like(logpk)
        for ops, rows in self._groups:
            if rows is None:
                fitted = self._apply_fit(ops, logpk)
            else:
                fitted[rows] = self._apply_fit(ops, logpk[rows].contiguous())
        res[:, mask] = 10**fitted
        self._pknow_rows = res


class SavGolPowerSpectrumBAOFilter(BasePowerSpectrumBAOFilter):

    """Savitzky-Golay smoothing of log(k P) along log k (reference bao_filter.py:244-266)."""
    name = 'savgol'

    @property
    def _op(self):
        """The filter as an operator on the CURRENT wavenumbers (the reference derives the window from them in ``_compute``, bao_filter.py:258-259: a filter
        re-used after ``set_k`` follows); it depends on (nk, window) only: built and uploaded once, shared by every filter object."""
        from .interpolator import _cached_operator
        self.nfilter = int(np.ceil(np.log(7) / np.log(self.k[-1] / self.k[-2])) // 2 * 2 + 1)
        nk, nfilter = self.k.size, self.nfilter
        return _cached_operator(('savgol', nk, nfilter, self.device.index), lambda: LinearOperator.dense(_savgol_operator(nk, nfilter), device=self.device))

    def _compute(self):
        torch = dv.torch()
        kt = dv.upload(self.k, self.device)
        res = torch.exp(self._op(torch.log(kt * self._pk_rows))) / kt
        h = self.nfilter // 2
        res[:, -h:] = self._pk_rows[:, -h:]
        self._pknow_rows = res


class EHNoWiggleSavGolPowerSpectrumBAOFilter(_OperatorFilterMixin, SavGolPowerSpectrumBAOFilter):

    """Savitzky-Golay smoothing of the ratio to the Eisenstein & Hu no-wiggle spectrum (reference bao_filter.py:269-286)."""
    name = 'ehsavgol'

    def _compute(self):
        pknow = self._eh_nowiggle(self.k)
        self._pknow_rows = self._op(self._pk_rows / pknow) * pknow


class EHNoWigglePolyPowerSpectrumBAOFilter(_OperatorFilterMixin, BasePowerSpectrumBAOFilter):

    """Ratio to the Eisenstein & Hu no-wiggle spectrum emulated by a constrained polynomial in k (reference bao_filter.py:289-342)."""
    name = 'ehpoly'

    def __init__(self, pk_interpolator, krange=(1e-3, 1.), rescale_krange=True, cosmo=None, **kwargs):
        self.krange = krange
        self.rescale_krange = rescale_krange
        super(EHNoWigglePolyPowerSpectrumBAOFilter, self).__init__(pk_interpolator, cosmo=cosmo, **kwargs)

    def _range_operator(self, first, last):
        """The constrained fit on the wavenumbers k[first:last] as a dense operator on the ratio there; kept per range (a batch of cosmologies has a
        few dozen different ranges)."""
        cache = self.__dict__.setdefault('_range_operators', {})
        if (first, last) not in cache:
            k = self.k[first:last]
            gradient = np.array([k**(i - 2) for i in range(6)])
            cg = np.column_stack([gradient[..., 0], gradient[..., 1] - gradient[..., 0], gradient[..., -1], gradient[..., -2] - gradient[..., -1]])
            A = _constrained_lsq_operator(gradient, k**2, cg, _end_constraints(k.size, order=2))
            cache[(first, last)] = LinearOperator.dense(A, device=self.device)
        return cache[(first, last)]

    def _compute(self):
        torch = dv.torch()
        krange = np.asarray(self.krange, dtype='f8')
        res = self._pk_rows.clone()
        if self._batch_size() is None:
            if self.rescale_krange:
                krange = krange / self._scalar_rs_drag_ratio()
            mask = (self.k >= krange[0]) & (self.k <= krange[1])
            first, last = int(np.flatnonzero(mask)[0]), int(np.flatnonzero(mask)[-1]) + 1
            pknow = self._eh_nowiggle(self.k[first:last])
            ratio = (self._pk_rows[:, first:last] / pknow).contiguous()
            res[:, first:last] = self._range_operator(first, last)(ratio) * pknow      # pk / (ratio / model)
            self._pknow_rows = res
            return
        # a batch of cosmologies: the fit depends on the cosmology through the RANGE of wavenumbers only (krange / its rs_drag ratio) -- the
        # cosmologies that share a range share the operator (the ranges are read back once: B pairs of integers)
        ratios = _host_value(self.rs_drag_ratio()) if self.rescale_krange else np.ones(self._batch_size())
        ratios = np.broadcast_to(ratios, (self._batch_size(),))
        inside = (self.k >= (krange[0] / ratios)[:, None]) & (self.k <= (krange[1] / ratios)[:, None])      # (B, nk): the reference's mask, cosmology by cosmology
        first, last = inside.argmax(axis=1), inside.shape[1] - inside[:, ::-1].argmax(axis=1)
        pknow = self._eh_nowiggle(self.k)                                                                   # (ncol, nk)
        first, last = self._per_row(first), self._per_row(last)
        for f, l in sorted(set(zip(first.tolist(), last.tolist()))):
            rows = dv.upload(np.flatnonzero((first == f) & (last == l)), self.device)
            now = pknow[rows, f:l]
            res[rows, f:l] = self._range_operator(f, l)((self._pk_rows[rows, f:l] / now).contiguous()) * now
        self._pknow_rows = res


class PeakAveragePowerSpectrumBAOFilter(_OperatorFilterMixin, BasePowerSpectrumBAOFilter):

    """Average of the splines through the maxima and through the minima of the wiggles, at the fiducial positions moved by the
    rs_drag ratio (reference bao_filter.py:512-580).  ``cosmo_fid`` is mandatory, with an engine."""
    name = 'peakaverage'

    @property
    def cosmo_fid(self):
        """Reference cosmology."""
        if self._cosmo_fid is None:
            raise ValueError('cosmo_fid must be provided, with an engine')
        return self._cosmo_fid

    def _prepare(self):
        """Knots of the two splines (reference bao_filter.py:536-563): the extrema of the fiducial wiggles above k = 0.01, with every
        sample below k = 1e-3 and above the last extremum (at least all of k > 1) kept as knots on either side."""
        index = np.flatnonzero((self.k >= 1e-3) & (self.k <= 1.))
        ratio, correction = _fiducial_wiggles(self.cosmo_fid, self.k[index])
        start = np.searchsorted(self.k[index], 1e-2, side='right') + 1
        first = index[0]
        self.k_peaks, self.pad_peaks = [], []
        for extrema in _wiggle_extrema(ratio / correction, start):
            extrema = extrema + first
            resume = max(index[-1], extrema[-1] + 1)
            self.pad_peaks.append((first, len(extrema), self.k.size - resume))
            self.k_peaks.append(np.concatenate([self.k[:first], self.k[extrema], self.k[resume:]]))

    def _operator(self, rescale):
        """``_interp`` (reference bao_filter.py:565-574): natural splines in log10 k, data -> moved knots (extrapolating) -> all k."""
        from .spline import dense_operator
        logx = np.log10(self.k)
        M = 0.
        for kp, npad in zip(self.k_peaks, self.pad_peaks):
            scale = np.concatenate([np.linspace(1., rescale, npad[0]), np.full(npad[1], rescale), np.linspace(rescale, 1., npad[2])])
            logxx = np.log10(kp / scale)
            M = M + 0.5 * dense_operator(logxx, logx, bc='natural').dot(dense_operator(logx, logxx, bc='natural', extrapolate=True))
        return M

    # one cosmology: False = the two splines run as kernels, as for a batch (0.9 ms per filter); True = their (nk, nk) product built on the host and
    # applied as a dense operator (four dense spline operators per filter object: 140 ms on the host, against 26 ms for the reference's own filter)
    _DENSE_OPERATOR = False

    def _compute(self):
        pknow = self._eh_nowiggle(self.k)
        if self._batch_size() is None and self._DENSE_OPERATOR:
            op = LinearOperator.dense(self._operator(self._scalar_rs_drag_ratio()), device=self.device)
            self._pknow_rows = op(self._pk_rows / pknow) * pknow
            return
        self._pknow_rows = self._interp_batch(self._pk_rows / pknow) * pknow

    def _interp_batch(self, ratio):
        """``_interp`` (reference bao_filter.py:565-574) for a batch of cosmologies, one rs_drag ratio each: the dense operator of one cosmology is two
        natural splines in a row -- data on log10 k -> the moved knots (extrapolating), moved knots -> all log10 k -- and the knots move with the
        ratio, so neither is shared.  First spline: shared knots, per-row queries (second derivatives of all rows by ``cp_spline_rows``, then
        ``cp_spline_rows_at_queries``); second: per-row knots, shared queries (``cp_spline_columns``).  ratio : (ncol, nk) on the device."""
        from .spline import SplineRows
        torch = dv.torch()
        lib = _lib.load()
        ncol, nk = ratio.shape
        logx = np.log10(self.k)
        key, solver = self.__dict__.get('_log_solver', (None, None))
        if key != self.k.tobytes():      # (the wavenumbers may have changed since: set_k)
            key, solver = self.__dict__['_log_solver'] = (self.k.tobytes(), SplineRows(logx, logx, bc='natural', device=self.device))
        ratio = ratio.contiguous()
        second = solver.second_derivatives(ratio)
        if self._batch_size() is None:      # one cosmology, any number of columns: one ratio
            rescale = torch.full((ncol,), self._scalar_rs_drag_ratio(), dtype=torch.float64, device=self.device)
        else:
            rescale = dv.to_device(self.rs_drag_ratio(), self.device).reshape(-1)
            rescale = rescale.repeat_interleave(self._columns_per_cosmology()) if self._columns_per_cosmology() > 1 else rescale
        tlogx = dv.upload(logx, self.device)
        total = torch.zeros((nk, ncol), dtype=torch.float64, device=self.device)
        for kp, npad in zip(self.k_peaks, self.pad_peaks):
            # the knots' scale: 1 -> rescale over the samples kept in front, rescale at the extrema, rescale -> 1 over the samples kept behind
            weight = np.concatenate([np.linspace(0., 1., npad[0]), np.ones(npad[1]), np.linspace(1., 0., npad[2])])
            scale = 1. + (rescale[:, None] - 1.) * dv.upload(weight, self.device)
            knots = (dv.upload(np.log10(kp), self.device) - torch.log10(scale)).contiguous()               # (ncol, nknots) = log10(kp / scale)
            nknots = knots.shape[1]
            values = torch.empty((nknots, ncol), dtype=torch.float64, device=self.device)                  # knot-major
            _lib.check(lib.cp_spline_rows_at_queries(tlogx.data_ptr(), ratio.data_ptr(), second.data_ptr(), ncol, nk, knots.data_ptr(), nknots, values.data_ptr(), 1,
                                                     self.device.index, dv.stream_of(self.device)))
            xk = knots.t().contiguous()
            out = torch.empty((nk, ncol), dtype=torch.float64, device=self.device)
            scratch = torch.empty(int(lib.cp_spline_columns_scratch_doubles(ncol, nknots)), dtype=torch.float64, device=self.device)
            _lib.check(lib.cp_spline_columns(xk.data_ptr(), values.data_ptr(), ncol, nknots, tlogx.data_ptr(), nk, out.data_ptr(), scratch.data_ptr(),
                                             self.device.index, dv.stream_of(self.device)))
            total += out
        return (0.5 * total).t().contiguous()


class BSplinePowerSpectrumBAOFilter(_OperatorFilterMixin, BasePowerSpectrumBAOFilter):

    """
    Ratio to the Eisenstein & Hu no-wiggle spectrum emulated by B-splines in log10 k, mixed so that the result keeps the sigma8 (, sigma_d)
    of the input (reference bao_filter.py:583-688; https://arxiv.org/pdf/1509.02120.pdf App. A).

    Each B-spline fit, pinned at its four end samples, is a fixed linear map of the ratio (one dense operator per fit, built on the host);
    the constraints are linear functionals of P (Simpson weights), so the mixing is a 1 x 1 ... 3 x 3 system per column, solved on the device
    in closed form.  The system is read as the reference wrote it for numpy < 2 -- one right-hand-side VECTOR per column: under numpy 2 its
    ``numpy.linalg.solve(system, target)`` (:685) raises for any constraint or several columns.
    """
    name = 'bspline'
    _functionals = ('sigma8', 'sigmad')

    def __init__(self, pk_interpolator, constraint=('sigma8',), cosmo=None, **kwargs):
        if not isinstance(constraint, (tuple, list)):
            constraint = [constraint]
        self.constraint = list(constraint)
        for name in self.constraint:
            if name not in self._functionals:
                raise ValueError('unknown constraint {}; choices are {}'.format(name, list(self._functionals)))
        if len(self.constraint) > 2:
            raise ValueError('at most two constraints (three B-spline models), got {}'.format(self.constraint))
        super(BSplinePowerSpectrumBAOFilter, self).__init__(pk_interpolator, cosmo=cosmo, **kwargs)

    def _prepare(self):
        kmin, kmax = 5e-3, 1.
        self.kmask_fid = (self.k >= kmin) & (self.k <= kmax)
        logk = np.log10(self.k[self.kmask_fid])
        weights = 1 + 1e6 * np.tanh(0.005 * (logk + 1.1)**16)
        weights /= np.sum(weights)
        ends = _end_constraints(logk.size, order=2)
        self._fits = []
        for nknots, degree in [(14, 5), (14, 6), (15, 7)][:1 + len(self.constraint)]:
            inner = nknots - 2 * degree
            ts = np.concatenate([np.zeros(degree + 1), np.arange(1, inner) / inner, np.ones(degree + 1)])
            gradient = _bspline_basis(np.log10((kmax - kmin) * ts + kmin), degree, logk).T       # (ncoef, nk): scipy BSpline(ts, e_i, degree)(logk)
            A = _constrained_lsq_operator(gradient, weights, gradient.dot(ends.T), ends, inverse=True)
            self._fits.append(LinearOperator.dense(A, device=self.device))
        # sigma8^2 and sigma_d^2 as weights on the samples of P (:665-672): the reference's Simpson rule on k, top-hat of radius 8
        kr = 8. * self.k
        simpson = _simpson_weights(self.k)
        weights = {'sigma8': 1. / (2. * np.pi**2) * simpson * self.k**2 * (3. * (np.sin(kr) - kr * np.cos(kr)) / kr**3)**2,
                   'sigmad': 1. / (6. * np.pi**2) * simpson}
        self._constraint_weights = np.array([weights[name] for name in self.constraint]).reshape(len(self.constraint), self.k.size)

    def _compute(self):
        torch = dv.torch()
        pk = self._pk_rows
        mask = dv.upload(self.kmask_fid, self.device)
        pknow = self._eh_nowiggle(self.k[self.kmask_fid])
        ratio = (pk[:, mask] / pknow).contiguous()
        models = []
        for fit in self._fits:
            model = pk.clone()
            model[:, mask] = fit(ratio) * pknow
            models.append(model)
        if len(models) == 1:
            self._pknow_rows = models[0]
            return
        # the constraint integrals of every model and of the input, (ncol, nc) each, through the package's dense-operator kernel
        weigh = _cached_dense_operator(self._constraint_weights, self.device)       # (nc, nk) operator
        integrals = [weigh(m.contiguous()) for m in models]
        target = weigh(pk.contiguous())
        nc = self._constraint_weights.shape[0]
        rows = [[torch.ones_like(pk[:, 0])] * len(models)] + [[integral[:, c] for integral in integrals] for c in range(nc)]       # system[i][j]: (ncol,)
        rhs = [torch.ones_like(pk[:, 0])] + [target[:, c] for c in range(nc)]
        coeffs = _solve_small(rows, rhs)
        self._pknow_rows = sum(c[:, None] * m for c, m in zip(coeffs, models))


def _cached_dense_operator(weights, device):
    """:class:`LinearOperator` of a small host matrix, kept by content (the constraint weights of a filter do not change between calls)."""
    weights = np.ascontiguousarray(weights, dtype='f8')
    key = (weights.shape, weights.tobytes(), device.index)
    if key not in _dense_operators:
        if len(_dense_operators) > 16:
            _dense_operators.clear()
        _dense_operators[key] = LinearOperator.dense(weights, device=device)
    return _dense_operators[key]


_dense_operators = {}


def _solve_small(a, b):
    """Solutions x_j (tensors over the batch) of the n x n systems sum_j a[i][j] x_j = b[i], n <= 3, by Cramer's rule on batch tensors."""
    n = len(b)

    def det(m):
        if len(m) == 1:
            return m[0][0]
        if len(m) == 2:
            return m[0][0] * m[1][1] - m[0][1] * m[1][0]
        return (m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
                + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]))

    d = det(a)
    return [det([[b[i] if jj == j else a[i][jj] for jj in range(n)] for i in range(n)]) / d for j in range(n)]


class RegisteredCorrelationFunctionBAOFilter(type):

    """Metaclass registering :class:`BaseCorrelationFunctionBAOFilter`-derived classes by ``name`` (reference bao_filter.py:691-700)."""
    _registry = {}

    def __new__(meta, name, bases, class_dict):
        cls = super().__new__(meta, name, bases, class_dict)
        meta._registry[cls.name] = cls
        return cls


class BaseCorrelationFunctionBAOFilter(dv.Copyable, metaclass=RegisteredCorrelationFunctionBAOFilter):

    """Base BAO filter for correlation function (reference bao_filter.py:703-832)."""
    name = 'base'

    def __init__(self, xi_interpolator, cosmo=None, cosmo_fid=None, **kwargs):
        self._cosmo_fid = cosmo_fid
        self.xi_interpolator = xi_interpolator
        self.device = xi_interpolator.device
        self.set_s(**kwargs)
        self.set_xi(xi_interpolator, cosmo=cosmo)
        self._prepare()
        self._compute()
        self._finalize()

    def _prepare(self):
        """Anything that can be done once."""

    def set_s(self, ns=1024):
        """Separations where the correlation function is evaluated (reference bao_filter.py:749-758)."""
        self.s = np.geomspace(self.xi_interpolator.extrap_smin, self.xi_interpolator.extrap_smax, ns)

    def set_xi(self, xi_interpolator, cosmo=None):
        """Set the input correlation function (reference bao_filter.py:760-770): device rows (ncol, ns)."""
        self._cosmo = cosmo
        self.xi_interpolator = xi_interpolator
        if isinstance(xi_interpolator, CorrelationFunctionInterpolator2D):
            out = xi_interpolator._eval_device(self.s, xi_interpolator.z, grid=True, ignore_growth=True)   # (ns, nz)
        else:
            out = xi_interpolator._eval_device(self.s)                                                    # (ns,) + columns
        self.shape = tuple(out.shape)
        self._xi_rows = out.reshape(self.s.size, -1).T.contiguous()

    def _finalize(self):
        self.xi = self._xi_rows.cpu().numpy().T.reshape(self.shape)
        self.xinow = self._xinow_rows.cpu().numpy().T.reshape(self.shape)

    def __call__(self, xi_interpolator, cosmo=None):
        self.set_xi(xi_interpolator, cosmo=cosmo)
        self._compute()
        self._finalize()
        return self

    def smooth_xi_interpolator(self, **kwargs):
        """Smooth (no-peak) correlation function interpolator (reference bao_filter.py:778-792)."""
        return self.xi_interpolator.clone(s=self.s, xi=self.xinow, **kwargs)

    def smooth_pk_interpolator(self, **kwargs):
        """Smooth (no-wiggle) power spectrum through FFTLog (reference bao_filter.py:794-808)."""
        return self.smooth_xi_interpolator().to_pk(**kwargs)

    @property
    def cosmo(self):
        """Cosmology."""
        if self._cosmo is None:
            self._cosmo = Cosmology()
        return self._cosmo

    @property
    def cosmo_fid(self):
        """Reference cosmology."""
        if self._cosmo_fid is None:
            self._cosmo_fid = Cosmology()
        return self._cosmo_fid

    rs_drag_ratio = BasePowerSpectrumBAOFilter.rs_drag_ratio
    _scalar_rs_drag_ratio = BasePowerSpectrumBAOFilter._scalar_rs_drag_ratio


class Kirkby2013CorrelationFunctionBAOFilter(BaseCorrelationFunctionBAOFilter):

    """
    Cut the BAO peak and bridge it with a polynomial in 1/s fitted on either side (reference bao_filter.py:835-909).
    For fixed boxes the whole filter is ONE linear map of xi(s): it is built on the host (a weighted 5-parameter least-squares
    projector on ~200 samples, blended with the identity) and applied to all columns on the device as a dense operator.
    """
    name = 'kirkby2013'

    def __init__(self, xi_interpolator, srange_left=(50., 82.), srange_right=(150., 190.), rescale_sbox=True, cosmo=None, **kwargs):
        self.srange_left = np.asarray(srange_left)
        self.srange_right = np.asarray(srange_right)
        self.rescale_sbox = rescale_sbox
        super(Kirkby2013CorrelationFunctionBAOFilter, self).__init__(xi_interpolator, cosmo=cosmo, **kwargs)

    # Shape of the fit (reference bao_filter.py:885-896, 898-909): a polynomial in 1/s of degree 3 times s, i.e. the five powers
    # s^1 .. s^-3, fitted with weight 1 inside the two side boxes; the weight ramps linearly to 0 over 1 % of a box edge outside
    # the boxes and over 1 % of the gap towards the peak.  The peak region is where the weight is 0 between the boxes: there (and on
    # the inner ramps, blended linearly) xi is replaced by the fitted curve.  Samples beyond a factor 2 of the outer box edges do
    # not enter the fit at all.
    _n_powers, _ramp_fraction, _fit_reach = 5, 0.01, 2.

    def _prepare(self):
        (l0, l1), (r0, r1) = self.srange_left, self.srange_right
        self.smask = (self.s >= l0 / self._fit_reach) & (self.s <= r1 * self._fit_reach)
        self.model = self.s[None, :]**(1. - np.arange(self._n_powers))[:, None]
        ramp = (r0 - l1) * self._ramp_fraction
        knots = np.array([l0 * (1. - self._ramp_fraction), l0, l1, l1 + ramp, r0 - ramp, r0, r1, r1 * (1. + self._ramp_fraction)])
        self.window = (knots, np.array([0., 1., 1., 0., 0., 1., 1., 0.]))

    def _operator(self, rescale):
        """xinow = A xi with A = diag(1 - center) + diag(center) model^T (G W G^T)^-1 G W (restricted to ``smask`` columns): the weighted
        least-squares fit of :class:`utils.LeastSquareSolver` written as a projector, so that the whole filter is one linear map."""
        knots, weights = self.window
        precision = np.interp(self.s[self.smask] / rescale, knots, weights, left=0., right=0.)
        center = np.interp(self.s / rescale, knots[2:-2], 1. - weights[2:-2], left=0., right=0.)   # 1 between the boxes, ramps on their inner edges
        g = self.model[:, self.smask]
        hv = g * precision
        proj = np.linalg.solve(hv.dot(g.T), hv)               # (5, nmask): parameters = proj . xi[smask]
        A = np.diag(1. - center)
        A[:, self.smask] += center[:, None] * self.model.T.dot(proj)
        return A

    def _compute(self):
        rescale = self._scalar_rs_drag_ratio() if self.rescale_sbox else 1.
        key = float(rescale)
        if getattr(self, '_op_key', None) != key:
            self._op, self._op_key = LinearOperator.dense(self._operator(key), device=self.device), key
        self._xinow_rows = self._op(self._xi_rows)


def CorrelationFunctionBAOFilter(xi_interpolator, engine='kirkby2013', **kwargs):
    """Run correlation function BAO filter ``engine`` (reference bao_filter.py:924-933); available here: 'kirkby2013'."""
    engine = engine.lower()
    if engine not in RegisteredCorrelationFunctionBAOFilter._registry or engine == 'base':
        raise ValueError('Correlation function BAO filter {} is unknown'.format(engine))
    return RegisteredCorrelationFunctionBAOFilter._registry[engine](xi_interpolator, **kwargs)
