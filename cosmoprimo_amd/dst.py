"""Batched orthonormal DST-II / DST-III of rows on the GPU (``cp_dst_*``): the scipy.fftpack.dst / idst(type=2, norm='ortho') of the
wallish2018 BAO filter (reference bao_filter.py:371-372, 412)."""
import ctypes

import numpy as np

from . import _lib
from . import _device as dv


class DST(object):

    """Plan for rows of length ``n`` (256, 1024 or 4096); ``kx``: optional abscissa for the fused log(kx x) / exp(y)/kx maps."""

    def __init__(self, n, kx=None, device=None):
        self.device = dv.resolve_device(device)
        self.n = int(n)
        self._handle = ctypes.c_void_p()
        kxp = None
        if kx is not None:
            kx = np.ascontiguousarray(kx, dtype='f8')
            if kx.size != self.n:
                raise ValueError('kx must have length {:d}'.format(self.n))
            kxp = _lib.as_double_p(kx)
        _lib.check(_lib.load().cp_dst_plan_create(ctypes.byref(self._handle), self.n, kxp, self.device.index))

    def __call__(self, x, inverse=False, fused=False, split=False):
        """x : (..., n) -> (..., n) device tensor: dst (or idst) type 2, norm='ortho', along the last axis.  ``split``: the coefficients are
        stored de-interleaved, even-indexed ones in the first half of the row and odd-indexed ones in the second."""
        torch = dv.torch()
        x = dv.to_device(x, self.device)
        if x.shape[-1] != self.n:
            raise ValueError('last dimension must be {:d}, got {}'.format(self.n, tuple(x.shape)))
        nrows = x.numel() // self.n
        # rows are independent, as in scipy's row-by-row transform: the kernel itself keeps a row that is not finite (or, for the fused log map,
        # not positive) away from the row it shares a complex FFT with, and stores NaN for it
        out = torch.empty_like(x)
        if nrows:
            _lib.check(_lib.load().cp_dst_execute(self._handle, x.data_ptr(), out.data_ptr(), nrows, int(bool(inverse)), int(bool(fused)) | (2 if split else 0),
                                                  dv.stream_of(self.device)))
        return out

    def forward_analytic(self, engine, bg, pk, split=False, box=None):
        """dst(log(kx P_c(kx))) of a batch of cosmologies of an analytic engine with the spectra evaluated inside the transform kernel
        (``cp_dst_forward_analytic``): (ncosmo, n), or None when the parameters are not a batch or the plan is not wallish2018's (n = 4096 with kx).
        box = (margin_first, margin_second, offset_first, offset_second): also the next step of the filter in the kernel's epilogue
        (``cp_dst_forward_analytic_box``: split layout, boxes rewritten); returns (coefficients, boxes (2 ncosmo, 2) int32)."""
        torch = dv.torch()
        from .background import DEFAULTS as bg_defaults
        from .power import PK_DEFAULTS
        if self.n != 4096 or engine not in _lib.ENGINES:
            return None
        cbg, n1, keep1 = dv.pack_params(_lib.BG_PARAMS, bg, bg_defaults, self.device)
        cpk, n2, keep2 = dv.pack_params(_lib.PK_PARAMS, pk, PK_DEFAULTS, self.device)
        sizes = {n for n in (n1, n2) if n is not None}
        if len(sizes) != 1:
            return None
        ncosmo = sizes.pop()
        lib = _lib.load()
        out = torch.empty((ncosmo, self.n), dtype=torch.float64, device=self.device)
        work = torch.empty(int(lib.cp_dst_forward_analytic_workspace_bytes(ncosmo)), dtype=torch.uint8, device=self.device)
        nu, keep_nu = dv.ncdm_arg(bg, ncosmo)
        if box is not None:
            boxes = torch.empty((2 * ncosmo, 2), dtype=torch.int32, device=self.device)
            _lib.check(lib.cp_dst_forward_analytic_box(self._handle, _lib.ENGINES[engine], ncosmo, dv.as_void_p(cbg), 0, nu, dv.as_void_p(cpk), out.data_ptr(),
                                                       work.data_ptr(), boxes.data_ptr(), int(box[0]), int(box[1]), int(box[2]), int(box[3]), dv.stream_of(self.device)))
            return out, boxes
        _lib.check(lib.cp_dst_forward_analytic(self._handle, _lib.ENGINES[engine], ncosmo, dv.as_void_p(cbg), 0, nu, dv.as_void_p(cpk), out.data_ptr(), work.data_ptr(),
                                               2 if split else 0, dv.stream_of(self.device)))
        return out

    def __del__(self):
        try:
            if self._handle:
                _lib.load().cp_dst_plan_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
