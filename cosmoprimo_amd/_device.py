"""Small helpers shared by the host-side modules: device resolution, numpy <-> torch plumbing, cp_param packing."""
import collections
import ctypes
import hashlib

import numpy as np

from . import _lib


class Copyable(object):

    """``copy()`` of the reference's BaseClass (utils.py): a shallow copy; device tables and plans are shared, not duplicated."""

    def copy(self):
        import copy
        return copy.copy(self)


def torch():
    import torch as _torch
    return _torch


def is_torch(x):
    return type(x).__module__.startswith('torch')


def resolve_device(device=None, *tensors):
    """torch.device to run on: explicit ``device``, else the device of the first CUDA tensor, else the current CUDA device."""
    t = torch()
    if device is None:
        for x in tensors:
            if is_torch(x) and x.is_cuda:
                device = x.device
                break
    if device is None:
        if not t.cuda.is_available():
            raise RuntimeError('cosmoprimo_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path')
        device = t.device('cuda', t.cuda.current_device())
    device = t.device(device if not isinstance(device, int) else 'cuda:{:d}'.format(device))
    if device.type != 'cuda':
        raise ValueError('cosmoprimo_amd runs on a GPU; got device {}'.format(device))
    if device.index is None:
        device = t.device('cuda', t.cuda.current_device())
    return device


_upload_cache = collections.OrderedDict()      # (device index, dtype, shape, digest of the bytes) -> (tensor, event, stream), least recently used first
_upload_held = {}                              # entries a HIP graph was recorded on: never evicted (a replay reads their raw pointers)
_UPLOAD_CACHE_ENTRIES, _UPLOAD_CACHE_BYTES = 1024, 1 << 16


def upload(x, device, cache=True):
    """Tensor on ``device`` from a numpy array / number / tensor, dtype kept.

    cache=True (grids, masks, index lists, weights: the constants pipelines upload again and again): host arrays up to 64 KiB are kept on the
    device, keyed by their content; a second upload of the same values is a dictionary lookup and returns the SAME tensor -- it is read-only by
    contract (an in-place write would change what every later upload of these values returns).  A function that has run once can then be recorded
    into a HIP graph (``torch.cuda.graph``) without any copy in it; entries that are created or served while a graph is being captured are held
    for the life of the process, because a replay reads their raw device pointers (the others are evicted least recently used first).  An entry
    served on another stream than the one its copy was queued on is ordered behind that copy by an event.

    cache=False (per-call parameter vectors): a private tensor, uploaded through a page-locked block and an asynchronous copy.

    Arrays up to 1 MiB go through a page-locked block of torch's caching host allocator: a copy from pageable memory blocks the host until
    everything queued on the stream before it has run, i.e. every small upload in the middle of a pipeline was a device synchronisation (21 of them
    per 16 384-vector chunk of the wallish2018 filter)."""
    t = torch()
    if is_torch(x):
        return x.to(device=device)
    shape = np.shape(x)
    a = np.ascontiguousarray(x)
    if a.dtype.byteorder not in '=|':
        a = a.astype(a.dtype.newbyteorder('='))
    capturing = t.cuda.is_current_stream_capturing()
    key = None
    if cache and a.nbytes <= _UPLOAD_CACHE_BYTES:
        key = (t.device(device).index, a.dtype.str, shape, hashlib.blake2b(a.tobytes(), digest_size=16).digest())
        hit = _upload_held.get(key)
        if hit is None:
            hit = _upload_cache.get(key)
            if hit is not None:
                if capturing:       # the graph being recorded will read this tensor at every replay: out of the evictable set
                    _upload_held[key] = _upload_cache.pop(key)
                else:
                    _upload_cache.move_to_end(key)
        if hit is not None:
            tensor, event, stream = hit
            if not capturing:
                current = t.cuda.current_stream(tensor.device)
                if current.cuda_stream != stream:
                    current.wait_event(event)       # the copy was queued on another stream
            return tensor
    if capturing:
        raise RuntimeError('a host array is uploaded while a HIP graph is being captured: call the function once before capturing it, so that its '
                           'constants are on the device (per-call values must be device tensors)')
    h = t.from_numpy(a) if a.flags.writeable else t.from_numpy(a.copy())
    out = None
    if 0 < a.nbytes <= (1 << 20):
        try:
            out = h.pin_memory().to(device, non_blocking=True).reshape(shape)
        except RuntimeError:    # no page-locked memory left: the blocking copy below is the same result
            out = None
    if out is None:
        out = h.to(device).reshape(shape)
    if key is not None:
        current = t.cuda.current_stream(out.device)
        event = t.cuda.Event()
        event.record(current)
        _upload_cache[key] = (out, event, current.cuda_stream)
        if len(_upload_cache) > _UPLOAD_CACHE_ENTRIES:
            _upload_cache.popitem(last=False)
    return out


def to_device(x, device, cache=True):
    """float64 contiguous tensor on ``device`` from a number / numpy array / tensor (shape preserved).  Small host arrays come from the content-keyed
    cache of :func:`upload` and are shared: treat the result as read-only, or pass ``cache=False`` for a private tensor (per-call parameters)."""
    t = torch()
    if is_torch(x):
        return x.to(device=device, dtype=t.float64).contiguous()
    return upload(np.asarray(x, dtype='f8'), device, cache=cache)


def to_host(x):
    """numpy array from a device tensor.  Results between 4 MiB and 1 GiB (batches of filtered spectra, P(k, z) tables) land in a page-locked
    buffer of torch's caching host allocator, which the returned array keeps alive: the copy then runs at the PCIe rate instead of the
    pageable-memory rate (which is a staged copy plus a host memcpy).  Larger results take the pageable path so that a caller holding
    several of them does not pin gigabytes of host memory."""
    t = torch()
    if not is_torch(x):
        return np.asarray(x)
    x = x.detach()
    if x.is_cuda and (1 << 22) <= x.numel() * x.element_size() <= (1 << 30):
        try:
            buf = t.empty(x.shape, dtype=x.dtype, pin_memory=True)
        except RuntimeError:    # no page-locked memory left: the pageable copy below is the same result, slower
            buf = None
        if buf is not None:
            buf.copy_(x, non_blocking=True)
            t.cuda.current_stream(x.device).synchronize()
            return buf.numpy()
    return x.cpu().numpy()


def screen_rows(x, require_positive=False, with_scale=False):
    """One pass over the rows of the contiguous device tensor ``x`` (..., n) (``cp_rows_screen``): ``ok`` (..., 1) bool, False for rows
    holding a NaN / Inf (or a value <= 0 with ``require_positive``), and -- ``with_scale`` -- the power of two >= max |row| (..., 1)."""
    t = torch()
    n = x.shape[-1]
    nrows = x.numel() // n if n else 0
    ok = t.empty(tuple(x.shape[:-1]) + (1,), dtype=t.uint8, device=x.device)
    scale = t.empty(tuple(x.shape[:-1]) + (1,), dtype=t.float64, device=x.device) if with_scale else None
    _lib.check(_lib.load().cp_rows_screen(x.data_ptr(), nrows, n, int(bool(require_positive)), ok.data_ptr(), scale.data_ptr() if with_scale else None,
                                          x.device.index, stream_of(x.device)))
    ok = ok.to(t.bool)
    return (ok, scale) if with_scale else ok


def stream_of(device):
    return torch().cuda.current_stream(device).cuda_stream


def float_dtype(*args):
    """Reference utils._bcast_dtype (utils.py:88-95): float32 only if every array input is float32, else float64."""
    t = torch()
    dts = []
    for a in args:
        if is_torch(a):
            dts.append({t.float32: np.float32, t.float64: np.float64}.get(a.dtype, np.float64))
        elif hasattr(a, 'dtype'):
            dts.append(a.dtype)
    if not dts:
        return np.dtype('f8')
    out = np.result_type(*dts)
    return out if np.issubdtype(out, np.floating) else np.dtype('f8')


def pack_params(names, params, defaults, device):
    """(ctypes cp_param array, ncosmo or None, keepalive list) for per-cosmology parameters given as floats or (ncosmo,) arrays."""
    carr = (_lib.cp_param * len(names))()
    keep, ncosmo = [], None
    for i, name in enumerate(names):
        v = params.get(name, defaults[name])
        if not is_torch(v) and np.ndim(v) == 0:
            carr[i].ptr, carr[i].value = None, float(v)
            continue
        tv = to_device(v, device, cache=False).reshape(-1)      # per-call parameter vectors: private, not cached
        if ncosmo is not None and tv.numel() != ncosmo:
            raise ValueError('parameter arrays must share one length, got {} and {}'.format(ncosmo, tv.numel()))
        ncosmo = tv.numel()
        keep.append(tv)
        carr[i].ptr, carr[i].value = tv.data_ptr(), 0.
    return carr, ncosmo, keep


def ncdm_arg(bg, ncosmo):
    """The ``const cp_ncdm*`` argument for a background parameter block: ``bg['ncdm']`` (a :class:`cosmoprimo_amd.background.NcdmTables`, what
    :meth:`BaseEngine.bg_params` adds for cosmologies with massive neutrinos) as (ctypes reference or None, the struct to keep alive)."""
    ncdm = bg.get('ncdm', None) if bg else None
    if ncdm is None or not ncdm.nspecies:
        return None, None
    if ncdm.ncosmo != ncosmo:
        raise ValueError('massive-neutrino tables hold {:d} cosmologies, the parameters {:d}'.format(ncdm.ncosmo, ncosmo))
    cn = ncdm.struct()
    return ctypes.byref(cn), cn


def as_void_p(carr):
    return ctypes.cast(carr, ctypes.c_void_p)
