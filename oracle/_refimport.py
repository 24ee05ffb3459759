"""Import the reference package from /root/reference (this container only; TEST INFRASTRUCTURE).

Shim from SURVEY.md App. B: the reference calls importlib.metadata.version("cosmoprimo") at import,
which fails for an un-installed tree.  Nothing from the reference is copied: it is only *run* to
produce golden vectors (plain float64 arrays).
"""
import os
import sys
import importlib.metadata as _md

REFERENCE_ROOT = '/root/reference'


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'cosmoprimo'))


def import_reference():
    if not available():
        raise ImportError('reference tree not present (expected on the build container only)')
    _v = _md.version
    if not getattr(_md.version, '_cp_shim', False):
        def version(name):
            return '1.0.0' if name == 'cosmoprimo' else _v(name)
        version._cp_shim = True
        _md.version = version
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import cosmoprimo
    return cosmoprimo
