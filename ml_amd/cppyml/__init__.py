"""Python surface mirroring the reference's `cppyml` package for the clustering hot path
(cppyml/cppyml/__init__.py, cppyml/clustering.cpp): `from ml_amd.cppyml import clustering`.

Not in the reference (which is CPU-only, single process): `device_context()` returns the GPU context the model classes
run on, so that a row-sharded job can install its all-reduce hook on it before every rank calls `fit(shard)`:

    from ml_amd import cppyml, dist as mldist          # torch.distributed initialised, one rank per GPU
    mldist.install_allreduce(cppyml.device_context(), world_size, rank)
    em = cppyml.clustering.EM(K); em.fit(X[lo:hi])     # identical parameters on every rank
"""
import ctypes as _C

from . import clustering  # noqa: F401

__all__ = ["clustering", "device_context"]

_context = None


def device_context():
    """The process-wide `ml_amd._lib.Context` of the facade classes (created on first use; MLHIP_DEVICE / LOCAL_RANK pick
    the GPU). Keep the returned object alive while a hook installed through it is in use."""
    global _context
    if _context is None:
        from .. import _lib
        h = _C.c_void_p()
        _lib.check(_lib.lib.mlpp_device_context(_C.byref(h)))
        _context = _lib.Context.borrow(h)
    return _context
