"""Python surface mirroring the reference's `cppyml` package for the clustering hot path
(cppyml/cppyml/__init__.py, cppyml/clustering.cpp): `from ml_amd.cppyml import clustering`."""
from . import clustering  # noqa: F401

__all__ = ["clustering"]
