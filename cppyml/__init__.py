"""Import path of the reference's Python package: `from cppyml import clustering` (reference cppyml/cppyml/__init__.py:17,
cppyml/tests/test_clustering.py:13) resolves to the MI355X implementation in ml_amd.cppyml -- scripts written against the
reference run unchanged. Only the clustering module is on this repository's path; the reference's other
submodules (decision_trees, linear_regression, logistic_regression) are out of scope and raise ImportError if asked for."""
from ml_amd.cppyml import clustering, device_context

__all__ = ["clustering", "device_context"]
