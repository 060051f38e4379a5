"""The C-ABI library loads without a GPU and exports every symbol the headers in include/ declare. CPU only."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(ml(?:hip|pp)_[a-zA-Z0-9_]+)\s*\(", text)
    return sorted(set(n for n in names if not n.endswith("_fn")))


@pytest.mark.parametrize("header", ["mlhip.h", "mlpp_c.h"])
def test_every_declared_symbol_is_exported(header):
    from ml_amd import _lib
    names = declared_functions(header)
    assert len(names) > 20
    missing = [n for n in names if not hasattr(_lib.lib, n)]
    assert not missing, missing


def test_library_is_the_in_tree_hip_build():
    from ml_amd import _lib
    # (MLHIP_LIBRARY selects another in-tree build -- the sanitizer build of tests/test_sanitizers.py -- never a library from elsewhere)
    assert os.path.dirname(os.path.realpath(_lib.LIB_PATH)) == os.path.realpath(os.path.join(ROOT, "ml_amd"))
    assert "MLHIP_LIBRARY" in os.environ or os.path.basename(_lib.LIB_PATH) == "libmlhip.so"
    assert b"gfx950" in _lib.lib.mlhip_version()
    # the code object for gfx950 is embedded in the shared library
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob


def test_no_cpu_fallback_without_device():
    """Without a GPU every compute entry point must fail loudly (skipped where a GPU is present)."""
    import numpy as np
    from ml_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.NoDeviceError):
        _lib.Context()
    with pytest.raises(_lib.NoDeviceError):
        _lib.Context.group(2, device_ids=[0, 0])                 # a device group needs its GPUs just the same
    with pytest.raises(ValueError):
        _lib.Context.group(0)                                    # (argument errors come first: 1 to 64 shards)
    from ml_amd.cppyml import clustering
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        clustering.EM(2).fit(np.zeros((10, 3)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        clustering.KMeans(2).fit(np.zeros((10, 3)))


def test_product_does_not_reference_the_oracle():
    """The shipped package must not import, link or mention the test oracle."""
    for base, _, files in os.walk(os.path.join(ROOT, "ml_amd")):
        if os.sep + "build" in base:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(base, f)
    for base, _, files in os.walk(os.path.join(ROOT, "include")):
        for f in files:
            assert "oracle" not in open(os.path.join(base, f)).read().lower()
    out = os.popen(f"ldd {os.path.join(ROOT, 'ml_amd', 'libmlhip.so')}").read()
    assert "oracle" not in out
