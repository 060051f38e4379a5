"""The whole facade / Python-surface test module once more with the process-wide context being a device GROUP picked up from the
environment (MLHIP_NUM_GPUS=3: three shards, on a one-GPU box all on GPU 0): every fit of tests/test_gpu_facade.py -- all
initialisers, both start modes, multi-initialisation K-means, verbose fits, the sklearn pin -- must come out the same when ONE
`fit(X)` call is row-sharded inside the library (reference API: ML/EM.cpp:91, ML/KMeans.cpp:25; cppyml/clustering.cpp:27-30)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_facade_tests_pass_with_a_group_as_the_default_context():
    env = dict(os.environ, MLHIP_NUM_GPUS="3")
    env.pop("LOCAL_RANK", None)
    env.pop("MLHIP_DEVICES", None)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_facade.py"),
                          os.path.join(ROOT, "tests", "test_cppyml_alias.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
