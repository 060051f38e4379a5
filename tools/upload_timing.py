import sys, time, numpy as np
sys.path.insert(0, '.')
from ml_amd import _lib, synth
mix = synth.Mixture(32, 64)
X, _ = mix.sample(10_000_000)
ctx = _lib.Context()
for rep in range(3):
    t0 = time.perf_counter(); dt = _lib.Data(ctx, X); t1 = time.perf_counter()
    m, c = dt.sample_covariance(); t2 = time.perf_counter()
    print("upload %.3f s (%.1f GB/s)  sample_cov %.3f s" % (t1 - t0, X.nbytes / (t1 - t0) / 1e9, t2 - t1))
    dt.close()
t0 = time.perf_counter(); Y = X.copy(); print("host memcpy %.3f s (%.1f GB/s)" % (time.perf_counter() - t0, X.nbytes / (time.perf_counter() - t0) / 1e9))
