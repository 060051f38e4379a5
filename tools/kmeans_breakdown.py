#!/usr/bin/env python3
"""Where the K-means step goes (N=12.5M, d=8, K=256 by default): kernel time with / without the update sums, per kernel variant."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from ml_amd import _lib, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    mix = synth.Mixture(d, K, seed=77, diagonal=True)
    X, _ = mix.sample(n)
    ctx = _lib.Context()
    data = _lib.Data(ctx, X)
    C = mix.means + 0.3 * np.random.default_rng(1).standard_normal((K, d))
    for _ in range(3):
        _, _, _, C = data.kmeans_step(C)
    out = {"N": n, "d": d, "K": K}
    ctx.timing_enable(True)
    for name, fn in (("step", lambda: data.kmeans_step(C)), ("assign_only", lambda: data.kmeans_assign(C))):
        ctx.timing_reset()
        for _ in range(5):
            fn()
        ms, cnt = ctx.timing_get("kmeans_assign")
        out[name + "_kernel_ms"] = ms
    print(json.dumps(out))


if __name__ == "__main__":
    main()
