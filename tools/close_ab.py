"""Closing kernel (device/em_close.hip) of two builds of the library, bit for bit: runs mlhip_em_iterate on a list of shapes and
writes every output to an .npz; run once per build (MLHIP_LIBRARY=...) and compare with --compare.
    python tools/close_ab.py out_a.npz ;  MLHIP_LIBRARY=old.so python tools/close_ab.py out_b.npz ;  python tools/close_ab.py --compare out_a.npz out_b.npz"""
import sys
import numpy as np

SHAPES = [(1, 3), (2, 3), (3, 5), (4, 16), (6, 7), (8, 32), (11, 5), (12, 9), (16, 16), (20, 6), (24, 5), (28, 3), (32, 64), (32, 4), (40, 3), (48, 3), (64, 4), (96, 3), (130, 2), (200, 2)]


def main():
    if sys.argv[1] == "--compare":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        bad = [k for k in a.files if not np.array_equal(a[k], b[k], equal_nan=True)]
        print("arrays", len(a.files), "differing", bad)
        sys.exit(1 if bad else 0)
    sys.path.insert(0, ".")
    from ml_amd import _lib
    ctx = _lib.Context()
    out = {}
    for d, K in SHAPES:
        rng = np.random.default_rng(100 * d + K)
        n = 400 * K + 1000
        means = 3.0 * rng.standard_normal((K, d))
        X = np.ascontiguousarray(means[rng.integers(0, K, n)] + rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, d) + 5.0)
        mu0 = means + 5.0 + 0.3 * rng.standard_normal((K, d))
        dt = _lib.Data(ctx, X)
        _, cov = dt.sample_covariance()
        got = dt.em_iterate(np.full(K, 1.0 / K), mu0, np.stack([cov] * K), 6, atol=0.0)
        for i, v in enumerate(got):
            out["d%d_K%d_%d" % (d, K, i)] = np.asarray(v)
        for i, v in enumerate(dt.em_step(np.full(K, 1.0 / K), mu0, np.stack([cov] * K))):      # (closing on the host at every d)
            out["d%d_K%d_step%d" % (d, K, i)] = np.asarray(v)
        dt.close()
    np.savez(sys.argv[1], **out)
    ctx.close()


if __name__ == "__main__":
    main()
