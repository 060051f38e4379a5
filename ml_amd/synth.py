"""Synthetic Gaussian-mixture workloads (SURVEY.md section 8d / BASELINE.md section 2).

K-component mixture: means mu_k = 6 z_k (z ~ N(0, I_d)); full covariances Sigma_k = A_k A_k^T / d + 0.5 I;
weights proportional to U[0.5, 1.5]. numpy's PCG64 Generator with a fixed seed makes the data identical on
every machine. Samples are drawn component by component and then shuffled with the same generator."""
import numpy as np

SEED = 20240601


class Mixture:
    def __init__(self, d, K, seed=SEED, diagonal=False):
        rng = np.random.default_rng(seed)
        self.d, self.K = d, K
        self.means = 6.0 * rng.standard_normal((K, d))
        self.covs = np.empty((K, d, d))
        for k in range(K):
            if diagonal:
                self.covs[k] = np.diag(rng.uniform(0.5, 2.0, d))
            else:
                A = rng.standard_normal((d, d))
                self.covs[k] = A @ A.T / d + 0.5 * np.eye(d)
        w = rng.uniform(0.5, 1.5, K)
        self.weights = w / w.sum()
        self.chols = np.linalg.cholesky(self.covs)
        self._seed = seed

    def sample(self, n, stream=0, threads=None):
        """n x d samples (C-contiguous float64) and their component labels. `stream` selects an independent
        substream (e.g. the rank of a row shard), so shards of different ranks never repeat samples. Blocks of 2^20
        samples have their own generator keyed by (seed, stream, block), so the result does not depend on `threads`."""
        import os
        from concurrent.futures import ThreadPoolExecutor
        X = np.empty((n, self.d))
        comp = np.empty(n, dtype=np.int32)
        block = 1 << 20

        def fill(b):
            lo, hi = b * block, min(n, (b + 1) * block)
            rng = np.random.default_rng([self._seed, 1000 + stream, b])
            c = rng.choice(self.K, size=hi - lo, p=self.weights).astype(np.int32)
            z = rng.standard_normal((hi - lo, self.d))
            order = np.argsort(c, kind="stable")          # group the block by component: x = mu_k + L_k z
            zs = z[order]
            edges = np.searchsorted(c[order], np.arange(self.K + 1))
            for k in range(self.K):
                a, e = edges[k], edges[k + 1]
                if e > a:
                    zs[a:e] = self.means[k] + zs[a:e] @ self.chols[k].T
            X[lo:hi][order] = zs
            comp[lo:hi] = c

        nblocks = (n + block - 1) // block
        if threads is None:
            threads = max(1, min(8, (os.cpu_count() or 2) // 2))
        if nblocks <= 1 or threads <= 1:
            for b in range(nblocks):
                fill(b)
        else:
            with ThreadPoolExecutor(threads) as pool:
                list(pool.map(fill, range(nblocks)))
        return X, comp

    def initial_means(self, seed=7):
        """Deterministic start: true means + 0.5 N(0, I) (a user-supplied centroids initialiser)."""
        rng = np.random.default_rng(seed)
        return self.means + 0.5 * rng.standard_normal((self.K, self.d))
