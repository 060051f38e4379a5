"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper over oracle/libmlpp_oracle.so (see oracle/README.md).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Data convention follows the Python surface of the reference (cppyml/clustering.cpp:27-30): numpy
float64 C-contiguous N x d, which *is* the d x N column-major matrix the C++ side wants.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmlpp_oracle.so")

_dp = C.POINTER(C.c_double)
_up = C.POINTER(C.c_uint)

FORGY, RANDOM_PARTITION, KPP, FIXED = 0, 1, 2, 3


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def _load():
    path = os.environ.get("MLPP_ORACLE_LIBRARY") or _LIB_PATH      # override: the sanitizer build (`make -C oracle SANITIZE=...`)
    if path == _LIB_PATH and not os.path.exists(_LIB_PATH):
        build()
    lib = C.CDLL(path)
    lib.orc_last_error.restype = C.c_char_p
    for name in ("orc_em_log_likelihood", "orc_km_inertia", "orc_em_time_iterations", "orc_km_time_steps"):
        getattr(lib, name).restype = C.c_double
    for name in ("orc_em_steps_done", "orc_km_steps_done"):
        getattr(lib, name).restype = C.c_uint
    return lib


lib = _load()


class OracleError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code  # -1 invalid_argument, -2 domain_error, -3 other


def _check(rc):
    if rc != 0:
        raise OracleError(rc, lib.orc_last_error().decode())


def _d(a):
    return a.ctypes.data_as(_dp)


def _as_data(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    assert x.ndim == 2
    return x


def xAx_symmetric(A, x):
    A = np.asfortranarray(A, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = C.c_double()
    _check(lib.orc_xAx_symmetric(_d(A), A.shape[0], A.shape[1], _d(x), x.size, C.byref(out)))
    return out.value


def xxT(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    dest = np.empty((x.size, x.size), order="F")
    _check(lib.orc_xxT(_d(x), x.size, _d(dest)))
    return dest


def add_a_xxT(x, dest, a):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.array(dest, dtype=np.float64, order="F")
    _check(lib.orc_add_a_xxT(_d(x), x.size, _d(out), out.shape[0], out.shape[1], C.c_double(a)))
    return out


def calculate_XXt_beta(X, y, lam):
    """X: N x q (rows = data points), y: N, lam: q. Returns (XXt q x q, beta q)."""
    X = _as_data(X)
    y = np.ascontiguousarray(y, dtype=np.float64)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    n, q = X.shape
    XXt, beta = np.empty((q, q)), np.empty(q)
    _check(lib.orc_calculate_XXt_beta(_d(X), q, n, _d(y), y.size, _d(lam), lam.size, _d(XXt), _d(beta)))
    return XXt, beta


def init_centroids(kind, data, K, seed=None):
    """Returns K x d (row-major view of the d x K column-major result)."""
    data = _as_data(data)
    n, d = data.shape
    out = np.empty((K, d))
    _check(lib.orc_init_centroids(kind, _d(data), d, n, d, K, int(seed is not None), C.c_uint(seed or 0), _d(out)))
    return out


def init_closest_centroid(kind, data, K, seed=None):
    data = _as_data(data)
    n, d = data.shape
    out = np.empty((n, K), order="F")
    _check(lib.orc_init_closest_centroid(kind, _d(data), d, n, d, K, int(seed is not None), C.c_uint(seed or 0), _d(out)))
    return out


def sample_covariance(data):
    data = _as_data(data)
    n, d = data.shape
    out = np.empty((d, d), order="F")
    _check(lib.orc_sample_covariance(_d(data), d, n, d, _d(out)))
    return out


class EM:
    def __init__(self, K):
        self.K = K
        self._h = C.c_void_p()
        _check(lib.orc_em_create(K, C.byref(self._h)))
        self.d = None
        self.n = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.orc_em_destroy(self._h)
            self._h = None

    def set_seed(self, s): _check(lib.orc_em_set_seed(self._h, C.c_uint(s)))
    def set_absolute_tolerance(self, t): _check(lib.orc_em_set_absolute_tolerance(self._h, C.c_double(t)))
    def set_relative_tolerance(self, t): _check(lib.orc_em_set_relative_tolerance(self._h, C.c_double(t)))
    def set_maximum_steps(self, m): _check(lib.orc_em_set_maximum_steps(self._h, C.c_uint(m)))
    def set_maximise_first(self, b): _check(lib.orc_em_set_maximise_first(self._h, int(b)))

    def set_covariance_type(self, kind):
        """'full' (the reference) or 'diag' (extension: BASELINE.json configs[1]; see mlpp_oracle.hpp)."""
        if kind not in ("full", "diag"):
            raise ValueError("covariance type must be 'full' or 'diag'")
        _check(lib.orc_em_set_diagonal(self._h, int(kind == "diag")))

    def set_means_initialiser(self, kind, fixed=None):
        """fixed: K x d array of initial means when kind == FIXED."""
        if kind == FIXED:
            fixed = np.ascontiguousarray(fixed, dtype=np.float64)
            _check(lib.orc_em_set_means_initialiser(self._h, kind, _d(fixed), fixed.shape[1]))
        else:
            _check(lib.orc_em_set_means_initialiser(self._h, kind, None, 0))

    def set_responsibilities_initialiser(self, kind, fixed=None):
        if kind == FIXED:
            fixed = np.ascontiguousarray(fixed, dtype=np.float64)
            _check(lib.orc_em_set_responsibilities_initialiser(self._h, kind, _d(fixed), fixed.shape[1]))
        else:
            _check(lib.orc_em_set_responsibilities_initialiser(self._h, kind, None, 0))

    def fit(self, data):
        data = _as_data(data)
        self.n, self.d = data.shape
        conv = C.c_int()
        _check(lib.orc_em_fit(self._h, _d(data), self.d, self.n, self.d, C.byref(conv)))
        return bool(conv.value)

    def set_parameters(self, means, covs, pis):
        """means: K x d; covs: K x d x d (symmetric); pis: K."""
        means = np.ascontiguousarray(means, dtype=np.float64)
        covs = np.ascontiguousarray(covs, dtype=np.float64)
        pis = np.ascontiguousarray(pis, dtype=np.float64)
        self.d = means.shape[1]
        _check(lib.orc_em_set_parameters(self._h, self.d, _d(means), _d(covs), _d(pis)))

    def set_responsibilities(self, R, d):
        R = np.asfortranarray(R, dtype=np.float64)
        self.n, self.d = R.shape[0], d
        _check(lib.orc_em_set_responsibilities(self._h, _d(R), d, self.n))

    def expectation_step(self, data):
        data = _as_data(data)
        self.n, self.d = data.shape
        _check(lib.orc_em_expectation_step(self._h, _d(data), self.d, self.n, self.d))

    def maximisation_step(self, data):
        data = _as_data(data)
        _check(lib.orc_em_maximisation_step(self._h, _d(data), self.d, self.n, self.d))

    def calculate_labels(self): _check(lib.orc_em_calculate_labels(self._h))

    def assign_responsibilities(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        u = np.empty(self.K)
        _check(lib.orc_em_assign_responsibilities(self._h, _d(x), x.size, _d(u), u.size))
        return u

    def time_iterations(self, data, iters):
        data = _as_data(data)
        self.n, self.d = data.shape
        return lib.orc_em_time_iterations(self._h, _d(data), self.d, self.n, self.d, iters)

    @property
    def log_likelihood(self): return lib.orc_em_log_likelihood(self._h)
    @property
    def converged(self): return bool(lib.orc_em_converged(self._h))
    @property
    def steps_done(self): return lib.orc_em_steps_done(self._h)

    @property
    def means(self):
        """K x d (row k = mean of component k)."""
        out = np.empty((self.K, self.d))
        lib.orc_em_get_means(self._h, _d(out))
        return out

    @property
    def mixing_probabilities(self):
        out = np.empty(self.K)
        lib.orc_em_get_mixing_probabilities(self._h, _d(out))
        return out

    @property
    def covariances(self):
        out = np.empty((self.K, self.d, self.d))
        lib.orc_em_get_covariances(self._h, _d(out))
        return out

    @property
    def inverse_covariances(self):
        out = np.empty((self.K, self.d, self.d))
        lib.orc_em_get_inverse_covariances(self._h, _d(out))
        return out

    @property
    def sqrt_dets(self):
        out = np.empty(self.K)
        lib.orc_em_get_sqrt_dets(self._h, _d(out))
        return out

    @property
    def responsibilities(self):
        out = np.empty((self.n, self.K), order="F")
        lib.orc_em_get_responsibilities(self._h, _d(out))
        return out

    @property
    def labels(self):
        out = np.empty(self.n, dtype=np.uint32)
        lib.orc_em_get_labels(self._h, out.ctypes.data_as(_up))
        return out


class KMeans:
    def __init__(self, K):
        self.K = K
        self._h = C.c_void_p()
        _check(lib.orc_km_create(K, C.byref(self._h)))
        self.d = None
        self.n = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.orc_km_destroy(self._h)
            self._h = None

    def set_seed(self, s): _check(lib.orc_km_set_seed(self._h, C.c_uint(s)))
    def set_absolute_tolerance(self, t): _check(lib.orc_km_set_absolute_tolerance(self._h, C.c_double(t)))
    def set_maximum_steps(self, m): _check(lib.orc_km_set_maximum_steps(self._h, C.c_uint(m)))
    def set_number_initialisations(self, n): _check(lib.orc_km_set_number_initialisations(self._h, C.c_uint(n)))

    def set_centroids_initialiser(self, kind, fixed=None):
        if kind == FIXED:
            fixed = np.ascontiguousarray(fixed, dtype=np.float64)
            _check(lib.orc_km_set_centroids_initialiser(self._h, kind, _d(fixed), fixed.shape[1]))
        else:
            _check(lib.orc_km_set_centroids_initialiser(self._h, kind, None, 0))

    def fit(self, data):
        data = _as_data(data)
        self.n, self.d = data.shape
        conv = C.c_int()
        _check(lib.orc_km_fit(self._h, _d(data), self.d, self.n, self.d, C.byref(conv)))
        return bool(conv.value)

    def set_centroids(self, c, n):
        c = np.ascontiguousarray(c, dtype=np.float64)
        self.d, self.n = c.shape[1], n
        _check(lib.orc_km_set_centroids(self._h, _d(c), self.d, n))

    def assignment_step(self, data):
        data = _as_data(data)
        _check(lib.orc_km_assignment_step(self._h, _d(data), self.d, self.n, self.d))

    def update_step(self, data):
        data = _as_data(data)
        _check(lib.orc_km_update_step(self._h, _d(data), self.d, self.n, self.d))

    def assign_label(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        label, dist = C.c_uint(), C.c_double()
        _check(lib.orc_km_assign_label(self._h, _d(x), C.byref(label), C.byref(dist)))
        return label.value, dist.value

    def time_steps(self, data, iters):
        data = _as_data(data)
        return lib.orc_km_time_steps(self._h, _d(data), self.d, self.n, self.d, iters)

    @property
    def inertia(self): return lib.orc_km_inertia(self._h)
    @property
    def converged(self): return bool(lib.orc_km_converged(self._h))
    @property
    def steps_done(self): return lib.orc_km_steps_done(self._h)

    @property
    def centroids(self):
        """K x d."""
        out = np.empty((self.K, self.d))
        lib.orc_km_get_centroids(self._h, _d(out))
        return out

    @property
    def labels(self):
        out = np.empty(self.n, dtype=np.uint32)
        lib.orc_km_get_labels(self._h, out.ctypes.data_as(_up))
        return out


# Numeric fixture of Tests/test_EM.cpp:16-23 / Tests/test_KMeans.cpp:16-23 (data, not code).
TWO_GAUSSIANS_MEANS = np.array([[0.4, 0.11, 0.5], [-1.2, 2.2, 1.6]])       # row k = mean of component k
TWO_GAUSSIANS_SIGMAS = np.array([[0.05, 0.04, 0.01], [0.2, 0.1, 0.2]])
TWO_GAUSSIANS_P0 = 0.25


def testdata_two_gaussians(n=400):
    data = np.empty((n, 3))
    truth = np.empty(n, dtype=np.uint32)
    lib.orc_testdata_two_gaussians(n, C.c_double(TWO_GAUSSIANS_P0), _d(np.ascontiguousarray(TWO_GAUSSIANS_MEANS)),
                                   _d(np.ascontiguousarray(TWO_GAUSSIANS_SIGMAS)), _d(data), truth.ctypes.data_as(_up))
    return data, truth


def testdata_mousie(n):
    data = np.empty((n, 2))
    classes = np.empty(n, dtype=np.uint32)
    lib.orc_testdata_mousie(n, _d(data), classes.ctypes.data_as(_up))
    return data, classes
