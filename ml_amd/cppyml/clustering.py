"""Clustering algorithms -- same classes, methods, properties and argument conventions as the reference's
pybind11 module `cppyml.clustering` (cppyml/clustering.cpp:75-185), running on an MI355X through libmlhip.so.

Conventions kept from the reference:
  * `fit(data)` takes a float64 C-contiguous `N x d` array and refuses anything else with TypeError
    (`py::arg("data").noconvert()`, clustering.cpp:115,161);
  * `EM.means` is `d x K` (NOT transposed, clustering.cpp:126), `KMeans.centroids` is `K x d` (clustering.cpp:66-69,172);
  * `KMeans.labels` is a Python list (pybind11/stl.h conversion, clustering.cpp:173);
  * std::invalid_argument / std::domain_error surface as ValueError.
Extensions (not in the reference surface): `EM.labels`, `EM.converged`, `EM.steps_done`, `EM.responsibilities_rows`, `KMeans.converged`,
`KMeans.steps_done`, `KMeans.labels_array`, `FixedCentroids`.
"""
import ctypes as C

import numpy as np

from .. import _lib

_l = _lib.lib
_check = _lib.check
_dp = _lib.dptr


def _require_data(data):
    if not (isinstance(data, np.ndarray) and data.dtype == np.float64 and data.ndim == 2 and data.flags.c_contiguous):
        raise TypeError("fit(): incompatible function arguments. data must be a C-contiguous float64 array of shape (N, d)")
    return data


def _vector(x):
    return np.ascontiguousarray(x, dtype=np.float64).ravel()


class CentroidsInitialiser:
    """Abstract centroids initialiser."""

    def __init__(self, *args, **kwargs):
        raise TypeError(f"{type(self).__module__}.{type(self).__qualname__}: No constructor defined!")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _l.mlpp_centroids_initialiser_destroy(h)
            self._h = None

    def _run(self, data, number_components, seed=None):
        """Test hook: runs the initialiser on N x d data, returns K x d centroids."""
        data = _require_data(data)
        n, d = data.shape
        out = np.empty((number_components, d))
        _check(_l.mlpp_centroids_initialiser_run(self._h, _dp(data), C.c_uint64(n), d, number_components,
                                                 int(seed is not None), C.c_uint32(seed or 0), _dp(out)))
        return out


    def _run_on_device(self, data, number_components, seed=None):
        """Test hook: the same through the path `fit` takes (data uploaded, the O(N) passes on the GPU)."""
        data = _require_data(data)
        n, d = data.shape
        out = np.empty((number_components, d))
        _check(_l.mlpp_centroids_initialiser_run_on_device(self._h, _dp(data), C.c_uint64(n), d, number_components,
                                                           int(seed is not None), C.c_uint32(seed or 0), _dp(out)))
        return out


class ResponsibilitiesInitialiser:
    """Abstract responsibilities initialiser."""

    def __init__(self, *args, **kwargs):
        raise TypeError(f"{type(self).__module__}.{type(self).__qualname__}: No constructor defined!")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _l.mlpp_responsibilities_initialiser_destroy(h)
            self._h = None

    def _run(self, data, number_components, seed=None):
        data = _require_data(data)
        n, d = data.shape
        out = np.empty((n, number_components), order="F")
        _check(_l.mlpp_responsibilities_initialiser_run(self._h, _dp(data), C.c_uint64(n), d, number_components,
                                                        int(seed is not None), C.c_uint32(seed or 0), _dp(out)))
        return out


class Forgy(CentroidsInitialiser):
    """Forgy initialisation algorithm."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(_l.mlpp_forgy_create(C.byref(self._h)))


class RandomPartition(CentroidsInitialiser):
    """Random Partition initialisation algorithm."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(_l.mlpp_random_partition_create(C.byref(self._h)))


class KPP(CentroidsInitialiser):
    """KMeans++ initialisation algorithm."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(_l.mlpp_kpp_create(C.byref(self._h)))


class FixedCentroids(CentroidsInitialiser):
    """Extension: returns the given centroids (K x d)."""

    def __init__(self, centroids):
        c = np.ascontiguousarray(centroids, dtype=np.float64)
        if c.ndim != 2:
            raise ValueError("centroids must be K x d")
        self._h = C.c_void_p()
        _check(_l.mlpp_fixed_centroids_create(_dp(c), c.shape[0], c.shape[1], C.byref(self._h)))


class ClosestCentroid(ResponsibilitiesInitialiser):
    """Assigns points to closest centroid."""

    def __init__(self, centroids_initialiser):
        if centroids_initialiser is not None and not isinstance(centroids_initialiser, CentroidsInitialiser):
            raise TypeError("centroids_initialiser must be a CentroidsInitialiser")
        self._ci = centroids_initialiser
        self._h = C.c_void_p()
        _check(_l.mlpp_closest_centroid_create(centroids_initialiser._h if centroids_initialiser is not None else None,
                                               C.byref(self._h)))


class EM:
    """Gaussian Expectation-Maximisation algorithm."""

    def __init__(self, number_components):
        self._h = C.c_void_p()
        _check(_l.mlpp_em_create(int(number_components), C.byref(self._h)))
        self._keep = []

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _l.mlpp_em_destroy(h)
            self._h = None

    def set_seed(self, seed):
        """Sets PRNG seed."""
        _check(_l.mlpp_em_set_seed(self._h, C.c_uint32(seed)))

    def set_absolute_tolerance(self, absolute_tolerance):
        """Sets absolute tolerance."""
        _check(_l.mlpp_em_set_absolute_tolerance(self._h, C.c_double(absolute_tolerance)))

    def set_relative_tolerance(self, relative_tolerance):
        """Sets relative tolerance."""
        _check(_l.mlpp_em_set_relative_tolerance(self._h, C.c_double(relative_tolerance)))

    def set_maximum_steps(self, maximum_steps):
        """Sets maximum number of iterations."""
        _check(_l.mlpp_em_set_maximum_steps(self._h, C.c_uint32(maximum_steps)))

    def set_means_initialiser(self, means_initialiser):
        """Sets the algorithm to initialise component means."""
        _check(_l.mlpp_em_set_means_initialiser(self._h, means_initialiser._h if means_initialiser is not None else None))
        self._keep.append(means_initialiser)

    def set_responsibilities_initialiser(self, responsibilities_initialiser):
        """Sets the algorithm to initialise responsibilities for data points."""
        _check(_l.mlpp_em_set_responsibilities_initialiser(
            self._h, responsibilities_initialiser._h if responsibilities_initialiser is not None else None))
        self._keep.append(responsibilities_initialiser)

    def set_verbose(self, verbose):
        """Turns on/off the verbose mode."""
        _check(_l.mlpp_em_set_verbose(self._h, int(bool(verbose))))

    def set_maximise_first(self, maximise_first):
        """Turns on/off doing an initial maximisation step before the E-M iterations."""
        _check(_l.mlpp_em_set_maximise_first(self._h, int(bool(maximise_first))))

    def set_covariance_type(self, covariance_type):
        """Extension (not in the reference surface): "full" (default, the reference's only mode) or "diag" -- every
        covariance restricted to its diagonal; `covariance(k)` then returns a diagonal matrix. One fused kernel per iteration for
        d <= 32, K <= 64; other shapes run the full-covariance kernels on diagonal matrices."""
        if covariance_type not in ("full", "diag"):
            raise ValueError("covariance_type must be 'full' or 'diag'")
        _check(_l.mlpp_em_set_covariance_type(self._h, int(covariance_type == "diag")))

    def fit(self, data):
        """Fits the components to the data (2D array with data points in rows). Returns True if EM converged."""
        data = _require_data(data)
        n, d = data.shape
        conv = C.c_int()
        _check(_l.mlpp_em_fit(self._h, _dp(data), C.c_uint64(n), d, C.byref(conv)))
        return bool(conv.value)

    def _dims(self):
        d, n = C.c_uint32(), C.c_uint64()
        _check(_l.mlpp_em_dims(self._h, C.byref(d), C.byref(n)))
        return d.value, n.value

    @property
    def number_components(self):
        """Number of Gaussian components."""
        k = C.c_uint32()
        _check(_l.mlpp_em_number_components(self._h, C.byref(k)))
        return k.value

    @property
    def means(self):
        """Fitted means (d x K)."""
        d, _ = self._dims()
        out = np.empty((d, self.number_components), order="F")
        _check(_l.mlpp_em_means(self._h, _dp(out)))
        return out

    @property
    def responsibilities(self):
        """Fitted responsibilities (N x K)."""
        _, n = self._dims()
        out = np.empty((n, self.number_components), order="F")
        _check(_l.mlpp_em_responsibilities(self._h, _dp(out)))
        return out

    def responsibilities_rows(self, first_row, number_rows):
        """Extension: rows [first_row, first_row + number_rows) of `responsibilities` without fetching the whole N x K block
        from the device (5 GB at N=10M, K=64)."""
        out = np.empty((int(number_rows), self.number_components), order="F")
        _check(_l.mlpp_em_responsibilities_rows(self._h, C.c_uint64(int(first_row)), C.c_uint64(int(number_rows)), _dp(out)))
        return out

    @property
    def log_likelihood(self):
        """Maximised log-likelihood."""
        v = C.c_double()
        _check(_l.mlpp_em_log_likelihood(self._h, C.byref(v)))
        return v.value

    @property
    def mixing_probabilities(self):
        """Mixing probabilities of components."""
        out = np.empty(self.number_components)
        _check(_l.mlpp_em_mixing_probabilities(self._h, _dp(out)))
        return out

    def covariance(self, k):
        """Returns k-th covariance matrix."""
        d, _ = self._dims()
        out = np.empty((d, d), order="F")
        _check(_l.mlpp_em_covariance(self._h, int(k), _dp(out)))
        return out

    def assign_responsibilities(self, x):
        """Given a data point x, calculate each component's responsibilities for x and return them."""
        x = _vector(x)
        u = np.empty(self.number_components)
        _check(_l.mlpp_em_assign_responsibilities(self._h, _dp(x), x.size, _dp(u), u.size))
        return u

    # ---- extensions ----
    @property
    def labels(self):
        _, n = self._dims()
        out = np.empty(n, dtype=np.uint32)
        _check(_l.mlpp_em_labels(self._h, _lib.u32ptr(out)))
        return out

    @property
    def converged(self):
        v = C.c_int()
        _check(_l.mlpp_em_converged(self._h, C.byref(v)))
        return bool(v.value)

    @property
    def steps_done(self):
        v = C.c_uint32()
        _check(_l.mlpp_em_steps_done(self._h, C.byref(v)))
        return v.value


class KMeans:
    """Naive K-Means algorithm."""

    def __init__(self, number_clusters):
        self._h = C.c_void_p()
        _check(_l.mlpp_kmeans_create(int(number_clusters), C.byref(self._h)))
        self._keep = []

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            _l.mlpp_kmeans_destroy(h)
            self._h = None

    def set_seed(self, seed):
        """Sets the PRNG seed."""
        _check(_l.mlpp_kmeans_set_seed(self._h, C.c_uint32(seed)))

    def set_absolute_tolerance(self, absolute_tolerance):
        """Sets absolute tolerance."""
        _check(_l.mlpp_kmeans_set_absolute_tolerance(self._h, C.c_double(absolute_tolerance)))

    def set_maximum_steps(self, maximum_steps):
        """Sets maximum number of iterations."""
        _check(_l.mlpp_kmeans_set_maximum_steps(self._h, C.c_uint32(maximum_steps)))

    def set_centroids_initialiser(self, centroids_initialiser):
        """Sets the algorithm to initialise cluster centroids."""
        _check(_l.mlpp_kmeans_set_centroids_initialiser(
            self._h, centroids_initialiser._h if centroids_initialiser is not None else None))
        self._keep.append(centroids_initialiser)

    def set_number_initialisations(self, centroids_initialiser):
        """Sets number of initialisations to try, to find the clusters with lowest inertia.
        (The keyword really is `centroids_initialiser` in the reference: cppyml/clustering.cpp:159.)"""
        _check(_l.mlpp_kmeans_set_number_initialisations(self._h, C.c_uint32(centroids_initialiser)))

    def set_verbose(self, verbose):
        """Turns on/off the verbose mode."""
        _check(_l.mlpp_kmeans_set_verbose(self._h, int(bool(verbose))))

    def fit(self, data):
        """Fits the clusters to the data (2D array with data points in rows). Returns True if the algorithm converged."""
        data = _require_data(data)
        n, d = data.shape
        conv = C.c_int()
        _check(_l.mlpp_kmeans_fit(self._h, _dp(data), C.c_uint64(n), d, C.byref(conv)))
        return bool(conv.value)

    def _dims(self):
        d, n = C.c_uint32(), C.c_uint64()
        _check(_l.mlpp_kmeans_dims(self._h, C.byref(d), C.byref(n)))
        return d.value, n.value

    @property
    def number_clusters(self):
        """Number of clusters."""
        k = C.c_uint32()
        _check(_l.mlpp_kmeans_number_clusters(self._h, C.byref(k)))
        return k.value

    @property
    def centroids(self):
        """Fitted centroids (K x d)."""
        d, _ = self._dims()
        out = np.empty((self.number_clusters, d))
        _check(_l.mlpp_kmeans_centroids(self._h, _dp(out)))
        return out

    @property
    def labels(self):
        """Fitted labels."""
        _, n = self._dims()
        out = np.empty(n, dtype=np.uint32)
        _check(_l.mlpp_kmeans_labels(self._h, _lib.u32ptr(out)))
        return out.tolist()

    @property
    def labels_array(self):
        """Extension: the fitted labels as a numpy uint32 array (the reference surface converts to a Python list, which
        costs seconds and gigabytes at N = 1e8)."""
        _, n = self._dims()
        out = np.empty(n, dtype=np.uint32)
        _check(_l.mlpp_kmeans_labels(self._h, _lib.u32ptr(out)))
        return out

    @property
    def inertia(self):
        """Minimised inertia."""
        v = C.c_double()
        _check(_l.mlpp_kmeans_inertia(self._h, C.byref(v)))
        return v.value

    def assign_label(self, x):
        """Given a data point x, assigns it to the closest cluster: (label, squared distance)."""
        x = _vector(x)
        label, dist = C.c_uint32(), C.c_double()
        _check(_l.mlpp_kmeans_assign_label(self._h, _dp(x), x.size, C.byref(label), C.byref(dist)))
        return label.value, dist.value

    # ---- extensions ----
    @property
    def converged(self):
        v = C.c_int()
        _check(_l.mlpp_kmeans_converged(self._h, C.byref(v)))
        return bool(v.value)

    @property
    def steps_done(self):
        v = C.c_uint32()
        _check(_l.mlpp_kmeans_steps_done(self._h, C.byref(v)))
        return v.value
