"""ctypes binding of the C ABI (include/mlhip.h). Loads ml_amd/libmlhip.so; never falls back to anything else."""
import ctypes as C
import weakref
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MLHIP_LIBRARY") or os.path.join(_HERE, "libmlhip.so")   # override: sanitizer / debug builds

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C ml_amd/csrc`). ml_amd has no CPU fallback.")



def _preload_shared_hip_runtime():
    """A process must run ONE HIP runtime. PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 with the
    same sonames as /opt/rocm's, and the dynamic loader keeps whichever copy is loaded first: if this library pulled in
    the system copy first, a later `import torch` would bind to a runtime it was not built against and report no GPU.
    So when a torch installation is present (it is only ever used for torch.distributed collectives), load its
    runtime libraries first -- without importing torch. MLHIP_HIP_RUNTIME=system opts out."""
    if os.environ.get("MLHIP_HIP_RUNTIME", "").lower() == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        for name in ("libhsa-runtime64.so", "libamdhip64.so"):
            path = os.path.join(libdir, name)
            if os.path.exists(path):
                C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        pass   # fall back to the system runtime named in libmlhip.so's DT_NEEDED


_preload_shared_hip_runtime()
lib = C.CDLL(LIB_PATH)

OK, E_INVALID_ARGUMENT, E_DOMAIN, E_RUNTIME, E_NO_DEVICE, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5

c_dp = C.POINTER(C.c_double)
c_u32p = C.POINTER(C.c_uint32)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

lib.mlhip_last_error.restype = C.c_char_p
lib.mlhip_version.restype = C.c_char_p


class MlhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class NoDeviceError(MlhipError):
    pass


def check(rc):
    if rc == OK:
        return
    msg = lib.mlhip_last_error().decode()
    if rc in (E_INVALID_ARGUMENT, E_DOMAIN):
        raise ValueError(msg)          # pybind11 maps invalid_argument / domain_error to ValueError
    if rc == E_NO_DEVICE:
        raise NoDeviceError(rc, msg)
    raise MlhipError(rc, msg)


def dptr(a):
    return a.ctypes.data_as(c_dp)


def u32ptr(a):
    return a.ctypes.data_as(c_u32p)


RCCL_UNIQUE_ID_BYTES = 128


def rccl_available():
    """Whether librccl can be loaded in this process (no GPU touched): agree on it across ranks BEFORE Context.init_rccl."""
    return bool(lib.mlhip_rccl_available())


def rccl_unique_id():
    """128 opaque bytes from ncclGetUniqueId (rank 0 calls this and hands them to every rank: Context.init_rccl)."""
    buf = C.create_string_buffer(RCCL_UNIQUE_ID_BYTES)
    check(lib.mlhip_rccl_unique_id(buf))
    return buf.raw


def device_count():
    n = C.c_int()
    check(lib.mlhip_device_count(C.byref(n)))
    return n.value


class Context:
    """One GPU + stream (mlhip_ctx) -- or, made by Context.group(...), a device GROUP: one context over several shards (GPUs), to
    which Data uploads row-shard the caller's block and every call fans out (mlhip_ctx_create_group)."""

    def __init__(self, device_id=-1):
        self._h = C.c_void_p()
        check(lib.mlhip_ctx_create(int(device_id), C.byref(self._h)))
        self._hook = None
        self._owned = True
        self._blocks = weakref.WeakSet()      # live Data objects: they hold a pointer to this context

    @classmethod
    def group(cls, n_shards=None, device_ids=None):
        """A device group of n_shards shards, shard s on GPU device_ids[s] (default: s mod the number of GPUs; ids may repeat --
        8 shards on one GPU rehearse the 8-GPU configurations on a one-GPU box)."""
        if device_ids is not None:
            ids = [int(v) for v in device_ids]
            n_shards = len(ids) if n_shards is None else int(n_shards)
            if n_shards != len(ids):
                raise ValueError("device_ids must name one GPU per shard")
            arr = (C.c_int * n_shards)(*ids)
        else:
            n_shards = device_count() if n_shards is None else int(n_shards)
            arr = None
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self._hook = None
        self._owned = True
        self._blocks = weakref.WeakSet()
        check(lib.mlhip_ctx_create_group(n_shards, arr, C.byref(self._h)))
        return self

    @property
    def shards(self):
        n = C.c_int()
        check(lib.mlhip_ctx_shards(self._h, C.byref(n)))
        return n.value

    @property
    def shard_devices(self):
        out = []
        for s in range(self.shards):
            d = C.c_int()
            check(lib.mlhip_ctx_shard_device(self._h, s, C.byref(d)))
            out.append(d.value)
        return out

    @property
    def reduce_kind(self):
        """How the statistics are summed: none / hook-host / hook-device / rccl / group-rccl / group-direct."""
        k = C.c_char_p()
        check(lib.mlhip_ctx_reduce_kind(self._h, C.byref(k)))
        return k.value.decode()

    @classmethod
    def borrow(cls, handle):
        """Wraps a context owned by someone else (the C++ facade's process-wide one): close() leaves it alive."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        self._hook = None
        self._owned = False
        self._blocks = weakref.WeakSet()
        return self

    def close(self):
        if getattr(self, "_h", None):
            for block in list(self._blocks):  # a sample block must not outlive its context
                block.close()
            if self._owned:
                lib.mlhip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._h

    @property
    def device(self):
        d = C.c_int()
        check(lib.mlhip_ctx_device(self._h, C.byref(d)))
        return d.value

    @property
    def stream(self):
        s = C.c_void_p()
        check(lib.mlhip_ctx_stream(self._h, C.byref(s)))
        return s.value or 0

    def synchronize(self):
        check(lib.mlhip_ctx_synchronize(self._h))

    def set_allreduce(self, fn, on_device, world_size, rank):
        """fn(ptr:int, count:int, on_device:bool, stream:int) -> None sums the buffer in place across ranks."""
        if fn is None:
            self._hook = None
            check(lib.mlhip_ctx_set_allreduce(self._h, C.cast(None, ALLREDUCE_FN), None, 0, 1, 0))
            return

        def trampoline(_user, buf, count, on_dev, stream):
            try:
                fn(buf or 0, count, bool(on_dev), stream or 0)
                return 0
            except Exception:  # surfaced as MLHIP_E_RUNTIME by the library
                import traceback
                traceback.print_exc()
                return 1

        self._hook = ALLREDUCE_FN(trampoline)
        check(lib.mlhip_ctx_set_allreduce(self._h, self._hook, None, int(on_device), int(world_size), int(rank)))

    def init_rccl(self, unique_id, world_size, rank):
        """The library's own RCCL communicator (collective call): statistics are summed by ncclAllReduce on the context's
        stream, no Python in the iteration."""
        if len(unique_id) != RCCL_UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be the 128 bytes of rccl_unique_id()")
        self._hook = None
        check(lib.mlhip_ctx_init_rccl(self._h, C.c_char_p(bytes(unique_id)), int(world_size), int(rank)))

    def init_rccl_file(self, path, world_size, rank):
        check(lib.mlhip_ctx_init_rccl_file(self._h, os.fsencode(path), int(world_size), int(rank)))

    @property
    def rccl_ranks(self):
        """Ranks of the library-owned communicator as RCCL counts them (0: none)."""
        n = C.c_int()
        check(lib.mlhip_ctx_rccl_ranks(self._h, C.byref(n)))
        return n.value

    def finalize_rccl(self):
        check(lib.mlhip_ctx_finalize_rccl(self._h))

    @property
    def world(self):
        w, r = C.c_int(), C.c_int()
        check(lib.mlhip_ctx_world(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def timing_enable(self, on=True):
        check(lib.mlhip_timing_enable(self._h, int(on)))

    def timing_reset(self):
        check(lib.mlhip_timing_reset(self._h))

    def timing_get(self, name):
        ms, cnt = C.c_double(), C.c_uint64()
        check(lib.mlhip_timing_get(self._h, name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value


class Data:
    """A d x N sample block resident in HBM (mlhip_data). `x` is N x d float64 C-contiguous."""

    def __init__(self, ctx, x=None, device_ptr=None, shape=None):
        self.ctx = ctx
        self._h = C.c_void_p()
        if x is not None:
            if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.ndim == 2 and x.flags.c_contiguous):
                raise TypeError("data must be a C-contiguous float64 N x d numpy array")
            n, d = x.shape
            check(lib.mlhip_data_upload(ctx.handle, dptr(x), d, C.c_uint64(n), C.c_int64(d), C.byref(self._h)))
        else:
            n, d = shape
            check(lib.mlhip_data_upload_dev(ctx.handle, C.cast(C.c_void_p(device_ptr), c_dp), d, C.c_uint64(n),
                                            C.c_int64(d), C.byref(self._h)))
        self.n, self.d = n, d
        ctx._blocks.add(self)

    def close(self):
        if getattr(self, "_h", None):
            lib.mlhip_data_free(self._h)
            self._h = None
            self.ctx._blocks.discard(self)

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._h

    @property
    def n_global(self):
        ng = C.c_uint64()
        check(lib.mlhip_data_shape(self._h, None, None, C.byref(ng)))
        return ng.value

    def shard_rows(self, shard):
        """(first_row, n_rows) of the rows shard `shard` of a device group holds."""
        lo, cnt = C.c_uint64(), C.c_uint64()
        check(lib.mlhip_data_shard_rows(self._h, int(shard), C.byref(lo), C.byref(cnt)))
        return lo.value, cnt.value

    @property
    def shift(self):
        out = np.empty(self.d)
        check(lib.mlhip_data_shift(self._h, dptr(out)))
        return out

    # ---- EM -----------------------------------------------------------------------------------------------
    # Parameter conventions (Python side): means K x d, covariances K x d x d, mixing K.
    def em_step(self, mixing, means, covs):
        K = len(mixing)
        mixing = np.ascontiguousarray(mixing, dtype=np.float64)
        means = np.ascontiguousarray(means, dtype=np.float64)
        covs = np.ascontiguousarray(covs, dtype=np.float64)
        ll = C.c_double()
        pi1, mu1, S1 = np.empty(K), np.empty((K, self.d)), np.empty((K, self.d, self.d))
        check(lib.mlhip_em_step(self.ctx.handle, self._h, K, dptr(mixing), dptr(means), dptr(covs), C.byref(ll),
                                dptr(pi1), dptr(mu1), dptr(S1)))
        return ll.value, pi1, mu1, S1

    def em_step_diag(self, mixing, means, variances):
        """Diagonal-covariance extension. variances: K x d (a K x d x d stack of diagonal matrices is accepted too and
        returned in the same form)."""
        K = len(mixing)
        mixing = np.ascontiguousarray(mixing, dtype=np.float64)
        means = np.ascontiguousarray(means, dtype=np.float64)
        v = np.asarray(variances, dtype=np.float64)
        stacked = v.ndim == 3
        if stacked:
            v = np.stack([np.diag(m) for m in v])
        v = np.ascontiguousarray(v)
        ll = C.c_double()
        pi1, mu1, v1 = np.empty(K), np.empty((K, self.d)), np.empty((K, self.d))
        check(lib.mlhip_em_step_diag(self.ctx.handle, self._h, K, dptr(mixing), dptr(means), dptr(v), C.byref(ll),
                                     dptr(pi1), dptr(mu1), dptr(v1)))
        if stacked:
            v1 = np.stack([np.diag(row) for row in v1])
        return ll.value, pi1, mu1, v1

    def em_iterate(self, mixing, means, covs, max_steps, atol=0.0, rtol=0.0, diagonal=False):
        """The EM loop in one call (mlhip_em_iterate): up to max_steps iterations with the reference's convergence test, the
        closing arithmetic on the device. covs: K x d x d (or K x d variances with diagonal=True).
        Returns (steps_done, converged, log_likelihood, mixing, means, covs, log_likelihood_history)."""
        K = len(mixing)
        pi = np.array(mixing, dtype=np.float64, order="C")
        mu = np.array(means, dtype=np.float64, order="C")
        S = np.array(covs, dtype=np.float64, order="C")
        assert mu.shape == (K, self.d) and S.shape == ((K, self.d) if diagonal else (K, self.d, self.d))
        steps, conv, ll = C.c_uint32(), C.c_int(), C.c_double()
        hist = np.full(int(max_steps), np.nan)
        check(lib.mlhip_em_iterate(self.ctx.handle, self._h, K, int(bool(diagonal)), dptr(pi), dptr(mu), dptr(S), C.c_uint32(max_steps),
                                   C.c_double(atol), C.c_double(rtol), C.byref(steps), C.byref(conv), C.byref(ll), dptr(hist)))
        return steps.value, bool(conv.value), ll.value, pi, mu, S, hist[:steps.value]

    def em_plan(self, K):
        """Which kernels a full-covariance EM iteration of K components is made of (mlhip_em_plan):
        {'fused', 'matrix_estep', 'self_norm'} -> bool."""
        f = C.c_uint32()
        check(lib.mlhip_em_plan(self._h, C.c_uint32(K), C.byref(f)))
        return {"fused": bool(f.value & 1), "matrix_estep": bool(f.value & 2), "self_norm": bool(f.value & 4)}

    def em_expectation(self, mixing, means, covs):
        K = len(mixing)
        mixing = np.ascontiguousarray(mixing, dtype=np.float64)
        means = np.ascontiguousarray(means, dtype=np.float64)
        covs = np.ascontiguousarray(covs, dtype=np.float64)
        ll = C.c_double()
        check(lib.mlhip_em_expectation(self.ctx.handle, self._h, K, dptr(mixing), dptr(means), dptr(covs), C.byref(ll)))
        return ll.value

    def em_maximisation(self, K):
        pi1, mu1, S1 = np.empty(K), np.empty((K, self.d)), np.empty((K, self.d, self.d))
        check(lib.mlhip_em_maximisation(self.ctx.handle, self._h, K, dptr(pi1), dptr(mu1), dptr(S1)))
        return pi1, mu1, S1

    def em_maximisation_from(self, resp):
        resp = np.asfortranarray(resp, dtype=np.float64)
        n, K = resp.shape
        pi1, mu1, S1 = np.empty(K), np.empty((K, self.d)), np.empty((K, self.d, self.d))
        check(lib.mlhip_em_maximisation_from(self.ctx.handle, self._h, K, dptr(resp), C.c_int64(n), dptr(pi1), dptr(mu1), dptr(S1)))
        return pi1, mu1, S1

    def em_maximisation_from_labels(self, labels, K):
        labels = np.ascontiguousarray(labels, dtype=np.uint32)
        pi1, mu1, S1 = np.empty(K), np.empty((K, self.d)), np.empty((K, self.d, self.d))
        check(lib.mlhip_em_maximisation_from_labels(self.ctx.handle, self._h, K, u32ptr(labels), dptr(pi1), dptr(mu1), dptr(S1)))
        return pi1, mu1, S1

    def em_responsibilities(self, K):
        out = np.empty((self.n, K), order="F")
        check(lib.mlhip_em_responsibilities(self.ctx.handle, self._h, K, dptr(out), C.c_int64(self.n)))
        return out

    def em_responsibilities_rows(self, K, first, count):
        """Rows [first, first + count) of the responsibilities only (mlhip_em_responsibilities_rows)."""
        out = np.empty((count, K), order="F")
        check(lib.mlhip_em_responsibilities_rows(self.ctx.handle, self._h, K, C.c_uint64(first), C.c_uint64(count), dptr(out),
                                                 C.c_int64(max(count, 1))))
        return out

    def em_labels(self, K):
        out = np.empty(self.n, dtype=np.uint32)
        check(lib.mlhip_em_labels(self.ctx.handle, self._h, K, u32ptr(out)))
        return out

    def sample_covariance(self):
        mean, cov = np.empty(self.d), np.empty((self.d, self.d))
        check(lib.mlhip_sample_covariance(self.ctx.handle, self._h, dptr(mean), dptr(cov)))
        return mean, cov

    # ---- K-means --------------------------------------------------------------------------------------------
    def kmeans_step(self, centroids):
        centroids = np.ascontiguousarray(centroids, dtype=np.float64)
        K = centroids.shape[0]
        inertia, changed = C.c_double(), C.c_uint64()
        counts, cout = np.empty(K), np.empty((K, self.d))
        check(lib.mlhip_kmeans_step(self.ctx.handle, self._h, K, dptr(centroids), C.byref(inertia), C.byref(changed),
                                    dptr(counts), dptr(cout)))
        return inertia.value, changed.value, counts, cout

    def kmeans_iterate(self, centroids, max_steps, atol=0.0):
        """The step loop of KMeans::fit_once in one call (mlhip_kmeans_iterate): the centroid table stays on the device
        between steps. Returns (steps_done, converged, inertia, counts, centroids, old_centroids)."""
        cur = np.array(centroids, dtype=np.float64, order="C")
        K = cur.shape[0]
        assert cur.shape == (K, self.d)
        old, counts = np.zeros((K, self.d)), np.zeros(K)
        steps, conv, inertia = C.c_uint32(), C.c_int(), C.c_double()
        check(lib.mlhip_kmeans_iterate(self.ctx.handle, self._h, K, dptr(cur), dptr(old), C.c_uint32(max_steps), C.c_double(atol),
                                       C.byref(steps), C.byref(conv), C.byref(inertia), dptr(counts)))
        return steps.value, bool(conv.value), inertia.value, counts, cur, old

    def kmeans_assign(self, centroids):
        centroids = np.ascontiguousarray(centroids, dtype=np.float64)
        K = centroids.shape[0]
        inertia, changed = C.c_double(), C.c_uint64()
        check(lib.mlhip_kmeans_assign(self.ctx.handle, self._h, K, dptr(centroids), C.byref(inertia), C.byref(changed)))
        return inertia.value, changed.value

    def kmeans_labels(self):
        out = np.empty(self.n, dtype=np.uint32)
        check(lib.mlhip_kmeans_labels(self.ctx.handle, self._h, u32ptr(out)))
        return out

    def kmeans_distances(self):
        """Per-sample squared distance to the assigned centroid from the last assignment."""
        out = np.empty(self.n)
        check(lib.mlhip_kmeans_distances(self.ctx.handle, self._h, dptr(out)))
        return out

    def min_squared_distances(self, centroids):
        centroids = np.ascontiguousarray(centroids, dtype=np.float64)
        out = np.empty(self.n)
        check(lib.mlhip_min_squared_distances(self.ctx.handle, self._h, centroids.shape[0], dptr(centroids), dptr(out)))
        return out


def calculate_XXt_beta(X, y, lam):
    """ml::LinearRegression::calculate_XXt_beta on an N x q float64 C-contiguous X: returns (XXt + diag(lam), beta)."""
    if not (isinstance(X, np.ndarray) and X.dtype == np.float64 and X.ndim == 2 and X.flags.c_contiguous):
        raise TypeError("X must be a C-contiguous float64 N x q numpy array")
    y = np.ascontiguousarray(y, dtype=np.float64)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    n, q = X.shape
    XXt, beta = np.empty((q, q)), np.empty(q)
    check(lib.mlpp_calculate_XXt_beta(dptr(X), C.c_uint64(n), q, dptr(y), C.c_uint64(y.size), dptr(lam), lam.size,
                                      dptr(XXt), dptr(beta)))
    return XXt, beta


def process_covariance(cov):
    cov = np.ascontiguousarray(cov, dtype=np.float64)
    d = cov.shape[0]
    inv, sd = np.empty((d, d)), C.c_double()
    check(lib.mlhip_process_covariance(d, dptr(cov), dptr(inv), C.byref(sd)))
    return inv, sd.value
