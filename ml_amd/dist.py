"""Row sharding of the N samples over ranks + the statistics all-reduce through torch.distributed
(backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for tests). One process per GPU.

The library calls the hook once per EM / K-means iteration with the fused statistics buffer
([K*F statistics, log-likelihood sum] -- 67 649 doubles at d=32, K=64): a latency-bound message, so a single
un-bucketed all-reduce is the right shape for point-to-point xGMI."""
import ctypes

import numpy as np


def shard_bounds(n, world_size, rank):
    """Contiguous, balanced row shard [lo, hi) of rank `rank`; the shards tile [0, n) exactly."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


class _DeviceBuffer:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


_tensor_cache = {}   # (ptr, count) -> zero-copy tensor view of a library-owned device buffer (stable across iterations)
_stream_cache = {}


def allreduce_sum(ptr, count, on_device, stream, group=None):
    """In-place sum over ranks of `count` doubles at `ptr` (device or host memory)."""
    import torch
    import torch.distributed as dist
    if on_device:
        ext = _stream_cache.get(stream)
        if ext is None:
            ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
            _stream_cache[stream] = ext
        t = _tensor_cache.get((ptr, count))
        with torch.cuda.stream(ext):
            if t is None:
                t = torch.as_tensor(_DeviceBuffer(ptr, count), device="cuda")
                if len(_tensor_cache) > 64:
                    _tensor_cache.clear()
                _tensor_cache[(ptr, count)] = t
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        ext.synchronize()
    else:
        buf = (ctypes.c_double * count).from_address(ptr)
        t = torch.from_numpy(np.frombuffer(buf, dtype=np.float64, count=count))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def install_allreduce(ctx, world_size, rank, on_device=None, group=None):
    """Registers the torch.distributed all-reduce as the context's statistics hook."""
    import torch.distributed as dist
    if on_device is None:
        on_device = dist.get_backend(group) == "nccl"
    ctx.set_allreduce(lambda ptr, count, dev, stream: allreduce_sum(ptr, count, dev, stream, group),
                      on_device, world_size, rank)


def install_native_rccl(ctx, world_size, rank, group=None):
    """The library's own RCCL communicator as the statistics all-reduce (mlhip_ctx_init_rccl): rank 0 draws the unique id,
    torch.distributed (any backend) only carries its 128 bytes to the other ranks; the iterations never enter Python."""
    import torch.distributed as dist
    from . import _lib
    # every rank must be able to load librccl, or the ranks that can would wait forever inside ncclCommInitRank for the one
    # that raised before it: agree on it first and fail (so that the caller can fall back) on ALL ranks together
    ok = _lib.rccl_available()
    if world_size > 1:
        flags = [None] * world_size
        dist.all_gather_object(flags, bool(ok), group=group)
        missing = [r for r, f in enumerate(flags) if not f]
    else:
        missing = [] if ok else [0]
    if missing:
        raise RuntimeError(f"RCCL is not available on rank(s) {missing}: librccl.so.1 could not be loaded (MLHIP_RCCL_LIBRARY)")
    box = [None]
    if rank == 0:
        try:
            box[0] = _lib.rccl_unique_id()
        except Exception as e:      # e.g. librccl cannot be loaded: every rank must learn it, or the others wait forever
            box[0] = ("error", str(e))
    if world_size > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    if isinstance(box[0], tuple):
        raise RuntimeError("rank 0 could not create an RCCL unique id: " + box[0][1])
    ctx.init_rccl(box[0], world_size, rank)
