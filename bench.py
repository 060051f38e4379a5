#!/usr/bin/env python3
"""Headline benchmark: Gaussian-mixture EM iterations/sec at N=10M, d=32, K=64, fp64 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

N > 1 either arrives already launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment, one rank per GPU) or -- as a bare command -- starts its own N rank processes: the parent does so BEFORE
anything touches a GPU (it imports neither torch nor the HIP library), waits for them and exits with their status;
rank 0's JSON line is the only thing on stdout either way.

One step = one EM iteration = E-step + M-step statistics + statistics all-reduce + M-step closing arithmetic
(K Cholesky/inverse) + convergence bookkeeping, i.e. one trip of the reference loop ML/EM.cpp:143-170, executed
through the C ABI entry point mlhip_em_step on data already resident in HBM. The N samples are row-sharded
over the ranks (strong scaling: the job size is fixed), the only exchange is one all-reduce of the K*561+1
sufficient statistics per iteration: ncclAllReduce on the library's own RCCL communicator (mlhip_ctx_init_rccl;
`--allreduce torch` routes it through torch.distributed's nccl backend instead). torch.distributed carries the
RCCL unique id, the barriers around the timed region and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0. At N=1 the line also carries `secondary`, a list: K-means (BASELINE.json configs[4]) at one
GPU's share of that job, N=12.5M, d=8, K=256, the diagonal-covariance GMM of configs[1] (N=1M, d=16, K=16) and the reference's own
benchmark case configs[0] (bm_EM.cpp: N=10k, d=4, K=3 -- latency-bound: three dependent launches per iteration), each with its
own roofline and cpu_baseline. Every line carries `allreduce_ms` (average device time of the statistics all-reduce, max over
ranks; 0 on one GPU) and the spread of the ranks' own time per step (`ms_per_step_min` / `_max`).

    python bench.py --gpus N --single-process              the same line from ONE process: the library's device group
        (mlhip_ctx_create_group: the caller's one N x d block row-sharded inside the library, one host thread per shard, RCCL
        between distinct GPUs, an in-process fixed-order sum when shards share a GPU) -- what ml::EM::fit / cppyml.clustering use
        when MLHIP_NUM_GPUS is set; the line then carries `processes: 1` and `device_group`.
    python bench.py --workload kmeans [--gpus N ...]       K-means steps/sec at N=100M, d=8, K=256 as the primary line;
        one step = mlhip_kmeans_step = assignment + exact update sums + all-reduce of counts/sums + new centroids
        (ML/KMeans.cpp:82-108).
    python bench.py --workload em-diag [...]               diagonal-covariance EM at N=1M, d=16, K=16 (BASELINE.json
        configs[1]; an extension: the reference has full covariances only)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TOTAL, DIM, COMPONENTS = 10_000_000, 32, 64
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector == matrix peak (spec); measured ceilings: tools/microbench_fp64
HBM_PEAK_GBS = 8000.0
CPU_ITERATIONS = 3           # the CPU baseline is timed over T = 3 iterations


def algorithmic_flops(n, d, K):
    """N*K*(2d^2 + 6d + 25) per EM iteration (full covariance)."""
    return float(n) * K * (2 * d * d + 6 * d + 25)


def estep_flops(n, d, K, self_norm=False):
    """E-step share of the algorithmic flops. The ~25 flops of the exponential / normalisation of a pair run where the
    exponential runs: in the E-step kernel, or -- self-normalising statistics kernel (mlhip_em_plan) -- in the statistics kernel."""
    return float(n) * K * (d * d + 3 * d + (0 if self_norm else 25))


def mstats_flops(n, d, K, self_norm=False):
    return float(n) * K * (d * d + 3 * d + (25 if self_norm else 0))


def kmeans_uses_matrix_cores(d, K, n_local=1 << 30):
    """device/kmeans.hip serves d = 1, 2, 3, 5, 6 with K < 128, few clusters (K <= 16 at d <= 32, K <= 24 at d <= 8, from 2^21 rows on) and d > 128 on the vector unit
    (direct form, 3d flops per pair); everything else runs kmeans_mfma.hip, which EXECUTES 2d flops per pair (scores
    x.c - |c|^2/2) plus the exact recheck."""
    return not ((d in (1, 2, 3, 5, 6) and K < 128) or (K <= (24 if d <= 8 else 16) and d <= 32 and n_local >= (1 << 21)) or d > 128)


def diag_flops(n, d, K):
    """SURVEY.md section 8(d), diagonal covariances: N*K*(8d + 25) per EM iteration."""
    return float(n) * K * (8 * d + 25)


def cpu_baseline(mix, d, K, n_cpu, iters, diagonal=False):
    """The CPU restatement of the reference (oracle/, single thread like the reference) on a bounded sample."""
    import numpy as np
    from oracle import oracle_ctypes as orc
    X, _ = mix.sample(n_cpu, stream=999)
    em = orc.EM(K)
    if diagonal:
        em.set_covariance_type("diag")
        cov0 = np.stack([np.diag(np.var(X, axis=0, ddof=1))] * K)
    else:
        cov0 = np.stack([np.cov(X.T)] * K)
    em.set_parameters(mix.initial_means(), cov0, np.full(K, 1.0 / K))
    sec_per_iter = em.time_iterations(X, iters)
    return sec_per_iter, n_cpu


# ---- the reference's OWN benchmark drivers (Benchmarks/bm_EM.cpp:9-48, bm_KMeans.cpp:9-48), timed as the reference times them:
# the 'mousie' sample (d = 2, K = 3), K-means++ start, tolerances 1e-14, the WHOLE fit incl. upload and initialisation. The CPU legs
# (and the sample generator, a restatement of the benchmark's own <random> calls) are the oracle's: they live here, in the
# cpu_baseline part of this file; tools/bm_clustering.py imports them from here.

def mousie(n):
    """The benchmark's sample (bm_EM.cpp:11-35): N x 2, same libstdc++ draws as the reference's generator."""
    from oracle import oracle_ctypes as orc
    return orc.testdata_mousie(int(n))[0]


def bm_em_fit(X):
    """ml::EM as bm_EM.cpp:38-44 sets it up, through the drop-in's Python surface: (seconds, steps, converged, log-likelihood)."""
    from ml_amd.cppyml import clustering as cl
    em = cl.EM(3)
    em.set_absolute_tolerance(1e-14)
    em.set_relative_tolerance(1e-14)
    em.set_means_initialiser(cl.KPP())
    em.set_maximise_first(False)
    t0 = time.perf_counter()
    conv = em.fit(X)
    return time.perf_counter() - t0, em.steps_done, bool(conv), em.log_likelihood


def bm_kmeans_fit(X):
    """ml::Clustering::KMeans as bm_KMeans.cpp:38-44 sets it up: (seconds, converged, inertia)."""
    from ml_amd.cppyml import clustering as cl
    km = cl.KMeans(3)
    km.set_absolute_tolerance(1e-14)
    km.set_centroids_initialiser(cl.KPP())
    km.set_number_initialisations(3)
    t0 = time.perf_counter()
    conv = km.fit(X)
    return time.perf_counter() - t0, bool(conv), km.inertia


def cpu_bm_em_fit(X):
    """The same fit by the single-threaded CPU restatement (oracle/): (seconds, steps, converged, log-likelihood)."""
    from oracle import oracle_ctypes as orc
    em = orc.EM(3)
    em.set_absolute_tolerance(1e-14)
    em.set_relative_tolerance(1e-14)
    em.set_means_initialiser(orc.KPP)
    em.set_maximise_first(False)
    t0 = time.perf_counter()
    conv = em.fit(X)
    return time.perf_counter() - t0, em.steps_done, bool(conv), em.log_likelihood


def cpu_bm_kmeans_fit(X):
    from oracle import oracle_ctypes as orc
    km = orc.KMeans(3)
    km.set_absolute_tolerance(1e-14)
    km.set_centroids_initialiser(orc.KPP)
    km.set_number_initialisations(3)
    t0 = time.perf_counter()
    conv = km.fit(X)
    return time.perf_counter() - t0, bool(conv), km.inertia


def bm_em_secondary(n, repeats):
    """`secondary` entry: whole-fit wall time of the reference's benchmark case at N = n (milliseconds, lower is better), the warm
    median of `repeats` fits after one untimed fit, next to the CPU restatement's time for the same fit."""
    import numpy as np
    X = mousie(n)
    first = bm_em_fit(X)                              # (untimed as `value`: kernels' code objects load, buffers are allocated)
    runs = [bm_em_fit(X) for _ in range(repeats)]
    sec = float(np.median([r[0] for r in runs]))
    cpu = cpu_bm_em_fit(X)
    return {
        "metric": f"EM.fit wall time, reference benchmark Benchmarks/bm_EM.cpp at N={n} (mousie d=2 K=3, K-means++ start, tolerances "
                  f"1e-14; upload + initialisation + all iterations + labels)",
        "value": sec * 1e3, "unit": "ms", "n_gpus": 1, "steps": repeats, "warmup": 1, "ms_per_step": sec * 1e3,
        "higher_is_better": False, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"bm_EM.cpp em_mousie/{n}: one whole ml::EM::fit through cppyml.clustering", "N": n, "d": 2, "K": 3,
                   "iterations_of_the_fit": runs[0][1], "converged": runs[0][2], "log_likelihood": runs[0][3],
                   "first_fit_of_the_process_ms": first[0] * 1e3, "us_per_iteration_incl_everything": sec * 1e6 / max(1, runs[0][1])},
        "cpu_baseline": {"value": cpu[0] * 1e3, "unit": "ms", "cores": 1, "kind": "port",
                         "sample": f"the same whole fit by the single-threaded CPU restatement (oracle/), not scaled: "
                                   f"{cpu[1]} iterations, converged {cpu[2]}, log-likelihood {cpu[3]!r}",
                         "iterations_of_the_fit": cpu[1]},
    }


def usable_cores():
    """Host cores this process can really use: the affinity mask, capped by the cgroup CPU quota of the container."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = f.read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:      # cgroup v1
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                cores = min(cores, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return cores


def cpu_baseline_all_cores(mix, d, K, n_per_thread, iters):
    """Row-parallel variant of the same restatement on every host core this process may use: one oracle instance per
    thread on its own sample shard (the E and M steps are sums over samples, so this is what an OpenMP `parallel for` over
    the reference's sample loops would do; the K x d x d combine is negligible). The reference itself is single-threaded."""
    import threading
    import numpy as np
    from oracle import oracle_ctypes as orc
    threads = min(usable_cores(), 32)         # bounded: every thread holds its own N_t x K responsibility block
    X, _ = mix.sample(n_per_thread * threads, stream=998)
    cov0 = np.stack([np.cov(X[:n_per_thread].T)] * K)
    models = []
    for _ in range(threads):
        em = orc.EM(K)
        em.set_parameters(mix.initial_means(), cov0, np.full(K, 1.0 / K))
        models.append(em)
    shards = [np.ascontiguousarray(X[t * n_per_thread:(t + 1) * n_per_thread]) for t in range(threads)]
    pool = [threading.Thread(target=models[t].time_iterations, args=(shards[t], iters)) for t in range(threads)]
    t0 = time.perf_counter()
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    return (time.perf_counter() - t0) / iters, n_per_thread * threads, threads


ROW_CHUNK = 1_250_000


def sample_rows(mix, lo, hi):
    """Rows [lo, hi) of the synthetic data set, which is defined chunk-wise (chunk c = ROW_CHUNK rows drawn from the
    generator's stream c): every sharding of the rows over ranks sees the same data, so the final log-likelihood of the
    1-, 2-, 4- and 8-GPU runs can be compared directly."""
    import numpy as np
    out = np.empty((hi - lo, mix.d))
    c = lo // ROW_CHUNK
    while c * ROW_CHUNK < hi:
        a, b = c * ROW_CHUNK, (c + 1) * ROW_CHUNK
        block = mix.sample(ROW_CHUNK, stream=c)[0]
        s0, s1 = max(lo, a), min(hi, b)
        out[s0 - lo:s1 - lo] = block[s0 - a:s1 - a]
        c += 1
    return out


class Job:
    """What one rank needs around a workload: its GPU context with the statistics all-reduce installed, the barrier that
    brackets the timed region, and the reductions over ranks for the report."""

    def __init__(self, args, rank, local_rank, world, dist, torch):
        from ml_amd import _lib
        from ml_amd import dist as mldist
        self.args, self.rank, self.local_rank, self.world, self.dist, self.torch = args, rank, local_rank, world, dist, torch
        self.units = args.gpus                       # GPUs (ranks, or shards of a device group) the rows are spread over
        self.allreduce = "none"
        self.rccl_ranks = 1
        self.last_ranks = {}
        if args.single_process and args.gpus > 1:
            # ONE process, a device group (mlhip_ctx_create_group): shard s on GPU s mod the number of visible GPUs -- the shards
            # share a GPU when there are fewer GPUs than shards (a one-GPU rehearsal of the multi-GPU configuration)
            self.ctx = _lib.Context.group(args.gpus)
            kind = self.ctx.reduce_kind                 # "group-direct (RCCL unavailable: ...)" when an automatic choice fell back
            self.allreduce = ("rccl-native (ncclCommInitAll communicators of the library's device group)" if kind == "group-rccl"
                              else "in-process fixed-order sum over the shards' buffers (device group)" + kind[len("group-direct"):])
            self.rccl_ranks = self.ctx.rccl_ranks
            self.shard_devices = self.ctx.shard_devices
            return
        self.ctx = _lib.Context(local_rank)
        self.shard_devices = None
        if world > 1 or args.force_hook:
            if args.allreduce == "native":
                try:
                    mldist.install_native_rccl(self.ctx, world, rank)
                    self.allreduce = "rccl-native (ncclAllReduce on the library's own communicator)"
                    self.rccl_ranks = self.ctx.rccl_ranks
                except Exception as e:                       # stay measurable: fall back to torch's communicator
                    print(f"[bench] native RCCL unavailable on rank {rank} ({e}); using torch.distributed", file=sys.stderr)
            if self.allreduce == "none" and args.allreduce == "gloo":
                mldist.install_allreduce(self.ctx, world, rank, on_device=False)
                self.allreduce = "gloo on the host (single-GPU rehearsal of the multi-rank path)"
                self.rccl_ranks = 0
            elif self.allreduce == "none":
                mldist.install_allreduce(self.ctx, world, rank)
                self.allreduce = "torch.distributed nccl hook"
                self.rccl_ranks = dist.get_world_size()
            # every rank must have taken the same route, or the collectives would not match
            routes = self.gather(self.allreduce)
            if len(set(routes)) != 1:
                raise SystemExit(f"ranks disagree on the all-reduce route: {routes}")

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.ctx.synchronize()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cpu" if self.args.allreduce == "gloo" else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        if self.world == 1:
            return [obj]
        box = [None] * self.world
        self.dist.all_gather_object(box, obj)
        return box

    def timed(self, run, steps, warmup, names=(), live=True):
        """W untimed steps, then exactly K steps between barrier + synchronize on both sides; max over ranks.
        run(k) executes k steps. Returns (seconds, {kernel family: average ms per launch}).
        live: the kernels of the timed steps themselves carry HIP event pairs on the stream they are launched on; the
        pairs are only recorded inside the region and read after its closing barrier, so the region runs without any
        extra synchronisation (cost: a few microseconds per pair -- 0.1 % of an EM iteration, 0.5 % of a K-means step).
        not live (iterations of ~0.1 ms, where those microseconds would be 10 % of the metric): the region runs without
        events and the same K steps are replayed with them right afterwards."""
        if warmup:
            run(warmup)
        self.ctx.timing_reset()
        live = live and not self.args.replay_events
        self.ctx.timing_enable(bool(names) and live)
        self.barrier()
        t0 = time.perf_counter()
        run(steps)
        own = time.perf_counter() - t0               # this rank's own steps, before it waits for the others
        self.barrier()
        elapsed = self.max_over_ranks(time.perf_counter() - t0)
        if names and not live:
            self.ctx.timing_enable(True)
            run(steps)
        names = list(names) + ["allreduce"]
        kernel_ms = {name: self.ctx.timing_get(name)[0] for name in names}
        self.ctx.timing_enable(False)
        # diagnosability of a multi-GPU run: every rank's own time per step and its average all-reduce time (which, on a rank
        # that arrives early, includes the wait for the slowest one)
        per_rank = self.gather({"ms_per_step": own / steps * 1e3, "allreduce_ms": kernel_ms["allreduce"]})
        self.last_ranks = {"ms_per_step_min": min(r["ms_per_step"] for r in per_rank),
                           "ms_per_step_max": max(r["ms_per_step"] for r in per_rank),
                           "allreduce_ms": max(r["allreduce_ms"] for r in per_rank),
                           "allreduce_ms_per_rank": [r["allreduce_ms"] for r in per_rank]}
        return elapsed, kernel_ms

    def rows_per_unit(self, data, own_rows):
        """Rows on every GPU: the ranks' shard sizes, or the device group's."""
        if self.shard_devices is not None:
            return [data.shard_rows(s)[1] for s in range(self.units)]
        return self.gather(own_rows)

    def physical_gpus(self):
        """`n_gpus` of the line: GPUs that really took part. Shards of a device group that share a GPU, or the ranks of a gloo
        rehearsal (all on cuda:0), are ONE GPU however many units the rows were spread over: a one-GPU box must never print an
        8-GPU-looking line; `units` says how many shards / ranks there were."""
        if self.shard_devices is not None:
            return len(set(self.shard_devices))
        return 1 if self.args.allreduce == "gloo" and self.world > 1 else self.units

    def layout(self):
        """Fields of the report that say how the GPUs were driven."""
        if self.shard_devices is None:
            out = {"processes": self.world, "units": self.units}
            if self.physical_gpus() != self.units:
                out["rehearsal"] = f"{self.units} ranks on ONE GPU over gloo: the multi-rank code path, not a multi-GPU measurement"
            return out
        shared = len(set(self.shard_devices)) < self.units
        out = {"processes": 1, "units": self.units,
               "device_group": {"shards": self.units, "devices": self.shard_devices, "shards_share_a_gpu": shared}}
        if shared:
            out["rehearsal"] = (f"{self.units} shards on {len(set(self.shard_devices))} GPU(s): the device group's code path at full size, "
                                f"not a multi-GPU measurement")
        return out

    def close(self):
        self.ctx.close()


def traffic_for(kernel, headline):
    """HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs of this same command at the
    headline shape; not re-measured inside this run). Returns (bytes or None, provenance string or None)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not (headline and os.path.exists(tpath)):
        return None, None
    try:
        t = json.load(open(tpath))
        return t.get(kernel), f"profiles/traffic.json@{t.get('_tag', 'r01_v8')} (rocprofv3 --pmc passes of this command, committed; not measured in this run)"
    except Exception:
        return None, None


def kmeans_measure(job, n, d, K, steps, warmup, with_cpu, cpu_samples):
    """K-means steps/sec, N row-sharded over the ranks, centroids replicated, one all-reduce of [inertia, n_changed,
    counts(K), sums(K*d)] per step. Returns the report dict on rank 0 (None elsewhere)."""
    import numpy as np
    from ml_amd import _lib, synth
    from ml_amd import dist as mldist
    lo, hi = mldist.shard_bounds(n, job.world, job.rank)
    mix = synth.Mixture(d, K, seed=77, diagonal=True)
    X = sample_rows(mix, lo, hi)             # chunk-wise definition: the same rows for any number of ranks
    data = _lib.Data(job.ctx, X)
    del X
    state = {"C": mix.means + 0.3 * np.random.default_rng(1).standard_normal((K, d)), "inertia": None}

    def run(k):
        if job.args.per_step_calls:
            for _ in range(k):
                state["inertia"], _, _, state["C"] = data.kmeans_step(state["C"])
            return
        # the reference's step loop (ML/KMeans.cpp:80-110) in one call, tolerance 0: it can only stop early on identical
        # labels; exactly k steps are executed either way -- a converged loop is entered again
        done = 0
        while done < k:
            n_steps, _, state["inertia"], _, state["C"], _ = data.kmeans_iterate(state["C"], k - done, 0.0)
            done += n_steps

    elapsed, ms = job.timed(run, steps, warmup, ["kmeans_assign"])
    k_ms = ms["kmeans_assign"]
    n_locals = job.rows_per_unit(data, hi - lo)
    out = None
    if job.rank == 0:
        n_local = n_locals[0]
        flops = float(n_local) * K * 3 * d                  # N*K*3d (direct-form distances)
        algorithmic = flops / (k_ms * 1e-3) / 1e12
        matrix = kmeans_uses_matrix_cores(d, K, n_local)
        # the matrix-core kernel executes 2d flops per pair: `frac` prices what the pipe executes, the 3d form is
        # reported beside it
        achieved = algorithmic * (2.0 / 3.0 if matrix else 1.0)
        out = {
            "metric": f"K-means steps/sec at N={n} d={d} K={K} (fp64)",
            "value": steps / elapsed, "unit": "steps/s", "n_gpus": job.physical_gpus(), "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"K-means (Lloyd) N={n} d={d} K={K}, row-sharded over {job.units} GPU(s)", "N": n, "d": d,
                       "K": K, "parallelism": f"dp{job.units}", "inertia": state["inertia"]},
            "rccl_ranks": job.rccl_ranks, "allreduce": job.allreduce, "n_local": n_locals, **job.layout(), **job.last_ranks,
            "roofline": {"bound": "mfma", "kernel": "kmeans_assign", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None,
                         "flops_counted": "executed: 2d per (sample, cluster) on the matrix cores" if matrix else "3d per (sample, cluster)",
                         "algorithmic_3d_tflops": algorithmic, "algorithmic_3d_frac": algorithmic / FP64_PEAK_TFLOPS,
                         "kernel_ms": {"kmeans_assign": k_ms},
                         "hbm_algorithmic_gbs": n_local * (8.0 * d + 4) / (k_ms * 1e-3) / 1e9, "hbm_peak_gbs": HBM_PEAK_GBS},
        }
        if with_cpu:
            from oracle import oracle_ctypes as orc
            n_cpu = min(n, 5 * cpu_samples)
            Xc, _ = mix.sample(n_cpu, stream=999)
            km = orc.KMeans(K)
            km.set_centroids(state["C"], n_cpu)
            sec = km.time_steps(Xc, CPU_ITERATIONS)
            out["cpu_baseline"] = {"value": 1.0 / (sec * n / n_cpu), "unit": "steps/s", "cores": 1, "kind": "port",
                                   "sample": f"{CPU_ITERATIONS} K-means steps of the single-threaded CPU restatement (oracle/) "
                                             f"on {n_cpu} samples, time per step scaled x{n / n_cpu:g} to N={n} (cost is "
                                             f"linear in N, ML/KMeans.cpp:173-177,187-191)",
                                   "seconds_per_step_on_sample": sec}
    data.close()
    return out


def elapsed_hint_ms(n, d, K, diagonal, world):
    """Rough iteration time (60 TFLOP/s full, 20 TFLOP/s diagonal / small shapes): only decides whether event pairs
    inside the timed region are negligible next to an iteration."""
    flops = (diag_flops if diagonal else algorithmic_flops)(n / world, d, K)
    return flops / ((20e12 if diagonal or d < 12 else 60e12)) * 1e3


def em_measure(job, n, d, K, steps, warmup, with_cpu, cpu_samples, diagonal=False):
    """EM iterations/sec (full covariances, or the diagonal extension), N row-sharded over the ranks."""
    import numpy as np
    from ml_amd import _lib, synth
    from ml_amd import dist as mldist
    lo, hi = mldist.shard_bounds(n, job.world, job.rank)
    mix = synth.Mixture(d, K, diagonal=diagonal)
    X = sample_rows(mix, lo, hi)             # the same N rows whatever the number of ranks
    data = _lib.Data(job.ctx, X)
    del X

    # Start exactly like EM::fit without maximise_first (ML/EM.cpp:127-135): given means, shared sample covariance.
    _, cov = data.sample_covariance()
    plan = {"fused": False, "matrix_estep": False, "self_norm": False} if diagonal else data.em_plan(K)
    S0 = np.stack([np.diag(cov)] * K) if diagonal else np.stack([cov] * K)
    state = {"ll": None, "pi": np.full(K, 1.0 / K), "mu": mix.initial_means(), "S": S0}

    def run(k):
        # k trips of the reference loop ML/EM.cpp:143-170 (E-step, M-step incl. the K Cholesky / inverse, all-reduce,
        # convergence test) in one library call; tolerances 0 => exactly k iterations (`ll_change < 0` is never true, :163)
        if job.args.per_step_calls:
            for _ in range(k):
                fn = data.em_step_diag if diagonal else data.em_step
                state["ll"], state["pi"], state["mu"], state["S"] = fn(state["pi"], state["mu"], state["S"])
            return
        done, _, state["ll"], state["pi"], state["mu"], state["S"], _ = data.em_iterate(state["pi"], state["mu"], state["S"], k,
                                                                                       0.0, 0.0, diagonal)
        assert done == k

    names = ["em_diag", "em_close"] if diagonal else ["em_estep", "em_mstats", "em_fused", "em_close", "em_resident"]
    elapsed, ms = job.timed(run, steps, warmup, names, live=(elapsed_hint_ms(n, d, K, diagonal, job.units) >= 1.0))
    ms.setdefault("em_fused_wide", 0.0)
    n_locals = job.rows_per_unit(data, hi - lo)
    if job.rank != 0:
        data.close()
        return None

    n_local = n_locals[0]
    headline = (n, d, K) == (N_TOTAL, DIM, COMPONENTS) and not diagonal
    it_tflops = (diag_flops if diagonal else algorithmic_flops)(n_local, d, K) / (elapsed / steps) / 1e12
    if diagonal:
        k_ms = ms["em_diag"]
        # N K (8d + 25) flops over N d 8 bytes (X once, fused) = 19 flop/B at d = K = 16, above the ridge (78.6 TFLOP/s /
        # 8 TB/s = 9.8): the BINDING roof is fp64 issue, the HBM fraction is reported beside it
        gbs = n_local * d * 8.0 / (k_ms * 1e-3) / 1e9
        tfl = diag_flops(n_local, d, K) / (k_ms * 1e-3) / 1e12
        roof = {"bound": "fp64", "kernel": "em_diag", "achieved": tfl, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tfl / FP64_PEAK_TFLOPS, "traffic": None, "kernel_ms": {"em_diag": k_ms},
                "hbm_algorithmic_gbs": gbs, "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": gbs / HBM_PEAK_GBS,
                "iteration_algorithmic_tflops": it_tflops}
    elif ms["em_fused_wide"] > 0:
        k_ms = ms["em_fused_wide"]                              # E-step + statistics in one kernel: the whole iteration's flops
        achieved = algorithmic_flops(n_local, d, K) / (k_ms * 1e-3) / 1e12
        traffic, source = traffic_for("em_fused_wide", headline and job.units == 1)
        roof = {"bound": "mfma", "kernel": "em_fused_wide", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": source,
                "kernel_ms": {"em_fused_wide": k_ms}, "iteration_algorithmic_tflops": it_tflops}
    elif ms.get("em_resident", 0.0) > 0 and ms["em_fused"] == 0 and ms["em_estep"] == 0:
        # short fits: the WHOLE loop is one launch of resident workgroups (em_resident.hip); its duration over the iterations it ran
        # is the iteration's device time. Neither roof says anything about it: an iteration is one pass over a few tiles per
        # workgroup, one hand-off of the partial sums between the workgroups (~1 - 2 us to the memory side and back) and the closing
        # arithmetic -- a chain of latencies (profiles/r05_resident_phases.txt); `frac` is reported for the contract.
        k_ms = ms["em_resident"] / steps
        gbs = n_local * d * 8.0 / (k_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "em_resident", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": None, "kernel_ms": {"em_resident_per_iteration": k_ms},
                "iteration_algorithmic_tflops": it_tflops,
                "note": "latency-bound: one launch runs every iteration on resident workgroups (pass, one hand-off of the partial "
                        "sums, closing arithmetic); `frac` is reported for the contract, not as a statement about the kernel"}
    elif ms["em_fused"] > 0 and ms["em_estep"] == 0:
        # fused small-shape kernel: algorithmic traffic = X once + LSE once; it is bound by that or by its exp work
        k_ms = ms["em_fused"]
        gbs = n_local * (d + 1) * 8.0 / (k_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "em_fused", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": None, "kernel_ms": {"em_fused": k_ms},
                "iteration_algorithmic_tflops": it_tflops}
        if k_ms * 1e3 < 20.0:
            # a kernel of a few microseconds: neither roof says anything -- the iteration is its three dependent dispatches
            roof["note"] = ("latency-bound: the iteration is three dependent launches (E+M kernel, reduction, closing) of a few "
                            "microseconds each; `frac` is reported for the contract, not as a statement about the kernel")
    else:
        e_ms, m_ms = ms["em_estep"], ms["em_mstats"]
        sn = plan["self_norm"]
        dom_name, dom_ms, dom_flops = ("em_estep", e_ms, estep_flops(n_local, d, K, sn)) if e_ms >= m_ms else \
                                      ("em_mstats", m_ms, mstats_flops(n_local, d, K, sn))
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        traffic, source = traffic_for(dom_name, headline and job.units == 1)
        roof = {"bound": "mfma", "kernel": dom_name, "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic, "traffic_source": source,
                "kernel_ms": {"em_estep": e_ms, "em_mstats": m_ms, "em_close": ms.get("em_close", 0.0)},
                "kernel_tflops": {"em_estep": estep_flops(n_local, d, K, sn) / (e_ms * 1e-3) / 1e12 if e_ms else None,
                                  "em_mstats": mstats_flops(n_local, d, K, sn) / (m_ms * 1e-3) / 1e12 if m_ms else None},
                "exp_runs_in": "em_mstats (self-normalising statistics kernel)" if sn else "em_estep",
                "iteration_algorithmic_tflops": it_tflops}
    kind = "diagonal covariance (extension; the reference is full-covariance only)" if diagonal else "full covariance"
    out = {
        "metric": f"GMM-EM iterations/sec at N={n} d={d} K={K} ({kind}, fp64)" if not headline
                  else "GMM-EM iterations/sec at N=10M d=32 K=64 (full covariance, fp64)",
        "value": steps / elapsed, "unit": "iterations/s", "n_gpus": job.physical_gpus(), "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"GMM-EM N={n} d={d} K={K} {kind}, row-sharded over {job.units} GPU(s)",
                   "N": n, "d": d, "K": K, "parallelism": f"dp{job.units}", "final_mean_log_likelihood": state["ll"]},
        "rccl_ranks": job.rccl_ranks, "allreduce": job.allreduce, "n_local": n_locals, **job.layout(), **job.last_ranks,
        "roofline": roof,
    }
    data.close()
    if with_cpu:
        sec, n_cpu = cpu_baseline(mix, d, K, cpu_samples, CPU_ITERATIONS, diagonal)
        out["cpu_baseline"] = {"value": 1.0 / (sec * n / n_cpu), "unit": "iterations/s", "cores": 1, "kind": "port",
                               "sample": f"{CPU_ITERATIONS} EM iterations of the single-threaded CPU restatement (oracle/) on "
                                         f"{n_cpu} samples (d={d}, K={K}), time per iteration scaled x{n / n_cpu:g} to N={n} "
                                         f"(cost is linear in N, ML/EM.cpp:205,245)",
                               "seconds_per_iteration_on_sample": sec}
        if not diagonal:
            psec, pn, threads = cpu_baseline_all_cores(mix, d, K, max(1000, cpu_samples // 4), 1)
            out["cpu_baseline_all_cores"] = {
                "value": 1.0 / (psec * n / pn), "unit": "iterations/s", "cores": threads, "kind": "port",
                "sample": f"1 EM iteration, row-parallel: {threads} threads x {pn // threads} samples each "
                          f"(one oracle instance per thread), time scaled x{n / pn:g} to N={n}; the reference is "
                          f"single-threaded, this is the all-cores bound for it",
                "seconds_per_iteration_on_sample": psec}
    return out


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args):
    """`python bench.py --gpus N` as a bare command: one fresh process per GPU through torch.distributed.run, started
    before this process has imported torch or opened a HIP context (a process that has touched the GPU is never
    re-executed). Rank 0 of the children prints the JSON line on the inherited stdout; their exit status is ours."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)]
    # the launcher's argparse would claim `--n` as an abbreviation of its own options: hand the children the long form
    cmd += ["--samples" if a == "--n" else ("--samples=" + a[4:] if a.startswith("--n=") else a) for a in sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # this pool's driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def dry_launch(rank, world):
    """--dry-launch: proves the launch plumbing without a GPU -- every rank joins a gloo group and reports what it was
    given; rank 0 prints the one JSON line."""
    import torch.distributed as dist
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)                                   # gloo reports its connections on the C-level stdout
    try:
        dist.init_process_group("gloo")
        if os.environ.get("BENCH_DRY_FAIL_RANK") == str(rank):      # (test) a rank that dies must fail the whole command
            raise SystemExit(3)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)
    box = [None] * world
    dist.all_gather_object(box, {"rank": rank, "world_size_env": int(os.environ["WORLD_SIZE"]), "pid": os.getpid(),
                                 "local_rank": int(os.environ.get("LOCAL_RANK", "0"))})
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "group_size": dist.get_world_size(), "ranks": box}))
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("em", "kmeans", "em-diag"), default="em")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    # (option names must not be abbreviations of torch.distributed.run's own options: its argparse would claim them)
    ap.add_argument("--samples", "--n", dest="n", type=int, default=None,
                    help="total samples (default: the BASELINE.json configuration)")
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--components", type=int, default=None, help="mixture components / clusters")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the K-means `secondary` object of the N=1 EM line")
    ap.add_argument("--cpu-samples", type=int, default=200_000)
    ap.add_argument("--allreduce", choices=("native", "torch", "gloo"), default="native",
                    help="native: ncclAllReduce on the library's own RCCL communicator; torch: torch.distributed nccl hook; "
                         "gloo: (rehearsal on ONE GPU) process group and statistics all-reduce over gloo on the host, every "
                         "rank on cuda:0 -- RCCL refuses two ranks on one device, this exercises everything else of the "
                         "multi-rank path")
    ap.add_argument("--force-hook", action="store_true",
                    help="(diagnostic) single rank, but with the RCCL all-reduce installed (1-rank communicator)")
    ap.add_argument("--per-step-calls", action="store_true",
                    help="(A/B) one mlhip_em_step / mlhip_kmeans_step call per iteration with the closing arithmetic on the host, instead of "
                         "mlhip_em_iterate / mlhip_kmeans_iterate (device-side closing)")
    ap.add_argument("--replay-events", action="store_true",
                    help="(A/B) no HIP events inside the timed region: the kernel times come from a replay of the same steps")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process through the library's device group (mlhip_ctx_create_group: one host thread per "
                         "shard, RCCL between distinct GPUs, an in-process sum when shards share a GPU) instead of one process per "
                         "GPU: what ml::EM::fit / cppyml.clustering.EM.fit use when MLHIP_NUM_GPUS is set")
    ap.add_argument("--dry-launch", action="store_true",
                    help="(test) start the ranks, join a gloo group, report the launch environment; no GPU work")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.single_process:
        return launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.single_process:
        if world != 1:
            raise SystemExit("--single-process is ONE process driving all GPUs: do not start it through a launcher")
    elif world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if args.dry_launch:
        return dry_launch(rank, world)

    import torch
    import torch.distributed as dist

    rehearsal = args.allreduce == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # RCCL prints a version banner on the C-level stdout when its first communicator comes up; stdout is reserved for
    # the one JSON line, so fd 1 points at stderr until the communicators exist.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        if world > 1 and rehearsal:
            dist.init_process_group("gloo")
        elif world > 1:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        elif args.force_hook:
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank))
        if (world > 1 or args.force_hook) and not rehearsal:
            warm = torch.zeros(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(warm)                      # creates torch's communicator (and its banner) now
            torch.cuda.synchronize()
        job = Job(args, rank, local_rank, world, dist, torch)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    with_cpu = args.gpus == 1 and not args.no_cpu_baseline
    if args.workload == "kmeans":
        out = kmeans_measure(job, args.n or 100_000_000, args.dim or 8, args.components or 256, args.steps, args.warmup,
                             with_cpu, args.cpu_samples)
    elif args.workload == "em-diag":
        out = em_measure(job, args.n or 1_000_000, args.dim or 16, args.components or 16, args.steps, args.warmup,
                         with_cpu, args.cpu_samples, diagonal=True)
    else:
        out = em_measure(job, args.n or N_TOTAL, args.dim or DIM, args.components or COMPONENTS, args.steps, args.warmup,
                         with_cpu, args.cpu_samples)
        default_shape = (args.n, args.dim, args.components) == (None, None, None)
        if args.gpus == 1 and default_shape and not args.no_secondary:
            # In the same driver-timed run (the EM block has been freed above), as a LIST:
            # [0] BASELINE.json configs[4] (K-means N=100M, d=8, K=256 on 8 GPUs) at ONE GPU's share of it;
            # [1] BASELINE.json configs[1] (N=1M, d=16, K=16 diagonal-covariance GMM on one GPU) -- an iteration is ~0.1 ms, so
            #     it gets fifty times the steps (a timed region of 18 ms -- ten times -- was at the mercy of one host hiccup: 6 242
            #     instead of 10 900 iterations/s in one run of round 4 with the kernel at its usual 79.6 us).
            # (1.1 ms steps: the clock of the chip ramps for ~15 ms after the idle gap of the upload -- profiles/r04_kmeans_clock.txt: every
            # dispatch takes the same 2.41e6 cycles at 2.03 ... 2.28 GHz -- so the warm-up is ten times and the timed region five times
            # the headline's step counts, as for the 0.09 ms iterations of the diagonal configuration below)
            sec = kmeans_measure(job, 12_500_000, 8, 256, 5 * args.steps, 10 * args.warmup, with_cpu, args.cpu_samples)
            diag = em_measure(job, 1_000_000, 16, 16, 50 * args.steps, 10 * args.warmup, with_cpu, args.cpu_samples, diagonal=True)
            # [2] BASELINE.json configs[0], the reference's own benchmark case (Benchmarks/bm_EM.cpp: N=10k, d=4, K=3): an iteration is
            #     ~18 us -- three dependent launches -- so it gets five hundred times the steps (a region of 0.18 s).
            small = em_measure(job, 10_000, 4, 3, 500 * args.steps, 100 * args.warmup, with_cpu, args.cpu_samples)
            # [3] the reference's own benchmark driver as written (Benchmarks/bm_EM.cpp: mousie d=2 K=3, K-means++ start, tolerances
            #     1e-14), N = 10 000: the WHOLE fit in milliseconds -- upload, initialisation, every iteration, labels -- next to the
            #     CPU restatement's time for the same fit
            bm = bm_em_secondary(10_000, 20) if with_cpu else None
            if out is not None and sec is not None and diag is not None and small is not None:
                sec["config"]["workload"] += " = one GPU's row shard of BASELINE.json configs[4] (N=100M on 8 GPUs)"
                diag["config"]["workload"] += " = BASELINE.json configs[1]"
                small["config"]["workload"] += " = BASELINE.json configs[0] (bm_EM.cpp)"
                out["secondary"] = [sec, diag, small] + ([bm] if bm else [])
        elif args.gpus > 1 and default_shape and not args.no_secondary:
            # multi-GPU runs: BASELINE.json configs[4] at 12.5M rows per GPU in the same driver-timed run -- N = 100M at 8 GPUs is
            # the configuration itself (weak scaling over the driver's 1 / 2 / 4 / 8 series: per-GPU work is fixed)
            sec = kmeans_measure(job, 12_500_000 * args.gpus, 8, 256, args.steps, args.warmup, False, args.cpu_samples)
            if out is not None and sec is not None:
                sec["config"]["workload"] += f" = BASELINE.json configs[4] at 12.5M rows per GPU (N=100M on 8 GPUs)"
                sec["scaling"] = "weak"
                out["secondary"] = [sec]
    if rank == 0:
        print(json.dumps(out))
    job.close()
    if world > 1 or args.force_hook:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
