#!/usr/bin/env python3
"""Headline benchmark: Gaussian-mixture EM iterations/sec at N=10M, d=32, K=64, fp64 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = one EM iteration = E-step + M-step statistics + statistics all-reduce + M-step closing arithmetic
(K Cholesky/inverse) + convergence bookkeeping, i.e. one trip of the reference loop ML/EM.cpp:143-170, executed
through the C ABI entry point mlhip_em_step on data already resident in HBM. The N samples are row-sharded
over the ranks (strong scaling: the job size is fixed), the only exchange is one all-reduce of the K*561+1
sufficient statistics per iteration through torch.distributed (backend nccl == RCCL over xGMI).

Prints ONE JSON line on rank 0 (see the fields below).

    python bench.py --workload kmeans [--gpus N ...]       second workload, same contract: K-means steps/sec at
        N=100M, d=8, K=256 (BASELINE.json configs[4]); one step = mlhip_kmeans_step = assignment + exact update sums +
        all-reduce of counts/sums + new centroids (ML/KMeans.cpp:82-108)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_TOTAL, DIM, COMPONENTS = 10_000_000, 32, 64
FP64_PEAK_TFLOPS = 78.6      # MI355X fp64 vector == matrix peak (spec); measured ceilings: tools/microbench_fp64
HBM_PEAK_GBS = 8000.0


def algorithmic_flops(n, d, K):
    """N*K*(2d^2 + 6d + 25) per EM iteration (full covariance)."""
    return float(n) * K * (2 * d * d + 6 * d + 25)


def estep_flops(n, d, K):
    return float(n) * K * (d * d + 3 * d + 25)


def mstats_flops(n, d, K):
    return float(n) * K * (d * d + 3 * d)


def cpu_baseline(mix, d, K, n_cpu, iters):
    """The CPU restatement of the reference (oracle/, single thread like the reference) on a bounded sample."""
    from oracle import oracle_ctypes as orc
    X, _ = mix.sample(n_cpu, stream=999)
    em = orc.EM(K)
    em.set_parameters(mix.initial_means(), np.stack([np.cov(X.T)] * K), np.full(K, 1.0 / K))
    sec_per_iter = em.time_iterations(X, iters)
    return sec_per_iter, n_cpu


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
    out = np.empty((hi - lo, mix.d))
    c = lo // ROW_CHUNK
    while c * ROW_CHUNK < hi:
        a, b = c * ROW_CHUNK, (c + 1) * ROW_CHUNK
        block = mix.sample(ROW_CHUNK, stream=c)[0]
        s0, s1 = max(lo, a), min(hi, b)
        out[s0 - lo:s1 - lo] = block[s0 - a:s1 - a]
        c += 1
    return out


def kmeans_workload(args, rank, local_rank, world, dist, torch):
    """K-means steps/sec, N row-sharded over the ranks, centroids replicated, one all-reduce of [inertia, n_changed,
    counts(K), sums(K*d)] per step."""
    from ml_amd import _lib, synth
    from ml_amd import dist as mldist
    n = args.n if args.n is not None else 100_000_000
    d = args.dim if args.dim is not None else 8
    K = args.components if args.components is not None else 256
    lo, hi = mldist.shard_bounds(n, world, rank)
    mix = synth.Mixture(d, K, seed=77, diagonal=True)
    X = sample_rows(mix, lo, hi)             # chunk-wise definition: the same rows for any number of ranks
    ctx = _lib.Context(local_rank)
    if world > 1 or args.force_hook:
        mldist.install_allreduce(ctx, world, rank)
    data = _lib.Data(ctx, X)
    del X
    C = mix.means + 0.3 * np.random.default_rng(1).standard_normal((K, d))

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    inertia = None
    for _ in range(args.warmup):
        inertia, _, _, C = data.kmeans_step(C)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        inertia, _, _, C = data.kmeans_step(C)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(3):
        inertia, _, _, C = data.kmeans_step(C)
    k_ms, _ = ctx.timing_get("kmeans_assign")
    ctx.timing_enable(False)
    if rank == 0:
        n_local = hi - lo
        flops = float(n_local) * K * 3 * d                  # N*K*3d (direct-form distances)
        achieved = flops / (k_ms * 1e-3) / 1e12
        out = {
            "metric": "K-means steps/sec at N=100M d=8 K=256 (fp64)" if (n, d, K) == (100_000_000, 8, 256)
                      else f"K-means steps/sec at N={n} d={d} K={K} (fp64; diagnostic shape)",
            "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"K-means (Lloyd) N={n} d={d} K={K}, row-sharded over {world} GPU(s)", "N": n, "d": d,
                       "K": K, "parallelism": f"dp{world}", "inertia": inertia},
            "roofline": {"bound": "mfma", "kernel": "kmeans_assign", "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": None,
                         "kernel_ms": {"kmeans_assign": k_ms},
                         "hbm_algorithmic_gbs": n_local * (8.0 * d + 4) / (k_ms * 1e-3) / 1e9, "hbm_peak_gbs": HBM_PEAK_GBS},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle_ctypes as orc
            n_cpu = min(n, 5 * args.cpu_samples)
            Xc, _ = mix.sample(n_cpu, stream=999)
            km = orc.KMeans(K)
            km.set_centroids(C, n_cpu)
            sec = km.time_steps(Xc, 1)
            out["cpu_baseline"] = {"value": 1.0 / (sec * n / n_cpu), "unit": "steps/s", "cores": 1, "kind": "port",
                                   "sample": f"1 K-means step of the single-threaded CPU restatement (oracle/) on {n_cpu} "
                                             f"samples, time scaled x{n / n_cpu:g} to N={n} (cost is linear in N, "
                                             f"ML/KMeans.cpp:173-177,187-191)",
                                   "seconds_per_step_on_sample": sec}
        print(json.dumps(out))
    data.close()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("em", "kmeans"), default="em")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=None, help="total samples (default: the BASELINE.json configuration)")
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--components", type=int, default=None, help="mixture components / clusters")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-samples", type=int, default=200_000)
    ap.add_argument("--force-hook", action="store_true",
                    help="(diagnostic) single rank, but with the torch.distributed/RCCL all-reduce hook installed")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist
    from ml_amd import _lib, synth
    from ml_amd import dist as mldist

    torch.cuda.set_device(local_rank)
    # RCCL prints a version banner on the C-level stdout when its first communicator comes up; stdout is reserved for
    # the one JSON line, so fd 1 points at stderr until the communicator exists.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        if world > 1:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        elif args.force_hook:
            dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29544", rank=0, world_size=1,
                                    device_id=torch.device("cuda", local_rank))
        if world > 1 or args.force_hook:
            warm = torch.zeros(1, dtype=torch.float64, device="cuda")
            dist.all_reduce(warm)                      # creates the communicator (and its banner) now
            torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    if args.workload == "kmeans":
        kmeans_workload(args, rank, local_rank, world, dist, torch)
        if world > 1 or args.force_hook:
            dist.destroy_process_group()
        return

    n = args.n if args.n is not None else N_TOTAL
    d = args.dim if args.dim is not None else DIM
    K = args.components if args.components is not None else COMPONENTS
    lo, hi = mldist.shard_bounds(n, world, rank)
    mix = synth.Mixture(d, K)
    X = sample_rows(mix, lo, hi)             # the same N rows whatever the number of ranks

    ctx = _lib.Context(local_rank)
    if world > 1 or args.force_hook:
        mldist.install_allreduce(ctx, world, rank)
    data = _lib.Data(ctx, X)
    del X

    # Start exactly like EM::fit without maximise_first (ML/EM.cpp:127-135): given means, shared sample covariance.
    _, cov = data.sample_covariance()
    pi = np.full(K, 1.0 / K)
    mu = mix.initial_means()
    S = np.stack([cov] * K)

    def barrier():
        if world > 1:
            dist.barrier()
        ctx.synchronize()
        torch.cuda.synchronize()

    ll = None
    for _ in range(args.warmup):
        ll, pi, mu, S = data.em_step(pi, mu, S)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ll, pi, mu, S = data.em_step(pi, mu, S)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Per-kernel device time (HIP events on the kernels' own stream), measured in a separate pass so that the
    # event synchronisation does not perturb the timed region above.
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(3):
        ll, pi, mu, S = data.em_step(pi, mu, S)
    e_ms, _ = ctx.timing_get("em_estep")
    m_ms, _ = ctx.timing_get("em_mstats")
    f_ms, _ = ctx.timing_get("em_fused")     # small shapes (d <= 8): E-step + statistics in one kernel, X read once
    ctx.timing_enable(False)

    if rank == 0:
        n_local = hi - lo
        dom_name, dom_ms, dom_flops = ("em_estep", e_ms, estep_flops(n_local, d, K)) if e_ms >= m_ms else \
                                      ("em_mstats", m_ms, mstats_flops(n_local, d, K))
        roof = None
        if f_ms > 0 and e_ms == 0:
            # fused small-shape kernel: algorithmic traffic = X once + LSE once; it is bound by that or by its exp work
            gbs = n_local * (d + 1) * 8.0 / (f_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "em_fused", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": None, "kernel_ms": {"em_fused": f_ms},
                    "iteration_algorithmic_tflops": algorithmic_flops(n_local, d, K) / (elapsed / args.steps) / 1e12}
            dom_name, dom_ms = "em_fused", f_ms
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        headline = (n, d, K) == (10_000_000, 32, 64) and world == 1     # the shape the PMC passes were collected on
        if os.path.exists(tpath) and headline:
            try:
                traffic = json.load(open(tpath)).get(dom_name)
            except Exception:
                traffic = None
        out = {
            "metric": "GMM-EM iterations/sec at N=10M d=32 K=64 (full covariance, fp64)" if (n, d, K) == (10_000_000, 32, 64)
                      else f"GMM-EM iterations/sec at N={n} d={d} K={K} (full covariance, fp64; diagnostic shape)",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"GMM-EM N={n} d={d} K={K} full covariance, row-sharded over {world} GPU(s)",
                       "N": n, "d": d, "K": K, "parallelism": f"dp{world}",
                       "final_mean_log_likelihood": ll},
            "roofline": roof or {"bound": "mfma", "kernel": dom_name, "achieved": achieved, "peak": FP64_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                                 "kernel_ms": {"em_estep": e_ms, "em_mstats": m_ms},
                                 "iteration_algorithmic_tflops": algorithmic_flops(n_local, d, K) / (elapsed / args.steps) / 1e12},
        }
        if world == 1 and not args.no_cpu_baseline:
            iters = 1
            sec, n_cpu = cpu_baseline(mix, d, K, args.cpu_samples, iters)
            out["cpu_baseline"] = {"value": 1.0 / (sec * n / n_cpu), "unit": "iterations/s", "cores": 1, "kind": "port",
                                   "sample": f"{iters} EM iteration(s) of the single-threaded CPU restatement (oracle/) on "
                                             f"{n_cpu} samples (d={d}, K={K}), time scaled x{n / n_cpu:g} to N={n} "
                                             f"(cost is linear in N, ML/EM.cpp:205,245)",
                                   "seconds_per_iteration_on_sample": sec}
            psec, pn, threads = cpu_baseline_all_cores(mix, d, K, max(1000, args.cpu_samples // 2), iters)
            out["cpu_baseline_all_cores"] = {
                "value": 1.0 / (psec * n / pn), "unit": "iterations/s", "cores": threads, "kind": "port",
                "sample": f"{iters} EM iteration(s), row-parallel: {threads} threads x {pn // threads} samples each "
                          f"(one oracle instance per thread), time scaled x{n / pn:g} to N={n}; the reference is "
                          f"single-threaded, this is the all-cores bound for it",
                "seconds_per_iteration_on_sample": psec}
        print(json.dumps(out))
    data.close()
    ctx.close()
    if world > 1 or args.force_hook:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
