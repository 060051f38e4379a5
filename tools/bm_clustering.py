#!/usr/bin/env python3
"""The reference's own benchmark drivers, timed as the reference times them (Benchmarks/bm_EM.cpp:9-48, bm_KMeans.cpp:9-48,
bm_LinearAlgebra.cpp:6-48): the 'mousie' sample (d = 2, K = 3), N in {100, 1 000, 10 000, 100 000}, K-means++ start, tolerances
1e-14 (K-means: 3 initialisations), the WHOLE fit -- upload, initialisation, every iteration, labels -- through the drop-in's
Python surface (cppyml.clustering), beside the single-threaded CPU restatement of the reference (bench.py's cpu_baseline legs).

  cold  = the first fit of a fresh process, after `import` and the context have been paid for (both reported once);
  warm  = median of the following fits in that process.

The three helpers of bm_LinearAlgebra.cpp (host symbols of libmlhip.so; n = 4 ... 1024) are timed by tools/bm_linear_algebra.cpp,
which this script builds with g++ and runs.      usage: python tools/bm_clustering.py [--sizes 100,1000,...] [--repeats R]"""
import argparse, json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import json, os, sys, time
t0 = time.perf_counter()
sys.path.insert(0, %(root)r)
import numpy as np
import bench
from ml_amd import _lib
from ml_amd.cppyml import clustering as cl
t1 = time.perf_counter()
import ctypes
ctx_t0 = time.perf_counter()
_lib.check(_lib.lib.mlpp_device_context(ctypes.byref(ctypes.c_void_p())))      # ml::device::context(): the facade's GPU context, created on first use
ctx_ms = (time.perf_counter() - ctx_t0) * 1e3
X = np.load(%(data)r)
fit = bench.bm_em_fit if %(algo)r == "em" else bench.bm_kmeans_fit
runs = [fit(X) for _ in range(1 + %(repeats)d)]
print(json.dumps({"import_ms": (t1 - t0) * 1e3, "context_ms": ctx_ms, "cold_ms": runs[0][0] * 1e3,
                  "warm_ms": float(np.median([r[0] for r in runs[1:]])) * 1e3, "detail": [list(r[1:]) for r in runs[:2]]}))
'''


def gpu_leg(algo, X, repeats):
    import numpy as np
    path = f"/tmp/bm_clustering_{os.getpid()}_{len(X)}.npy"
    np.save(path, X)
    try:
        out = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "data": path, "algo": algo, "repeats": repeats}],
                             capture_output=True, text=True, timeout=600)
    finally:
        os.unlink(path)
    if out.returncode != 0:
        raise SystemExit(out.stderr[-2000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100,1000,10000,100000")
    ap.add_argument("--repeats", type=int, default=10)
    ap.add_argument("--no-helpers", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import bench
    print("# whole-fit wall time in ms; GPU = cppyml.clustering on one MI355X, CPU = single-threaded restatement of the reference on this host")
    for algo, cpu_fit in (("em", bench.cpu_bm_em_fit), ("kmeans", bench.cpu_bm_kmeans_fit)):
        for n in [int(v) for v in args.sizes.split(",")]:
            X = bench.mousie(n)
            g = gpu_leg(algo, X, args.repeats)
            reps = 3 if n <= 10000 else 1
            cpu = [cpu_fit(X) for _ in range(reps)]
            cpu_ms = float(np.median([c[0] for c in cpu])) * 1e3
            name = "bm_EM em_mousie" if algo == "em" else "bm_KMeans km_mousie"
            print(f"{name}/{n}: GPU cold {g['cold_ms']:.2f} ms, warm {g['warm_ms']:.3f} ms | CPU {cpu_ms:.3f} ms | "
                  f"CPU/GPU warm x{cpu_ms / g['warm_ms']:.2f}, cold x{cpu_ms / g['cold_ms']:.2f} | GPU {g['detail'][1]} CPU {list(cpu[0][1:])} | "
                  f"(process: import {g['import_ms']:.0f} ms, context {g['context_ms']:.0f} ms)", flush=True)
    if not args.no_helpers:
        exe = os.path.join(ROOT, "tools", "bm_linear_algebra")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "bm_linear_algebra.cpp"),
                               "-o", exe, "-L", os.path.join(ROOT, "ml_amd"), "-lmlhip", "-Wl,-rpath," + os.path.join(ROOT, "ml_amd"),
                               "-Wl,-rpath,/opt/rocm/lib"])
        sys.stdout.flush()
        subprocess.check_call([exe])


if __name__ == "__main__":
    main()
