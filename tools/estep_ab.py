#!/usr/bin/env python3
"""A/B of the two matrix-core E-step kernels (em_estep_mfma4.hip: sample-stationary, experiments/em_estep_cs.hip:
component-stationary, only in the `make EXPERIMENTS=1` library) on one GPU: same data, same parameters, T iterations of mlhip_em_iterate with tolerances 0 in one child process per setting
of MLHIP_ESTEP_CS; the children report the E-step / statistics kernel times (HIP events on the kernels' stream), the
log-likelihood history and a checksum of the final responsibilities. The two kernels execute the same matrix instructions in
the same order, so everything must agree BITWISE.

    python tools/estep_ab.py [--shapes N,d,K ...] [--steps T] [--reps R]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n, d, K, steps, reps):
    sys.path.insert(0, ROOT)
    import numpy as np
    from ml_amd import _lib, synth
    mix = synth.Mixture(d, K, seed=20240601)
    X, _ = mix.sample(n)
    ctx = _lib.Context(0)
    dt = _lib.Data(ctx, X)
    mu0 = mix.initial_means()
    _, cov = dt.sample_covariance()
    S0 = np.stack([cov] * K)
    pi0 = np.full(K, 1.0 / K)
    out = {"estep_ms": [], "mstats_ms": []}
    dt.em_iterate(pi0, mu0, S0, 2)   # warm-up
    for _ in range(reps):
        ctx.timing_reset()
        ctx.timing_enable(True)
        st, conv, ll, pi, mu, S, hist = dt.em_iterate(pi0, mu0, S0, steps)
        ctx.synchronize()
        e_ms, e_n = ctx.timing_get("em_estep")
        m_ms, m_n = ctx.timing_get("em_mstats")
        ctx.timing_enable(False)
        out["estep_ms"].append(e_ms / max(1, e_n))
        out["mstats_ms"].append(m_ms / max(1, m_n))
    R = dt.em_responsibilities(K)
    out["hist"] = [float(h).hex() for h in hist]
    out["resp_sha"] = hashlib.sha256(np.ascontiguousarray(R).tobytes()).hexdigest()[:16]
    out["labels_sha"] = hashlib.sha256(dt.em_labels(K).tobytes()).hexdigest()[:16]
    out["mu_sha"] = hashlib.sha256(mu.tobytes()).hexdigest()[:16]
    dt.close()
    ctx.close()
    print("ABRESULT " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=["2000000,32,64", "300000,32,16", "200000,28,30", "200000,24,64"])
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        n, d, K = (int(v) for v in a.child.split(","))
        child(n, d, K, a.steps, a.reps)
        return 0
    bad = 0
    for shape in a.shapes:
        res = {}
        for cs in ("0", "1"):
            env = dict(os.environ, MLHIP_ESTEP_CS=cs, MLHIP_LIBRARY=os.path.join(ROOT, "ml_amd", "libmlhip_exp.so"))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", shape, "--steps", str(a.steps), "--reps", str(a.reps)],
                               env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("ABRESULT ")]
            if p.returncode != 0 or not line:
                print(f"shape {shape} cs={cs}: FAILED rc={p.returncode}\n{p.stdout[-2000:]}\n{p.stderr[-4000:]}", flush=True)
                bad += 1
                continue
            res[cs] = json.loads(line[0][len("ABRESULT "):])
        if len(res) == 2:
            same = all(res["0"][k] == res["1"][k] for k in ("hist", "resp_sha", "labels_sha", "mu_sha"))
            bad += 0 if same else 1
            print(json.dumps({"shape": shape, "bitwise_equal": same,
                              "estep_ms_mfma4": [round(v, 4) for v in res["0"]["estep_ms"]],
                              "estep_ms_cs": [round(v, 4) for v in res["1"]["estep_ms"]],
                              "mstats_ms": [round(v, 4) for v in res["1"]["mstats_ms"]]}), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
