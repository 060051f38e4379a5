#!/usr/bin/env python3
"""Knock-out timing of configuration B's kernel (em_diag_mixed_kernel, N = 1M, d = 16, K = 16 diagonal): which phase costs what.
Needs the experiments build (`make -C ml_amd/csrc EXPERIMENTS=1`): MLHIP_DIAG_KO=mask switches phases of the kernel off at run
time (results wrong by construction; device/em_diag.hip lists the bits). Every mask is timed on the SAME parameters (per-step calls
of mlhip_em_step_diag with fixed inputs: a knocked-out kernel must not steer the next iteration onto another code path); the time
is the kernel's own (HIP events around the launch, `mlhip_timing_get("em_diag")`), median of `reps` launches after a warm-up.
    usage: python tools/diag_knockout.py [N] [reps]        -> one line per mask"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MLHIP_LIBRARY", os.path.join(ROOT, "ml_amd", "libmlhip_exp.so"))
import numpy as np
from ml_amd import _lib, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
d = K = 16
mix = synth.Mixture(d, K, diagonal=True)
X, _ = mix.sample(n)
ctx = _lib.Context()
dt = _lib.Data(ctx, X)
pi0 = np.full(K, 1.0 / K)
mu0 = mix.initial_means()
var0 = np.stack([np.diag(c) for c in mix.covs])
NAMES = {1: "density", 2: "exp", 4: "statistics", 8: "x loads after first tile", 16: "lse store", 32: "prologue", 64: "flush", }
masks = [0, 1, 2, 4, 1 | 2, 1 | 2 | 4, 1 | 2 | 4 | 8, 1 | 2 | 4 | 8 | 16, 1 | 2 | 4 | 8 | 16 | 32, 1 | 2 | 4 | 8 | 16 | 32 | 64,
         1 | 2 | 4 | 16, 1 | 2 | 4 | 32, 1 | 2 | 4 | 64, 8, 16, 32, 64, 128, 1 | 2 | 4 | 128, 0]
ctx.timing_enable(True)
for mask in masks:
    os.environ["MLHIP_DIAG_KO"] = str(mask)
    ts = []
    for r in range(reps + 5):
        ctx.timing_reset()
        dt.em_step_diag(pi0, mu0, var0)
        ms, cnt = ctx.timing_get("em_diag")
        assert cnt == 1
        if r >= 5:
            ts.append(ms * 1e3)
    label = " + ".join(NAMES[b] for b in NAMES if mask & b) or "nothing (full kernel)"
    extra = "; WITH the next tile's samples requested during the statistics phase" if mask & 128 else ""
    print(f"mask {mask:3d}: {np.median(ts):7.2f} us (min {min(ts):7.2f})   without: {label}{extra}", flush=True)
