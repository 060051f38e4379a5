#!/usr/bin/env python3
"""Condenses the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/<tag>_*) into profiles/:
   <tag>_bench.json, <tag>_bench_kernel_stats.csv (as is), <tag>_pmc_{fetch,write,sq}.csv (per-kernel averages of every
   counter over the EM kernels' dispatches) and profiles/traffic.json (HBM bytes per launch of the two hot kernels: FETCH_SIZE
   is in KiB and is doubled for gfx950 streaming reads as MI355X_MICROARCH.md prescribes; WRITE_SIZE in KiB, exact).
   usage: python tools/summarise_profiles.py r01_v5"""
import csv
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for key in ("em_estep_mfma4_kernel", "em_estep_mfma_kernel", "em_estep_kernel", "em_mstats_wide_kernel", "em_mstats_kernel",
                "em_reduce_kernel", "kmeans", "transpose_kernel"):
        if key in name:
            tmpl = name[name.index(key) + len(key):].split("(")[0]
            return key + tmpl
    return name.split("(")[0][-60:]


def condense(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    tag = sys.argv[1]
    src, dst = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
    for name in ("bench.json", "bench_kernel_stats.csv", "kmeans_bench.json", "kmeans_kernel_stats.csv", "diag_bench.json",
                 "diag_kernel_stats.csv"):
        if os.path.exists(os.path.join(src, f"{tag}_{name}")):
            shutil.copy(os.path.join(src, f"{tag}_{name}"), os.path.join(dst, f"{tag}_{name}"))
    summary = {}
    for p in ("fetch", "write", "sq", "stall", "kmeans_sq"):
        raw = f"{tag}_kmeans_pmc_sq.csv" if p == "kmeans_sq" else f"{tag}_pmc_{p}.csv"
        if not os.path.exists(os.path.join(src, raw)):
            continue
        c = condense(os.path.join(src, raw))
        summary[p] = c
        with open(os.path.join(dst, raw), "w") as f:
            f.write("kernel,counter,average_per_dispatch,dispatches\n")
            for k in sorted(c):
                for cn in sorted(c[k]):
                    f.write(f"\"{k}\",{cn},{c[k][cn][0]:.6g},{c[k][cn][1]}\n")

    def pick(table, prefix, counter):
        best = None
        for k, cs in table.items():
            if k.startswith(prefix) and counter in cs and (best is None or cs[counter][1] > best[1]):
                best = cs[counter]
        return best[0] if best else None

    e_f, e_w = pick(summary["fetch"], "em_estep", "FETCH_SIZE"), pick(summary["write"], "em_estep", "WRITE_SIZE")
    m_f, m_w = pick(summary["fetch"], "em_mstats_wide", "FETCH_SIZE"), pick(summary["write"], "em_mstats_wide", "WRITE_SIZE")
    traffic = {
        "_tag": tag,
        "_comment": "HBM bytes per kernel launch at N=10M d=32 K=64 from rocprofv3 PMC (separate --pmc passes): FETCH_SIZE (KiB "
                    "units) doubled as MI355X_MICROARCH.md prescribes for gfx950 streaming reads, plus WRITE_SIZE (KiB, exact). "
                    f"Sources: profiles/{tag}_pmc_fetch.csv, profiles/{tag}_pmc_write.csv",
        "em_estep": int(2 * e_f * 1024 + e_w * 1024),
        "em_mstats": int(2 * m_f * 1024 + m_w * 1024),
        "detail": {"em_estep": {"FETCH_SIZE_KiB": e_f, "WRITE_SIZE_KiB": e_w, "algorithmic_read_bytes": 2.56e9, "algorithmic_write_bytes": 5.2e9},
                   "em_mstats": {"FETCH_SIZE_KiB": m_f, "WRITE_SIZE_KiB": m_w, "algorithmic_read_bytes": 7.76e9}},
    }
    with open(os.path.join(dst, "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic, indent=1))
    for k, cs in summary["sq"].items():
        if k.startswith("em_estep") or k.startswith("em_mstats"):
            busy, mf = cs.get("SQ_BUSY_CYCLES"), cs.get("SQ_VALU_MFMA_BUSY_CYCLES")
            print(k, {c: f"{v[0]:.4g}" for c, v in cs.items()})


if __name__ == "__main__":
    main()
