#!/usr/bin/env python3
"""Condenses a rocprofv3 counter_collection.csv of tools/diag_knockout.py into one row per knock-out mask: the dispatches of
em_diag_mixed_kernel in order, `per_mask` launches each (reps + 5 warm-up), counters averaged over a mask's launches.
    usage: diag_knockout_pmc.py counter_collection.csv per_mask"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2])
masks = [0, 1, 2, 4, 3, 7, 15, 31, 63, 127, 23, 39, 71, 8, 16, 32, 64, 128, 135, 0]       # the order of tools/diag_knockout.py
disp = collections.OrderedDict()
for r in rows:
    if "em_diag_mixed" not in r["Kernel_Name"]:
        continue
    disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(disp)
names = sorted({c for d in disp.values() for c in d})
print("mask," + ",".join(names))
for m, mask in enumerate(masks):
    chunk = ids[m * per:(m + 1) * per][2:]                                                   # (skip the first launches of a mask)
    if not chunk:
        break
    print(f"{mask}," + ",".join(f"{sum(disp[i].get(c, 0.0) for i in chunk) / len(chunk):.4g}" for c in names))
