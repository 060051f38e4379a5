#!/bin/bash
# HBM traffic and durations of the two EM kernels at a mid-dimension shape (is the lw round trip what they wait for?)
#   gpurun -- 'bash tools/midd_traffic.sh'   -> gpurun_out/midd_traffic.txt
set -eu -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ARGS="--samples 5000000 --dim 16 --components 64 --steps 3 --warmup 1 --no-cpu-baseline"
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  name=${pass%%:*}; counters=${pass#*:}
  rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$O/midd_$name" -- python3 "$R/bench.py" $ARGS > "$O/midd_$name.txt" 2>&1
  find "$O/midd_$name" -name '*counter_collection.csv' -exec cp {} "$O/midd_$name.csv" \;
  find "$O/midd_$name" -name '*kernel_trace.csv' -exec cp {} "$O/midd_${name}_trace.csv" \;
  rm -rf "$O/midd_$name"
done
python3 - "$O" <<'PY'
import csv, sys, collections
O = sys.argv[1]
dur = collections.defaultdict(list)
for row in csv.DictReader(open(f"{O}/midd_fetch_trace.csv")):
    n = row["Kernel_Name"]
    key = "em_estep" if "em_estep" in n else ("em_mstats" if "em_mstats_wide" in n else None)
    if key: dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("fetch", "write", "sq"):
    for row in csv.DictReader(open(f"{O}/midd_{p}.csv")):
        n = row["Kernel_Name"]
        key = "em_estep" if "em_estep" in n else ("em_mstats" if "em_mstats_wide" in n else None)
        if key: agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(f"{O}/midd_traffic.txt", "w") as f:
    for k in ("em_estep", "em_mstats"):
        d = sorted(dur[k])[len(dur[k]) // 2] / 1e6
        c = {n: sum(v) / len(v) for n, v in agg[k].items()}
        rd, wr = 2 * c.get("FETCH_SIZE", 0) * 1024 / 1e9, c.get("WRITE_SIZE", 0) * 1024 / 1e9   # (gfx950: FETCH_SIZE counts 64-B requests as 32 B: MI355X_MICROARCH.md)
        line = (f"{k}: {d:.3f} ms, HBM read {rd:.2f} GB + write {wr:.2f} GB = {(rd + wr) / d:.2f} TB/s; "
                f"MFMA busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024 / (c.get('SQ_BUSY_CYCLES', 1) / 32):.2f}, VALU issue {4 * c.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / (c.get('SQ_BUSY_CYCLES', 1) / 32):.2f}, "
                f"waiting {c.get('SQ_WAIT_ANY', 0) / max(1.0, c.get('SQ_WAVE_CYCLES', 1)):.2f} of the wave cycles")
        print(line); f.write(line + "\n")
PY
