#!/bin/bash
# Per-dispatch clock of the K-means scoring kernel: GRBM_GUI_ACTIVE (cycles the GPU was busy during the dispatch)
# against the dispatch's duration from the kernel trace -> the clock the chip held for THAT launch; SQ_BUSY_CYCLES beside it.
#   gpurun -- 'bash tools/kmeans_clock.sh'      -> gpurun_out/kmclock_{counters,trace}.csv, gpurun_out/kmclock_summary.txt
set -eu -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d "$O/kmclock" -- python3 "$R/bench.py" --workload kmeans --samples 12500000 --steps 12 --warmup 2 --no-cpu-baseline ${KM_EXTRA:-} > "$O/kmclock.txt" 2>&1
find "$O/kmclock" -name '*counter_collection.csv' -exec cp {} "$O/kmclock_counters.csv" \;
find "$O/kmclock" -name '*kernel_trace.csv' -exec cp {} "$O/kmclock_trace.csv" \;
rm -rf "$O/kmclock"
python3 - "$O" <<'PY'
import csv, sys, collections
O = sys.argv[1]
trace = {}
for row in csv.DictReader(open(f"{O}/kmclock_trace.csv")):
    trace[row["Dispatch_Id"]] = (row["Kernel_Name"], int(row["Start_Timestamp"]), int(row["End_Timestamp"]))
ctr = collections.defaultdict(dict)
for row in csv.DictReader(open(f"{O}/kmclock_counters.csv")):
    ctr[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
out = open(f"{O}/kmclock_summary.txt", "w")
prev_end = None
for did in sorted(trace, key=lambda k: trace[k][1]):
    name, t0, t1 = trace[did]
    gap = (t0 - prev_end) / 1e3 if prev_end else 0.0
    prev_end = t1
    if "kmeans_mfma" not in name and "kmeans_assign" not in name:
        continue
    c = ctr.get(did, {})
    dur = t1 - t0
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    line = (f"dispatch {did:>5s}  gap before {gap:9.1f} us  duration {dur / 1e3:9.1f} us  GRBM_GUI_ACTIVE {gui:.4g}  "
            f"-> {gui / dur if dur else 0:.3f} GHz   SQ_BUSY_CYCLES/32 {c.get('SQ_BUSY_CYCLES', 0) / 32:.4g} ({c.get('SQ_BUSY_CYCLES', 0) / 32 / dur if dur else 0:.3f} GHz)")
    print(line)
    out.write(line + "\n")
PY
