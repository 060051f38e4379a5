#!/bin/bash
# Kernel trace of one small-d, large-N EM shape (bench.py --samples N --dim d --components K): per-kernel average durations,
# with and without the vector-unit statistics form (MLHIP_FUSED_VALU).   usage: tools/small_shape_trace.sh N d K
# Run on the GPU box:  gpurun -- 'bash tools/small_shape_trace.sh 10000000 2 3 > gpurun_out/small_shape.txt'
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
N=${1:-10000000}; D=${2:-2}; K=${3:-3}
for v in 0 1; do
    export MLHIP_FUSED_VALU=$v
    rm -rf "$O/sst_$v"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/sst_$v" -- python3 "$R/bench.py" --samples $N --dim $D --components $K --no-cpu-baseline --steps 50 --warmup 10 > "$O/sst_$v.txt" 2>&1 || exit 1
    echo "== N=$N d=$D K=$K MLHIP_FUSED_VALU=$v"
    tail -1 "$O/sst_$v.txt" | cut -c1-330
    python3 - "$O/sst_$v" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("  %-72s calls %5s avg %10.1f ns  %5s%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
done
