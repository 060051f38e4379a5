#!/bin/bash
# rocprofv3 --kernel-trace --stats of the shapes above d = 64 (top kernels by total time):
#   gpurun -- 'bash tools/big_dim_kernel_stats.sh [tag]'   -> gpurun_out/<tag>_big_dim_kernel_stats.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$R/gpurun_out/${1:-r05}_big_dim_kernel_stats.txt
O=$R/gpurun_out/bdks
cd /tmp && export TMPDIR=/tmp
: > "$T"
for cfg in "1000000 128 32" "100000 256 8" "50000 1024 4"; do
  set -- $cfg
  rm -rf "$O"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --no-cpu-baseline --no-secondary --steps 10 --warmup 2 > /dev/null 2>&1
  echo "== N=$1 d=$2 K=$3 (rocprofv3 --kernel-trace --stats of bench.py --steps 10 --warmup 2; 12 iterations + the first records)" >> "$T"
  python3 - "$O" >> "$T" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-46:]
    print("  %-46s calls %5s avg %9.1f us total %8.2f ms" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
rm -rf "$O"
cat "$T"
