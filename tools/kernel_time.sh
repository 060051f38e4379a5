#!/bin/bash
# Average duration of the kernels whose name contains SUBSTR in one EM shape's iteration (rocprofv3 --kernel-trace --stats over
# bench.py --samples N --dim d --components K):   tools/kernel_time.sh SUBSTR N d K      (MLHIP_LIBRARY selects the build)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/kt
cd /tmp && export TMPDIR=/tmp
rm -rf "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -- python3 "$R/bench.py" --samples $2 --dim $3 --components $4 --no-cpu-baseline --no-secondary --steps 30 --warmup 5 > /dev/null 2>&1
python3 - "$O" "$1" "lib=${MLHIP_LIBRARY:-default} N=$2 d=$3 K=$4" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Name"]:
            print("%s  %-50s calls %4s avg %9.1f ns  min %9.1f" % (sys.argv[3], r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-50:], r["Calls"], float(r["AverageNs"]), float(r["MinNs"])))
PY
rm -rf "$O"
