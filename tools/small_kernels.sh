#!/bin/bash
# Kernel durations (rocprofv3) of the small-shape EM iteration (tools/small_latency.py) -> gpurun_out/<tag>_small_kernels.txt
set -e -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/smallk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/smallk -- python3 "$R/tools/small_latency.py" > /tmp/smallk.json 2> /tmp/smallk.err
f=$(find /tmp/smallk -name '*kernel_stats.csv' | head -1)
test -n "$f"
python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    print('%-70s calls=%6s avg_us=%8.2f min=%7.2f max=%8.2f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))" > "$O/${TAG}_small_kernels.txt"
tail -1 /tmp/smallk.json >> "$O/${TAG}_small_kernels.txt"
cat "$O/${TAG}_small_kernels.txt"
