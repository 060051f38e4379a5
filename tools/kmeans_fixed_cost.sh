#!/bin/bash
# Kernel duration of the K-means scoring kernel by rocprofv3 at several N (d = 8, K = 256): where does the per-launch fixed cost
# (~0.11 ms at N = 12.5M) come from? -> gpurun_out/<tag>_kmeans_fixed.txt
set -e -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
OUT=$O/${TAG}_kmeans_fixed.txt
: > "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in 3125000 12500000 25000000 50000000; do
    rm -rf /tmp/kmfix
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kmfix -- python3 "$R/bench.py" --workload kmeans --samples $n --dim 8 --components 256 --steps 10 --warmup 2 --no-cpu-baseline > /tmp/kmfix.json 2> /tmp/kmfix.err
    f=$(find /tmp/kmfix -name '*kernel_stats.csv' | head -1)
    echo "N=$n" >> "$OUT"
    grep -v rocprof /tmp/kmfix.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  bench: ms_per_step', d['ms_per_step'], d['roofline']['kernel_ms'])" >> "$OUT" || true
    if [ -n "$f" ]; then python3 -c "
import csv, sys
for r in csv.DictReader(open('$f')):
    if 'kmeans' in r['Name']:
        print('  %-50s calls=%s avg_ns=%s min=%s max=%s' % (r['Name'][:50], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs']))" >> "$OUT"; fi
    echo "[fixed] $n done"
done
cat "$OUT"
