#!/bin/bash
# A/B of the balanced unit dealing in em_mstats_wide (MLHIP_MSTATS_BALANCED=0 / 1) -> gpurun_out/<tag>_mstats_balance.txt
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}_mstats_balance.txt
: > "$O"
for cfg in "5000000 16 64" "5000000 12 64" "5000000 16 32" "5000000 16 48" "5000000 12 48"; do
    set -- $cfg
    for b in 0 1; do
        MLHIP_MSTATS_BALANCED=$b python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null |
            python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
k = d['roofline']['kernel_ms']
print('N=$1 d=$2 K=$3 balanced=$b  it/s=%.2f  estep=%.3f ms  mstats=%.3f ms  frac=%.3f' % (d['value'], k['em_estep'], k['em_mstats'], d['roofline']['frac']))" >> "$O"
    done
done
cat "$O"
