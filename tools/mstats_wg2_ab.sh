#!/bin/bash
# A/B of two statistics-kernel workgroups per CU for few-component shapes (MLHIP_MSTATS_WG2=0 / 1) -> gpurun_out/<tag>_mstats_wg2.txt
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}_mstats_wg2.txt
: > "$O"
for cfg in "1000000 16 16" "5000000 12 16" "5000000 20 8" "5000000 24 16"; do
    set -- $cfg
    for b in 0 1 0 1; do
        MLHIP_MSTATS_WG2=$b python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null |
            python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
k = d['roofline']['kernel_ms']
print('N=$1 d=$2 K=$3 wg2=$b  it/s=%.2f  estep=%.4f ms  mstats=%.4f ms  frac=%.3f' % (d['value'], k['em_estep'], k['em_mstats'], d['roofline']['frac']))" >> "$O"
    done
done
cat "$O"
