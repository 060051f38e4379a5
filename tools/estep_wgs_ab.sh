#!/bin/bash
# A/B of the E-step's workgroups per CU at small d (MLHIP_ESTEP_WGS=2 / 4) -> gpurun_out/<tag>_estep_wgs.txt
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}_estep_wgs.txt
: > "$O"
for cfg in "5000000 16 64" "5000000 12 64" "5000000 24 64" "5000000 20 32" "1000000 16 16" "5000000 16 128"; do
    set -- $cfg
    for b in 2 4 2 4; do
        MLHIP_ESTEP_WGS=$b python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null |
            python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
k = d['roofline']['kernel_ms']
print('N=$1 d=$2 K=$3 wgs=$b  it/s=%.2f  estep=%.4f ms  mstats=%.4f ms' % (d['value'], k['em_estep'], k['em_mstats']))" >> "$O"
    done
done
cat "$O"
