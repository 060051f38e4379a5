#!/bin/bash
# Same-box A/B of two builds of the library (MLHIP_LIBRARY) over a list of EM shapes: tools/lib_ab.sh <old.so> "<N d K>" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OLD=$1; shift
for cfg in "$@"; do
    set -- $cfg
    for lib in "$OLD" "" "$OLD" ""; do
        MLHIP_LIBRARY=$lib python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null |
            python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
k = d['roofline']['kernel_ms']
print('N=$1 d=$2 K=$3 %-4s it/s=%.2f  estep=%.4f  mstats=%.4f' % ('old' if '$lib' else 'new', d['value'], k.get('em_estep', 0), k.get('em_mstats', 0)))"
    done
done
