#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/km_sweep.jsonl
: > "$O"
for cfg in "12500000 8 256" "12500000 8 1024" "12500000 16 256" "12500000 32 64" "12500000 4 64" "5000000 64 256" "5000000 128 256" "12500000 8 250"; do
    set -- $cfg
    python3 "$R/bench.py" --workload kmeans --samples $1 --dim $2 --components $3 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null >> "$O"
done
python3 - <<'PY'
import json,os
for l in open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/km_sweep.jsonl'):
    j=json.loads(l); print(j['config']['workload'][:40], round(j['ms_per_step'],3), j.get('roofline',{}).get('frac'))
PY
