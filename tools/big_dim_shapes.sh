#!/bin/bash
# The d > 64 tiers, whole iterations + kernel times:   gpurun -- 'bash tools/big_dim_shapes.sh [tag]'  -> gpurun_out/<tag>_big_dim.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r05}_big_dim.txt
: > "$O"
for cfg in "1000000 128 32" "1000000 96 32" "1000000 72 32" "100000 256 8" "100000 192 8" "100000 512 4" "50000 1024 4"; do
  set -- $cfg
  python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('N=$1 d=$2 K=$3: ms/it %.3f'%b['ms_per_step'], {k:round(v,3) for k,v in r['kernel_ms'].items()}, {k:round(v,1) for k,v in r.get('kernel_tflops',{}).items()}, 'frac %.3f'%r['frac'], 'll', b['config'].get('final_mean_log_likelihood'))" >> "$O"
done
cat "$O"
