#!/bin/bash
# Closing arithmetic at d > 64 on the device (em_close_big.hip) against the host thread team (MLHIP_DEVICE_CLOSE=0), whole iterations:
#   gpurun -- 'bash tools/close_big_ab.sh'  -> gpurun_out/close_big_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/close_big_ab.txt
: > "$O"
for cfg in "1000000 128 32" "1000000 72 32" "100000 256 8" "100000 512 4" "50000 1024 4"; do
  set -- $cfg
  for dc in 0 1; do
    MLHIP_DEVICE_CLOSE=$dc python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('N=$1 d=$2 K=$3 MLHIP_DEVICE_CLOSE=$dc: ms/it %.3f'%b['ms_per_step'], {k:round(v,3) for k,v in r['kernel_ms'].items()}, 'frac %.3f'%r['frac'], 'll', b['config'].get('final_mean_log_likelihood'))" >> "$O"
  done
done
cat "$O"
