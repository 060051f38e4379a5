#!/bin/bash
# Tiled E-step above d = 128: a unit's row blocks cut into P parts (MLHIP_ESTEP_PARTS, experiments library) -- E-step kernel time per P and shape.
#   gpurun -- 'bash tools/estep_parts_ab.sh' -> gpurun_out/estep_parts_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/estep_parts_ab.txt
: > "$O"
export MLHIP_LIBRARY=$R/ml_amd/libmlhip_exp.so
for cfg in "50000 1024 4" "100000 512 4" "100000 256 8" "8192 512 4" "4096 1024 4" "20000 320 8" "200000 1024 4"; do
  set -- $cfg
  for P in 1 2 4 8 auto; do
    if [ $P = auto ]; then unset MLHIP_ESTEP_PARTS; else export MLHIP_ESTEP_PARTS=$P; fi
    python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']
print('N=$1 d=$2 K=$3 P=$P: em_estep %.3f ms  ms/it %.3f'%(r['kernel_ms']['em_estep'], b['ms_per_step']))" >> "$O"
  done
done
cat "$O"
