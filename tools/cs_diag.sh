#!/bin/bash
# timing diagnostics of em_estep_cs (MLHIP_CS_DIAG variants compute WRONG results): gpurun -- 'bash tools/cs_diag.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export MLHIP_LIBRARY=$R/ml_amd/libmlhip_exp.so   # make -C ml_amd/csrc EXPERIMENTS=1
for d in 0 1 2 3 4 7; do
  echo "diag=$d: $(MLHIP_ESTEP_CS=1 MLHIP_CS_DIAG=$d python3 $R/tools/estep_ab.py --child 2500000,32,64 --steps 4 --reps 3 | sed 's/.*"estep_ms": \(\[[^]]*\]\).*/\1/')"
done
echo "old: $(MLHIP_ESTEP_CS=0 python3 $R/tools/estep_ab.py --child 2500000,32,64 --steps 4 --reps 3 | sed 's/.*"estep_ms": \(\[[^]]*\]\).*/\1/')"
