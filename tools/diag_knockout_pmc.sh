#!/bin/bash
# Counters of configuration B's kernel per knock-out mask (tools/diag_knockout.py under rocprofv3; `make EXPERIMENTS=1` build):
#   gpurun -- 'bash tools/diag_knockout_pmc.sh'   -> gpurun_out/dko_<pass>.csv  (one row per mask: per-launch averages)
set -eu -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
export MLHIP_LIBRARY=$R/ml_amd/libmlhip_exp.so
for pass in "busy:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "lds:SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  name=${pass%%:*}; counters=${pass#*:}
  rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $R/gpurun_out/dko_$name -- python3 $R/tools/diag_knockout.py 1000000 6 > $R/gpurun_out/dko_$name.txt 2>&1
  find $R/gpurun_out/dko_$name -name '*counter_collection.csv' -exec cp {} $R/gpurun_out/dko_$name.raw.csv \;
  rm -rf $R/gpurun_out/dko_$name
  python3 $R/tools/diag_knockout_pmc.py $R/gpurun_out/dko_$name.raw.csv 11 > $R/gpurun_out/dko_$name.csv
  rm -f $R/gpurun_out/dko_$name.raw.csv
done
