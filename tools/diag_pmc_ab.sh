#!/bin/bash
# PMC passes of the diagonal-covariance workload for two builds / switches side by side:
#   gpurun -- 'bash tools/diag_pmc_ab.sh "MLHIP_DIAG_TWO_OP=0" "MLHIP_DIAG_TWO_OP=1"'   -> gpurun_out/dgab_<i>_<pass>.csv
set -eu -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
i=0
for setting in "$@"; do
  export $setting
  for pass in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "stall:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SMEM" "mem:SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
    name=${pass%%:*}; counters=${pass#*:}
    rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $R/gpurun_out/dgab_${i}_$name -- python3 $R/bench.py --workload em-diag --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/dgab_${i}_$name.txt 2>&1
    find $R/gpurun_out/dgab_${i}_$name -name '*counter_collection.csv' -exec cp {} $R/gpurun_out/dgab_${i}_$name.csv \;
    rm -rf $R/gpurun_out/dgab_${i}_$name
  done
  unset ${setting%%=*}
  i=$((i+1))
done
