#!/bin/bash
# PMC passes of the diagonal-covariance workload (bench.py --workload em-diag): gpurun -- 'bash tools/diag_pmc.sh'
set -eu -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for pass in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU" "stall:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "mem:SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  name=${pass%%:*}; counters=${pass#*:}
  rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $R/gpurun_out/dg_$name -- python3 $R/bench.py --workload em-diag --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/dg_$name.txt 2>&1
  find $R/gpurun_out/dg_$name -name '*counter_collection.csv' -exec cp {} $R/gpurun_out/dg_$name.csv \;
  rm -rf $R/gpurun_out/dg_$name
done
