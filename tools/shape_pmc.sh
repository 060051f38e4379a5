#!/bin/bash
# Counter passes of one EM shape (bench.py --samples N --dim d --components K), summed per kernel name over the timed dispatches:
# instruction counts, pipe-busy and wait cycles of whichever kernels the shape's iteration is made of.
#   usage: tools/shape_pmc.sh N d K [ENV=VALUE ...]      run on the GPU box: gpurun -- 'bash tools/shape_pmc.sh 10000000 4 16 > gpurun_out/pmc.txt'
set -eu -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out; mkdir -p "$O"
N=$1; D=$2; K=$3; shift 3
for kv in "$@"; do export "$kv"; done
W=${WORKLOAD:+--workload $WORKLOAD}      # WORKLOAD=kmeans: the K-means step of the shape instead of the EM iteration
cd /tmp && export TMPDIR=/tmp
echo "== N=$N d=$D K=$K $*"
rm -rf "$O/spmc_trace"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/spmc_trace" -- python3 "$R/bench.py" $W --samples $N --dim $D --components $K --no-cpu-baseline --steps 20 --warmup 5 > "$O/spmc_trace.txt" 2>&1
python3 - "$O/spmc_trace" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print("  %-60s calls %5s avg %10.1f ns  %5s%%" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
for pass in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU" "stall:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "mem:SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR"; do
    name=${pass%%:*}; counters=${pass#*:}
    rm -rf "$O/spmc_$name"
    rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$O/spmc_$name" -- python3 "$R/bench.py" $W --samples $N --dim $D --components $K --no-cpu-baseline --steps 3 --warmup 1 > "$O/spmc_$name.txt" 2>&1
    python3 - "$O/spmc_$name" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-48:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:3]:
    print("  %-48s per dispatch (%d):" % (k, cnt[k]), "  ".join("%s=%.4g" % (c, x / cnt[k]) for c, x in sorted(v.items())))
PY
    rm -rf "$O/spmc_$name"
done
