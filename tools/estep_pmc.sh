#!/bin/bash
# PMC comparison of the two matrix-core E-step kernels (MLHIP_ESTEP_CS=0 / 1) on one shape:
#   gpurun -- 'bash tools/estep_pmc.sh TAG [N,d,K]'   -> gpurun_out/TAG_cs{0,1}_{stats,sq,stall}.csv
set -eu -o pipefail
TAG=${1:-estep}
SHAPE=${2:-2500000,32,64}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export MLHIP_LIBRARY=$R/ml_amd/libmlhip_exp.so   # make -C ml_amd/csrc EXPERIMENTS=1
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for cs in 0 1; do
    export MLHIP_ESTEP_CS=$cs
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_t$cs" -- python3 "$R/tools/estep_ab.py" --child "$SHAPE" --steps 4 --reps 2 > "$O/${TAG}_cs${cs}_stats.txt" 2>&1
    find "$O/${TAG}_t$cs" -name '*kernel_stats.csv' -exec cp {} "$O/${TAG}_cs${cs}_stats.csv" \;
    rm -rf "$O/${TAG}_t$cs"
    for pass in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "stall:SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
        name=${pass%%:*}; counters=${pass#*:}
        rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$O/${TAG}_p$cs" -- python3 "$R/tools/estep_ab.py" --child "$SHAPE" --steps 2 --reps 1 > "$O/${TAG}_cs${cs}_$name.txt" 2>&1
        find "$O/${TAG}_p$cs" -name '*counter_collection.csv' -exec cp {} "$O/${TAG}_cs${cs}_$name.csv" \;
        rm -rf "$O/${TAG}_p$cs"
    done
    echo "[estep_pmc] cs=$cs done"
done
python3 - "$O" "$TAG" <<'PY'
import csv, sys, collections
O, TAG = sys.argv[1], sys.argv[2]
for cs in "01":
    for name in ("sq", "stall"):
        try:
            rows = list(csv.DictReader(open(f"{O}/{TAG}_cs{cs}_{name}.csv")))
        except OSError as e:
            print("missing", e); continue
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in rows:
            if "estep" not in r["Kernel_Name"]: continue
            a = acc[(r["Kernel_Name"][:60], r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(f"cs={cs} {k} {c} {v / n:.4g} (n={n})")
    try:
        for r in csv.DictReader(open(f"{O}/{TAG}_cs{cs}_stats.csv")):
            if "estep" in r["Name"] or "mstats" in r["Name"]: print(f"cs={cs} {r['Name'][:70]} avg_ns={r['AverageNs']} calls={r['Calls']}")
    except OSError as e:
        print("missing", e)
PY
