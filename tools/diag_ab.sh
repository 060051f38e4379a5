#!/bin/bash
# A/B of the two diagonal-covariance kernels (MLHIP_DIAG_SGPR=0 / 1) with SQ counters: gpurun -- 'bash tools/diag_ab.sh'
set -eu -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export MLHIP_DIAG_SGPR=$v
  python3 "$R/bench.py" --workload em-diag --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sgpr=$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  for pass in "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU" "mem:SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
    name=${pass%%:*}; counters=${pass#*:}
    rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$O/dgab_$name$v" -- python3 "$R/bench.py" --workload em-diag --steps 5 --warmup 1 --no-cpu-baseline > "$O/dgab_$name$v.txt" 2>&1 || true
    find "$O/dgab_$name$v" -name '*counter_collection.csv' -exec cp {} "$O/dgab_$name$v.csv" \;
    rm -rf "$O/dgab_$name$v"
  done
done
python3 - "$O" <<'PY'
import csv, sys, collections
O = sys.argv[1]
for v in "01":
    for name in ("sq", "mem"):
        try: rows = list(csv.DictReader(open(f"{O}/dgab_{name}{v}.csv")))
        except OSError as e: print("missing", e); continue
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in rows:
            if "em_diag" not in r["Kernel_Name"]: continue
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        print(f"sgpr={v}", {k: f"{s / n:.4g}" for k, (s, n) in sorted(acc.items())})
PY
