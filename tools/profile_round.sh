#!/bin/bash
# Collects the measurement set that DESIGN.md / bench.py cite, on the GPU box:
#   gpurun -- 'bash tools/profile_round.sh r01_v5'
# 1. bench.py (default size, with the CPU baseline)                         -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command (no CPU baseline) -> gpurun_out/<tag>_bench_kernel_stats.csv
# 4. K-means workload (bench.py --workload kmeans): bench line + kernel stats  -> gpurun_out/<tag>_kmeans_*
# 3. separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ busy counters)       -> gpurun_out/<tag>_pmc_{fetch,write,sq}.csv
# Copy what should be judged into profiles/ afterwards (tools/summarise_profiles.py does it and rebuilds traffic.json).
set -e -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 20 --warmup 3 > "$O/${TAG}_bench.json" 2> "$O/${TAG}_bench.err"
echo "[profile] bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_stats" -- python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$O/${TAG}_stats.txt" 2>&1
find "$O/${TAG}_stats" -name '*kernel_stats.csv' -exec cp {} "$O/${TAG}_bench_kernel_stats.csv" \;
echo "[profile] kernel stats done"
for pass in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU" "stall:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    name=${pass%%:*}; counters=${pass#*:}
    rocprofv3 --kernel-trace --pmc $counters --output-format csv -d "$O/${TAG}_pmc_$name" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$O/${TAG}_pmc_$name.txt" 2>&1
    find "$O/${TAG}_pmc_$name" -name '*counter_collection.csv' -exec cp {} "$O/${TAG}_pmc_$name.csv" \;
    echo "[profile] pmc $name done"
done
# second workload: K-means config E on one GPU (bench line + kernel trace)
python3 "$R/bench.py" --workload kmeans --steps 10 --warmup 2 > "$O/${TAG}_kmeans_bench.json" 2> "$O/${TAG}_kmeans_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_kmstats" -- python3 "$R/bench.py" --workload kmeans --steps 5 --warmup 1 --no-cpu-baseline > "$O/${TAG}_kmstats.txt" 2>&1
find "$O/${TAG}_kmstats" -name '*kernel_stats.csv' -exec cp {} "$O/${TAG}_kmeans_kernel_stats.csv" \;
# SQ counters of the K-means scoring kernel at one GPU's share of config E (the `secondary[0]` shape)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --output-format csv -d "$O/${TAG}_kmpmc" -- python3 "$R/bench.py" --workload kmeans --samples 12500000 --steps 3 --warmup 1 --no-cpu-baseline > "$O/${TAG}_kmpmc.txt" 2>&1
find "$O/${TAG}_kmpmc" -name '*counter_collection.csv' -exec cp {} "$O/${TAG}_kmeans_pmc_sq.csv" \;
rm -rf "$O/${TAG}_kmpmc"
echo "[profile] kmeans done"
rm -rf "$O/${TAG}_kmstats"
# third workload: diagonal-covariance EM (BASELINE.json configs[1]): bench line + kernel trace
python3 "$R/bench.py" --workload em-diag --steps 50 --warmup 5 > "$O/${TAG}_diag_bench.json" 2> "$O/${TAG}_diag_bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${TAG}_dgstats" -- python3 "$R/bench.py" --workload em-diag --steps 50 --warmup 5 --no-cpu-baseline > "$O/${TAG}_dgstats.txt" 2>&1
find "$O/${TAG}_dgstats" -name '*kernel_stats.csv' -exec cp {} "$O/${TAG}_diag_kernel_stats.csv" \;
echo "[profile] diag done"
rm -rf "$O/${TAG}_dgstats"
rm -rf "$O/${TAG}_stats" "$O/${TAG}"_pmc_fetch "$O/${TAG}"_pmc_write "$O/${TAG}"_pmc_sq "$O/${TAG}"_pmc_stall
ls -la "$O" | grep "$TAG"
