#!/bin/bash
# Round-5 measurement set for the short-fit work (device-resident loop, cold start, the reference's own benchmark drivers) and the
# shard sizes of the predicted multi-GPU lines:   gpurun -- 'bash tools/round5_small.sh r05_v1'   -> gpurun_out/<tag>_*
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p "$O"
cd "$R"
python3 tools/first_call.py > "$O/${TAG}_first_call.txt" 2>/dev/null
python3 tools/first_call.py >> "$O/${TAG}_first_call.txt" 2>/dev/null          # (a second fresh process: the spread)
python3 tools/fit_small.py > "$O/${TAG}_fit_small.txt" 2>/dev/null
python3 tools/bm_clustering.py > "$O/${TAG}_bm_clustering.txt" 2>/dev/null
MLHIP_RESIDENT_PROFILE=1 python3 tools/resident_phases.py > "$O/${TAG}_resident_phases.txt" 2>&1
python3 tools/small_latency.py > "$O/${TAG}_small_latency.json" 2>/dev/null
echo "[round5] small done"
: > "$O/${TAG}_shards.jsonl"
for n in 10000000 5000000 2500000 1250000; do
    python3 bench.py --samples $n --steps 20 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null >> "$O/${TAG}_shards.jsonl"
done
for n in 100000000 50000000 25000000 12500000; do
    python3 bench.py --workload kmeans --samples $n --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null >> "$O/${TAG}_shards.jsonl"
done
echo "[round5] shards done"
ls -la "$O" | grep "$TAG"
