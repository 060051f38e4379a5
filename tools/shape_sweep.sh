#!/bin/bash
# Per-shape timings beyond the headline (diagnostic): EM iteration and K-means step on one GPU -> gpurun_out/<tag>_shapes.jsonl
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}_shapes.jsonl
: > "$O"
for cfg in "10000000 2 8" "10000000 4 16" "10000000 8 32" "1000000 16 16" "5000000 16 64" "5000000 24 64" "10000000 32 64" "2500000 48 64" "2500000 64 64" "1000000 128 32" "2500000 32 256" "2500000 32 128" "1250000 32 64"; do
    set -- $cfg
    python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null >> "$O"
done
for cfg in "10000000 2 16" "12500000 8 256" "12500000 8 1024" "12500000 16 256" "12500000 32 64" "5000000 128 256"; do
    set -- $cfg
    python3 "$R/bench.py" --workload kmeans --samples $1 --dim $2 --components $3 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null >> "$O"
done
wc -l "$O"
