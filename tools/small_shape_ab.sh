#!/bin/bash
# Iteration time of EM shapes at the border of the vector-unit form (device/em_fused_small.hip: valu_max_k), with it (default)
# and without (MLHIP_FUSED_VALU=0):  tools/small_shape_ab.sh "N d K" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "$@"; do
    set -- $shape
    for v in 0 1 0 1; do
        ms=$(MLHIP_FUSED_VALU=$v python3 "$R/bench.py" --samples $1 --dim $2 --components $3 --no-cpu-baseline --no-secondary --steps 50 --warmup 10 2>/dev/null | tail -1 | python3 -c 'import sys,json; print("%.4f" % json.loads(sys.stdin.readline())["ms_per_step"])')
        echo "N=$1 d=$2 K=$3 valu=$v ms_per_iteration=$ms"
    done
done
