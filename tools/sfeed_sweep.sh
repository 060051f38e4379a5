#!/bin/bash
# EM iteration time over N of shapes on the matrix-core fused kernel, component records from LDS (MLHIP_FUSED_SFEED=0) against
# scalar registers (=1):   tools/sfeed_sweep.sh "d K" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "$@"; do
    set -- $shape
    for n in 16384 131072 1048576 8388608; do
        for v in 0 1; do
            ms=$(MLHIP_FUSED_SFEED=$v python3 "$R/bench.py" --samples $n --dim $1 --components $2 --no-cpu-baseline --no-secondary --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c 'import sys,json; print("%.4f" % json.loads(sys.stdin.readline())["ms_per_step"])')
            echo "d=$1 K=$2 N=$n sfeed=$v ms_per_iteration=$ms"
        done
    done
done
