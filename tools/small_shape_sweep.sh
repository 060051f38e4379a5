#!/bin/bash
# Iteration time of small-d EM shapes over N, matrix-core form (MLHIP_FUSED_VALU=0) against the vector-unit form (the default)
# (MLHIP_FUSED_VALU=1): where the crossover lies (kValuMinSamples, device/em_fused_small.hip).
#   usage: tools/small_shape_sweep.sh ["d K" ...]
# Run on the GPU box:  gpurun -- 'bash tools/small_shape_sweep.sh > gpurun_out/small_shape_sweep.txt'
R=${GRAFT_REPO_ROOT:-/root/repo}
SHAPES=("$@"); [ ${#SHAPES[@]} -eq 0 ] && SHAPES=("2 3" "2 8" "1 16" "3 6" "4 4")
for shape in "${SHAPES[@]}"; do
    set -- $shape
    for n in 16384 65536 262144 1048576 4194304; do
        for v in 0 1; do
            ms=$(MLHIP_FUSED_VALU=$v python3 "$R/bench.py" --samples $n --dim $1 --components $2 --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python3 -c 'import sys,json; print("%.4f" % json.loads(sys.stdin.readline())["ms_per_step"])')
            echo "d=$1 K=$2 N=$n valu=$v ms_per_iteration=$ms"
        done
    done
done
