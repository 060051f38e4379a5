#!/bin/bash
# The same EM shape timed REP times in fresh processes (run-to-run spread of bench.py --samples N --dim d --components K):
#   tools/repeat_shape.sh N d K [REP] [ENV=V ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=$1; D=$2; K=$3; REP=${4:-6}; shift; shift; shift; shift || true
for kv in "$@"; do export "$kv"; done
for r in $(seq $REP); do
    python3 "$R/bench.py" --samples $N --dim $D --components $K --no-cpu-baseline --no-secondary --steps 100 --warmup 20 2>/dev/null | tail -1 | python3 -c 'import sys,json; j=json.loads(sys.stdin.readline()); print("N='$N' d='$D' K='$K' '"$*"' ms_per_iteration=%.4f kernels=%s" % (j["ms_per_step"], j["roofline"].get("kernel_ms")))'
done
