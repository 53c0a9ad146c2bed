#!/bin/bash
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-gt}
mkdir -p "$OUT"
cd /tmp
LGA_GRID_TRACE=1 LGA_TREE_TIMING=1 timeout -k 10 240 python -u $ROOT/examples/gridsearch_diamonds.py --full --fresh --checkpoint /tmp/pgt.jsonl --rows-in-flight 48 > "$OUT/grid_trace.log" 2>&1 || { tail -30 "$OUT/grid_trace.log"; exit 1; }
grep -E "grid search|grid trace" "$OUT/grid_trace.log"; grep "tree kernel" "$OUT/grid_trace.log" | tail -2
