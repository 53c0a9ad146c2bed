#!/bin/bash
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-gq}
mkdir -p "$OUT"
cd /tmp
for cfg in "16 16" "16 32" "32 32" "32 64"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$1 timeout -k 10 240 python -u $ROOT/examples/gridsearch_diamonds.py --full --fresh --checkpoint /tmp/pgq_$1_$2.jsonl --rows-in-flight $2 > "$OUT/grid_q$1_rows$2.log" 2>&1 || { tail -30 "$OUT/grid_q$1_rows$2.log"; exit 1; }
  echo "queues $1 rows $2: $(grep -E 'grid search' $OUT/grid_q$1_rows$2.log)"
done
