#!/bin/bash
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-cold}
mkdir -p "$OUT"
cd /tmp
timeout -k 10 200 python -u $ROOT/bench/cold_start.py > "$OUT/cold_start_1.json" 2> "$OUT/cold.err" || { tail -20 "$OUT/cold.err"; exit 1; }
timeout -k 10 200 python -u $ROOT/bench/cold_start.py > "$OUT/cold_start_2.json" 2>> "$OUT/cold.err" || { tail -20 "$OUT/cold.err"; exit 1; }
cat "$OUT/cold_start_2.json"
