#!/bin/bash
# A/B of environment switches on the headline benchmark, interleaved and repeated:
#   tools/runs/ab_env.sh <tag> <repeats> "<bench args>" "ENV_A=.. ENV_B=.." "ENV_C=.." ...
# every setting is a quoted list of NAME=value pairs ("" = defaults); prints ms per tree of every run
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; REP=$2; ARGS=$3; shift 3
OUT=$ROOT/gpurun_out/ab_$TAG
mkdir -p "$OUT"
cd "$ROOT"
for r in $(seq 1 "$REP"); do
  i=0
  for setting in "$@"; do
    # shellcheck disable=SC2086
    env $setting timeout -k 10 200 python bench.py $ARGS --no-other-precision > "$OUT/run_${i}_$r.json" 2> "$OUT/run_${i}_$r.err"
    echo "rep $r [$setting] $(grep -o '"ms_per_step": [0-9.]*' "$OUT/run_${i}_$r.json") $(grep -o '"value": [0-9.]*' "$OUT/run_${i}_$r.json")" | tee -a "$OUT/summary.txt"
    i=$((i + 1))
  done
done
