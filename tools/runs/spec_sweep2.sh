#!/bin/bash
# speculation settings against rounds / splits / time per tree over the 500-tree run:
#   tools/runs/spec_sweep2.sh <tag> "ENV=.. ENV=.." "ENV=.." ...     ("LGA_X=1" = defaults)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/spec_$TAG.txt
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for setting in "$@"; do
  echo "== $setting" >> "$OUT"
  # shellcheck disable=SC2086
  env $setting timeout -k 10 200 python "$ROOT/tools/runs/rounds_per_tree.py" > /tmp/rpt.txt 2>/tmp/rpt.err || { tail -5 /tmp/rpt.err; exit 1; }
  sed -n 1p /tmp/rpt.txt >> "$OUT"
done
cat "$OUT"
