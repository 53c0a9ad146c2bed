#!/bin/bash
# per-kernel statistics of 33 trees of one wide shape: tools/runs/kstats_wide.sh <shape prefix> [wide_shapes args...]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
SHAPE=$1; shift
OUT=$ROOT/gpurun_out/r3/kstats_$SHAPE
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/kt_$SHAPE && timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt_$SHAPE -o kt -- python $ROOT/tools/runs/wide_shapes.py $SHAPE "$@" > "$OUT/run.txt" 2> /tmp/kt_$SHAPE.log
DB="$(find /tmp/kt_$SHAPE -name '*.db' | sort | tail -1)"
python $ROOT/tools/rocpd_stats.py "$DB" "$OUT/kernel_stats.csv" > /dev/null
python $ROOT/tools/rocpd_rounds.py "$DB" "$OUT/rounds_last_tree.csv" > /dev/null || true
head -12 "$OUT/kernel_stats.csv"
