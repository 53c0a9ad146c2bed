#!/bin/bash
# per-kernel statistics of 32 trees of the headline benchmark (after 100 warm-up trees): tools/runs/kstats.sh <tag> [bench args...]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/r3/kstats_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/kt_$TAG && timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt_$TAG -o kt -- python $ROOT/bench.py --steps 32 --warmup 100 --no-full-run "$@" > "$OUT/bench.json" 2> /tmp/kt_$TAG.log
DB="$(find /tmp/kt_$TAG -name '*.db' | sort | tail -1)"
python $ROOT/tools/rocpd_stats.py "$DB" "$OUT/kernel_stats.csv" > /dev/null
python $ROOT/tools/rocpd_rounds.py "$DB" "$OUT/rounds_last_tree.csv" > /dev/null || true
head -12 "$OUT/kernel_stats.csv"
