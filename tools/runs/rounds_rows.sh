#!/bin/bash
# the last tree of a profiled run round by round: kernel times (tools/rocpd_rounds.py) next to the rows the grower sent
#   tools/runs/rounds_rows.sh <tag> [bench args...]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/rounds_rows_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
export LGA_GROW_TRACE=1 LGA_GROW_DUMP=130
rm -rf /tmp/rr_$TAG && timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/rr_$TAG -o kt -- python $ROOT/bench.py --steps 32 --warmup 100 --no-full-run --no-other-precision "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
DB="$(find /tmp/rr_$TAG -name '*.db' | sort | tail -1)"
python $ROOT/tools/rocpd_rounds.py "$DB" "$OUT/rounds_last_tree.csv" > /dev/null
grep "grow dump\] round" "$OUT/err.txt" > "$OUT/rounds_dump.txt"
paste -d' ' <(tail -n +3 "$OUT/rounds_last_tree.csv" | grep -v '^#') <(cut -d' ' -f3- "$OUT/rounds_dump.txt") | cut -c1-220
