#!/bin/bash
# kernel trace of trees 100..132 of the headline benchmark: per-kernel statistics, the dispatch timeline of
# two trees and the grower's own per-tree counters
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/trace_now
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
LGA_GROW_TRACE=1 timeout -k 10 200 python $ROOT/bench.py --steps 200 --warmup 100 --no-full-run --stats > "$OUT/bench_stats.json" 2> "$OUT/grow_trace.txt"
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $ROOT/bench.py --steps 32 --warmup 100 --no-full-run > /tmp/kt.log 2>&1
DB="$(find /tmp/kt -name '*.db' | sort | tail -1)"
python $ROOT/tools/rocpd_stats.py "$DB" "$OUT/kernel_stats.csv" > /dev/null
python $ROOT/tools/rocpd_dispatches.py "$DB" "$OUT/dispatches.csv" 0.9 400 > /dev/null
echo done
