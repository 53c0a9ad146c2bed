#!/bin/bash
# the data-parallel code path with one rank through RCCL: per-tree time at the full size and at a 1/8 shard,
# next to the serial learner on the same shard, and a kernel trace of the shard run
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/dp_trace
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup 30 --no-full-run --stats > $OUT/rccl_1rank_full.json 2> $OUT/err.txt
cut -c1-220 $OUT/rccl_1rank_full.json
LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup 30 --rows 1312500 --no-full-run --stats > $OUT/rccl_1rank_shard.json 2>> $OUT/err.txt
cut -c1-220 $OUT/rccl_1rank_shard.json
timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup 30 --rows 1312500 --no-full-run --stats > $OUT/serial_shard.json 2>> $OUT/err.txt
cut -c1-220 $OUT/serial_shard.json
rm -rf /tmp/kt && LGA_BENCH_FORCE_DIST=1 timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $ROOT/bench.py --steps 32 --warmup 30 --rows 1312500 --no-full-run > /tmp/kt.log 2>&1
DB="$(find /tmp/kt -name '*.db' | sort | tail -1)"
python $ROOT/tools/rocpd_stats.py "$DB" "$OUT/kernel_stats_shard_dp.csv" > /dev/null
python $ROOT/tools/rocpd_dispatches.py "$DB" "$OUT/dispatches_shard_dp.csv" 0.9 600 > /dev/null
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $ROOT/bench.py --steps 32 --warmup 30 --rows 1312500 --no-full-run > /tmp/kt.log 2>&1
DB="$(find /tmp/kt -name '*.db' | sort | tail -1)"
python $ROOT/tools/rocpd_stats.py "$DB" "$OUT/kernel_stats_shard_serial.csv" > /dev/null
python $ROOT/tools/rocpd_dispatches.py "$DB" "$OUT/dispatches_shard_serial.csv" 0.9 600 > /dev/null
echo done
