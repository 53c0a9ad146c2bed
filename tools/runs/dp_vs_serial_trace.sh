#!/bin/bash
# the grower's own per-tree counters (means over each 100 trees): serial learner against one RCCL rank, one round in flight each
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/dp_vs_serial
mkdir -p "$OUT"
cd /tmp
LGA_GROW_TRACE=1 LGA_ROUNDS_IN_FLIGHT=1 timeout -k 10 200 python $ROOT/bench.py --steps 500 --warmup 0 > $OUT/t_serial.json 2> $OUT/t_serial.txt
LGA_GROW_TRACE=1 LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python $ROOT/bench.py --steps 500 --warmup 0 > $OUT/t_dp1.json 2> $OUT/t_dp1.txt
grep "grow trace" $OUT/t_serial.txt; grep "grow trace" $OUT/t_dp1.txt
