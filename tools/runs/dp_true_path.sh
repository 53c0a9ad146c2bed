#!/bin/bash
# the distributed learner itself on one GPU: a communicator of ONE RCCL rank that the engine really trains through
# (LGA_DP_KEEP_SINGLE_RANK, set by bench.py under LGA_BENCH_FORCE_DIST) - rounds in flight 1 against 2, both
# exchanges, at the full size and at a 1/8 shard; the serial learner next to it
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/dp_true_path
mkdir -p "$OUT"
cd /tmp
val() { python -c "import json;d=json.load(open('$1'));print(d['ms_per_step'], d['config']['parallelism'])"; }
for rows in 10500000 1312500; do
  timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup 30 --rows $rows --no-full-run > $OUT/r${rows}_serial.json 2>> $OUT/err.txt
  echo "rows $rows serial: $(val $OUT/r${rows}_serial.json)" | tee -a $OUT/summary.txt
  for ex in allreduce reduce_scatter; do
    for fl in 1 2; do
      LGA_DP_EXCHANGE=$ex LGA_ROUNDS_IN_FLIGHT=$fl LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup 30 --rows $rows --no-full-run > $OUT/r${rows}_${ex}_f${fl}.json 2>> $OUT/err.txt
      echo "rows $rows $ex flight $fl: $(val $OUT/r${rows}_${ex}_f${fl}.json)" | tee -a $OUT/summary.txt
    done
  done
done
