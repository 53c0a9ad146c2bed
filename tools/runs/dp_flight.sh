#!/bin/bash
# data-parallel path, one RCCL rank: rounds in flight 1 (everything on one stream) against 2 (collectives and
# searches on their own stream, joined by events), at the full size and at a 1/8 shard
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/dp_flight
mkdir -p "$OUT"
cd /tmp
for rows in 10500000 5250000 2625000 1312500; do
  for fl in 1 2; do
    LGA_ROUNDS_IN_FLIGHT=$fl LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup 30 --rows $rows --no-full-run > $OUT/r${rows}_f${fl}.json 2>> $OUT/err.txt
    echo "rows $rows flight $fl: $(python -c "import json;print(json.load(open('$OUT/r${rows}_f${fl}.json'))['ms_per_step'])") ms/tree" | tee -a $OUT/summary.txt
  done
  timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup 30 --rows $rows --no-full-run > $OUT/r${rows}_serial.json 2>> $OUT/err.txt
  echo "rows $rows serial: $(python -c "import json;print(json.load(open('$OUT/r${rows}_serial.json'))['ms_per_step'])") ms/tree" | tee -a $OUT/summary.txt
done
