#!/bin/bash
# serial learner: one against two growth rounds in flight, over row counts, and the complete 500-tree run
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/serial_flight
mkdir -p "$OUT"
cd /tmp
for rows in 10500000 2625000 1312500 600000; do
  for fl in 1 2; do
    LGA_ROUNDS_IN_FLIGHT=$fl timeout -k 10 200 python $ROOT/bench.py --steps 100 --warmup 30 --rows $rows --no-full-run > $OUT/r${rows}_f${fl}.json 2>> $OUT/err.txt
    echo "rows $rows flight $fl: $(python -c "import json;print(json.load(open('$OUT/r${rows}_f${fl}.json'))['ms_per_step'])") ms/tree" | tee -a $OUT/summary.txt
  done
done
for fl in 1 2; do
  LGA_ROUNDS_IN_FLIGHT=$fl timeout -k 10 200 python $ROOT/bench.py --steps 500 --warmup 5 > $OUT/full500_f${fl}.json 2>> $OUT/err.txt
  echo "500 trees flight $fl: $(python -c "import json;print(json.load(open('$OUT/full500_f${fl}.json'))['value'])") s" | tee -a $OUT/summary.txt
done
