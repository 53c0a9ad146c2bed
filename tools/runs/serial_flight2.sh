#!/bin/bash
# repeats of the complete 500-tree run with one and two growth rounds in flight (serial learner), interleaved
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/serial_flight
mkdir -p "$OUT"
cd /tmp
for rep in 1 2 3; do
  for fl in 1 2; do
    LGA_ROUNDS_IN_FLIGHT=$fl timeout -k 10 200 python $ROOT/bench.py --steps 500 --warmup 5 > $OUT/rep${rep}_f${fl}.json 2>> $OUT/err.txt
    echo "rep $rep, 500 trees, flight $fl: $(python -c "import json;print(json.load(open('$OUT/rep${rep}_f${fl}.json'))['value'])") s" | tee -a $OUT/repeats.txt
  done
done
LGA_BENCH_FORCE_DIST=1 timeout -k 10 200 python $ROOT/bench.py --steps 500 --warmup 5 > $OUT/dp1_default.json 2>> $OUT/err.txt
echo "one RCCL rank, default: $(python -c "import json;print(json.load(open('$OUT/dp1_default.json'))['value'])") s" | tee -a $OUT/repeats.txt
