#!/bin/bash
# sensitivity of the headline benchmark to the unit count of multi-unit leaves and to the solo limit
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/sweep_units.txt
: > $OUT
for tu in 256 384 512 640 1024 2048; do
  echo "LGA_TARGET_UNITS=$tu" >> $OUT
  LGA_TARGET_UNITS=$tu timeout -k 10 120 python $ROOT/bench.py --steps 100 --warmup 30 --no-full-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $OUT
done
for sr in 4096 8192 16384; do
  echo "LGA_SOLO_ROWS=$sr" >> $OUT
  LGA_SOLO_ROWS=$sr timeout -k 10 120 python $ROOT/bench.py --steps 100 --warmup 30 --no-full-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $OUT
done
cat $OUT
