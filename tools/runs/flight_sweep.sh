#!/bin/bash
# headline benchmark against the number of growth rounds kept in flight and the size from which a leaf gets a round of its own
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/flight_sweep.txt
: > $OUT
run() {
  echo "$*" >> $OUT
  env "$@" timeout -k 10 120 python $ROOT/bench.py --steps 100 --warmup 30 --no-full-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $OUT
}
run LGA_ROUNDS_IN_FLIGHT=1
run LGA_ROUNDS_IN_FLIGHT=2
run LGA_ROUNDS_IN_FLIGHT=3
run LGA_ROUNDS_IN_FLIGHT=4
run LGA_ROUNDS_IN_FLIGHT=4 LGA_OWN_ROUND_ROWS=65536
run LGA_ROUNDS_IN_FLIGHT=4 LGA_OWN_ROUND_ROWS=1000000
run LGA_ROUNDS_IN_FLIGHT=3 LGA_OWN_ROUND_ROWS=1000000
cat $OUT
