#!/bin/bash
# a 1/8 shard of the headline shape (1.3 M rows) against the speculation bar of the grower: its rounds are pure latency
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/spec_sweep_shard.txt
: > $OUT
run() {
  echo "$*" >> $OUT
  env "$@" LGA_GROW_TRACE=1 timeout -k 10 120 python $ROOT/bench.py --steps 100 --warmup 100 --rows 1312500 --no-full-run 2>/tmp/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $OUT
  grep "grow trace" /tmp/err.txt | tail -1 | sed 's/.*rows partitioned/rows partitioned/' >> $OUT
}
run LGA_X=1
run LGA_SPEC_LO=0.05
run LGA_SPEC_LO=0.0
run LGA_SPEC_HI=0.5
run LGA_SPEC_HI=0.2
run LGA_SPEC_LO=0.0 LGA_SPEC_HI=0.0
run LGA_SPEC_ROWS_LO=1000000 LGA_SPEC_ROWS_HI=8000000
run LGA_SOLO_ROWS=4096
run LGA_SOLO_ROWS=8192
run LGA_ROUNDS_IN_FLIGHT=1
run LGA_ROUNDS_IN_FLIGHT=3
cat $OUT
