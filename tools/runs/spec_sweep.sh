#!/bin/bash
# headline benchmark against the speculation bar of the grower
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/spec_sweep.txt
: > $OUT
run() {
  echo "$*" >> $OUT
  env "$@" timeout -k 10 120 python $ROOT/bench.py --steps 150 --warmup 50 --no-full-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $OUT
}
run LGA_X=1
run LGA_SPEC_LO=0.05
run LGA_SPEC_LO=0.1
run LGA_SPEC_LO=0.4
run LGA_SPEC_HI=0.7
run LGA_SPEC_HI=0.5
run LGA_SPEC_LO=0.05 LGA_SPEC_HI=0.7
run LGA_SPEC_ROWS_LO=32768
run LGA_SPEC_ROWS_LO=524288
run LGA_SPEC_ROWS_HI=8000000
run LGA_SPEC_ROWS_HI=1000000
run LGA_SOLO_ROWS=4096
run LGA_SOLO_ROWS=8192
run LGA_TARGET_UNITS=320
run LGA_TARGET_UNITS=1280
cat $OUT
