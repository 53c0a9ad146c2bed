#!/bin/bash
# does the placement of kernel arguments matter?  (descriptors of up to 48 leaves travel as kernel arguments)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/kernarg_sweep.txt
: > $OUT
run() {
  echo "$*" >> $OUT
  env "$@" timeout -k 10 120 python $ROOT/bench.py --steps 100 --warmup 30 --no-full-run 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> $OUT
}
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run LGA_X=1
cat $OUT
