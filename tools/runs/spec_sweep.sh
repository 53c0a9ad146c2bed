#!/bin/bash
# speculation settings against rounds / splits / time per tree over the 500-tree run
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/r3/spec_sweep.txt
mkdir -p $(dirname $OUT); : > $OUT
run() { echo "== $*" >> $OUT; env "$@" timeout -k 10 200 python $ROOT/tools/runs/rounds_per_tree.py > /tmp/rpt.txt 2>/dev/null || exit 1; sed -n 1p /tmp/rpt.txt >> $OUT; }
run LGA_X=1
run LGA_FILL_ROWS=0
run LGA_SPEC_LO=0.05
run LGA_SPEC_LO=0.0
run LGA_SPEC_LO=0.0 LGA_SPEC_ROWS_LO=1000000
run LGA_SPEC_LO=0.1 LGA_SPEC_HI=0.5
cat $OUT
