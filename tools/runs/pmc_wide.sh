#!/bin/bash
# hardware counters of the kernels of one wide shape (6 trees): tools/runs/pmc_wide.sh <shape prefix> <tag> <counters...>
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
SHAPE=$1; TAG=$2; shift; shift
OUT=$ROOT/gpurun_out/r3/pmc_$SHAPE
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/pmcw_$TAG
timeout -s KILL 240 rocprofv3 --pmc "$@" -d /tmp/pmcw_$TAG -o p -- python $ROOT/tools/runs/wide_shapes.py $SHAPE --precision full --trees 3 > /tmp/pmcw_$TAG.log 2>&1
python $ROOT/tools/rocpd_pmc.py "$(find /tmp/pmcw_$TAG -name '*.db' | sort | tail -1)" "$OUT/pmc_$TAG.csv" > /dev/null
grep "hist_kernel\|scan_pairs" "$OUT/pmc_$TAG.csv" | head -40
