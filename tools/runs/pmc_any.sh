#!/bin/bash
# any counter set over 7 trees of the headline benchmark: PMC="SQ_WAVES ..." KERNEL=grow_partition tools/runs/pmc_any.sh <tag> [bench args...]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/r3/pmc_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout -s KILL 240 rocprofv3 --pmc $PMC -d /tmp/pmc_$TAG -o p -- python $ROOT/bench.py --steps 7 --warmup 1 --no-full-run "$@" > /tmp/pmc_$TAG.log 2>&1
python $ROOT/tools/rocpd_pmc.py "$(find /tmp/pmc_$TAG -name '*.db' | sort | tail -1)" "$OUT/pmc.csv" > /dev/null
head -1 "$OUT/pmc.csv"
grep "${KERNEL:-grow}" "$OUT/pmc.csv" | head -${LINES_MAX:-30}
