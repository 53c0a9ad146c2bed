#!/bin/bash
# LDS / VALU counters of the growth kernels over 7 trees: tools/runs/pmc_lds.sh <tag> [bench args...]
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/r3/pmc_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES -d /tmp/pmc_$TAG -o p -- python $ROOT/bench.py --steps 7 --warmup 1 --no-full-run "$@" > /tmp/pmc_$TAG.log 2>&1
python $ROOT/tools/rocpd_pmc.py "$(find /tmp/pmc_$TAG -name '*.db' | sort | tail -1)" "$OUT/pmc_sq_lds.csv" > /dev/null
grep "grow_kernel" "$OUT/pmc_sq_lds.csv" | head -40
