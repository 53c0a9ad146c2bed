#!/bin/bash
# single-pass partition: rows per workgroup forced / chosen per round (tools/runs/kstats.sh per variant)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
run() { tag=$1; shift; env "$@" bash $ROOT/tools/runs/kstats.sh $tag --no-other-precision | grep -E "partition|flags" ; tail -1 $ROOT/gpurun_out/r3/kstats_$tag/rounds_last_tree.csv; }
run pauto LGA_X=1
run p2048 LGA_PART_ROWS=2048
run p8192 LGA_PART_ROWS=8192
run pauto_b LGA_PART_ROWS4K_ABOVE=300000 LGA_PART_ROWS8K_ABOVE=2000000
run pauto_c LGA_PART_ROWS4K_ABOVE=100000000 LGA_PART_ROWS8K_ABOVE=1000000
