#!/bin/bash
# First GPU contact of the whole-tree kernel: parity tests, then the small-booster throughput bench.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-tk}
mkdir -p "$OUT"
cd $ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu.py -x -q --timeout 120 --timeout-method thread -k "whole_tree or regression_trees_equal" > "$OUT/pytest.log" 2>&1 || { tail -40 "$OUT/pytest.log"; exit 1; }
tail -3 "$OUT/pytest.log"
timeout -k 10 200 python -u bench/small_tree_bench.py --threads 1,4,8,16 > "$OUT/small_tree_q4.jsonl" 2> "$OUT/small_tree_q4.err" || { tail -20 "$OUT/small_tree_q4.err"; exit 1; }
cat "$OUT/small_tree_q4.jsonl"
GPU_MAX_HW_QUEUES=16 timeout -k 10 200 python -u bench/small_tree_bench.py --threads 8,16,32 > "$OUT/small_tree_q16.jsonl" 2> "$OUT/small_tree_q16.err" || { tail -20 "$OUT/small_tree_q16.err"; exit 1; }
cat "$OUT/small_tree_q16.jsonl"
