#!/bin/bash
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-tt}
mkdir -p "$OUT"
cd $ROOT
timeout -k 10 200 python -u -m pytest tests/test_gpu.py -x -q --timeout 120 --timeout-method thread -k "whole_tree or chunks or regression_trees_equal or parity_sweep or scan_kernel" > "$OUT/pytest.log" 2>&1 || { tail -40 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
LGA_TREE_TIMING=1 timeout -k 10 120 python -u bench/small_tree_bench.py --threads 1 --iters 200 > "$OUT/timing.jsonl" 2> "$OUT/timing.err" || { tail -20 "$OUT/timing.err"; exit 1; }
cat "$OUT/timing.jsonl"; grep "tree kernel" "$OUT/timing.err" | tail -2
