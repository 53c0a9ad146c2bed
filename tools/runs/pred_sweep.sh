#!/bin/bash
# device predictor: tree-tile size (LGA_PRED_TILE_KB) and rows per lane (LGA_PRED_ROWS) against rows per second (bench/predict_bench.py)
for cfg in ${PRED_CFGS:-"7 1" "13 1" "20 1" "25 1" "32 1" "13 2" "20 2"}; do
  set -- $cfg
  echo "tile $1 KB rows/lane $2: $(LGA_PRED_TILE_KB=$1 LGA_PRED_ROWS=$2 timeout -k 10 200 python bench/predict_bench.py 2>/dev/null | grep -o '"rows_per_second": [0-9]*\|"max_abs_diff_vs_host": [0-9.e-]*' | tr '\n' ' ')"
done
