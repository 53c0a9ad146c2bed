#!/bin/bash
# Parity tests of the small-data paths, the cold-start breakdown and the full grid.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-q}
mkdir -p "$OUT"
cd $ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu.py -x -q --timeout 120 --timeout-method thread -k "whole_tree or chunks or batched or parity_sweep" > "$OUT/pytest.log" 2>&1 || { tail -40 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
cd /tmp
timeout -k 10 200 python -u $ROOT/bench/cold_start.py > "$OUT/cold_start.json" 2> "$OUT/cold.err" || { tail -20 "$OUT/cold.err"; exit 1; }
grep -E "first_model|lgb_train_200|dataset_construct|native_library" "$OUT/cold_start.json"
LGA_GRID_TRACE=1 timeout -k 10 240 python -u $ROOT/examples/gridsearch_diamonds.py --full --fresh --checkpoint /tmp/pg.jsonl > "$OUT/grid_full.log" 2>&1 || { tail -30 "$OUT/grid_full.log"; exit 1; }
grep -E "grid search|grid trace|lgb.train|lgb.cv " "$OUT/grid_full.log"
python - $ROOT <<PY
import json, sys
root = sys.argv[1]
def table(p):
    return {r["row"]: (r["iteration"], r["score"]) for r in map(json.loads, open(p))}
print("table == batched-rounds table:", table("/tmp/pg.jsonl") == table(root + "/profiles/r2/grid_full_table_rounds_path.jsonl"))
PY
cd $ROOT && timeout -k 10 200 python -u tools/runs/train_modes.py 2>/dev/null | tee "$OUT/train_modes.jsonl"
