#!/bin/bash
# Full GPU test suite, the benchmark as the driver runs it, and the full grid search.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-fc}
mkdir -p "$OUT"
cd $ROOT
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
tail -3 "$OUT/pytest.log"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
tail -1 "$OUT/bench.json"
cd /tmp
timeout -k 10 240 python -u $ROOT/examples/gridsearch_diamonds.py --full --fresh --checkpoint /tmp/pg.jsonl > "$OUT/grid_full.log" 2>&1 || { tail -30 "$OUT/grid_full.log"; exit 1; }
grep -E "grid search|lgb.train|lgb.cv |trend|mean of top" "$OUT/grid_full.log"
cp /tmp/pg.jsonl "$OUT/paramGrid.jsonl"
python - $ROOT <<PY
import json, sys
root = sys.argv[1]
def table(p):
    return {r["row"]: (r["iteration"], r["score"]) for r in map(json.loads, open(p))}
print("table == batched-rounds table:", table("/tmp/pg.jsonl") == table(root + "/profiles/r2/grid_full_table_rounds_path.jsonl"))
PY
