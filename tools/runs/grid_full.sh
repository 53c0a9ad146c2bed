#!/bin/bash
# The reference's 108-point x 5-fold grid on one GPU with several numbers of rows in flight; the tables must agree
# with each other and with the table the batched-rounds engine produced (profiles/r2/grid_full_table_rounds_path.jsonl).
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-grid}
shift || true
ROWS=${@:-"1 8 16"}
mkdir -p "$OUT"
cd /tmp
for r in $ROWS; do
  timeout -k 10 240 python -u $ROOT/examples/gridsearch_diamonds.py --full --fresh --checkpoint /tmp/pg_$r.jsonl --rows-in-flight $r > "$OUT/grid_full_rows$r.log" 2>&1 || { tail -30 "$OUT/grid_full_rows$r.log"; exit 1; }
  grep -E "grid search|lgb.train|lgb.cv|trend|mean of top" "$OUT/grid_full_rows$r.log"
  cp /tmp/pg_$r.jsonl "$OUT/paramGrid_rows$r.jsonl"
done
python - $ROOT $ROWS <<PY
import json, sys
root, rows = sys.argv[1], sys.argv[2:]
def table(p):
    return {r["row"]: (r["iteration"], r["score"]) for r in map(json.loads, open(p))}
ref = table(root + "/profiles/r2/grid_full_table_rounds_path.jsonl")
for o in rows:
    t = table(f"/tmp/pg_{o}.jsonl")
    same = t == ref
    print("rows in flight", o, ": table == batched-rounds table:", same)
    if not same:
        bad = [k for k in ref if t.get(k) != ref[k]]
        print("  differing rows:", len(bad), [(k, t.get(k), ref[k]) for k in bad[:3]])
PY
