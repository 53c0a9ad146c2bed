#!/bin/bash
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-bc}
mkdir -p "$OUT"
cd $ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu.py -x -q --timeout 170 --timeout-method thread -k "parity_sweep or regression_trees_equal or binary_matches or quantised_auc or multiclass_matches or row_line_layout or rf_and_rollback" > "$OUT/pytest.log" 2>&1 || { tail -40 "$OUT/pytest.log"; exit 1; }
tail -2 "$OUT/pytest.log"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python - "$OUT/bench.json" <<PY
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"])
PY
