#!/bin/bash
# GPU test suite (optionally a -k expression), then nothing else.
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-gt}
mkdir -p "$OUT"
cd $ROOT
if [ -n "$2" ]; then
  timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread -k "$2" > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
else
  timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 170 --timeout-method thread > "$OUT/pytest.log" 2>&1 || { tail -60 "$OUT/pytest.log"; exit 1; }
fi
tail -5 "$OUT/pytest.log"
