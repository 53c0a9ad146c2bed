#!/bin/bash
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-ck}
mkdir -p "$OUT"
cd $ROOT
timeout -k 10 300 python -u tools/runs/chunk_check.py > "$OUT/chunk_check.log" 2>&1 || { tail -30 "$OUT/chunk_check.log"; exit 1; }
cat "$OUT/chunk_check.log" | grep -v amdgpu.ids
