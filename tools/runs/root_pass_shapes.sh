#!/bin/bash
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/root_pass_shapes.txt
: > "$OUT"
cd /tmp; export TMPDIR=/tmp
for k in higgs higgs_cont randn28 randn32; do
  rm -rf /tmp/rps && timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/rps -o kt -- python $ROOT/tools/runs/root_pass_shapes.py $k > /tmp/rps.log 2>&1
  DB="$(find /tmp/rps -name '*.db' | sort | tail -1)"
  python $ROOT/tools/rocpd_stats.py "$DB" /tmp/rps.csv > /dev/null
  echo "== $k" >> "$OUT"; grep -E "grow_kernel<true|grow_kernel<false|walk_update" /tmp/rps.csv >> "$OUT"
done
cat "$OUT"
