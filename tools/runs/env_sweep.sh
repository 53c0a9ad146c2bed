#!/bin/bash
# ms per tree (trees 41..100 of the headline benchmark) under environment settings: tools/runs/env_sweep.sh <precision> "VAR=a VAR2=b" "VAR=c" ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
PREC=${1:-full}; shift
for combo in "$@"; do
  ms=$(env $combo timeout -k 10 120 python $ROOT/bench.py --steps 60 --warmup 40 --no-full-run --precision $PREC --stats 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); p=d['config']['profile']
print(d['ms_per_step'], 'rounds/tree', round(p['rounds']/p['trees'],1), 'splits/tree', round(p['split_tasks']/p['trees']), 'part_rows/tree(M)', round(p['partition_rows']/p['trees']/1e6,1), 'hist_rows/tree(M)', round(p['hist_rows']/p['trees']/1e6,1))")
  echo "$PREC [$combo] ms_per_tree=$ms"
done
