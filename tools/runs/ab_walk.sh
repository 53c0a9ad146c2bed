#!/bin/bash
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for d in 0 8 10 12 14 18 100; do
  LGA_WALK_DEPTH=$d timeout -k 10 300 python bench.py --steps 500 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walk depth <= $d', 'value', d['value'])"
done
