#!/bin/bash
# ms per tree over (LGA_MAX_UNIT_ROWS, LGA_TARGET_UNITS) for the in-tree HIP core and an A/B build
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
PREC=${1:-q16}; shift
for lib in _lga_hip.so _lga_hip_ab.so; do
  [ -f $ROOT/lightgbm_amd/$lib ] || continue
  for combo in "$@"; do
    m=${combo%%:*}; u=${combo##*:}
    ms=$(LGA_HIP_LIB=$lib LGA_MAX_UNIT_ROWS=$m LGA_TARGET_UNITS=$u timeout -k 10 120 python $ROOT/bench.py --steps 60 --warmup 40 --no-full-run --precision $PREC 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$lib $PREC max_unit=$m units=$u ms_per_tree=$ms"
  done
done
