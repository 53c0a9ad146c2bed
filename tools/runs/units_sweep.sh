#!/bin/bash
# ms per tree of the headline benchmark over LGA_TARGET_UNITS, for the in-tree HIP core and (if present) an A/B build
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
PREC=${1:-q16}
for lib in _lga_hip.so _lga_hip_ab.so; do
  [ -f $ROOT/lightgbm_amd/$lib ] || continue
  for u in 512 640 768 1024 1536 2048; do
    ms=$(LGA_HIP_LIB=$lib LGA_TARGET_UNITS=$u timeout -k 10 120 python $ROOT/bench.py --steps 60 --warmup 40 --no-full-run --precision $PREC 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$lib $PREC units=$u ms_per_tree=$ms"
  done
done
